// Recurrent half of one BiLSTM layer (both directions), fp32-exact on MFMA.
//
// Reference semantics: CudnnCompatibleLSTMCell(128) under stack_bidirectional_dynamic_rnn
// (clair/model.py:299-312, 423-451): per step z = [x_t, h_{t-1}].W + b, gates (i, c~, f, o),
// c_t = sig(f) c_{t-1} + sig(i) tanh(c~), h_t = sig(o) tanh(c_t); the backward direction walks
// t = 32..0; zero initial state.  The x-part (x_t.Wx + b) arrives precomputed from the
// projection GEMM in fragment-major order (gemm.hip.h: zx_block_offset), so each step only
// adds h_{t-1}.Wh, a [16,128]x[128,512] product per 16-candidate tile.
//
// Mapping to the CU: one 256-thread workgroup (one wave per SIMD) owns TILES tiles of 16
// candidates of one direction for all 33 steps.  Wave w owns hidden units 32w..32w+31 of all
// four gates (8 column blocks of 16), so the gate non-linearities are lane-local in the MFMA
// C layout.  Its [128 x 128] slice of Wh stays in 256 VGPRs for the whole kernel (the register
// file is the only on-chip store big enough for the 256 KiB fp32 Wh); c_t stays in registers;
// h_t is exchanged between the four waves through a double-buffered LDS tile, one barrier per
// step.  K is visited in the order k = q*32 + kk (q = lane>>4) so that a lane's A operands for
// 4 consecutive MFMAs are one ds_read_b128.
#pragma once
#include "common.hip.h"

namespace clair {

constexpr int H_LDS_ROW = HID + 4;  // 132 floats: rows 16 B apart in bank space -> conflict-free b128 reads

struct LstmArgs {
    const float *zx;   // fragment-major x-projection [2][33][ntiles][4][8][64][4]
    const float *whp;  // packed recurrent weights [2][4][8][8][64][4]  (dir, wave, nb, kk/4, lane, kk%4)
    float *aout;       // [33][n_pad][256]  (fw -> cols 0..127, bw -> 128..255)
    int n_pad;
    int ntiles;
};

template <int TILES>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void lstm_rec_kernel(LstmArgs p) {
    __shared__ __attribute__((aligned(16))) float hbuf[2][TILES][16][H_LDS_ROW];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lq = lane >> 4;
    const int d = blockIdx.x & 1;
    const int tile0 = (blockIdx.x >> 1) * TILES;

    // resident weights: Bw[nb][kk] = Wh[k = lq*32 + kk][col = g*128 + 32w + 16hh + li], nb = g*2+hh
    float Bw[8][32];
    {
        const f32x4 *wp = (const f32x4 *)p.whp + (size_t)(d * 4 + w) * (8 * 8 * 64) + lane;
#pragma unroll
        for (int nb = 0; nb < 8; ++nb)
#pragma unroll
            for (int k4 = 0; k4 < 8; ++k4) {
                f32x4 v = wp[(nb * 8 + k4) * 64];
                Bw[nb][k4 * 4 + 0] = v[0];
                Bw[nb][k4 * 4 + 1] = v[1];
                Bw[nb][k4 * 4 + 2] = v[2];
                Bw[nb][k4 * 4 + 3] = v[3];
            }
    }

    float cst[TILES][8];
#pragma unroll
    for (int tl = 0; tl < TILES; ++tl)
#pragma unroll
        for (int e = 0; e < 8; ++e) cst[tl][e] = 0.0f;

    // zx fragments of this wave: block (t, tile) is 4 waves x 8 nb x 256 floats
    auto zx_ptr = [&](int t, int tile) {
        return (const f32x4 *)(p.zx + ((((size_t)(d * T_POS + t) * p.ntiles + tile) * 4 + w) * 8) * 256) + lane;
    };

    f32x4 znext[TILES][8];
    {
        const int t = d ? T_POS - 1 : 0;
#pragma unroll
        for (int tl = 0; tl < TILES; ++tl) {
            const f32x4 *z = zx_ptr(t, tile0 + tl);
#pragma unroll
            for (int nb = 0; nb < 8; ++nb) znext[tl][nb] = z[nb * 64];
        }
    }

    for (int s = 0; s < T_POS; ++s) {
        const int t = d ? T_POS - 1 - s : s;
        f32x4 acc[TILES][8];
#pragma unroll
        for (int tl = 0; tl < TILES; ++tl)
#pragma unroll
            for (int nb = 0; nb < 8; ++nb) acc[tl][nb] = znext[tl][nb];
        if (s + 1 < T_POS) {  // prefetch next step's x-projection under this step's MFMAs
            const int tn = d ? t - 1 : t + 1;
#pragma unroll
            for (int tl = 0; tl < TILES; ++tl) {
                const f32x4 *z = zx_ptr(tn, tile0 + tl);
#pragma unroll
                for (int nb = 0; nb < 8; ++nb) znext[tl][nb] = z[nb * 64];
            }
        }
#pragma unroll
        for (int tl = 0; tl < TILES; ++tl) {
            if (s > 0) {
                const float *hrow = &hbuf[(s - 1) & 1][tl][li][lq * 32];
#pragma unroll
                for (int k4 = 0; k4 < 8; ++k4) {
                    const f32x4 a = *(const f32x4 *)(hrow + k4 * 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int nb = 0; nb < 8; ++nb) acc[tl][nb] = mfma16(a[j], Bw[nb][k4 * 4 + j], acc[tl][nb]);
                }
            }
            // gates: element (row = 4*lq + r, unit = 32w + 16hh + li)
            float *orow = p.aout + ((size_t)t * p.n_pad + (size_t)(tile0 + tl) * 16 + lq * 4) * (2 * HID) + d * HID + w * 32 + li;
#pragma unroll
            for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float ig = sigmoid_f(acc[tl][0 + hh][r]);
                    const float gg = tanh_f(acc[tl][2 + hh][r]);
                    const float fg = sigmoid_f(acc[tl][4 + hh][r]);
                    const float og = sigmoid_f(acc[tl][6 + hh][r]);
                    const float c = fg * cst[tl][hh * 4 + r] + ig * gg;
                    cst[tl][hh * 4 + r] = c;
                    const float h = og * tanh_f(c);
                    hbuf[s & 1][tl][lq * 4 + r][w * 32 + hh * 16 + li] = h;
                    orow[(size_t)r * (2 * HID) + hh * 16] = h;
                }
        }
        __syncthreads();
    }
}

}  // namespace clair
