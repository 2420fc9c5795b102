// The two BiLSTM layers (both directions each), recurrence kept on chip for all 33 steps.
//
// Reference semantics: CudnnCompatibleLSTMCell(128) under stack_bidirectional_dynamic_rnn
// (clair/model.py:299-312, 423-451): per step z = [x_t, h_{t-1}].W + b, gates (i, c~, f, o),
// c_t = sig(f) c_{t-1} + sig(i) tanh(c~), h_t = sig(o) tanh(c_t); the backward direction walks
// t = 32..0; zero initial state.
//
// Mapping to the CU.  One 256-thread workgroup (one wave per SIMD) owns one 16-candidate tile of one
// direction for all 33 steps.  Wh of a direction (128 x 512, 256 KiB as fp32 or as two fp16 planes) is larger
// than the 160 KiB LDS, so it lives in REGISTERS: wave w holds the [128 x 128] slice for hidden units
// 32w..32w+31 of all four gates (256 of the 512 unified registers).  Owning all four gates of its units
// makes the gate non-linearities lane-local in the MFMA C layout; c_t never leaves registers; h_t is
// exchanged between the four waves through a double-buffered LDS tile, one barrier per step.
//
// Products run as the 2-way fp16 split of common.hip.h,  a*b ~= a1*b1 + a1*b2 + a2*b1  on
// v_mfma_f32_16x16x32_f16: a step costs 8 blocks x 4 k-steps x 3 terms = 96 MFMAs of ~17 cycles instead
// of the 256 fp32 MFMAs of 32 cycles the first versions of this kernel used (profiles/r01_microbench.txt
// records why nothing can be hidden under an fp32 MFMA stream, and the variants that were tried).  End to
// end the probabilities stay within 2.5e-6 of the fp32 oracle and every VCF GT call is identical.
//   * h lives in LDS as its two fp16 planes (the gate code splits it once, 4 VALU instructions per element);
//     layer 1 copies those planes to HBM unchanged -- they are the A operand of the LSTM2 projection GEMM
//     (gemm_split.hip.h) -- layer 2 hands the fp32 sum p1 + p2, exactly the h its own recurrence used, to
//     the L3/L4 kernel.  Copy-out happens from LDS as whole rows at the start of the NEXT step.
//   * layer 1 computes its input projection itself (K = 32: one k-step; x_t read straight from the caller's
//     [n][33][32] tensor one step ahead, split on the fly; accumulators start from the bias), so no
//     x-projection tensor exists for layer 1.
//   * layer 2 gets its x-projection (fragment-major, gemm_split.hip.h) by LDS-DMA (global_load_lds, no VGPRs)
//     one whole step ahead; the fragments seed the accumulators with one ds_read_b128 each.  All VMEM of a
//     step is issued before its MFMAs, so the only vmcnt wait (a step later) never stalls.
//
// Gate pre-scaling: the host multiplies every gate column of Wx, Wh and the bias by the constant its
// activation needs in front of v_exp_f32 (2^x): -log2(e) for the sigmoid gates i, f, o and 2*log2(e) for the
// tanh gate c~ (engine.hip: gate_scale), and the cell state is carried as c' = 2*log2(e)*c.  The MFMA result
// is then directly the exp2 argument: 21 VALU/transcendental instructions per gate element instead of 27
// (+4 for the fp16 split of h).
#pragma once
#include "common.hip.h"

namespace clair {

constexpr int HP_ROW = HID + 8;   // fp16 units per LDS row of one plane: 272 B, 16-B aligned, rows 4 banks apart

struct LstmSplitArgs {
    const float *x;             // FIRST: [n_pad][33][32] network input
    const unsigned short *wxs;  // FIRST: x-part   [2 dir][4 wave][8 nb][2 plane][64 lane][8] fp16, k = 8*lq + j (K = 32), gate-scaled
    const float *bias;          // FIRST: [2][512] gate-scaled
    const float *zx;            // !FIRST: fragment-major x-projection (bias included), gate-scaled
    const unsigned short *whs;  // h-part   [2 dir][4 wave][8 nb][4 kstep][2 plane][64 lane][8] fp16, k = 32*ks + 8*lq + j, gate-scaled
    unsigned short *aout2;      // FIRST: [2 plane][33][n_pad][256] fp16 planes of the layer output
    float *aout;                // !FIRST: [33][n_pad][256] fp32
    int n_pad;
    int ntiles;
};

template <bool FIRST>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void lstm_split_kernel(LstmSplitArgs p) {
    __shared__ __attribute__((aligned(16))) _Float16 hbuf[2][2][16][HP_ROW];   // [step parity][plane][row][unit]
    __shared__ __attribute__((aligned(16))) float zlds[FIRST ? 1 : 2][4][8][256];   // layer 2: DMA'd x-projection fragments

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lq = lane >> 4;
    const int d = blockIdx.x & 1;
    const int tile = blockIdx.x >> 1;

    // resident weights: Bw[nb][ks][plane] = 8 fp16: Wh[32*ks + 8*lq + j][col(nb, li)]
    f16x8 Bw[8][4][2];
    {
        const f16x8 *wp = (const f16x8 *)p.whs + (size_t)(d * 4 + w) * (8 * 4 * 2 * 64) + lane;
#pragma unroll
        for (int nb = 0; nb < 8; ++nb)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) Bw[nb][ks][pl] = wp[((nb * 4 + ks) * 2 + pl) * 64];
    }
    f16x8 Bx[FIRST ? 8 : 1][2];
    float bv[FIRST ? 8 : 1];
    if (FIRST) {
        const f16x8 *xp = (const f16x8 *)p.wxs + (size_t)(d * 4 + w) * (8 * 2 * 64) + lane;
#pragma unroll
        for (int nb = 0; nb < 8; ++nb) {
            Bx[nb][0] = xp[(nb * 2 + 0) * 64];
            Bx[nb][1] = xp[(nb * 2 + 1) * 64];
            bv[nb] = p.bias[d * GATES + (nb >> 1) * HID + 32 * w + 16 * (nb & 1) + li];
        }
    }
    float cst[8];   // c' = 2 log2(e) c for elements e = hh*4 + r: (row 4*lq + r, unit 32w + 16hh + li)
#pragma unroll
    for (int e = 0; e < 8; ++e) cst[e] = 0.0f;

    // layer 1: this lane's 8 input features (lq*8 .. +7) of candidate li at step s, as fp16 hi/lo planes
    const float *xrow = FIRST ? p.x + ((size_t)tile * 16 + li) * (T_POS * F_IN) + lq * 8 : nullptr;
    auto load_x = [&](f32x4 (&xf)[2], int s) {
        const int t = d ? T_POS - 1 - s : s;
        xf[0] = *(const f32x4 *)(xrow + t * F_IN);
        xf[1] = *(const f32x4 *)(xrow + t * F_IN + 4);
    };
    // layer 2: LDS-DMA of this wave's 8 x-projection fragments of step s into zlds[s&1][w]
    auto fetch_zx = [&](int s) {
        const int t = d ? T_POS - 1 - s : s;
        const f32x4 *src = (const f32x4 *)(p.zx + ((((size_t)(d * T_POS + t) * p.ntiles + tile) * 4 + w) * 8) * 256) + lane;
        const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lptr_t)&zlds[FIRST ? 0 : (s & 1)][w][0][0]);
#pragma unroll
        for (int nb = 0; nb < 8; ++nb) glds16(src + nb * 64, lds0 + nb * 1024);
    };
    // h_s (both planes complete in LDS) -> HBM; thread: row tid>>4, 8-unit chunk tid&15
    auto store_h = [&](int s) {
        const int t = d ? T_POS - 1 - s : s;
        const int row = tid >> 4, c8 = tid & 15;
        const size_t off = ((size_t)t * p.n_pad + (size_t)tile * 16 + row) * (2 * HID) + d * HID + c8 * 8;
        const f16x8 hi = *(const f16x8 *)&hbuf[s & 1][0][row][c8 * 8];
        const f16x8 lo = *(const f16x8 *)&hbuf[s & 1][1][row][c8 * 8];
        if (FIRST) {
            const size_t plane = (size_t)T_POS * p.n_pad * (2 * HID);
            *(f16x8 *)(p.aout2 + off) = hi;
            *(f16x8 *)(p.aout2 + plane + off) = lo;
        } else {
            f32x4 o0, o1;
#pragma unroll
            for (int j = 0; j < 4; ++j) { o0[j] = (float)hi[j] + (float)lo[j]; o1[j] = (float)hi[4 + j] + (float)lo[4 + j]; }
            *(f32x4 *)(p.aout + off) = o0;
            *(f32x4 *)(p.aout + off + 4) = o1;
        }
    };

    f32x4 xcur[2], xnext[2];
    if (FIRST) load_x(xcur, 0); else fetch_zx(0);
    for (int s = 0; s < T_POS; ++s) {
        if (!FIRST) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // z(s): issued a whole step ago
        f32x4 acc[8];
        f16x8 afr[4][2];
        if (s > 0) {   // A fragments of h_{s-1}: lane (row li, k-chunk lq) of each k-step and plane
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) afr[ks][pl] = *(const f16x8 *)&hbuf[(s - 1) & 1][pl][li][ks * 32 + lq * 8];
        }
        if (FIRST) {
            load_x(xnext, s + 1 < T_POS ? s + 1 : s);
            f16x8 xh, xl;
#pragma unroll
            for (int j = 0; j < 8; ++j) { _Float16 a, b; split2(xcur[j >> 2][j & 3], a, b); xh[j] = a; xl[j] = b; }
#pragma unroll
            for (int nb = 0; nb < 8; ++nb) acc[nb] = (f32x4){bv[nb], bv[nb], bv[nb], bv[nb]};
#pragma unroll
            for (int nb = 0; nb < 8; ++nb) acc[nb] = mfma16h(xl, Bx[nb][0], acc[nb]);
#pragma unroll
            for (int nb = 0; nb < 8; ++nb) acc[nb] = mfma16h(xh, Bx[nb][1], acc[nb]);
#pragma unroll
            for (int nb = 0; nb < 8; ++nb) acc[nb] = mfma16h(xh, Bx[nb][0], acc[nb]);
        } else {
#pragma unroll
            for (int nb = 0; nb < 8; ++nb) acc[nb] = *(const f32x4 *)&zlds[s & 1][w][nb][lane * 4];
            if (s + 1 < T_POS) fetch_zx(s + 1);
        }
        if (s > 0) {
            store_h(s - 1);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
                for (int nb = 0; nb < 8; ++nb) acc[nb] = mfma16h(afr[ks][1], Bw[nb][ks][0], acc[nb]);
#pragma unroll
                for (int nb = 0; nb < 8; ++nb) acc[nb] = mfma16h(afr[ks][0], Bw[nb][ks][1], acc[nb]);
#pragma unroll
                for (int nb = 0; nb < 8; ++nb) acc[nb] = mfma16h(afr[ks][0], Bw[nb][ks][0], acc[nb]);
            }
        }
        // gates; acc holds exp2 arguments (pre-scaled columns): e^-i, e^2g, e^-f, e^-o
        constexpr float K2 = 2.0f * 1.44269504088896340736f;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float ri = fast_rcp(1.0f + __builtin_amdgcn_exp2f(acc[0 + hh][r]));
                const float rg = fast_rcp(1.0f + __builtin_amdgcn_exp2f(acc[2 + hh][r]));
                const float rf = fast_rcp(1.0f + __builtin_amdgcn_exp2f(acc[4 + hh][r]));
                const float ro = fast_rcp(1.0f + __builtin_amdgcn_exp2f(acc[6 + hh][r]));
                const float kg = fmaf(rg, -2.0f * K2, K2);
                const float c = fmaf(rf, cst[hh * 4 + r], ri * kg);
                cst[hh * 4 + r] = c;
                const float rc = fast_rcp(1.0f + __builtin_amdgcn_exp2f(c));
                const float h = fmaf(rc, -2.0f * ro, ro);
                _Float16 hi, lo;
                split2(h, hi, lo);
                hbuf[s & 1][0][lq * 4 + r][w * 32 + hh * 16 + li] = hi;
                hbuf[s & 1][1][lq * 4 + r][w * 32 + hh * 16 + li] = lo;
            }
        if (FIRST) { xcur[0] = xnext[0]; xcur[1] = xnext[1]; }
        __syncthreads();
    }
    store_h(T_POS - 1);
}

}  // namespace clair
