// LSTM2 input projection  zx2 = a1[33n,256] . Wx2[256,1024] + b2  (clair/model.py:443-450, x-part)
// with fp32-grade accuracy on the fp16 matrix cores ("2-way split", common.hip.h):
//     a*b ~= a1*b1 + a1*b2 + a2*b1,   x1 = fp16(x), x2 = fp16(x - x1)
// Three f16 MFMAs per block replace the fp32 MFMAs of the same block at a sixth of the matrix-pipe time.
//
// The product is computed TRANSPOSED, zx^T = Wx2^T . a1^T, on v_mfma_f32_32x32x16_f16: the A operand is the
// weight tile (rows = gate rows in the order the recurrent kernel wants them), the B operand the activation
// tile (columns = (t, candidate) rows of a1).  One 32x32 accumulator block is then exactly one
// (direction, t, 32-candidate tile, wave, block) piece of lstm32_kernel's accumulator layout
// (lstm32.hip.h): the epilogue adds the bias and stores it as four contiguous 1 KiB pieces.
//
// Operands arrive pre-split: LSTM1 writes its output as two fp16 planes, the host splits the gate-scaled
// Wx2 (engine.hip).  Tile: 128 gate rows x 128 activation rows per 256-thread workgroup, 2x2 waves of
// 64 x 64 (four accumulator blocks), K in steps of 32 through LDS.
#pragma once
#include "common.hip.h"
#include "lstm32.hip.h"

namespace clair {

typedef unsigned short f16bits_t;   // raw fp16 storage

struct GemmSplitArgs {
    const f16bits_t *X3;    // [2][33*n_pad][256]  fp16 planes of a1 (rows in (t, n) order)
    const f16bits_t *W3;    // [8 ksteps][2 planes][1024 gate rows][32 k]  fp16 planes of gate-scaled Wx2^T;
                            // gate row R = ((d*4 + w)*4 + b)*32 + 8a + 4h' + c  <->  column d*512 + c*128 + 32w + 8b + 4h' + a
    const float *bias;      // [1024] gate-scaled, in gate-row order
    float *C;               // zx in lstm32_kernel's layout: [2 dir][n_pad/32][33][4 wave][4 b][4 a][64 lane][4 c]
    int n_pad;
    int ntiles;             // n_pad / 32
    int m_rows;             // 33 * n_pad
};

// LDS tile of one plane: 128 rows x 32 fp16 = 64 B per row, four 16-byte chunks per row.  The fragment read
// of the 32x32x16 MFMA is ds_read_b128 of row l%32, chunk 2*kk + l/32; the hardware serves it in the lane
// groups {0-3,12-15,20-27}, {4-11,16-19,28-31} (+32) (MI355X_MICROARCH.md, LDS): XOR-ing the chunk with
// (row >> 3) & 3 puts the 16 rows of every group on 16 different 16-byte slots of the 256-byte bank row.
__device__ __forceinline__ int split_lds_off(int row, int chunk) { return row * 32 + ((chunk ^ ((row >> 3) & 3)) << 3); }   // in fp16 units

// PROBE != 0 only in tools/ubench/gemm_split_probe.hip: 1 = no zx store, 2 = no MFMAs, 3 = no global loads after the first
template <int PROBE = 0>
__global__ __launch_bounds__(256, 3) void gemm_split_kernel(GemmSplitArgs p) {
    __shared__ __attribute__((aligned(16))) f16bits_t Ws[2][128 * 32];
    __shared__ __attribute__((aligned(16))) f16bits_t Xs[2][128 * 32];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l32 = lane & 31, lh = lane >> 5;
    // XCD-aware tile order.  Workgroups go round-robin over the 8 XCDs (one L2 each) by linear id, so the
    // eight gate-row tiles that share one 128-row activation tile are given ids with the same id & 7 and
    // consecutive id >> 3: they run back to back on ONE XCD and the activation tile crosses the fabric once
    // instead of up to eight times (PMC FETCH_SIZE 170 MB -> 65 MB per launch, profiles/r01_pmc_hbm_traffic.txt).
    const int wg = blockIdx.x;
    const int xcd = wg & 7, seq = wg >> 3;
    const int x_tile = (seq >> 3) * 8 + xcd;
    if (x_tile * 128 >= p.m_rows) return;
    const int xrow0 = x_tile * 128;          // first activation row ((t, n) order)
    const int grow0 = (seq & 7) * 128;       // first gate row

    // staging map: chunk id f = tid + 256*h (h = 0,1) of a plane tile: row f>>2, 16-byte chunk f&3
    const f16bits_t *xsrc[2];
    int lds_dst[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int f = tid + 256 * h, r = f >> 2, c = f & 3;
        int gr = xrow0 + r;
        gr = gr < p.m_rows ? gr : p.m_rows - 1;
        xsrc[h] = p.X3 + (size_t)gr * 256 + c * 8;
        lds_dst[h] = split_lds_off(r, c);
    }
    const size_t x_plane = (size_t)p.m_rows * 256;
    const f16bits_t *wsrc = p.W3 + ((size_t)grow0 * 32) + (size_t)tid * 8;   // + (kstep*2 + plane)*1024*32 + h*256*8

    f32x16 acc[2][2];   // [gate block mi][activation block ni]
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    f32x4 rx[2][2], rw[2][2];   // raw 16-byte chunks in flight (f16x8 each)
    auto gload = [&](int ks) {
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                rx[pl][h] = *(const f32x4 *)(xsrc[h] + pl * x_plane + ks * 32);
                rw[pl][h] = *(const f32x4 *)(wsrc + ((size_t)(ks * 2 + pl) * 1024 * 32) + h * 256 * 8);
            }
    };
    gload(0);
    constexpr int KSTEPS = 8;   // 256 / 32
    for (int ks = 0; ks < KSTEPS; ++ks) {
        __syncthreads();   // previous step's fragments are all in registers / consumed
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                *(f32x4 *)&Xs[pl][lds_dst[h]] = rx[pl][h];
                *(f32x4 *)&Ws[pl][lds_dst[h]] = rw[pl][h];
            }
        __syncthreads();
        if (ks + 1 < KSTEPS && PROBE != 3) gload(ks + 1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            // fragments of this wave's two gate blocks and two activation blocks, both planes:
            // lane (l32, lh) = row l32, k-chunk 2*kk + lh
            f16x8 wf[2][2], xf[2][2];
#pragma unroll
            for (int pl = 0; pl < 2; ++pl)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    wf[pl][i] = *(const f16x8 *)&Ws[pl][split_lds_off(wm * 64 + i * 32 + l32, 2 * kk + lh)];
                    xf[pl][i] = *(const f16x8 *)&Xs[pl][split_lds_off(wn * 64 + i * 32 + l32, 2 * kk + lh)];
                }
            if (PROBE == 2) {
                acc[0][0][kk] += (float)wf[0][0][0] + (float)xf[0][1][1] + (float)wf[1][1][2] + (float)xf[1][0][3];
                continue;
            }
            // three product terms, small ones first; the four blocks alternate so that consecutive MFMAs
            // never wait on each other's accumulator
#define SPLIT_TERM(PW, PX)                                                                     \
    _Pragma("unroll") for (int mi = 0; mi < 2; ++mi) _Pragma("unroll") for (int ni = 0; ni < 2; ++ni) \
        acc[mi][ni] = mfma32h(wf[PW][mi], xf[PX][ni], acc[mi][ni]);
            SPLIT_TERM(1, 0)
            SPLIT_TERM(0, 1)
            SPLIT_TERM(0, 0)
#undef SPLIT_TERM
        }
    }

    // epilogue: bias, then each accumulator block is four contiguous 1 KiB pieces of the recurrent kernel's layout
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const int xblk = (xrow0 >> 5) + wn * 2 + ni;      // = t * ntiles + tile
        if (xblk * 32 >= p.m_rows) continue;
        const int t = xblk / p.ntiles, tile = xblk - t * p.ntiles;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            const int gblk = (grow0 >> 5) + wm * 2 + mi;  // = (d*4 + w)*4 + b
            const int d = gblk >> 4, wb = gblk & 15;
            float *dst = p.C + ((((size_t)(d * p.ntiles + tile) * T_POS + t) * 16 + wb) * 1024) + lane * 4;
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const f32x4 bv = *(const f32x4 *)(p.bias + gblk * 32 + 8 * a + 4 * lh);
                f32x4 v = {acc[mi][ni][4 * a + 0], acc[mi][ni][4 * a + 1], acc[mi][ni][4 * a + 2], acc[mi][ni][4 * a + 3]};
                v += bv;
                if (PROBE == 1 && v[0] != 12345.678f) continue;
                *(f32x4 *)(dst + a * 256) = v;
            }
        }
    }
}

}  // namespace clair
