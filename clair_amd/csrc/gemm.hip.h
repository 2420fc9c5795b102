// fp32-exact MFMA GEMM used for the LSTM input projections and the split-K L4 layer.
//
//   PROJ1:  zx1 = X[33n,32]   . Wx1[32,1024]  + b1   (x-part of LSTM1 kernels, clair/model.py:423-430)
//   PROJ2:  zx2 = a1[33n,256] . Wx2[256,1024] + b2   (x-part of LSTM2 kernels, clair/model.py:443-450)
//   L4   :  part[s] = l3[n,7680(slice s)] . W4[7680(slice s),192]          (clair/model.py:482-488)
//
// The contraction runs on v_mfma_f32_16x16x4_f32 (f32 in / f32 accumulate, bit-exact fmaf chain),
// so results carry plain fp32 round-off.  Tile: (2*MI*16) rows x (2*NI*16) columns per 256-thread
// workgroup, 2x2 waves, each wave MI x NI blocks of 16x16; K is consumed in slabs of 16 staged
// through LDS (rows padded to 20 floats so the ds_read_b128 fragment reads spread over banks).
//
// B operands are pre-packed on the host as Bp[slab][column][16] so a slab of the workgroup's
// columns is one contiguous chunk (pack_b_slabs in engine.hip).
#pragma once
#include "common.hip.h"

namespace clair {

enum GemmMode { GEMM_PROJ1 = 0, GEMM_PROJ2 = 1, GEMM_L4 = 2 };

struct GemmArgs {
    const float *A;     // PROJ1: X [n_pad][33][32]; PROJ2: a1 [33][n_pad][256]; L4: l3 [n_pad][7680]
    const float *Bp;    // packed [K/16][N][16]
    const float *bias;  // [N] (projections) or nullptr
    float *C;           // PROJ: fragment-major zx; L4: partial [splits][n_pad][192]
    int n_pad;          // candidates rounded up to 16
    int ntiles;         // n_pad / 16
    int m_rows;         // total rows of A (33*n_pad or n_pad)
    int slabs_per_wg;   // K-slabs each workgroup consumes
};

constexpr int LDS_ROW = 20;  // floats per staged row (16 + 4 pad)

// Fragment-major address of the 16x16 block (row-block rb = t*ntiles+tile, column block cb):
// zx[d][t][tile][w][nb][lane][4] with column cb*16 = d*512 + g*128 + w*32 + hh*16, nb = g*2+hh.
__device__ __forceinline__ size_t zx_block_offset(int rb, int cb, int ntiles) {
    int t = rb / ntiles, tile = rb - t * ntiles;
    int d = cb >> 5, rem = cb & 31;
    int g = rem >> 3, w = (rem >> 1) & 3, hh = rem & 1;
    return ((((size_t)(d * T_POS + t) * ntiles + tile) * 4 + w) * 8 + (g * 2 + hh)) * 256;
}

template <int MODE, int MI, int NI>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs p) {
    constexpr int NROWS = 2 * MI * 16;
    constexpr int NCOLS = 2 * NI * 16;
    constexpr int NTOT = (MODE == GEMM_L4) ? L4_UNITS : 2 * GATES;
    constexpr int KTOT = (MODE == GEMM_PROJ1) ? F_IN : (MODE == GEMM_PROJ2 ? 2 * HID : L3_OUT);
    // double-buffered slab staging: one barrier per slab
    __shared__ __attribute__((aligned(16))) float As[2][NROWS * LDS_ROW];
    __shared__ __attribute__((aligned(16))) float Bs[2][NCOLS * LDS_ROW];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 15, lq = lane >> 4;

    const int row0 = blockIdx.x * NROWS;
    const int col0 = blockIdx.y * NCOLS;
    const int slab0 = blockIdx.z * p.slabs_per_wg;

    // global source of this thread's A float4s (rows h*64 + (tid>>2), 16-B column tid&3)
    constexpr int A_PER_THREAD = NROWS / 64;
    const float *arow[A_PER_THREAD];
#pragma unroll
    for (int h = 0; h < A_PER_THREAD; ++h) {
        int r = row0 + h * 64 + (tid >> 2);
        r = r < p.m_rows ? r : p.m_rows - 1;  // clamp: rows beyond M are computed but never stored
        if (MODE == GEMM_PROJ1) {
            int t = r / p.n_pad, n = r - t * p.n_pad;
            arow[h] = p.A + (size_t)n * (T_POS * F_IN) + t * F_IN;
        } else {
            arow[h] = p.A + (size_t)r * KTOT;
        }
        arow[h] += (tid & 3) * 4;
    }
    constexpr int B_F4 = NCOLS * 4;               // float4s per B slab of this workgroup
    constexpr int B_PER_THREAD = B_F4 / 256;      // 2 (NI=4) or 3 (NI=6)
    const f32x4 *bsrc = (const f32x4 *)(p.Bp + ((size_t)slab0 * NTOT + col0) * 16) + tid;

    f32x4 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};

    f32x4 ra[A_PER_THREAD], rb[B_PER_THREAD];
    auto gload = [&](int s) {
#pragma unroll
        for (int h = 0; h < A_PER_THREAD; ++h) ra[h] = *(const f32x4 *)(arow[h] + (size_t)(slab0 + s) * 16);
#pragma unroll
        for (int h = 0; h < B_PER_THREAD; ++h) rb[h] = bsrc[(size_t)s * NTOT * 4 + h * 256];
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int h = 0; h < A_PER_THREAD; ++h)
            *(f32x4 *)&As[buf][(h * 64 + (tid >> 2)) * LDS_ROW + (tid & 3) * 4] = ra[h];
#pragma unroll
        for (int h = 0; h < B_PER_THREAD; ++h) {
            const int f = h * 256 + tid;  // float4 index inside the slab chunk: column f>>2, quarter f&3
            *(f32x4 *)&Bs[buf][(f >> 2) * LDS_ROW + (f & 3) * 4] = rb[h];
        }
    };
    const int S = p.slabs_per_wg;
    gload(0);
    lstore(0);
    if (S > 1) gload(1);
    __syncthreads();
    for (int s = 0; s < S; ++s) {
        const int buf = s & 1;
        f32x4 a[MI], b[NI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) a[mi] = *(const f32x4 *)&As[buf][(wm * MI * 16 + mi * 16 + li) * LDS_ROW + lq * 4];
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) b[ni] = *(const f32x4 *)&Bs[buf][(wn * NI * 16 + ni * 16 + li) * LDS_ROW + lq * 4];
        // slab s+1 (in registers since the previous iteration) -> the other buffer, whose readers all
        // passed the barrier that ended iteration s-1; then fetch slab s+2 under this slab's MFMAs
        if (s + 1 < S) lstore(buf ^ 1);
        if (s + 2 < S) gload(s + 2);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = mfma16(a[mi][j], b[ni][j], acc[mi][ni]);
        __syncthreads();
    }

    // epilogue
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int rblk = (row0 >> 4) + wm * MI + mi;  // 16-row block index
        if (rblk * 16 >= p.m_rows) continue;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int cblk = (col0 >> 4) + wn * NI + ni;
            if (MODE == GEMM_L4) {
                float *dst = p.C + ((size_t)blockIdx.z * p.n_pad + rblk * 16 + lq * 4) * L4_UNITS + cblk * 16 + li;
#pragma unroll
                for (int r = 0; r < 4; ++r) dst[(size_t)r * L4_UNITS] = acc[mi][ni][r];
            } else {
                const float bv = p.bias[cblk * 16 + li];
                f32x4 v = acc[mi][ni];
                v += (f32x4){bv, bv, bv, bv};
                *(f32x4 *)(p.C + zx_block_offset(rblk, cblk, p.ntiles) + lane * 4) = v;
            }
        }
    }
}

}  // namespace clair
