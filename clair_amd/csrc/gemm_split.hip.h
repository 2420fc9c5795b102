// LSTM2 input projection  zx2 = a1[33n,256] . Wx2[256,1024] + b2  (clair/model.py:443-450, x-part)
// with fp32-grade accuracy on the fp16 matrix cores ("2-way split", common.hip.h):
//     a*b ~= a1*b1 + a1*b2 + a2*b1,   x1 = fp16(x), x2 = fp16(x - x1)
// Three v_mfma_f32_16x16x32_f16 per 16x16x32 block replace eight v_mfma_f32_16x16x4_f32 (51 instead of 256
// matrix-pipe cycles), and -- unlike fp32 MFMAs -- they leave issue slots for the LDS reads in between.
//
// Operands arrive pre-split: LSTM1 writes its output as two fp16 planes (lstm.hip.h), the host splits the
// gate-scaled Wx2 (engine.hip).  Tile: 128 x 128 per 256-thread workgroup, 2x2 waves of 64 x 64
// (16 accumulator blocks), K in steps of 32 through LDS.
#pragma once
#include "common.hip.h"

namespace clair {

// Fragment-major address of the 16x16 block (row-block rb = t*ntiles+tile, column block cb):
// zx[d][t][tile][w][nb][lane][4] with column cb*16 = d*512 + g*128 + w*32 + hh*16, nb = g*2+hh.
__device__ __forceinline__ size_t zx_block_offset(int rb, int cb, int ntiles) {
    int t = rb / ntiles, tile = rb - t * ntiles;
    int d = cb >> 5, rem = cb & 31;
    int g = rem >> 3, w = (rem >> 1) & 3, hh = rem & 1;
    return ((((size_t)(d * T_POS + t) * ntiles + tile) * 4 + w) * 8 + (g * 2 + hh)) * 256;
}


typedef unsigned short f16bits_t;   // raw fp16 storage

struct GemmSplitArgs {
    const f16bits_t *A3;    // [2][33*n_pad][256]  fp16 planes of a1 (rows in (t, n) order)
    const f16bits_t *B3;    // [8 ksteps][2 planes][1024 cols][32 k]  fp16 planes of gate-scaled Wx2
    const float *bias;   // [1024] gate-scaled
    float *C;            // fragment-major zx (zx_block_offset)
    int n_pad;
    int ntiles;
    int m_rows;          // 33 * n_pad
};

// LDS tile of one plane: 128 rows (or columns) x 32 fp16 = 64 B per row, four 16-byte chunks per row,
// chunk index XOR-swizzled with (row >> 2) & 3 so that the ds_read_b128 of 16 consecutive rows at one
// k-chunk spreads over all banks (unswizzled: 4-way conflict, row pitch = 16 dwords).
__device__ __forceinline__ int split_lds_off(int row, int chunk) { return row * 32 + ((chunk ^ ((row >> 2) & 3)) << 3); }   // in fp16 units

// PROBE != 0 only in tools/ubench/gemm_split_probe.hip: 1 = no zx store, 2 = no MFMAs, 3 = no global loads after the
// first, 4 = per-workgroup phase timestamps
__device__ long long *gemm_probe_stamps;
template <int PROBE = 0>
__global__ __launch_bounds__(256, 3) void gemm_split_kernel(GemmSplitArgs p) {
    __shared__ __attribute__((aligned(16))) f16bits_t As[2][128 * 32];
    __shared__ __attribute__((aligned(16))) f16bits_t Bs[2][128 * 32];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 15, lq = lane >> 4;
    // XCD-aware tile order.  Workgroups go round-robin over the 8 XCDs (one L2 each) by linear id, so the
    // eight column tiles that share one 128-row A tile are given ids with the same id & 7 and consecutive
    // id >> 3: they run back to back on ONE XCD and the A tile crosses the fabric once instead of up to eight
    // times (PMC FETCH_SIZE of the (x = rows, y = cols) grid was 5x the A + B bytes, profiles/r01_pmc_hbm_traffic.txt).
    const int wg = blockIdx.x;
    long long stamp[4];
    if (PROBE == 4) stamp[0] = __builtin_readcyclecounter();
    const int xcd = wg & 7, seq = wg >> 3;
    const int row_tile = (seq >> 3) * 8 + xcd;
    if (row_tile * 128 >= p.m_rows) return;
    const int row0 = row_tile * 128;
    const int col0 = (seq & 7) * 128;

    // staging map: chunk id f = tid + 256*h (h = 0,1) of a plane tile: row f>>2, 16-byte chunk f&3
    const f16bits_t *asrc[2];
    int lds_dst[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int f = tid + 256 * h, r = f >> 2, c = f & 3;
        int gr = row0 + r;
        gr = gr < p.m_rows ? gr : p.m_rows - 1;
        asrc[h] = p.A3 + (size_t)gr * 256 + c * 8;
        lds_dst[h] = split_lds_off(r, c);
    }
    const size_t a_plane = (size_t)p.m_rows * 256;
    const f16bits_t *bsrc = p.B3 + ((size_t)col0 * 32) + (size_t)tid * 8;   // + (kstep*2 + plane)*1024*32 + h*256*8

    f32x4 acc[4][4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};

    f32x4 ra[2][2], rb[2][2];   // raw 16-byte chunks in flight (f16x8 each)
    auto gload = [&](int ks) {
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                ra[pl][h] = *(const f32x4 *)(asrc[h] + pl * a_plane + ks * 32);
                rb[pl][h] = *(const f32x4 *)(bsrc + ((size_t)(ks * 2 + pl) * 1024 * 32) + h * 256 * 8);
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
                *(f32x4 *)&As[pl][lds_dst[h]] = ra[pl][h];
                *(f32x4 *)&Bs[pl][lds_dst[h]] = rb[pl][h];   // B tile: "row" = column index inside the tile
            }
        __syncthreads();
        if (PROBE == 4 && ks == 0) stamp[1] = __builtin_readcyclecounter();
        if (ks + 1 < KSTEPS && PROBE != 3) gload(ks + 1);
        // A fragments of this wave's four 16-row blocks, both planes: lane (li, lq) = row li, k-chunk lq
        f16x8 af[2][4];
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) af[pl][mi] = *(const f16x8 *)&As[pl][split_lds_off(wm * 64 + mi * 16 + li, lq)];
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            f16x8 bfr[2];
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) bfr[pl] = *(const f16x8 *)&Bs[pl][split_lds_off(wn * 64 + ni * 16 + li, lq)];
            // three product terms, small ones first; the four row blocks alternate so that consecutive MFMAs
            // never wait on each other's accumulator
#define SPLIT_TERM(PA, PB)                                                    \
    _Pragma("unroll") for (int mi = 0; mi < 4; ++mi) acc[mi][ni] = mfma16h(af[PA][mi], bfr[PB], acc[mi][ni]);
            if (PROBE == 2) {
                acc[ni][0] += (f32x4){(float)af[0][ni][0], (float)bfr[0][1], (float)af[1][ni][2], (float)bfr[1][3]};
                continue;
            }
            SPLIT_TERM(1, 0)
            SPLIT_TERM(0, 1)
            SPLIT_TERM(0, 0)
#undef SPLIT_TERM
        }
    }

    if (PROBE == 4) {
        asm volatile("" : "+v"(acc[3][3]));
        stamp[2] = __builtin_readcyclecounter();
    }
    // epilogue: bias, fragment-major store: each accumulator block is one contiguous 1 KiB piece
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        const int rblk = (row0 >> 4) + wm * 4 + mi;
        if (rblk * 16 >= p.m_rows) continue;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const int cblk = (col0 >> 4) + wn * 4 + ni;
            const float bv = p.bias[cblk * 16 + li];
            f32x4 v = acc[mi][ni];
            v += (f32x4){bv, bv, bv, bv};
            if (PROBE == 1 && v[0] != 12345.678f) continue;
            *(f32x4 *)(p.C + zx_block_offset(rblk, cblk, p.ntiles) + lane * 4) = v;
        }
    }
    if (PROBE == 4) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamp[3] = __builtin_readcyclecounter();
        if (tid == 0) {
            for (int i = 0; i < 4; ++i) gemm_probe_stamps[(size_t)wg * 6 + i] = stamp[i];
            gemm_probe_stamps[(size_t)wg * 6 + 4] = __builtin_amdgcn_s_getreg((4 /*HW_ID*/) | (0 << 6) | (31 << 11));
            gemm_probe_stamps[(size_t)wg * 6 + 5] = __builtin_amdgcn_s_getreg((20 /*XCC_ID*/) | (0 << 6) | (31 << 11));
        }
    }
}

}  // namespace clair
