// LSTM2 input projection  zx2 = a1[33n,256] . Wx2[256,1024] + b2  (clair/model.py:443-450, x-part)
// with fp32-grade accuracy on the fp16 matrix cores ("2-way split", common.hip.h):
//     a*b ~= a1*b1 + a1*b2 + a2*b1,   x1 = fp16(x), x2 = fp16(x - x1)
// Three f16 MFMAs per block replace the fp32 MFMAs of the same block at a sixth of the matrix-pipe time.
//
// The product is computed TRANSPOSED, zx^T = Wx2^T . a1^T, on v_mfma_f32_32x32x16_f16: the A operand is the
// weight tile (rows = gate rows in the order the recurrent kernel wants them), the B operand the activation
// tile (columns = (t, candidate) rows of a1).  One 32x32 accumulator block is then exactly one
// (direction, 32-candidate tile, t, wave, block) piece of lstm32_kernel's accumulator layout
// (lstm32.hip.h): the epilogue stores it as four contiguous 1 KiB pieces (the bias seeds the accumulators).
//
// Weight-stationary, persistent.  K is only 256, so the whole K extent of a wave's 64 gate rows (both fp16
// planes: 64 KiB) fits the 256 accumulation VGPRs of a one-wave-per-SIMD kernel -- the same trick as the
// recurrent kernels.  A workgroup is four waves with FOUR DIFFERENT 64-row weight slices (256 gate rows) that all
// multiply the SAME 64-row activation tile.  The activation tile streams through a four-deep LDS ring of 64-wide
// k-phases (16 KiB each) by LDS-DMA, issued THREE phases (144 MFMAs, ~2 us) ahead: with several batches in flight a
// load takes microseconds to come back, and the round-1 kernel (registers two 24-MFMA slabs ahead, then ds_write)
// stalled on it at every slab (73 % matrix-pipe utilisation in its main loop).  One barrier per phase of 48 MFMAs.
//
// Workgroup -> work.  id & 7 = XCD (hardware round-robin); on an XCD, local id l = id >> 3: gate tile l & 3 (256 rows),
// group l >> 2.  The four workgroups of a group walk the SAME activation tiles (x_tile = xcd + 8 * (group + groups * j)),
// so an activation tile is fetched from HBM once per XCD and served to the other three from that XCD's L2.
#pragma once
#include "common.hip.h"
#include "lstm32.hip.h"
#include <type_traits>

namespace clair {

typedef unsigned short f16bits_t;   // raw fp16 storage

struct GemmSplitArgs {
    const f16bits_t *X3;    // [2][33*n_pad][256]  fp16 planes of a1 (rows in (t, n) order)
    const f16bits_t *W3;    // [16 slices of 64 gate rows][2 mi][16 kk][2 planes][64 lanes][8]  A fragments of the gate-scaled Wx2^T:
                            // gate row R = slice*64 + mi*32 + lane%32, k = 16*kk + 8*(lane/32) + j;
                            // R = ((d*4 + w)*4 + b)*32 + 8a + 4h' + c  <->  column d*512 + c*128 + 32w + 8b + 4h' + a
    const float *bias;      // [1024] gate-scaled, in gate-row order
    float *C;               // zx in lstm32_kernel's layout: [2 dir][n_pad/32][33][4 wave][4 b][4 a][64 lane][4 c]
    int n_pad;
    int ntiles;             // n_pad / 32
    int m_rows;             // 33 * n_pad
    int groups;             // workgroup groups per XCD (grid = 32 * groups)
};

constexpr int GS_ROWS = 64;                       // activation rows per tile
constexpr int GS_PLANE = GS_ROWS * 32;            // fp16 units of one plane of one 32-wide k-slab
constexpr int GS_PHASE = 2 * 2 * GS_PLANE;        // one phase = two slabs x two planes = 16 KiB

// LDS image of one activation slab plane: 64 rows x 32 fp16 = 64 B per row, four 16-byte chunks per row.  The
// fragment read of the 32x32x16 MFMA is ds_read_b128 of row l%32, chunk 2*kk + l/32; the hardware serves it in the
// lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31} (+32) (MI355X_MICROARCH.md, LDS): XOR-ing the chunk with
// (row >> 3) & 3 puts the 16 rows of every group on 16 different 16-byte slots of the 256-byte bank row.
__device__ __forceinline__ int split_lds_off(int row, int chunk) { return row * 32 + ((chunk ^ ((row >> 3) & 3)) << 3); }   // in fp16 units

constexpr int GS_RING = 4;                        // phases resident in LDS

// FUSED (lstm2_fused.hip.h): the workgroup's tiles are those of the candidate-tile pairs its XCD owns (pair q -> XCD q % 8,
// whatever the batch size), in the TIME order of its direction (gate tiles 2, 3 walk t = 32..0), and every wave publishes a
// ticket word per (direction, tile, t) block once its sixteen pieces of it are in L2.
template <bool FUSED>
__device__ __forceinline__ void gemm_split_body(const GemmSplitArgs &p, const int block, const FuseArgs &fz) {
    __shared__ __attribute__((aligned(16))) f16bits_t Xs[GS_RING][GS_PHASE];   // [ring slot][slab][plane][row][k]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l32 = lane & 31, lh = lane >> 5;
    const int xcd = block & 7, local = block >> 3;
    const int gtile = local & 3, group = local >> 2;
    const int x_tiles = (p.m_rows + GS_ROWS - 1) / GS_ROWS;
    const int x_step = 8 * p.groups;
    const int x_first = xcd + 8 * group;
    // fused: items i = s * nq + k of this XCD (step s of this gate tile's direction, pair q = xcd + 8k); group g takes i = g, g + groups, ...
    const int npairs = p.ntiles >> 1;
    const int nq = FUSED ? (xcd < npairs ? (npairs - xcd + 7) >> 3 : 0) : 0;
    const int n_items = T_POS * nq;
    if (FUSED ? group >= n_items : x_first >= x_tiles) return;
    const int n_my = FUSED ? (n_items - group + p.groups - 1) / p.groups : (x_tiles - x_first + x_step - 1) / x_step;   // activation tiles of this workgroup
    const int slice = gtile * 4 + wave;                           // this wave's 64 gate rows
    auto xt_of = [&](int it) -> int {                             // activation tile (64 rows of a1) number it of this workgroup
        if (!FUSED) return x_first + it * x_step;
        const int i = group + it * p.groups, s_ = i / nq, k = i - s_ * nq;
        return ((gtile >> 1) ? T_POS - 1 - s_ : s_) * npairs + xcd + 8 * k;
    };

    // resident weights: Wr[mi][kk][plane]
    f16x8 Wr[2][16][2];
    {
        const f16x8 *wp = (const f16x8 *)p.W3 + ((size_t)slice * (2 * 16 * 2 * 64)) + lane;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int kk = 0; kk < 16; ++kk)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) Wr[mi][kk][pl] = wp[((mi * 16 + kk) * 2 + pl) * 64];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int kk = 0; kk < 16; ++kk)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) asm volatile("" : "+a"(Wr[mi][kk][pl]));   // AGPR-resident, never re-loaded (lstm32.hip.h)
    }
    f32x4 bq[2][4];   // bias quads of this wave's two gate blocks: rows 8a + 4h' + c
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int a = 0; a < 4; ++a) bq[mi][a] = *(const f32x4 *)(p.bias + slice * 64 + mi * 32 + 8 * a + 4 * lh);
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int a = 0; a < 4; ++a) asm volatile("" : "+v"(bq[mi][a]));   // hipcc waits for its own loads HERE, not after the DMA prologue
    CLAIR_VMWAIT(0);   // from here on this kernel counts its vector-memory operations itself

    // Staging.  A phase is 2 slabs x 2 planes x 64 rows x 64 B = 16 one-KiB DMA pieces; wave w moves the four pieces of
    // plane-slab w (slab w>>1, plane w&1): piece j = rows 16j .. 16j+15, lane l -> LDS row 16j + l/4, 16-byte slot l%4, which
    // holds chunk (l%4) ^ ((row>>3)&3) of that row (split_lds_off).  Per-lane byte offsets are fixed for the kernel's life; the
    // base is wave-uniform scalar arithmetic.  Rows past m_rows of a ragged last tile are read as they come (the a1 workspace
    // carries slack rows) -- their accumulator blocks are never stored.
    const int my_slab = wave >> 1, my_plane = wave & 1;
    unsigned lane_off[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = 16 * j + (lane >> 2), c = (lane & 3) ^ ((row >> 3) & 3);
        lane_off[j] = (unsigned)(row * 512 + c * 16);
    }
    const f16bits_t *my_base = p.X3 + (size_t)my_plane * p.m_rows * 256 + my_slab * 32;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lptr_t)&Xs[0][0]) + (unsigned)(wave * GS_PLANE * 2);
    const int n_phases = n_my * 4;
    auto phase_base = [&](int P) -> const f16bits_t * {
        const int q = P < n_phases ? P : n_phases - 1;   // the prefetch past the end re-reads the last phase
        return my_base + ((size_t)xt_of(q >> 2) * GS_ROWS * 256) + (q & 3) * 64;
    };
    auto dma = [&](int P, int j) { glds16_s(lane_off[j], phase_base(P), lds0 + (unsigned)((P & (GS_RING - 1)) * GS_PHASE * 2 + j * 1024)); };
    // B fragments of one slab: [kk][plane][activation block]
    auto fread = [&](f16x8 (&xf)[2][2][2], int slot, int sl) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    xf[kk][pl][ni] = *(const f16x8 *)&Xs[slot][(sl * 2 + pl) * GS_PLANE + split_lds_off(ni * 32 + l32, 2 * kk + lh)];
    };

    auto fread1 = [&](f16x8 (&xf)[2][2][2], int slot, int sl, int i) {   // piece i = kk*4 + pl*2 + ni of the same
        const int kk = i >> 2, pl = (i >> 1) & 1, ni = i & 1;
        xf[kk][pl][ni] = *(const f16x8 *)&Xs[slot][(sl * 2 + pl) * GS_PLANE + split_lds_off(ni * 32 + l32, 2 * kk + lh)];
    };

    // Pipeline state at the top of phase P: ring slot P&3 = phase P visible to every wave, phases P+1 and P+2 in flight,
    // xa = fragments of phase P's first slab.
    f16x8 xa[2][2][2], xb[2][2][2];
#pragma unroll
    for (int P = 0; P < 3; ++P)
#pragma unroll
        for (int j = 0; j < 4; ++j) dma(P, j);
    CLAIR_VMWAIT(8);
    __syncthreads();
    fread(xa, 0, 0);

    // Two accumulator sets: tile i accumulates in set i & 1 while the sixteen 1 KiB stores of tile i-1 (the other set) ride in the
    // MFMA shadows of its first two phases -- as a burst after the last MFMA they were 11 % of the kernel (tools/gpu/gemm_ablate.sh).
    f32x16 accs[2][2][2];   // [set][gate block mi][activation block ni]
    f32x16 bias16[2];       // the bias in accumulator layout: C operand of every accumulator's first MFMA (D != C there)
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            bias16[mi][4 * a] = bq[mi][a][0]; bias16[mi][4 * a + 1] = bq[mi][a][1];
            bias16[mi][4 * a + 2] = bq[mi][a][2]; bias16[mi][4 * a + 3] = bq[mi][a][3];
        }
    asm volatile("" : "+v"(bias16[0]), "+v"(bias16[1]));
    // piece j = 0..15 of a tile's output: activation block ni = j >> 3, gate block mi = (j >> 2) & 1, quad a = j & 3; each accumulator
    // block is four contiguous 1 KiB pieces of the recurrent kernel's layout.  The address is wave-uniform but for lane * 16 bytes: the
    // part that belongs to the wave's gate block is computed once per kernel, the part that belongs to the activation block (one integer
    // division) once per tile -- per piece it was 25 scalar instructions and a branch in front of every store, a third of what a store
    // phase cost beyond its MFMAs (tools/gpu/gemm_stamps.py, profiles/r04_gemm_stamps.txt).
    size_t gate_off[2];   // elements: ((d * ntiles) * 33 * 16 + wb) * 1024 of gate block mi
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        const int gblk = slice * 2 + mi;   // = (d*4 + w)*4 + b
        gate_off[mi] = ((size_t)(gblk >> 4) * p.ntiles * T_POS * 16 + (gblk & 15)) * 1024;
    }
    auto tile_off = [&](int xt, int ni) -> size_t {          // elements: ((tile * 33 + t) * 16) * 1024 of activation block ni of tile xt
        const int xblk = xt * 2 + ni;      // = t * ntiles + tile
        const int t = xblk / p.ntiles, tile = xblk - t * p.ntiles;
        return ((size_t)tile * T_POS + t) * 16 * 1024;
    };
    auto store_piece = [&](const f32x16 (&acc)[2][2], const size_t (&toff)[2], int j) {
        const int ni = j >> 3, mi = (j >> 2) & 1, a = j & 3;
        f32x4 *dst = (f32x4 *)(p.C + gate_off[mi] + toff[ni] + a * 256) + lane;
        __builtin_nontemporal_store((f32x4){acc[mi][ni][4 * a], acc[mi][ni][4 * a + 1], acc[mi][ni][4 * a + 2], acc[mi][ni][4 * a + 3]}, dst);
    };
    // fused: this wave's ticket words of the two candidate tiles of activation tile xt
    auto publish = [&](int xt) {
        if (!FUSED) return;
        const int t = xt / npairs, q = xt - t * npairs;
        unsigned *fw = fz.flags + (((size_t)((gtile >> 1) * p.ntiles + 2 * q) * T_POS + t) * 8) + (gtile & 1) * 4 + wave;
        if (lane == 0) {
            __hip_atomic_store(fw, fz.ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(fw + T_POS * 8, fz.ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    auto run_tile = [&](auto set_c, auto first_c, int it) {
        constexpr int SET = decltype(set_c)::value;
        constexpr bool FIRST_TILE = decltype(first_c)::value;      // nothing to store yet: no branch in front of the stores of the others
        f32x16 (&acc)[2][2] = accs[SET];
        const int xt_prev = xt_of(FIRST_TILE ? 0 : it - 1);
        const size_t toff[2] = {tile_off(xt_prev, 0), tile_off(xt_prev, 1)};
#pragma unroll
        for (int ph = 0; ph < 4; ++ph) {
            const int P = it * 4 + ph;
            // ---- first slab of the phase; its partner's eight fragments are read one per MFMA shadow (as a burst in front of the
            //      MFMAs they held the wave's issue for ~130 cycles per slab)
#pragma unroll
            for (int m = 0; m < 24; ++m) {
                const int kk = m / 12, term = (m % 12) / 4, mi = (m >> 1) & 1, ni = m & 1;
                // three product terms per k-step, small ones first; the four blocks alternate so consecutive MFMAs never chain
                if (ph == 0 && m < 4) mfma32_av_first(acc[mi][ni], Wr[mi][0][1], xa[0][0][ni], bias16[mi]);
                else mfma32_av(acc[mi][ni], Wr[mi][ph * 4 + kk][term == 0 ? 1 : 0], xa[kk][term == 1 ? 1 : 0][ni]);
                __builtin_amdgcn_sched_barrier(0);
                CLAIR_DBG_FENCE();
                if (m < 8) fread1(xb, ph, 1, m);
                __builtin_amdgcn_sched_barrier(0);
            }
            // Phase P+1 must have landed (this wave's four pieces; the barrier covers the other waves').  Vector-memory operations
            // retire in issue order, and what was issued after those pieces is the second slab of phase P-1: up to eight stores of
            // the previous tile, THEN phase P+2's four pieces.  "At most four outstanding" therefore covers the pieces and the
            // stores before them; it does not lean on how stores and loads retire relative to each other ("at most twelve" in the
            // phases that carry stores would), and costs nothing measurable (tools/gpu/ab_libs.sh: 7.57-7.60 against 7.54-7.61 M/s).
            CLAIR_VMWAIT(4);
            __syncthreads();
            // fused: at most phase P+2's four pieces are outstanding here, so in phase 3 the previous tile's stores (issued in phases
            // 0 and 1) have retired -- its blocks are in L2; the two ticket stores are older than this slab's DMA pieces, so the
            // counts above hold
            if (FUSED && ph == 3 && !FIRST_TILE) publish(xt_prev);
            // ---- second slab: the next phase's first fragments (one per shadow), the previous tile's output (eight pieces in each of
            //      the first two phases), then phase P+3 into the slot phase P-1 has left
#pragma unroll
            for (int m = 0; m < 24; ++m) {
                const int kk = m / 12, term = (m % 12) / 4, mi = (m >> 1) & 1, ni = m & 1;
                mfma32_av(acc[mi][ni], Wr[mi][ph * 4 + 2 + kk][term == 0 ? 1 : 0], xb[kk][term == 1 ? 1 : 0][ni]);
                __builtin_amdgcn_sched_barrier(0);
                CLAIR_DBG_FENCE();
                if (m < 8) fread1(xa, (ph + 1) & 3, 0, m);
                // (four pieces in every phase instead of eight in the first two were measured: the same 8 270 cycles per tile, tools/gpu/gemm_stamps.py)
                if (!FIRST_TILE && ph < 2 && m >= 1 && m <= 8) store_piece(accs[SET ^ 1], toff, ph * 8 + m - 1);
                if (m >= 10 && m < 18 && (m & 1) == 0) dma(P + 3, (m - 10) >> 1);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    run_tile(std::integral_constant<int, 0>(), std::true_type(), 0);
    for (int it = 1; it < n_my; it += 2) {
        run_tile(std::integral_constant<int, 1>(), std::false_type(), it);
        if (it + 1 < n_my) run_tile(std::integral_constant<int, 0>(), std::false_type(), it + 1);
    }
    // the last tile's output (12 wait states between the last MFMA and the first read of its result); only this one can be ragged
    {
        const int last = n_my - 1, xt = xt_of(last);
        const size_t toff[2] = {tile_off(xt, 0), tile_off(xt, 1)};
        asm volatile("s_nop 11" : "+v"(accs[0][0][0]), "+v"(accs[0][0][1]), "+v"(accs[0][1][0]), "+v"(accs[0][1][1]),
                                  "+v"(accs[1][0][0]), "+v"(accs[1][0][1]), "+v"(accs[1][1][0]), "+v"(accs[1][1][1]));
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if ((xt * 2 + (j >> 3)) * 32 >= p.m_rows) continue;
            if (last & 1) store_piece(accs[1], toff, j); else store_piece(accs[0], toff, j);
        }
    }
    CLAIR_VMWAIT(0);   // the clamped prefetches past the end still target this workgroup's LDS
    if (FUSED) publish(xt_of(n_my - 1));
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm_split_kernel(GemmSplitArgs p) {
    gemm_split_body<false>(p, blockIdx.x, FuseArgs{nullptr, 0u, nullptr, nullptr});
}

}  // namespace clair
