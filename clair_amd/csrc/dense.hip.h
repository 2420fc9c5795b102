// Position-mixing "slice dense" layer (L3) and the classifier tail (L4 reduce, L5 x4, heads, softmax).
#pragma once
#include "common.hip.h"

namespace clair {

// ---- L3 + L4 fused: slice dense (256 x dense 33->30, selu) feeding the split-K 7680->192 product --------
// clair/model.py:225-244 (slice_dense_layer), :464-479 (L3 + flatten, flat index u*256 + c), :482-488 (L4).
//   l3[n][u*256 + c] = selu( sum_t a2[t][n][c] * W3[c][t][u] + b3[c][u] )
//   part[sp][n][j]   = sum_{u<30, c in the 32 channels of split sp} l3[n][u*256 + c] * W4[u*256 + c][j]
// A UNIT is 64 candidates x one group of 8 channels, i.e. the K-slice {u*256 + c} of L4 (240 rows of W4).  Both products run as the
// 2-way fp16 split on v_mfma_f32_32x32x16_f16:
//   * the a2 tile [33 t][64 cand][8 ch] (67.6 KB) arrives by LDS-DMA;
//   * L3, transposed: D[u][cand] = sum_t W3^T[u][t] a2[t][cand] per channel, K = t padded 33 -> 48 with the BIAS as row 33 (the
//     activation operand carries 1.0 there).  The MFMA wants eight consecutive k per lane, the tile has t outermost: a lane
//     gathers its eight t values of FOUR channels with eight ds_read_b128 and splits them into the two fp16 planes in registers;
//   * selu + split of the 64 x 240 outputs become L4's A operand in LDS [plane][cand][u*8 + ch];
//   * L4: [64 x 240] x [240 x 192], the W4 fragments streamed from L2; 64 candidates, so that every W4 fragment feeds two MFMAs
//     (the stream is bound by the 64 B/clk L1 return path).
// Round 4: ONE 512-thread workgroup per CU WALKS L34_WALK = 4 units (the four channel groups its XCD owns, same
// 64 candidates) with the L4 accumulators RESIDENT, and the two halves of the work run side by side on every SIMD:
//   * waves 0-3, the PRODUCERS (channel quad x candidate block): wait for the tile, L3, selu + split into registers, store the l3
//     tile -- and start the NEXT unit's DMA as soon as the four of them are done reading the a2 buffer;
//   * waves 4-7, the CONSUMERS ((K half) x (N half): three 32-column blocks x two 32-candidate blocks, 96 accumulator registers):
//     L4 of unit g over the l3 tile while the producers work on unit g+1 -- the matrix pipe runs L4's MFMAs under the producers'
//     selu VALU, and the a2 tile of unit g+2 is in flight under both.
//   Two workgroup barriers per unit hand the single l3 buffer over (B1: consumers done with it, producers hold the next one in
//   registers; B2: stored, and the next a2 tile has landed); the producers meet among themselves on an LDS counter before the DMA
//   overwrites the a2 buffer.  Round 3 ran one unit per 256-thread workgroup, two per CU in lock-step: all DMA at once at the HBM
//   rate, then everybody's VALU, then everybody's MFMAs (matrix pipe 16 % busy), 32 split-K partials (24.6 KB per candidate
//   written and read back; now 8 = 6.1 KB).
// Summation order of L4 (fixed; what "bit-identical" means from round 4 on): per (K half, split) one MFMA chain over the split's
// four channel groups in ascending order; lower + upper K half; the tail adds the eight splits in ascending order.
// The 30 KB/candidate l3 tensor never exists in HBM.
constexpr int L34_CAND = 64;
constexpr int L34_CH = 8;                        // channels per group
constexpr int L34_GROUPS = 2 * HID / L34_CH;     // 32 channel groups
constexpr int L34_WALK = L34_GROUPS / L4_SPLITS; // 4 channel groups per workgroup = per split-K partial
constexpr int L34_K = L3_UNITS * L34_CH;         // 240: one group's rows of W4
constexpr int L34_KS = L34_K / 16;               // 15 k-steps
constexpr int L34_ROW = L34_K + 8;               // fp16 units per candidate row of one l3 plane: 496 B, 16-B aligned, conflict-free ds_read_b128 over 16 rows
constexpr int L34_A2_BYTES = T_POS * L34_CAND * L34_CH * 4;   // the a2 staging tile: 67 584 B
constexpr int L34_L3_BYTES = 2 * L34_CAND * L34_ROW * 2;      // the l3 tile, two planes: 63 488 B
constexpr int L34_THREADS = 512;
static_assert(2 * 6 * 64 * 16 * 4 <= L34_L3_BYTES, "the K-half exchange fits the l3 tile's buffer");
constexpr int L34_SUB = L34_WALK >= 4 ? 1 : 4 / L34_WALK;   // workgroups that share an XCD's four channel groups for one candidate block (1: the workgroup walks all four)
constexpr int L34_XPS = L34_WALK > 4 ? L34_WALK / 4 : 1;    // XCDs that serve one split (a walk of more than four groups: the split's W4 rows sit in that many L2s, each for its own candidate blocks)
static_assert(L34_WALK * L4_SPLITS * L34_CH == 2 * HID && (L34_WALK >= 4 ? L34_WALK % 4 == 0 && 8 % L34_XPS == 0 : L34_SUB * L34_WALK == 4),
              "channel groups cover the 256 LSTM2 features; an XCD owns four of them, or shares a longer walk with its neighbours");
// workgroups of a launch: per candidate block one per split; with L34_XPS > 1 the blocks are dealt over the XCDs of their split (rounded up to whole rounds of 8)
__host__ __device__ constexpr int l34_grid(int nblk) { return L34_XPS == 1 ? nblk * L4_SPLITS : (nblk + L34_XPS - 1) / L34_XPS * 8; }
static_assert(L34_A2_BYTES + L34_L3_BYTES + 16 <= 160 * 1024, "one workgroup per CU");
// l3 is multiplied by 2^4 before its 2-way fp16 split and the L4 reduction by 2^-4 (folded into TailArgs::l4_scale): a selu output
// of 0.01 would otherwise have a subnormal low plane (3e-8 absolute = 3e-6 relative); 2^4 keeps 22 bits down to |y| ~ 0.008 and
// overflows only beyond |y| = 4 094.
constexpr float L34_ACT_SCALE = 16.0f;

struct L3L4Args {
    const float *a2;    // [32 groups of 8 features][33][n_pad][8] (lstm32.hip.h: copy_write)
    const unsigned short *w3s;   // [256 c][3 kk][2 plane][64 lane][8] fp16 split A fragments of (W3[c]^T | b3[c]) * 2^w3_shift: row u = lane%32 (0 beyond 30),
                                 // k = 16*kk + 8*(lane/32) + j = t for t < 33, the bias for k = 33, 0 beyond
    const unsigned short *w4s;   // [32 cg][15 ks][6 nb][2 plane][64 lane][8] fp16 split B fragments of W4 * 2^w4_shift: row (2*ks + lane/32)*256 + cg*8 + j,
                                 // column nb*32 + lane%32
    float *part;        // [8 splits][blocks of 64][2 nh][6 mb*3+nb][4 a][64 lane][4 r]  split-K partial sums, in the accumulator layout
    int n_pad;
    float l3_unscale;   // 2^-w3_shift
    float *dbg;         // parity tap (NULL in production): l3 as this kernel holds it, hi + lo, [n_pad][7680]
    unsigned long long *stamps;   // phase clock of every wave: [workgroup][8 waves][16] s_memtime values (probe build -DCLAIR_L34_STAMPS with CLAIR_AMD_L34_STAMPS=1; unused otherwise)
};

// workgroup barrier with the LDS traffic of this wave retired first (nothing else: global loads and LDS-DMA stay in flight across it)
__device__ __forceinline__ void l34_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__global__ __launch_bounds__(L34_THREADS) void l3l4_kernel(L3L4Args p) {
#ifdef CLAIR_L34_STAMPS   // probe build only (clair_amd/build.py: build_probe; tools/gpu/l34_stamps.py): per-wave clock at every phase boundary
#define L34_STAMP(i) if (p.stamps && (threadIdx.x & 63) == 0) p.stamps[((size_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * 16 + (i)] = __builtin_amdgcn_s_memtime();
#else
#define L34_STAMP(i)
#endif
    L34_STAMP(0)
    __shared__ __attribute__((aligned(16))) float a2buf[L34_A2_BYTES / 4];          // [33 t][64 cand][8 ch] fp32, filled by LDS-DMA
    __shared__ __attribute__((aligned(16))) _Float16 l3h[2][L34_CAND][L34_ROW];      // [plane][cand][u*8 + ch]; at the very end the K-half exchange
    __shared__ unsigned psync;                                                       // producers' rendezvous counter
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l32 = lane & 31, hq = lane >> 5;
    // XCD-aware order (workgroups go round-robin over the 8 XCDs by linear id): XCD x owns split x = channel groups 4x .. 4x+3 for every
    // candidate block, so each L2 holds only its own 1/8 of the W4 fragments (740 KB) instead of every L2 streaming all 5.9 MB.
    const int xcd = blockIdx.x & 7;
    const int nblk = (p.n_pad + L34_CAND - 1) / L34_CAND;
    const int blk = L34_XPS == 1 ? (int)(blockIdx.x >> 3) / L34_SUB : (int)(blockIdx.x >> 3) * L34_XPS + xcd % L34_XPS;
    const int split = L34_XPS == 1 ? xcd * L34_SUB + (int)(blockIdx.x >> 3) % L34_SUB : xcd / L34_XPS;
    if (blk >= nblk) return;                       // the last round of a launch whose blocks do not fill it (L34_XPS > 1 only)
    const int n0 = blk * L34_CAND;
    const int cg0 = split * L34_WALK;
    if (tid == 0) psync = 0u;
    // a2 tile -> LDS.  Row q = t*64 + cand is the 32 bytes a2[cg][t][n0 + cand][0 .. 7]; one DMA piece moves 32 rows = one contiguous KiB of
    // the group-major tensor (lane l: row 32*piece + l/2, 16-byte slot l%2).  Slot s of a row holds channel half s ^ ((cand >> 3) & 1): the
    // fragment gather reads 16 bytes of every 32-byte row, and the swizzle puts candidates c and c + 8 on different banks.
    // Candidates beyond n_pad (the ragged half of the last block) re-read the last row; their partials are never used.
    // Four waves share a tile; wave q's pieces are q, q + 4, ...: always the same half of the 64 candidates (piece parity = q & 1), so the
    // per-lane part of the source address is one 32-bit offset for the kernel's life and a piece costs scalar arithmetic only (glds16_s:
    // wave-uniform base in SGPRs).  Round 5: the FIRST tile is issued by the producers (q = w), every later one by the CONSUMERS
    // (q = w - 4) from inside their MFMA stream -- 17 pieces were 1.7 k ticks of the producers' 13.8 k per unit, the critical path of the
    // kernel (tools/gpu/l34_stamps.py), and the consumers wait a third of every unit anyway.
    constexpr int L34_NPIECE = T_POS * L34_CAND / 32;   // 66: piece = 2 t + (candidate half)
    const int q4 = w & 3;
    const unsigned lds_a2 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lptr_t)a2buf);
    const int dcand = (q4 & 1) * 32 + (lane >> 1);
    const unsigned dma_lane_off = (unsigned)((min(n0 + dcand, p.n_pad - 1) * L34_CH + (((lane & 1) ^ ((dcand >> 3) & 1)) * 4)) * (int)sizeof(float));
    auto issue_pieces = [&](int cg, int first, int count) {      // pieces q4 + 4 * i, i in [first, first + count), of channel group cg's tile
        const char *base = (const char *)(p.a2 + ((size_t)cg * T_POS + (q4 >> 1)) * p.n_pad * L34_CH);
        const size_t step = (size_t)2 * p.n_pad * L34_CH * sizeof(float);          // two positions on
        for (int i = first; i < first + count && q4 + 4 * i < L34_NPIECE; ++i) {
            const size_t at = (size_t)(base + (size_t)i * step);        // wave-uniform by construction; said so to the register allocator
            const void *sbase = (const void *)(((size_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(at >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)at));
            glds16_s_nt(dma_lane_off, sbase, lds_a2 + (q4 + 4 * i) * 1024);   // a2 is read once
        }
    };
    constexpr int L34_MY_PIECES = (L34_NPIECE + 3) / 4;   // 17 (waves 0, 1) or 16 (waves 2, 3): issue_pieces stops at the tile's end

    if (w < 4) {
        // =================================== producers: a2 tile -> L3 -> selu + split -> l3 tile ===================================
        // wave w = (channel quad cq = w >> 1, candidate block mb3 = w & 1): channels cg*8 + 4cq .. +3 of candidates 32mb3 .. +31
        // The producers are the critical path of a unit (~1 300 VALU instructions of selu + split per wave against the consumers' 144 MFMAs):
        // they win the issue arbitration of their SIMD, the consumers' MFMAs fill the gaps.
        __builtin_amdgcn_s_setprio(3);
        const int cq = w >> 1, mb3 = w & 1;
        const int cand = mb3 * 32 + l32;
        const float *arow = a2buf + (size_t)cand * 8 + ((cq ^ ((cand >> 3) & 1)) * 4);     // + t * 512 floats
        issue_pieces(cg0, 0, L34_MY_PIECES);
        f16x8 wf0[4][2], wf1[4][2], wf2[4][2];   // W3 fragments of k-step 0 / 1 / 2: [channel][plane]; k-step 0 of the NEXT unit is fetched a phase ahead
        {
            const f16x8 *wp = (const f16x8 *)p.w3s + (size_t)(cg0 * L34_CH + cq * 4) * (3 * 2 * 64) + lane;   // + ((cc*3 + kk)*2 + plane)*64
#pragma unroll
            for (int cc = 0; cc < 4; ++cc)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) wf0[cc][pl] = wp[((cc * 3 + 0) * 2 + pl) * 64];
        }
        CLAIR_VMWAIT(0);
        L34_STAMP(1)
        l34_barrier();                              // Bp: the first tile is in LDS, psync is zero
#pragma unroll 1
        for (int g = 0; g < L34_WALK; ++g) {
            const int cg = cg0 + g;
            const f16x8 *wp = (const f16x8 *)p.w3s + (size_t)(cg * L34_CH + cq * 4) * (3 * 2 * 64) + lane;
            f32x16 acc3[4];
#pragma unroll
            for (int cc = 0; cc < 4; ++cc)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc3[cc][i] = 0.0f;
#pragma unroll
            for (int kk = 0; kk < 3; ++kk) {
                if (kk == 0) {
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc)
#pragma unroll
                        for (int pl = 0; pl < 2; ++pl) wf1[cc][pl] = wp[((cc * 3 + 1) * 2 + pl) * 64];
                } else if (kk == 1) {
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc)
#pragma unroll
                        for (int pl = 0; pl < 2; ++pl) wf2[cc][pl] = wp[((cc * 3 + 2) * 2 + pl) * 64];
                }
                // the lane's eight k values (t = 16kk + 8hq + j) of the four channels -> B fragments, hi and lo plane
                union { f16x8 v; unsigned u[4]; } bh[4], bl[4];
                if (kk < 2) {
                    f32x4 v[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = *(const f32x4 *)(arow + (size_t)(16 * kk + 8 * hq + j) * (L34_CAND * L34_CH));
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc)
#pragma unroll
                        for (int jp = 0; jp < 4; ++jp) {
                            const float x0 = v[2 * jp][cc], x1 = v[2 * jp + 1][cc];
                            float r0, r1;
                            asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(bh[cc].u[jp]) : "v"(x0), "v"(x1));
                            asm volatile("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(r0) : "v"(x0), "v"(bh[cc].u[jp]));
                            asm volatile("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r1) : "v"(x1), "v"(bh[cc].u[jp]));
                            asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(bl[cc].u[jp]) : "v"(r0), "v"(r1));
                        }
                } else {   // k = 32 is t = 32, k = 33 the bias row (activation 1.0), the rest of the padding is zero; lanes of the upper k half hold zeros only
                    const f32x4 v = *(const f32x4 *)(arow + (size_t)32 * (L34_CAND * L34_CH));
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc) {
                        const float x0 = hq ? 0.0f : v[cc], x1 = hq ? 0.0f : 1.0f;
                        float r0;
                        asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(bh[cc].u[0]) : "v"(x0), "v"(x1));
                        asm volatile("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(r0) : "v"(x0), "v"(bh[cc].u[0]));
                        asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(bl[cc].u[0]) : "v"(r0), "v"(0.0f));
                        bh[cc].u[1] = bh[cc].u[2] = bh[cc].u[3] = 0u;
                        bl[cc].u[1] = bl[cc].u[2] = bl[cc].u[3] = 0u;
                    }
                }
                // three product terms per k-step, small ones first: w_lo.a_hi, w_hi.a_lo, w_hi.a_hi
                const f16x8 (&wf)[4][2] = kk == 0 ? wf0 : (kk == 1 ? wf1 : wf2);
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) acc3[cc] = mfma32h(wf[cc][1], bh[cc].v, acc3[cc]);
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) acc3[cc] = mfma32h(wf[cc][0], bl[cc].v, acc3[cc]);
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) acc3[cc] = mfma32h(wf[cc][0], bh[cc].v, acc3[cc]);
            }
            asm volatile("" : "+v"(acc3[3]));
            L34_STAMP(2 + 3 * g)
            // this producer is done reading the a2 buffer (its ds_reads have returned): once all four have said so the consumers let the next
            // tile in (the count is only ever read by them: no rendezvous among the producers any more)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_fetch_add(&psync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            asm volatile("" ::: "memory");
            if (g + 1 < L34_WALK) {
                const f16x8 *wn = (const f16x8 *)p.w3s + (size_t)((cg + 1) * L34_CH + cq * 4) * (3 * 2 * 64) + lane;
#pragma unroll
                for (int cc = 0; cc < 4; ++cc)
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl) wf0[cc][pl] = wn[((cc * 3 + 0) * 2 + pl) * 64];
            }
            // selu, 2-way fp16 split (the L4 product runs on the fp16 matrix cores) into registers: accumulator register 4a + r of a lane is row
            // u = 8a + 4hq + r of candidate 32mb3 + lane%32; the wave's four channels of one (candidate, u) make one 8-byte LDS store per plane
            uint2 hi[4][4], lo[4][4];
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const f32x2 y01 = selu_scaled2((f32x2){acc3[0][4 * a + r], acc3[1][4 * a + r]} * p.l3_unscale, L34_ACT_SCALE);
                    const f32x2 y23 = selu_scaled2((f32x2){acc3[2][4 * a + r], acc3[3][4 * a + r]} * p.l3_unscale, L34_ACT_SCALE);
                    const float y[4] = {y01[0], y01[1], y23[0], y23[1]};
                    split2_pk4(y, hi[a][r], lo[a][r]);
                }
            L34_STAMP(3 + 3 * g)
            l34_barrier();                          // B1: the consumers are done with the previous l3 tile
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int u = 8 * a + 4 * hq + r;
                    if (a < 3 || u < L3_UNITS) {
                        *(uint2 *)&l3h[0][cand][u * L34_CH + cq * 4] = hi[a][r];
                        *(uint2 *)&l3h[1][cand][u * L34_CH + cq * 4] = lo[a][r];
                    }
                }
            CLAIR_VMWAIT(0);                        // this wave's W3 fragments of the next unit have landed
            l34_barrier();                          // B2: the l3 tile is complete, the next a2 tile is in LDS (the consumers waited for their pieces)
            L34_STAMP(4 + 3 * g)
        }
        l34_barrier();                              // E1, E2: the consumers' K-half exchange
        l34_barrier();
        L34_STAMP(14)
    } else {
        // =================================== consumers: L4 over the l3 tile, accumulators resident over the walk ===================================
        // wave = (K half kh, N half nh): output columns 96nh .. +95 (three 32-column blocks), both 32-candidate blocks, k-steps [0, 8) or [8, 15)
        // of every unit.  D[cand][col] += sum_k l3[cand][k] W4[k][col] as the 2-way fp16 split.
        const int kh = (w - 4) >> 1, nh = (w - 4) & 1;
        const int ks0 = kh ? 8 : 0, nks = kh ? 7 : 8;
        f32x16 acc[2][3];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int nb = 0; nb < 3; ++nb)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[mb][nb][i] = 0.0f;
        L34_STAMP(1)
        l34_barrier();                              // Bp
        // The a2 tile of unit t (t >= 1) may enter the buffer once the four producers have finished the L3 product of unit t - 1
        // (psync >= 4 t); it has to be there when they start unit t, right after barrier B2 of unit t - 1.  Tile 1: issued in front of
        // unit 0's barriers (the consumers have nothing else to do yet); tile g + 2: in the shadows of unit g's MFMAs, nine pieces per
        // k-step from the moment the count allows it, the rest (a producer that was late) right behind the loop.
        int dma_done = L34_MY_PIECES;               // pieces of the tile in flight / being issued that this wave has issued
        auto dma_ready = [&](int t) -> bool {       // one LDS word: the same answer in every lane, said so to the compiler (a scalar branch)
            return (unsigned)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(&psync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) >= 4u * (unsigned)t;
        };
        if (L34_WALK > 1) {
            while (!dma_ready(1)) __builtin_amdgcn_s_sleep(1);
            asm volatile("" ::: "memory");
            issue_pieces(cg0 + 1, 0, L34_MY_PIECES);
        }
#pragma unroll 1
        for (int g = 0; g < L34_WALK; ++g) {
            const int cg = cg0 + g;
            const f16x8 *bsrc = (const f16x8 *)p.w4s + ((size_t)cg * L34_KS * 6 + nh * 3) * 2 * 64 + lane;   // + (ks*6 + nb)*2*64 + plane*64
            // B fragments stream from L2 with a prefetch distance of PF - 1 k-steps (the first ones across the two barriers); fully unrolled, static ring
            constexpr int PF = 3;
            f16x8 bq[PF][3][2];
#pragma unroll
            for (int i = 0; i < PF - 1; ++i)
#pragma unroll
                for (int nb = 0; nb < 3; ++nb)
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl) bq[i][nb][pl] = bsrc[((size_t)(ks0 + i) * 6 + nb) * 128 + pl * 64];
            l34_barrier();                          // B1: this wave is done reading the previous l3 tile
            L34_STAMP(2 + 3 * g)
            CLAIR_VMWAIT(0);                        // this wave's pieces of the a2 tile of unit g + 1 have landed (and the W4 fragments above)
            l34_barrier();                          // B2: the l3 tile of unit g is complete, the producers may start on the next a2 tile
            const bool more = g + 2 < L34_WALK;     // unit g + 2's tile goes in during this unit's MFMAs
            dma_done = more ? 0 : L34_MY_PIECES;
            L34_STAMP(3 + 3 * g)
            if (p.dbg) {   // debug tap: this unit's 64 x (30 u x 8 channels) slice of l3
                for (int f = tid - 256; f < L34_CAND * L34_K; f += 256) {
                    const int row = f / L34_K, k = f - row * L34_K, u = k >> 3, ch = k & 7;
                    if (n0 + row < p.n_pad)
                        p.dbg[(size_t)(n0 + row) * L3_OUT + u * 256 + cg * L34_CH + ch] = ((float)l3h[0][row][k] + (float)l3h[1][row][k]) * (1.0f / L34_ACT_SCALE);
                }
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (i < nks) {
                    const int ks = ks0 + i;
                    if (i + PF - 1 < nks) {
#pragma unroll
                        for (int nb = 0; nb < 3; ++nb)
#pragma unroll
                            for (int pl = 0; pl < 2; ++pl) bq[(i + PF - 1) % PF][nb][pl] = bsrc[((size_t)(ks + PF - 1) * 6 + nb) * 128 + pl * 64];
                    }
                    f16x8 a[2][2];   // [m-block][plane]: candidate 32mb + lane%32, k = 16ks + 8hq ..
#pragma unroll
                    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                        for (int pl = 0; pl < 2; ++pl) a[mb][pl] = *(const f16x8 *)&l3h[pl][mb * 32 + l32][16 * ks + 8 * hq];
#pragma unroll
                    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                        for (int nb = 0; nb < 3; ++nb) acc[mb][nb] = mfma32h(a[mb][1], bq[i % PF][nb][0], acc[mb][nb]);
#pragma unroll
                    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                        for (int nb = 0; nb < 3; ++nb) acc[mb][nb] = mfma32h(a[mb][0], bq[i % PF][nb][1], acc[mb][nb]);
#pragma unroll
                    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                        for (int nb = 0; nb < 3; ++nb) acc[mb][nb] = mfma32h(a[mb][0], bq[i % PF][nb][0], acc[mb][nb]);
                    if (i >= 2 && dma_done < L34_MY_PIECES && dma_ready(g + 2)) {   // wave-uniform; the producers need ~3 k-steps for their L3 product
                        asm volatile("" ::: "memory");
                        issue_pieces(cg + 2, dma_done, 9);      // the wave's 16-17 pieces over two k-steps: four per k-step left the tile landing 2 k ticks
                        dma_done += 9;                           // after the producers wanted it (tools/gpu/l34_stamps.py, profiles/r05_l34_stamps.txt)
                    }
                }
            }
            asm volatile("" : "+v"(acc[1][2]));
            if (dma_done < L34_MY_PIECES) {          // what the k-steps did not get to
                while (!dma_ready(g + 2)) __builtin_amdgcn_s_sleep(1);
                asm volatile("" ::: "memory");
                issue_pieces(cg + 2, dma_done, L34_MY_PIECES - dma_done);
                dma_done = L34_MY_PIECES;
            }
            L34_STAMP(4 + 3 * g)
        }
        // the two K halves meet in LDS (fixed order: lower + upper), then go out as this split's partial, fragment-major:
        // [split][block of 64][nh][mb*3 + nb][a][lane][4 r] -- a lane's accumulator quad is 16 contiguous bytes and a wave instruction one contiguous KiB
        l34_barrier();                              // E1: every consumer is done reading the last l3 tile
        f32x4 *xch = (f32x4 *)&l3h[0][0][0] + (size_t)nh * (6 * 4 * 64) + lane;   // [nh][mb*3 + nb][a][lane]
        if (kh) {
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int nb = 0; nb < 3; ++nb)
#pragma unroll
                    for (int a = 0; a < 4; ++a)
                        xch[((mb * 3 + nb) * 4 + a) * 64] = (f32x4){acc[mb][nb][4 * a], acc[mb][nb][4 * a + 1], acc[mb][nb][4 * a + 2], acc[mb][nb][4 * a + 3]};
        }
        l34_barrier();                              // E2
        if (!kh) {
            float *dst = p.part + (((size_t)split * nblk + blk) * 2 + nh) * (6 * 4 * 256) + lane * 4;
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int nb = 0; nb < 3; ++nb)
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        const f32x4 up = xch[((mb * 3 + nb) * 4 + a) * 64];
                        *(f32x4 *)(dst + ((mb * 3 + nb) * 4 + a) * 256) =
                            (f32x4){acc[mb][nb][4 * a] + up[0], acc[mb][nb][4 * a + 1] + up[1], acc[mb][nb][4 * a + 2] + up[2], acc[mb][nb][4 * a + 3] + up[3]};
                    }
        }
        L34_STAMP(14)
    }
#undef L34_STAMP
}

// ---- tail: L4 split-K reduce + selu, L5_1..4 + selu, heads + selu + softmax ---------------------
// clair/model.py:482-488 (L4), :507-569 (L5_k), :582-620 (heads: selu on the logits, then softmax).
// Output rows are packed gt21(21) | genotype(3) | len1(33) | len2(33).
//
// One workgroup of eight waves per 32-candidate tile; waves k and k + 4 share branch k's L5 product by K halves, wave k then owns the branch to
// the end (head k -> softmax).  Round 4: both products run as the
// 2-way fp16 split on v_mfma_f32_32x32x16_f16, TRANSPOSED (weights = A operand, candidates = columns) like L3 -- round 3 ran them on
// v_mfma_f32_16x16x4_f32, a sixteenth of the rate: 360 MFMAs of 32 cycles per wave and 16 candidates, 11.5 k cycles of a 20 us kernel that
// held 64 CUs.  Now 144 MFMAs per wave and 32 candidates, and with 32 candidates per workgroup the W5 fragments a wave streams from L2
// (73.7 KB per branch, bound by the 64 B/clk L1 return path) take as long as the MFMAs they feed.  In the transposed product a lane's
// accumulator quad is four consecutive outputs of ONE candidate: bias, selu and the split are per-lane, the l5 tile is written with 8-byte
// LDS stores, and the softmax of a candidate lives in two lanes (l and l ^ 32) -- one cross-lane exchange instead of a 16-lane butterfly.
// Activations are multiplied by 2^4 before their split and the weight images by a per-tensor power of two, as for L3 / L4 (engine.hip).
constexpr int TAIL_TILE = 32;
constexpr int TL4_ROW = L4_UNITS + 8;   // fp16 units per candidate row of one l4 plane: 400 B (16-B aligned, conflict-free ds_read_b128)
constexpr int TL5_ROW = L5_UNITS + 8;   // 208 B
constexpr float TAIL_ACT_SCALE = 16.0f;

struct TailArgs {
    const float *l4part;        // [8 splits][blocks of 64][2 nh][6 mb*3+nb][4 a][64 lane][4 r]  (l3l4_kernel above)
    const float *b4;            // [192]
    const unsigned short *w5s;  // [4 branch][12 ks][3 nb][2 plane][64 lane][8] fp16 split A fragments of W5_k^T * 2^w5_shift[k]: row n = nb*32 + lane%32,
                                // k = 16*ks + 8*(lane/32) + j
    const float *b5;            // [4][96]
    const unsigned short *whs;  // [4 branch][6 ks][2 nb][2 plane][64 lane][8] fp16 split A fragments of Wh_k^T * 2^wh_shift[k]: row class = nb*32 + lane%32 (0 beyond
                                // the head's size), k = 16*ks + 8*(lane/32) + j
    const float *bh;            // [4][64] head biases, 0-padded
    float *out;                 // [n][90]
    int n_pad;
    int n;                      // valid candidates
    float l4_scale;             // 2^-w4_shift / L34_ACT_SCALE: the partials are sums over the W4 image and the scaled l3 planes
    float l5_scale[4];          // 2^-w5_shift[k] / TAIL_ACT_SCALE
    float head_scale[4];        // 2^-wh_shift[k] / TAIL_ACT_SCALE
};

constexpr int TAIL_THREADS = 512;

__global__ __launch_bounds__(TAIL_THREADS) void tail_kernel(TailArgs p) {
    __shared__ __attribute__((aligned(16))) _Float16 l4h[2][TAIL_TILE][TL4_ROW];        // [plane][cand][unit]
    __shared__ __attribute__((aligned(16))) _Float16 l5h[4][2][TAIL_TILE][TL5_ROW];     // [branch][plane][cand][unit]; before that the K-half exchange of the branch
    static_assert(3 * 4 * 64 * 16 <= 2 * TAIL_TILE * TL5_ROW * 2, "a branch's upper-half accumulators fit its l5 tile");
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int w = wv & 3, kh = wv >> 2;         // eight waves: branch w, K half kh of its L5 product (k-steps 6kh .. 6kh + 5)
    const int l32 = lane & 31, hq = lane >> 5;
    const int n0 = blockIdx.x * TAIL_TILE;

    // This kernel is a chain of dependent memory round trips on 32 CUs (nothing to hide them behind at one workgroup per CU), so it is built to
    // make as few of them as it can: the split-K partials and the first four of the wave's six k-steps of W5 fragments leave together, the last
    // two as soon as the partials' registers are free -- the whole weight stream of the L5 product is two round trips, the reduction one.
    const f16x8 *wp = (const f16x8 *)p.w5s + ((size_t)w * 12 + 6 * kh) * (3 * 2 * 64) + lane;   // + ((i*3 + nb)*2 + plane)*64, i = k-step within the half
    f16x8 wq[6][3][2];
    // L4: fixed-order reduction of the split-K partials, bias, selu, 2-way split.  A thread takes accumulator quads of the producing
    // kernel's layout -- 16-byte loads, a wave instruction one contiguous KiB.  This tile is candidate block mb of the 64-candidate block
    // n0 / 64: quad (nh, nb, a) of lane ln holds candidates 8a + 4*(ln/32) + r of column 96nh + 32nb + ln%32: 24 quads x 64 lanes over 512
    // threads = 3 per thread, all their partial loads in flight at once.
    {
        const int blk = n0 / L34_CAND, mb = (n0 >> 5) & 1, nblk = (p.n_pad + L34_CAND - 1) / L34_CAND;
        constexpr int QB = L4_SPLITS <= 8 ? 3 : 1;          // quads per round
#pragma unroll
        for (int round = 0; round < 3 / QB; ++round) {
            f32x4 part[QB][L4_SPLITS];
            float b4[QB];
#pragma unroll
            for (int q = 0; q < QB; ++q) {
                const int f = tid + TAIL_THREADS * (QB * round + q);
                const int g = f >> 6, ln = f & 63, nh = g / 12, rem = g - nh * 12, nb = rem >> 2, a = rem & 3;
                const size_t at = ((((size_t)blk * 2 + nh) * 6 + mb * 3 + nb) * 4 + a) * 256 + ln * 4;
#pragma unroll
                for (int sp = 0; sp < L4_SPLITS; ++sp) part[q][sp] = *(const f32x4 *)(p.l4part + (size_t)sp * nblk * (2 * 6 * 4 * 256) + at);
                b4[q] = p.b4[nh * 96 + nb * 32 + (ln & 31)];
            }
            if (round == 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int nb = 0; nb < 3; ++nb)
#pragma unroll
                        for (int pl = 0; pl < 2; ++pl) wq[i][nb][pl] = wp[((i * 3 + nb) * 2 + pl) * 64];
            }
#pragma unroll
            for (int q = 0; q < QB; ++q) {
                const int f = tid + TAIL_THREADS * (QB * round + q);
                const int g = f >> 6, ln = f & 63, nh = g / 12, rem = g - nh * 12, nb = rem >> 2, a = rem & 3;
                f32x4 s = part[q][0];
#pragma unroll
                for (int sp = 1; sp < L4_SPLITS; ++sp) s += part[q][sp];      // split order
                const int col = nh * 96 + nb * 32 + (ln & 31);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    _Float16 hi, lo;
                    split2(selu_scaled(s[r] * p.l4_scale + b4[q], TAIL_ACT_SCALE), hi, lo);
                    const int cand = 8 * a + 4 * (ln >> 5) + r;
                    l4h[0][cand][col] = hi;
                    l4h[1][cand][col] = lo;
                }
            }
        }
    }
#pragma unroll
    for (int i = 4; i < 6; ++i)
#pragma unroll
        for (int nb = 0; nb < 3; ++nb)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) wq[i][nb][pl] = wp[((i * 3 + nb) * 2 + pl) * 64];
    __syncthreads();

    // L5 branch w, K half kh, transposed: D[n][cand] = sum_k W5_w[k][n] l4[cand][k]: three 32-row blocks, six k-steps, three product terms
    // (small ones first).  Summation order of a branch (fixed): each half one MFMA chain over its k-steps, lower + upper half.
    f32x16 acc[3];
#pragma unroll
    for (int nb = 0; nb < 3; ++nb)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[nb][i] = 0.0f;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        f16x8 b[2];
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) b[pl] = *(const f16x8 *)&l4h[pl][l32][16 * (6 * kh + i) + 8 * hq];
#pragma unroll
        for (int nb = 0; nb < 3; ++nb) acc[nb] = mfma32h(wq[i][nb][1], b[0], acc[nb]);
#pragma unroll
        for (int nb = 0; nb < 3; ++nb) acc[nb] = mfma32h(wq[i][nb][0], b[1], acc[nb]);
#pragma unroll
        for (int nb = 0; nb < 3; ++nb) acc[nb] = mfma32h(wq[i][nb][0], b[0], acc[nb]);
    }
    // the upper half hands its accumulators to the lower half's wave through the branch's (still unused) l5 tile and is done
    f32x4 *xch = (f32x4 *)&l5h[w][0][0][0] + lane;      // [nb*4 + a][lane]
    if (kh) {
#pragma unroll
        for (int nb = 0; nb < 3; ++nb)
#pragma unroll
            for (int a = 0; a < 4; ++a) xch[(nb * 4 + a) * 64] = (f32x4){acc[nb][4 * a], acc[nb][4 * a + 1], acc[nb][4 * a + 2], acc[nb][4 * a + 3]};
    }
    __syncthreads();
    if (kh) return;
#pragma unroll
    for (int nb = 0; nb < 3; ++nb)
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const f32x4 up = xch[(nb * 4 + a) * 64];
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[nb][4 * a + r] += up[r];
        }
    // the head's fragments on their way while the l5 tile is made: [ks][nb][plane]
    const int nh = w == 0 ? 21 : (w == 1 ? 3 : 33);
    const int off = w == 0 ? 0 : (w == 1 ? 21 : (w == 2 ? 24 : 57));
    const bool two = nh > 32;                 // wave-uniform: the 33-class heads have one class in a second 32-row block
    f16x8 hw[6][2][2];
    {
        const f16x8 *hp = (const f16x8 *)p.whs + (size_t)w * (6 * 2 * 2 * 64) + lane;   // + ((ks*2 + nb)*2 + plane)*64
#pragma unroll
        for (int ks = 0; ks < 6; ++ks)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
                hw[ks][0][pl] = hp[((ks * 2 + 0) * 2 + pl) * 64];
                hw[ks][1][pl] = hw[ks][0][pl];
                if (two) hw[ks][1][pl] = hp[((ks * 2 + 1) * 2 + pl) * 64];
            }
    }
    // bias, selu, split: accumulator register 4a + r of block nb is unit nb*32 + 8a + 4hq + r of candidate lane%32 -- one 8-byte store per plane
    {
        const float sc = p.l5_scale[w];
#pragma unroll
        for (int nb = 0; nb < 3; ++nb)
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const int u0 = nb * 32 + 8 * a + 4 * hq;
                const f32x4 bias = *(const f32x4 *)(p.b5 + w * L5_UNITS + u0);
                const f32x2 y01 = selu_scaled2((f32x2){acc[nb][4 * a], acc[nb][4 * a + 1]} * sc + (f32x2){bias[0], bias[1]}, TAIL_ACT_SCALE);
                const f32x2 y23 = selu_scaled2((f32x2){acc[nb][4 * a + 2], acc[nb][4 * a + 3]} * sc + (f32x2){bias[2], bias[3]}, TAIL_ACT_SCALE);
                const float y[4] = {y01[0], y01[1], y23[0], y23[1]};
                uint2 hi, lo;
                split2_pk4(y, hi, lo);
                *(uint2 *)&l5h[w][0][l32][u0] = hi;
                *(uint2 *)&l5h[w][1][l32][u0] = lo;
            }
    }
    __builtin_amdgcn_wave_barrier();       // the branch's l5 tile is written and read by this wave alone (LDS serves a wave in order)

    // head w, transposed: D[class][cand] = sum_k Wh_w[k][class] l5[cand][k], 6 k-steps; selu on the logits (model.py:586), softmax over the classes
    f32x16 hacc[2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int i = 0; i < 16; ++i) hacc[nb][i] = 0.0f;
#pragma unroll
    for (int ks = 0; ks < 6; ++ks) {
        f16x8 b[2];
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) b[pl] = *(const f16x8 *)&l5h[w][pl][l32][16 * ks + 8 * hq];
        hacc[0] = mfma32h(hw[ks][0][1], b[0], hacc[0]);
        if (two) hacc[1] = mfma32h(hw[ks][1][1], b[0], hacc[1]);
        hacc[0] = mfma32h(hw[ks][0][0], b[1], hacc[0]);
        if (two) hacc[1] = mfma32h(hw[ks][1][0], b[1], hacc[1]);
        hacc[0] = mfma32h(hw[ks][0][0], b[0], hacc[0]);
        if (two) hacc[1] = mfma32h(hw[ks][1][0], b[0], hacc[1]);
    }
    {
        // class of register 4a + r of block nb: nb*32 + 8a + 4hq + r.  Block 1 exists for the 33-class heads only and holds class 32 alone (a = r = 0, hq = 0).
        const float sc = p.head_scale[w];
        float v[17];
        float mx = -INFINITY;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const f32x4 bias = *(const f32x4 *)(p.bh + w * 64 + 8 * a + 4 * hq);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = 8 * a + 4 * hq + r;
                v[4 * a + r] = c < nh ? selu_f(hacc[0][4 * a + r] * sc + bias[r]) : -INFINITY;
                mx = fmaxf(mx, v[4 * a + r]);
            }
        }
        v[16] = (two && hq == 0) ? selu_f(hacc[1][0] * sc + p.bh[w * 64 + 32]) : -INFINITY;
        mx = fmaxf(mx, v[16]);
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        float sum = 0.0f;
#pragma unroll
        for (int i = 0; i < 17; ++i) {
            const bool valid = i < 16 ? (8 * (i >> 2) + 4 * hq + (i & 3)) < nh : (two && hq == 0);
            v[i] = valid ? __expf(v[i] - mx) : 0.0f;
            sum += v[i];
        }
        sum += __shfl_xor(sum, 32);
        const int row = n0 + l32;
        if (row < p.n) {
            float *o = p.out + (size_t)row * OUT_FLOATS + off;
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int c = 8 * a + 4 * hq + r;
                    if (c < nh) o[c] = v[4 * a + r] / sum;  // true division, as tf.nn.softmax
                }
            if (two && hq == 0) o[32] = v[16] / sum;
        }
    }
}

}  // namespace clair
