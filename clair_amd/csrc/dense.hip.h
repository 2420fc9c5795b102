// Position-mixing "slice dense" layer (L3) and the classifier tail (L4 reduce, L5 x4, heads, softmax).
#pragma once
#include "common.hip.h"

namespace clair {

// ---- L3 + L4 fused: slice dense (256 x dense 33->30, selu) feeding the split-K 7680->192 product --------
// clair/model.py:225-244 (slice_dense_layer), :464-479 (L3 + flatten, flat index u*256 + c), :482-488 (L4).
//   l3[n][u*256 + c] = selu( sum_t a2[t][n][c] * W3[c][t][u] + b3[c][u] )
//   part[cg][n][j]   = sum_{u<30, c in group cg} l3[n][u*256 + c] * W4[u*256 + c][j]        (cg = 8 channels)
// A workgroup owns 64 candidates x one group of 8 channels, i.e. the K-slice {u*256 + c} of L4 (240 rows of W4).  Both
// products run as the 2-way fp16 split on v_mfma_f32_32x32x16_f16 (round 2 ran L3 on v_mfma_f32_16x16x4_f32: a sixteenth of the
// rate, and a wave streaming fp32 MFMAs starves its SIMD partner; the 16x16x32 form of L4 hid at most two VALU instructions):
//   * the a2 tile [33 t][64 cand][8 ch] (67.6 KB) arrives by LDS-DMA;
//   * L3, transposed: D[u][cand] = sum_t W3^T[u][t] a2[t][cand] per channel, K = t padded 33 -> 48 with the BIAS as row 33 (the
//     activation operand carries 1.0 there).  The MFMA wants eight consecutive k per lane, the tile has t outermost: a lane
//     gathers its eight t values of FOUR channels with eight ds_read_b128 and splits them into the two fp16 planes in registers;
//   * selu + split of the 64 x 240 outputs go back into the same LDS buffer as L4's A operand [plane][cand][u*8 + ch];
//   * L4: [64 x 240] x [240 x 192], the W4 fragments streamed from L2.  64 candidates per workgroup, so that every W4 fragment feeds
//     two MFMAs: the stream is bound by the 64 B/clk L1 return path, and at 32 candidates it took as long as the MFMAs it fed
//     (360 KB per 32 candidates in round 2, 180 KB per 64 now).  The four waves are (K half) x (N half): three 32-column blocks x
//     two 32-candidate blocks each, the K halves summed through LDS in a fixed order.
// The 30 KB/candidate l3 tensor never exists in HBM.
constexpr int L34_CAND = 64;
constexpr int L34_CH = 8;                        // channels per group; L4_SPLITS groups
constexpr int L34_K = L3_UNITS * L34_CH;         // 240: this group's rows of W4
constexpr int L34_KS = L34_K / 16;               // 15 k-steps
constexpr int L34_ROW = L34_K + 8;               // fp16 units per candidate row of one l3 plane: 496 B, 16-B aligned, conflict-free ds_read_b128 over 16 rows
constexpr int L34_LDS_BYTES = T_POS * L34_CAND * L34_CH * 4;   // the a2 staging tile (67 584 B) is the largest of the buffer's three lives
static_assert(2 * L34_CAND * L34_ROW * 2 <= L34_LDS_BYTES && 2 * 6 * 64 * 16 * 4 <= L34_LDS_BYTES, "l3 tile and K-half exchange fit the a2 tile's buffer");
static_assert(L4_SPLITS * L34_CH == 2 * HID, "channel groups cover the 256 LSTM2 features");
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
    float *part;        // [32 cg][blocks of 64][2 nh][6 mb*3+nb][4 a][64 lane][4 r]  split-K partial sums, in the accumulator layout
    int n_pad;
    float l3_unscale;   // 2^-w3_shift
    float *dbg;         // parity tap (NULL in production): l3 as this kernel holds it, hi + lo, [n_pad][7680]
    unsigned long long *stamps;   // phase clock of every wave: [workgroup][wave][16] s_memtime values (probe build -DCLAIR_L34_STAMPS with CLAIR_AMD_L34_STAMPS=1; unused otherwise)
};

__global__ __launch_bounds__(256) void l3l4_kernel(L3L4Args p) {
#ifdef CLAIR_L34_STAMPS   // probe build only (clair_amd/build.py: build_probe; tools/gpu/l34_stamps.py): per-wave clock at every phase boundary
#define L34_STAMP(i) if (p.stamps && (threadIdx.x & 63) == 0) p.stamps[((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 16 + (i)] = __builtin_amdgcn_s_memtime();
#else
#define L34_STAMP(i)
#endif
    L34_STAMP(0)
    // one LDS buffer, three lives: the a2 tile [33 t][64 cand][8 ch] fp32 (filled by LDS-DMA), then -- after every wave has pulled its
    // fragments out of it -- the l3 tile that feeds L4, then the accumulators of the upper K half on their way to the lower half's waves
    __shared__ __attribute__((aligned(16))) float lds_buf[L34_LDS_BYTES / 4];
    _Float16 (*l3h)[L34_CAND][L34_ROW] = (_Float16 (*)[L34_CAND][L34_ROW])lds_buf;   // [plane][cand][u*8 + ch]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l32 = lane & 31, hq = lane >> 5;
    // XCD-aware order (workgroups go round-robin over the 8 XCDs by linear id): XCD x owns channel groups 4x .. 4x+3 for every
    // candidate block, so each L2 holds only its own 1/8 of the W4 fragments (740 KB) instead of every L2 streaming all 5.9 MB.
    const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
    const int blk = seq >> 2, nblk = (p.n_pad + L34_CAND - 1) / L34_CAND;
    const int n0 = blk * L34_CAND;
    const int cg = xcd * 4 + (seq & 3);        // channels cg*8 .. cg*8+7

    // ---- a2 tile -> LDS.  Row q = t*64 + cand is the 32 bytes a2[cg][t][n0 + cand][0 .. 7]; one DMA piece moves 32 rows = one contiguous KiB
    //      of the group-major tensor (lane l: row 32*piece + l/2, 16-byte slot l%2).  Slot s of a row holds channel half s ^ ((cand >> 3) & 1): the fragment gather below reads
    //      16 bytes of every 32-byte row, and the swizzle puts candidates c and c + 8 on different banks (conflict-free ds_read_b128).
    //      Candidates beyond n_pad (the ragged half of the last block) re-read the last row; their partials are never used.
    {
        constexpr int NPIECE = T_POS * L34_CAND / 32;   // 66
        const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lptr_t)lds_buf);
        for (int piece = w; piece < NPIECE; piece += 4) {
            const int q = piece * 32 + (lane >> 1);
            const int t = q >> 6, cand = q & 63;
            const int half = (lane & 1) ^ ((cand >> 3) & 1);
            const int row = min(n0 + cand, p.n_pad - 1);
            const float *src = p.a2 + (((size_t)cg * T_POS + t) * p.n_pad + row) * L34_CH + half * 4;
            glds16((const f32x4 *)src, lds0 + piece * 1024);
        }
    }
    L34_STAMP(1)

    // ---- L3: wave w = (channel quad cq = w >> 1, candidate block mb = w & 1): channels cg*8 + 4cq .. +3 of candidates 32mb .. +31 ----
    const int cq = w >> 1, mb3 = w & 1;
    f32x16 acc3[4];
#pragma unroll
    for (int cc = 0; cc < 4; ++cc)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc3[cc][i] = 0.0f;
    {
        const f16x8 *wp = (const f16x8 *)p.w3s + (size_t)(cg * L34_CH + cq * 4) * (3 * 2 * 64) + lane;   // + ((cc*3 + kk)*2 + plane)*64
        f16x8 wf[2][4][2];   // [kk parity][channel][plane], fetched a k-step ahead (the first under the DMA wait)
#pragma unroll
        for (int cc = 0; cc < 4; ++cc)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) wf[0][cc][pl] = wp[((cc * 3 + 0) * 2 + pl) * 64];
        CLAIR_VMWAIT(0);
        L34_STAMP(2)
        __syncthreads();
        L34_STAMP(3)
        const int cand = mb3 * 32 + l32;
        const float *arow = lds_buf + (size_t)cand * 8 + ((cq ^ ((cand >> 3) & 1)) * 4);     // + t * 512 floats
#pragma unroll
        for (int kk = 0; kk < 3; ++kk) {
            if (kk < 2) {
#pragma unroll
                for (int cc = 0; cc < 4; ++cc)
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl) wf[(kk + 1) & 1][cc][pl] = wp[((cc * 3 + kk + 1) * 2 + pl) * 64];
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
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) acc3[cc] = mfma32h(wf[kk & 1][cc][1], bh[cc].v, acc3[cc]);
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) acc3[cc] = mfma32h(wf[kk & 1][cc][0], bl[cc].v, acc3[cc]);
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) acc3[cc] = mfma32h(wf[kk & 1][cc][0], bh[cc].v, acc3[cc]);
        }
    }
    asm volatile("" : "+v"(acc3[3]));   // the MFMAs stay on this side of the barrier
    L34_STAMP(4)
    __syncthreads();                        // every wave is done reading the a2 tile: the buffer becomes the l3 tile
    L34_STAMP(5)
    // selu, 2-way fp16 split (the L4 product runs on the fp16 matrix cores), then one 8-byte LDS store per plane and (candidate, u):
    // the wave's four channels together.  Accumulator register 4a + r of a lane is row u = 8a + 4hq + r of candidate 32mb + lane%32.
    {
        const int cand = mb3 * 32 + l32;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int u = 8 * a + 4 * hq + r;
                if (a < 3 || u < L3_UNITS) {
                    const f32x2 y01 = selu_scaled2((f32x2){acc3[0][4 * a + r], acc3[1][4 * a + r]} * p.l3_unscale, L34_ACT_SCALE);
                    const f32x2 y23 = selu_scaled2((f32x2){acc3[2][4 * a + r], acc3[3][4 * a + r]} * p.l3_unscale, L34_ACT_SCALE);
                    const float y[4] = {y01[0], y01[1], y23[0], y23[1]};
                    uint2 hi, lo;
                    split2_pk4(y, hi, lo);
                    *(uint2 *)&l3h[0][cand][u * L34_CH + cq * 4] = hi;
                    *(uint2 *)&l3h[1][cand][u * L34_CH + cq * 4] = lo;
                }
            }
    }
    L34_STAMP(6)
    __syncthreads();
    L34_STAMP(7)
    if (p.dbg) {   // debug tap: this workgroup's 64 x (30 u x 8 channels) slice of l3
        for (int f = tid; f < L34_CAND * L34_K; f += 256) {
            const int row = f / L34_K, k = f - row * L34_K, u = k >> 3, ch = k & 7;
            if (n0 + row < p.n_pad)
                p.dbg[(size_t)(n0 + row) * L3_OUT + u * 256 + cg * L34_CH + ch] = ((float)l3h[0][row][k] + (float)l3h[1][row][k]) * (1.0f / L34_ACT_SCALE);
        }
    }

    // ---- L4 over this K-slice: wave w = (K half kh = w >> 1, N half nh = w & 1): output columns 96nh .. +95 (three 32-column blocks), both
    //      32-candidate blocks, k-steps [0, 8) or [8, 15).  D[cand][col] = sum_k l3[cand][k] W4[k][col] as the 2-way fp16 split.
    const int kh = w >> 1, nh = w & 1;
    f32x16 acc[2][3];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int nb = 0; nb < 3; ++nb)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mb][nb][i] = 0.0f;
    {
        const int ks0 = kh ? 8 : 0, nks = kh ? 7 : 8;
        const f16x8 *bsrc = (const f16x8 *)p.w4s + ((size_t)cg * L34_KS * 6 + nh * 3) * 2 * 64 + lane;   // + (ks*6 + nb)*2*64 + plane*64
        // B fragments stream from L2 with a prefetch distance of PF - 1 k-steps; the loop is fully unrolled so the ring is static
        constexpr int PF = 3;
        f16x8 bq[PF][3][2];
#pragma unroll
        for (int i = 0; i < PF - 1; ++i)
#pragma unroll
            for (int nb = 0; nb < 3; ++nb)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) bq[i][nb][pl] = bsrc[((size_t)(ks0 + i) * 6 + nb) * 128 + pl * 64];
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
            }
        }
    }
    // the two K halves meet in LDS (fixed order: lower + upper), then go out as split-K partials, fragment-major:
    // [cg][block of 64][nh][mb*3 + nb][a][lane][4 r] -- a lane's accumulator quad is 16 contiguous bytes and a wave instruction one contiguous KiB
    asm volatile("" : "+v"(acc[1][2]));
    L34_STAMP(8)
    __syncthreads();                        // every wave is done reading the l3 tile
    L34_STAMP(9)
    f32x4 *xch = (f32x4 *)lds_buf + (size_t)nh * (6 * 4 * 64) + lane;   // [nh][mb*3 + nb][a][lane]
    if (kh) {
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int nb = 0; nb < 3; ++nb)
#pragma unroll
                for (int a = 0; a < 4; ++a)
                    xch[((mb * 3 + nb) * 4 + a) * 64] = (f32x4){acc[mb][nb][4 * a], acc[mb][nb][4 * a + 1], acc[mb][nb][4 * a + 2], acc[mb][nb][4 * a + 3]};
    }
    __syncthreads();
    if (!kh) {
        float *dst = p.part + (((size_t)cg * nblk + blk) * 2 + nh) * (6 * 4 * 256) + lane * 4;
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
    L34_STAMP(10)
#undef L34_STAMP
}

// ---- tail: L4 split-K reduce + selu, L5_1..4 + selu, heads + selu + softmax ---------------------
// clair/model.py:482-488 (L4), :507-569 (L5_k), :582-620 (heads: selu on the logits, then softmax).
// Output rows are packed gt21(21) | genotype(3) | len1(33) | len2(33).
//
// One workgroup per 16-candidate tile; wave k owns branch k end to end (L5_k -> head k -> softmax),
// both products on v_mfma_f32_16x16x4_f32 with the weights streamed from L2 as pre-packed B
// fragments (engine.hip: pack_tail_weights).  K is visited as k = (lane>>4)*(K/4) + kk so a lane's A
// operands for four MFMAs are one ds_read_b128 of the LDS activation tile.
constexpr int TAIL_TILE = 16;
constexpr int L4S_ROW = L4_UNITS + 4;  // padded LDS rows (16 B aligned, conflict-free b128 reads)
constexpr int L5S_ROW = L5_UNITS + 4;

struct TailArgs {
    const float *l4part;  // [32 cg][blocks of 64][2 nh][6 mb*3+nb][4 a][64 lane][4 r]  (l3l4_kernel above)
    const float *b4;      // [192]
    const float *w5f;     // [4][12][6][64][4]  B fragments of L5_k: W5[k5][lq*48 + k4*4 + j][nb*16 + li]
    const float *b5;      // [4][96]
    const float *whf;     // [4][6][3][64][4]   B fragments of head k: Wh[lq*24 + k4*4 + j][nb*16 + li], 0-padded
    const float *bhf;     // [4][48]            head biases, 0-padded
    float *out;           // [n][90]
    int n_pad;
    int n;                // valid candidates
    float l4_scale;       // 2^-w4_shift: the partials are sums over the W4 image scaled by 2^w4_shift (engine.hip)
};

__global__ __launch_bounds__(256) void tail_kernel(TailArgs p) {
    __shared__ __attribute__((aligned(16))) float l4s[TAIL_TILE][L4S_ROW];
    __shared__ __attribute__((aligned(16))) float l5s[4][TAIL_TILE][L5S_ROW];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lq = lane >> 4;
    const int n0 = blockIdx.x * TAIL_TILE;

    // L4: fixed-order reduction of the split-K partials, bias, selu.  A thread takes accumulator quads of the producing kernel's
    // layout -- 16-byte loads, a wave instruction one contiguous KiB.  This tile is rows 16h .. 16h+15 of candidate block mb of the
    // 64-candidate block n0 / 64: accumulator quads a = 2h, 2h+1 (rows 8a + 4*(lane/32) + r), columns 96nh + 32nb + lane%32:
    // 12 (nh, nb, a) triples x 64 lanes = 768 quads over 256 threads.
    {
        const int blk = n0 / L34_CAND, mb = (n0 >> 5) & 1, h = (n0 >> 4) & 1, nblk = (p.n_pad + L34_CAND - 1) / L34_CAND;
        for (int f = tid; f < 12 * 64; f += 256) {
            const int g = f >> 6, ln = f & 63, nh = g / 6, rem = g - nh * 6, nb = rem >> 1, ai = rem & 1;
            const size_t at = ((((size_t)blk * 2 + nh) * 6 + mb * 3 + nb) * 4 + 2 * h + ai) * 256 + ln * 4;
            // all 32 partials of the quad in flight at once (one memory round trip per quad instead of four), summed in split order
            f32x4 part[L4_SPLITS];
#pragma unroll
            for (int sp = 0; sp < L4_SPLITS; ++sp) part[sp] = *(const f32x4 *)(p.l4part + (size_t)sp * nblk * (2 * 6 * 4 * 256) + at);
            f32x4 s = part[0];
#pragma unroll
            for (int sp = 1; sp < L4_SPLITS; ++sp) s += part[sp];
            const int col = nh * 96 + nb * 32 + (ln & 31);
            const float b4 = p.b4[col];
#pragma unroll
            for (int r = 0; r < 4; ++r) l4s[8 * ai + 4 * (ln >> 5) + r][col] = selu_f(s[r] * p.l4_scale + b4);
        }
    }
    __syncthreads();

    // L5 branch w: [16,192] x [192,96]
    f32x4 acc[6];
#pragma unroll
    for (int nb = 0; nb < 6; ++nb) acc[nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    {
        const f32x4 *wp = (const f32x4 *)p.w5f + (size_t)w * (12 * 6 * 64) + lane;
#pragma unroll 2
        for (int k4 = 0; k4 < 12; ++k4) {
            const f32x4 a = *(const f32x4 *)&l4s[li][lq * 48 + k4 * 4];
            f32x4 b[6];
#pragma unroll
            for (int nb = 0; nb < 6; ++nb) b[nb] = wp[(k4 * 6 + nb) * 64];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int nb = 0; nb < 6; ++nb) acc[nb] = mfma16(a[j], b[nb][j], acc[nb]);
        }
    }
#pragma unroll
    for (int nb = 0; nb < 6; ++nb) {
        const float bias = p.b5[w * L5_UNITS + nb * 16 + li];
#pragma unroll
        for (int r = 0; r < 4; ++r) l5s[w][lq * 4 + r][nb * 16 + li] = selu_f(acc[nb][r] + bias);
    }
    __syncthreads();

    // head w: [16,96] x [96,nh], selu on the logits (model.py:586), softmax over the nh classes
    f32x4 hacc[3];
#pragma unroll
    for (int nb = 0; nb < 3; ++nb) hacc[nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    {
        const f32x4 *hp = (const f32x4 *)p.whf + (size_t)w * (6 * 3 * 64) + lane;
#pragma unroll
        for (int k4 = 0; k4 < 6; ++k4) {
            const f32x4 a = *(const f32x4 *)&l5s[w][li][lq * 24 + k4 * 4];
            f32x4 b[3];
#pragma unroll
            for (int nb = 0; nb < 3; ++nb) b[nb] = hp[(k4 * 3 + nb) * 64];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int nb = 0; nb < 3; ++nb) hacc[nb] = mfma16(a[j], b[nb][j], hacc[nb]);
        }
    }
    const int nh = w == 0 ? 21 : (w == 1 ? 3 : 33);
    const int off = w == 0 ? 0 : (w == 1 ? 21 : (w == 2 ? 24 : 57));
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float v[3];
        float mx = -INFINITY;
#pragma unroll
        for (int nb = 0; nb < 3; ++nb) {
            const int col = nb * 16 + li;
            v[nb] = col < nh ? selu_f(hacc[nb][r] + p.bhf[w * 48 + col]) : -INFINITY;
            mx = fmaxf(mx, v[nb]);
        }
#pragma unroll
        for (int sh = 1; sh < 16; sh <<= 1) mx = fmaxf(mx, __shfl_xor(mx, sh));
        float sum = 0.0f;
#pragma unroll
        for (int nb = 0; nb < 3; ++nb) {
            v[nb] = (nb * 16 + li) < nh ? __expf(v[nb] - mx) : 0.0f;
            sum += v[nb];
        }
#pragma unroll
        for (int sh = 1; sh < 16; sh <<= 1) sum += __shfl_xor(sum, sh);
        const int row = n0 + lq * 4 + r;
        if (row < p.n) {
            float *o = p.out + (size_t)row * OUT_FLOATS + off;
#pragma unroll
            for (int nb = 0; nb < 3; ++nb)
                if (nb * 16 + li < nh) o[nb * 16 + li] = v[nb] / sum;  // true division, as tf.nn.softmax
        }
    }
}

}  // namespace clair
