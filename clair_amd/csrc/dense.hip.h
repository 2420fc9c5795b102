// Position-mixing "slice dense" layer (L3) and the classifier tail (L4 reduce, L5 x4, heads, softmax).
#pragma once
#include "common.hip.h"

namespace clair {

// ---- L3 + L4 fused: slice dense (256 x dense 33->30, selu) feeding the split-K 7680->192 product --------
// clair/model.py:225-244 (slice_dense_layer), :464-479 (L3 + flatten, flat index u*256 + c), :482-488 (L4).
//   l3[n][u*256 + c] = selu( sum_t a2[t][n][c] * W3[c][t][u] + b3[c][u] )
//   part[cg][n][j]   = sum_{u<30, c in group cg} l3[n][u*256 + c] * W4[u*256 + c][j]        (cg = 16 channels)
// A workgroup owns 32 candidates x one channel group, i.e. the K-slice {u*256 + c}: for a fixed u its 16
// channels are 16 CONSECUTIVE rows of W4, exactly one slab of the packed [K/16][192][16] operand.  So L3 is
// computed on the MFMA ([32 cand x 36 t] x [36 t x 32 u] per channel, operands straight from global) into
// an LDS tile laid out as L4's A operand, and L4 runs out of that tile with its B fragments read directly
// from L2: the 30 KB/candidate l3 tensor never exists in HBM and a kernel launch disappears.
constexpr int L34_CAND = 32;
constexpr int L34_ROW = 30 * 16 + 8;    // fp16 units per candidate row of one l3 plane: 976 B, 16-B aligned, rows 52 banks apart
constexpr int L34_LDS_BYTES = 33 * 32 * 16 * 4;   // the a2 staging tile (67 584 B) is the larger of the buffer's two lives
// l3 is multiplied by 2^4 before its 2-way fp16 split and the L4 reduction by 2^-4 (folded into TailArgs::l4_scale): a selu output
// of 0.01 would otherwise have a subnormal low plane (3e-8 absolute = 3e-6 relative); 2^4 keeps 22 bits down to |y| ~ 0.008 and
// overflows only beyond |y| = 4 094.
constexpr float L34_ACT_SCALE = 16.0f;

struct L3L4Args {
    const float *a2;    // [33][n_pad][256]
    const float *w3f;   // [256][64][20]  B fragments of L3: slot kk*2 + nbk = W3[c][t = lq*9 + kk][u = nbk*16 + li] (0 beyond 33 / 30)
    const float *b3;    // [256][30]
    const unsigned short *w4s;   // [16 cg][15 ks][12 nb][2 plane][64 lane][8] fp16 split of W4: row u*256 + cg*16 + ch with
                                 // u = 2*ks + (lq>>1), ch = 8*(lq&1) + j; column nb*16 + li
    float *part;        // [16 cg][n_pad/32][4 wave][6 mb*3+nb][64 lane][4 r]  split-K partial sums, in the accumulator layout
    int n_pad;
    float *dbg;         // parity tap (NULL in production): l3 as this kernel holds it, hi + lo, [n_pad][7680]
};

__global__ __launch_bounds__(256) void l3l4_kernel(L3L4Args p) {
    // one LDS buffer, two lives: first the a2 tile [33 t][32 cand][16 ch] (67.6 KB, filled by LDS-DMA),
    // then -- after every wave has pulled its A fragments out of it -- the l3 tile that feeds L4
    __shared__ __attribute__((aligned(16))) float lds_buf[L34_LDS_BYTES / 4];
    _Float16 (*l3h)[L34_CAND][L34_ROW] = (_Float16 (*)[L34_CAND][L34_ROW])lds_buf;   // [plane][cand][u*16 + ch]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lq = lane >> 4;
    // XCD-aware order (workgroups go round-robin over the 8 XCDs by linear id): XCD x owns channel groups 2x and
    // 2x+1 for every candidate block, so (a) the two 64-byte halves of each 128-byte a2 line are fetched by
    // neighbouring workgroups of ONE L2, and (b) each L2 holds only its own 1/8 of the W4 fragments (740 KB)
    // instead of every L2 streaming all 5.9 MB.
    const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
    const int n0 = (seq >> 1) * L34_CAND;
    const int cg = xcd * 2 + (seq & 1);        // channels cg*16 .. cg*16+15; this wave: 4 of them, cg*16 + 4w ..

    // ---- L3 for this wave's four channels --------------------------------------------------------------
    f32x4 acc3[4][2][2];   // [channel][m-block][u-block]
    {
        // a2 tile -> LDS: row q = t*32 + cand is the 64 bytes a2[t][n0 + cand][cg*16 .. +15]; one DMA piece moves
        // 16 rows (lane l: row 16*piece + l/4, 16-byte chunk l%4), so every fetched half-line is fully used and
        // fetched once per workgroup (per-lane float4 loads of 4 channels touched each line from all four waves)
        // the first channel's W3 fragments do not depend on the tile: fetch them under the DMA wait (-1.5 us; the biases, hoisted
        // the same way, gained nothing)
        f32x4 bf0[5];
        {
            const f32x4 *bp0 = (const f32x4 *)(p.w3f + ((size_t)(cg * 16 + w * 4) * 64 + lane) * 20);
#pragma unroll
            for (int i = 0; i < 5; ++i) bf0[i] = bp0[i];
        }
        constexpr int NPIECE = T_POS * L34_CAND / 16;   // 66
        const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lptr_t)lds_buf);
        for (int piece = w; piece < NPIECE; piece += 4) {
            const int q = piece * 16 + (lane >> 2);
            const int t = q >> 5, cand = q & 31;
            const float *src = p.a2 + ((size_t)t * p.n_pad + n0 + cand) * 256 + cg * 16 + (lane & 3) * 4;
            glds16((const f32x4 *)src, lds0 + piece * 1024);
        }
        CLAIR_VMWAIT(0);
        __syncthreads();
        // A operands: candidate li (of block mb), positions t = lq*9 + kk, this wave's four channels per float4
        f32x4 av[2][9];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int kk = 0; kk < 9; ++kk) {
                const int t = lq * 9 + kk;
                av[mb][kk] = t < T_POS ? *(const f32x4 *)&lds_buf[((t * 32 + mb * 16 + li) * 16) + w * 4]
                                       : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
            const int c = cg * 16 + w * 4 + cc;
            f32x4 bf[5];
            const f32x4 *bp = (const f32x4 *)(p.w3f + ((size_t)c * 64 + lane) * 20);
#pragma unroll
            for (int i = 0; i < 5; ++i) bf[i] = cc == 0 ? bf0[i] : bp[i];
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int nbk = 0; nbk < 2; ++nbk) acc3[cc][mb][nbk] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < 9; ++kk)
#pragma unroll
                for (int nbk = 0; nbk < 2; ++nbk) {
                    const float b = bf[(kk * 2 + nbk) >> 2][(kk * 2 + nbk) & 3];
#pragma unroll
                    for (int mb = 0; mb < 2; ++mb) acc3[cc][mb][nbk] = mfma16(av[mb][kk][cc], b, acc3[cc][mb][nbk]);
                }
        }
        asm volatile("" : "+v"(acc3[3][1][1]));   // the MFMAs stay on this side of the barrier
        __syncthreads();                              // every wave is done reading the a2 tile: the buffer becomes l3s
        // bias + selu, 2-way fp16 split (the L4 product runs on the fp16 matrix cores), then one 8-byte LDS store
        // per plane and (candidate row, u): the wave's four channels together
#pragma unroll
        for (int nbk = 0; nbk < 2; ++nbk) {
            const int u = nbk * 16 + li;
            if (u < L3_UNITS) {
                float bias[4];
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) bias[cc] = p.b3[(cg * 16 + w * 4 + cc) * L3_UNITS + u];
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float y[4];
#pragma unroll
                        for (int cc = 0; cc < 4; ++cc) y[cc] = selu_scaled(acc3[cc][mb][nbk][r] + bias[cc], L34_ACT_SCALE);
                        uint2 hi, lo;
                        split2_pk4(y, hi, lo);
                        *(uint2 *)&l3h[0][mb * 16 + lq * 4 + r][u * 16 + w * 4] = hi;
                        *(uint2 *)&l3h[1][mb * 16 + lq * 4 + r][u * 16 + w * 4] = lo;
                    }
            }
        }
    }
    __syncthreads();
    if (p.dbg) {   // debug tap: this workgroup's 32 x (30 u x 16 channels) slice of l3
        for (int f = tid; f < L34_CAND * 480; f += 256) {
            const int row = f / 480, k = f - row * 480, u = k >> 4, ch = k & 15;
            p.dbg[(size_t)(n0 + row) * L3_OUT + u * 256 + cg * 16 + ch] = ((float)l3h[0][row][k] + (float)l3h[1][row][k]) * (1.0f / L34_ACT_SCALE);
        }
    }

    // ---- L4 over this K-slice: wave w owns output columns 48w .. 48w+47 (3 blocks), both 16-row blocks ----------
    // 2-way fp16 split product (common.hip.h): K = 480 = 15 k-steps of 32 = two u values x 16 channels each
    f32x4 acc[2][3];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int nb = 0; nb < 3; ++nb) acc[mb][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const f16x8 *bsrc = (const f16x8 *)p.w4s + ((size_t)cg * 15 * 12 + w * 3) * 2 * 64 + lane;   // + (ks*12 + nb)*2*64 + plane*64
    // B fragments stream from L2 with a prefetch distance of PF k-steps; the loop is fully unrolled so the ring is static
    constexpr int PF = 4, KS = 15;
    f16x8 bq[PF][3][2];
#pragma unroll
    for (int i = 0; i < PF - 1; ++i)
#pragma unroll
        for (int nb = 0; nb < 3; ++nb)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) bq[i][nb][pl] = bsrc[((size_t)i * 12 + nb) * 128 + pl * 64];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        if (ks + PF - 1 < KS) {
#pragma unroll
            for (int nb = 0; nb < 3; ++nb)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) bq[(ks + PF - 1) % PF][nb][pl] = bsrc[((size_t)(ks + PF - 1) * 12 + nb) * 128 + pl * 64];
        }
        f16x8 a[2][2];   // [m-block][plane]: row li, k-chunk lq -> u = 2*ks + (lq>>1), channels 8*(lq&1) ..
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) a[mb][pl] = *(const f16x8 *)&l3h[pl][mb * 16 + li][(2 * ks + (lq >> 1)) * 16 + (lq & 1) * 8];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int nb = 0; nb < 3; ++nb) acc[mb][nb] = mfma16h(a[mb][1], bq[ks % PF][nb][0], acc[mb][nb]);
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int nb = 0; nb < 3; ++nb) acc[mb][nb] = mfma16h(a[mb][0], bq[ks % PF][nb][1], acc[mb][nb]);
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int nb = 0; nb < 3; ++nb) acc[mb][nb] = mfma16h(a[mb][0], bq[ks % PF][nb][0], acc[mb][nb]);
    }
    // split-K partials, fragment-major: [cg][block of 32][wave][mb*3 + nb][lane][4 r] -- a lane's accumulator quad is 16 contiguous
    // bytes and a wave instruction one contiguous KiB (round 1: 24 dword stores per lane, four 64-byte segments each)
    const int block = n0 / L34_CAND, nblocks = p.n_pad / L34_CAND;
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int nb = 0; nb < 3; ++nb)
            *(f32x4 *)(p.part + ((((size_t)cg * nblocks + block) * 4 + w) * 6 + mb * 3 + nb) * 256 + lane * 4) = acc[mb][nb];
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
    const float *l4part;  // [16 cg][n_pad/32][4 wave][6 mb*3+nb][64 lane][4 r]  (l3l4_kernel above)
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
    // layout -- 16-byte loads, a wave instruction one contiguous KiB -- i.e. rows 4*lq' .. +3 (r) of column 48w' + 16nb + li':
    // this tile is half mb of candidate block n0 / 32; 12 (w', nb) pairs x 64 lanes = 768 quads over 256 threads.
    {
        const int blk = n0 / 32, mb = (n0 >> 4) & 1, nblocks = p.n_pad / 32;
        for (int f = tid; f < 12 * 64; f += 256) {
            const int wn = f >> 6, ln = f & 63, wq = wn / 3, nb = wn - wq * 3;
            const size_t at = (((size_t)blk * 4 + wq) * 6 + mb * 3 + nb) * 256 + ln * 4;
            f32x4 s = *(const f32x4 *)(p.l4part + at);
#pragma unroll
            for (int sp = 1; sp < L4_SPLITS; ++sp) s += *(const f32x4 *)(p.l4part + (size_t)sp * nblocks * (4 * 6 * 256) + at);
            const int col = wq * 48 + nb * 16 + (ln & 15);
            const float b4 = p.b4[col];
#pragma unroll
            for (int r = 0; r < 4; ++r) l4s[(ln >> 4) * 4 + r][col] = selu_f(s[r] * p.l4_scale + b4);
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
