// BiLSTM layer 2 with TWO 32-candidate tiles per workgroup (same direction), their steps alternating.
//
// lstm32_kernel (lstm32.hip.h) hides the gate math of block b-1 behind the MFMAs of block b, which leaves a quarter of every step
// exposed: the last block's gates (~100 instructions with no MFMA to hide behind), the step barrier and the 16 fragment reads of
// h_s.  With two tiles A and B per workgroup the step of A is followed by the step of B:
//     step of X, block 0 : MFMAs of X   | in their shadows: the LAST block's gates of the OTHER tile's previous step
//     barrier            : the other tile's h is complete (mid-stream: nobody waits long, every wave has MFMAs queued)
//     blocks 1, 2, 3     : MFMAs of X   | gates of X's blocks 0, 1, 2; the copy-out of X's h_{s-1}; and in block 3, each k-step's
//                                         fragment registers are refilled with the OTHER tile's h right after their last use
// so nothing but MFMAs and their shadows is left.  The resident weights (256 AGPRs) serve both tiles; one set of seed registers
// is enough (a block's seeds are refilled, for the other tile's next step, right after the block consumed them); the second
// tile costs 16 VGPRs of cell state and 35 KB of LDS.  Reference semantics as lstm32.hip.h (clair/model.py:299-312, 443-450).
#pragma once
#include "lstm32.hip.h"

namespace clair {

struct Lstm32PairArgs {
    const float *zx;            // [2 dir][n_pad/32][33][4 wave][4 b][4 a][64 lane][4 c]  x-projection, bias included (gemm_split.hip.h)
    const unsigned short *whs;  // [2 dir][4 wave][4 b][8 kk][2 plane][64 lane][8] fp16  A fragments of Wh^T, gate-scaled
    float *aout;                // [32 groups of 8 features][33][n_pad][8] fp32
    int n_pad;
    int ntiles;                 // n_pad / 32; workgroup id = 2 * pair + direction, tiles 2*pair and min(2*pair + 1, ntiles - 1)
};

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void lstm32_pair_kernel(Lstm32PairArgs p) {
    __shared__ __attribute__((aligned(16))) unsigned char lds_raw[2 * L32_HBUF_BYTES];
    _Float16 (*hbuf)[2][2][L32_TILE][HP_ROW] = (_Float16 (*)[2][2][L32_TILE][HP_ROW])lds_raw;   // [tile][step parity][plane][cand][unit]

    int tid = threadIdx.x;      // not const: re-defined (to itself) in front of the epilogue, see there
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cand = lane & 31, hq = lane >> 5;
    const int d = blockIdx.x & 1;
    const int tile_of[2] = {2 * (int)(blockIdx.x >> 1), min(2 * (int)(blockIdx.x >> 1) + 1, p.ntiles - 1)};

    // resident weights: Aw[b][kk][plane] = 8 fp16 of gate row (b, lane%32), k = 16*kk + 8*(lane/32) + j
    f16x8 Aw[4][8][2];
    {
        const f16x8 *wp = (const f16x8 *)p.whs + (size_t)(d * 4 + w) * (4 * 8 * 2 * 64) + lane;
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int kk = 0; kk < 8; ++kk)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) Aw[b][kk][pl] = wp[((b * 8 + kk) * 2 + pl) * 64];
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int kk = 0; kk < 8; ++kk)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) asm volatile("" : "+a"(Aw[b][kk][pl]));   // AGPR-resident (lstm32.hip.h)
    }
    float cst[2][4][4];   // [tile][block b][element a]: c' = 2 log2(e) c of unit 32w + 8b + 4h' + a of candidate lane%32
#pragma unroll
    for (int tl = 0; tl < 2; ++tl)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int a = 0; a < 4; ++a) cst[tl][b][a] = 0.0f;

    const float *zx0[2];
#pragma unroll
    for (int tl = 0; tl < 2; ++tl) zx0[tl] = p.zx + ((((size_t)d * p.ntiles + tile_of[tl]) * T_POS * 4 + w) * 4) * 1024 + lane * 4;
    auto load_seed = [&](f32x16 &z, int tl, int s, int b) {
        const int sc = s < T_POS ? s : T_POS - 1;   // the prefetch past the last step re-reads the last one
        const int t = d ? T_POS - 1 - sc : sc;
        const float *src = zx0[tl] + ((size_t)t * 16 + b) * 1024;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const f32x4 v = __builtin_nontemporal_load((const f32x4 *)(src + a * 256));
            z[4 * a + 0] = v[0]; z[4 * a + 1] = v[1]; z[4 * a + 2] = v[2]; z[4 * a + 3] = v[3];
        }
    };
    // h (both planes complete in LDS) -> HBM as fp32, piece j = 16-byte chunk g = 256 j + tid of [16 groups][32 candidates][2 halves of 4 units],
    //   in two halves of two pieces each (registers: the whole tile at once would spill)
    f16x4 c4[2][2];
    f32x4 co[2];
    auto copy_read = [&](int tl, int s, int half) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int g = (half * 2 + j) * 256 + tid;
            c4[j][0] = *(const f16x4 *)&hbuf[tl][s & 1][0][(g & 63) >> 1][(g >> 6) * 8 + (g & 1) * 4];   // chunk g: group g >> 6, candidate (g & 63) >> 1, half g & 1
            c4[j][1] = *(const f16x4 *)&hbuf[tl][s & 1][1][(g & 63) >> 1][(g >> 6) * 8 + (g & 1) * 4];
        }
    };
    auto copy_cvt = [&](int i) {   // micro-step i = 0..3 of a half: two units each
        const int j = i >> 1, q = (i & 1) * 2;
        co[j][q] = (float)c4[j][0][q] + (float)c4[j][1][q];
        co[j][q + 1] = (float)c4[j][0][q + 1] + (float)c4[j][1][q + 1];
    };
    auto copy_write = [&](int tl, int s, int half, int j) {
        const int t = d ? T_POS - 1 - s : s;
        const int g = (half * 2 + j) * 256 + tid;
        __builtin_nontemporal_store(co[j], (f32x4 *)(p.aout + ((((size_t)(d * 16 + (g >> 6)) * T_POS + t) * p.n_pad + (size_t)tile_of[tl] * L32_TILE) * 8) + (g & 63) * 4));   // group-major, non-temporal (lstm32.hip.h)
    };

    f32x16 acc[2];        // block b accumulates in acc[b & 1]
    f32x16 zq[4];         // seeds: [b] = block b of the next step (of whichever tile comes next) it is needed in
    f16x8 hf[8][2];       // B fragments of the current tile's h_{s-1}: [kk][plane]
    unsigned hp[2], lp[2];
    float eg[4], ei[4], ef[4], eo[4], tt[4], ng[4], hh[4];

    // h_{-1} = 0 for both tiles
    for (int i = tid; i < 2 * L32_TILE * HP_ROW / 8; i += 256) {
        ((f32x4 *)&hbuf[0][1][0][0][0])[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        ((f32x4 *)&hbuf[1][1][0][0][0])[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int b = 0; b < 4; ++b) load_seed(zq[b], 0, 0, b);
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 8; ++kk)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) hf[kk][pl] = *(const f16x8 *)&hbuf[0][1][pl][cand][kk * 16 + hq * 8];
    // the very first block 0 runs "the other tile's last block of step -1": harmless garbage into a buffer that is rewritten before it
    // is read, but its accumulator must at least be finite
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[1][i] = 0.0f;
    asm volatile("" : "+v"(acc[1]));

    // gap G of the shared gate schedule (lstm32.hip.h: CLAIR_GATE_GAP) for block PB of tile TL, whose h goes to step parity PAR of that tile
#define P2_GAP(G, TL, PB, PAR)                                                                                    \
    {                                                                                                             \
        const f32x16 &Z = acc[(PB) & 1];                                                                          \
        float (&C_)[4] = cst[TL][PB];                                                                             \
        CLAIR_GATE_GAP(G, *(uint2 *)&hbuf[TL][PAR][0][cand][w * 32 + (PB) * 8 + hq * 4] = make_uint2(hp[0], hp[1]);  \
                          *(uint2 *)&hbuf[TL][PAR][1][cand][w * 32 + (PB) * 8 + hq * 4] = make_uint2(lp[0], lp[1]);) \
    }
    // One block of tile X's step s (X, Y = 1 - X compile-time): 24 MFMAs (k-step kk: w_lo.h_hi, w_hi.h_lo, w_hi.h_hi) and what rides in
    // their shadows.  sy = the step the other tile runs next (its seeds are fetched here, its fragments read in block 3);
    // ypar = parity of the step whose last block of Y is finished in block 0.
#define P2_BLOCK(X, Y, b)                                                                                         \
    {                                                                                                             \
        const f32x16 zold = zq[b];                                                                                \
_Pragma("unroll")                                                                                                 \
        for (int m = 0; m < 24; ++m) {                                                                            \
            const int kk = m / 3, term = m % 3;                                                                   \
            if (m == 0) mfma32_av_first(acc[(b) & 1], Aw[b][kk][1], hf[kk][0], zold);                             \
            else mfma32_av(acc[(b) & 1], Aw[b][kk][term == 0 ? 1 : 0], hf[kk][term == 1 ? 1 : 0]);                \
            __builtin_amdgcn_sched_barrier(0);                                                                    \
            CLAIR_DBG_FENCE();                                                                                    \
            if (m == 3) load_seed(zq[b], Y, sy, b);   /* after the keep-alive below: the refill can land in the registers it replaces */ \
            if ((b) == 0 && m >= 1) P2_GAP(m, Y, 3, ypar)                                                         \
            if ((b) > 0 && m >= 1) P2_GAP(m, X, (b) - 1, s & 1)                                                   \
            if (m == 2) asm volatile("" :: "v"(zold));   /* the first MFMA's C registers stay untouched until here */ \
            if ((b) == 1 || (b) == 2) {   /* copy-out of X's h_{s-1}, one half per block (at s = 0: of the still uninitialised buffer, rewritten a step later) */ \
                if (m == 1) copy_read(X, s_prev, (b) - 1);                                                        \
                if (m >= 6 && m < 14 && (m & 1) == 0) copy_cvt((m - 6) >> 1);                                     \
                if (m == 16 || m == 20) copy_write(X, s_prev, (b) - 1, (m - 16) >> 2);                            \
            }                                                                                                     \
            if ((b) == 3 && term == 2) {   /* k-step kk is done for this step: its registers take the other tile's h */ \
                hf[kk][0] = *(const f16x8 *)&hbuf[Y][ypar_next][0][cand][kk * 16 + hq * 8];                       \
                hf[kk][1] = *(const f16x8 *)&hbuf[Y][ypar_next][1][cand][kk * 16 + hq * 8];                       \
            }                                                                                                     \
            __builtin_amdgcn_sched_barrier(0);                                                                    \
        }                                                                                                         \
    }
#define P2_HALF(X, Y)                                                                                             \
    {                                                                                                             \
        P2_BLOCK(X, Y, 0)                                                                                         \
        __syncthreads();   /* the other tile's h (step parity ypar) is complete; every wave has the next block queued */ \
        P2_BLOCK(X, Y, 1)                                                                                         \
        P2_BLOCK(X, Y, 2)                                                                                         \
        P2_BLOCK(X, Y, 3)                                                                                         \
    }

    for (int s = 0; s < T_POS; ++s) {
        const int s_prev = s > 0 ? s - 1 : 0;
        {   // tile 0, step s; the other tile (1) is at step s-1 (its last block finishes here) and runs step s next
            const int sy = s, ypar = s > 0 ? (s - 1) & 1 : 0, ypar_next = (s - 1) & 1;   // s = 0: the bogus gates write the buffer step 0 rewrites; the fragments come from the zeroed h_{-1} (parity 1)
            P2_HALF(0, 1)
            if (s == 0) {   // forget what the bogus "step -1" gates did to tile 1's last block
#pragma unroll
                for (int a = 0; a < 4; ++a) cst[1][3][a] = 0.0f;
            }
        }
        {   // tile 1, step s; the other tile (0) finished step s except for its last block and runs step s+1 next
            const int sy = s + 1, ypar = s & 1, ypar_next = s & 1;
            P2_HALF(1, 0)
        }
    }
    // tile 1's last block of the last step has no MFMAs left to hide behind (12 wait states after its last MFMA)
    {
        const int s = T_POS - 1;
        asm volatile("s_nop 11" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 1; g <= 23; ++g) P2_GAP(g, 1, 3, s & 1)
    }
    __syncthreads();
    // The epilogue's per-thread offsets (chunk g = 256 j + tid and what follows from it) are pure functions of tid: hipcc computed them in
    // the prologue and carried them across the step loop, where every one of the 256 VGPRs is taken -- three of them went to scratch
    // (16 B per lane, the only scratch of the whole pass; profiles/r05_kernel_resource_usage.txt).  An opaque re-definition of tid here
    // makes them values of the epilogue.
    asm volatile("" : "+v"(tid));
#pragma unroll
    for (int tl = 0; tl < 2; ++tl)
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            copy_read(tl, T_POS - 1, half);
#pragma unroll
            for (int i = 0; i < 4; ++i) copy_cvt(i);
#pragma unroll
            for (int j = 0; j < 2; ++j) copy_write(tl, T_POS - 1, half, j);
        }
#undef P2_HALF
#undef P2_BLOCK
#undef P2_GAP
}

}  // namespace clair
