// Recurrent half of one BiLSTM layer (both directions), fp32-exact on MFMA.
//
// Reference semantics: CudnnCompatibleLSTMCell(128) under stack_bidirectional_dynamic_rnn
// (clair/model.py:299-312, 423-451): per step z = [x_t, h_{t-1}].W + b, gates (i, c~, f, o),
// c_t = sig(f) c_{t-1} + sig(i) tanh(c~), h_t = sig(o) tanh(c_t); the backward direction walks
// t = 32..0; zero initial state.  The x-part (x_t.Wx + b) arrives precomputed from the
// projection GEMM in fragment-major order (gemm.hip.h: zx_block_offset), so each step only
// adds h_{t-1}.Wh, a [16,128]x[128,512] product per 16-candidate tile.
//
// Mapping to the CU: the fp32 Wh of one direction (256 KiB) is larger than the LDS, so it lives
// in registers, split over the waves of one workgroup by hidden unit; a wave owns all four gates
// of its units, which makes the gate non-linearities lane-local in the MFMA C layout; c_t stays
// in registers; h_t is exchanged through LDS.  K is visited in the order k = q*32 + kk
// (q = lane>>4) so that a lane's A operands for 4 consecutive MFMAs are one ds_read_b128.
//
// Gate pre-scaling: the host packs every gate column of Wx, Wh and the bias multiplied by the
// constant its activation needs in front of v_exp_f32 (2^x): -log2(e) for the sigmoid gates i, f, o
// and 2*log2(e) for the tanh gate c~ (engine.hip: gate_scale).  The MFMA result is then directly
// the exp2 argument, and the cell state is carried as c' = 2*log2(e)*c for the same reason.  This
// removes 6 of ~27 VALU instructions per gate element -- the budget that matters, see below.
#pragma once
#include "common.hip.h"

namespace clair {

constexpr int H_LDS_ROW = HID + 4;  // 132 floats: rows 16 B apart in bank space -> conflict-free b128 reads

// ---- the kernel ---------------------------------------------------------------------------------
//
// What the hardware allows (measured, profiles/r01_microbench.txt): an fp32 MFMA occupies its SIMD for
// 32 cycles and nothing hides under it -- VALU/transcendental instructions of the SAME wave add their
// issue time (~4 / ~13 cycles) to the stream, and a SECOND wave on the SIMD is starved outright while
// the first streams MFMAs (0.01 instructions per MFMA, whatever its s_setprio).  A two-waves-per-SIMD
// variant that alternated MFMA and gate slots between the waves was built, verified and measured
// slower (290 us vs 161 us per layer for batch 1024) for exactly that reason.  So the recipe is: one
// wave per SIMD, as few non-MFMA instructions as possible, and no exposed memory latency:
//   * one 256-thread workgroup = one 16-candidate tile of one direction; wave w owns hidden units
//     32w..32w+31 of all four gates: a [128 x 128] slice of Wh, 256 registers;
//   * the step's x-projection fragments arrive by LDS-DMA (global_load_lds, no VGPRs) one whole step
//     ahead and seed the accumulators with a ds_read_b128 each;
//   * h_t goes to a double-buffered LDS tile (one barrier per step) and leaves for HBM from there as
//     whole 512-byte rows at the start of the NEXT step -- all VMEM of a step is issued before its
//     MFMAs, so the only vmcnt wait (before the DMA'd data is read, a step later) never stalls;
//   * gates: 21 VALU/transcendental instructions per element thanks to the pre-scaled columns.
struct LstmArgs {
    const float *zx;   // fragment-major x-projection [2][33][ntiles][4][8][64][4]  (gemm.hip.h), gate-scaled
    const float *whp;  // packed recurrent weights [2][4][8][8][64][4]  (dir, wave, nb, kk/4, lane, kk%4), gate-scaled
    float *aout;       // [33][n_pad][256]  (fw -> cols 0..127, bw -> 128..255)
    int n_pad;
    int ntiles;
};

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void lstm_rec_kernel(LstmArgs p) {
    __shared__ __attribute__((aligned(16))) float hbuf[2][16][H_LDS_ROW];   // [step parity]
    __shared__ __attribute__((aligned(16))) float zlds[2][4][8][256];       // [step parity][wave][nb][lane*4 + r]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lq = lane >> 4;
    const int d = blockIdx.x & 1;
    const int tile = blockIdx.x >> 1;

    // resident weights: Bw[nb][kk] = Wh[k = lq*32 + kk][col = g*128 + 32w + 16hh + li], nb = g*2 + hh
    float Bw[8][32];
    {
        const f32x4 *wp = (const f32x4 *)p.whp + (size_t)(d * 4 + w) * (8 * 8 * 64) + lane;
#pragma unroll
        for (int nb = 0; nb < 8; ++nb)
#pragma unroll
            for (int k4 = 0; k4 < 8; ++k4) {
                const f32x4 x = wp[(nb * 8 + k4) * 64];
                Bw[nb][k4 * 4 + 0] = x[0];
                Bw[nb][k4 * 4 + 1] = x[1];
                Bw[nb][k4 * 4 + 2] = x[2];
                Bw[nb][k4 * 4 + 3] = x[3];
            }
    }
    float cst[8];   // c' = 2 log2(e) c  for elements e = hh*4 + r: (row 4*lq + r, unit 32w + 16hh + li)
#pragma unroll
    for (int e = 0; e < 8; ++e) cst[e] = 0.0f;

    auto fetch_zx = [&](int s) {   // LDS-DMA of this wave's 8 fragments of step s into zlds[s&1][w]
        const int t = d ? T_POS - 1 - s : s;
        const f32x4 *src = (const f32x4 *)(p.zx + ((((size_t)(d * T_POS + t) * p.ntiles + tile) * 4 + w) * 8) * 256) + lane;
        const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lptr_t)&zlds[s & 1][w][0][0]);
#pragma unroll
        for (int nb = 0; nb < 8; ++nb) glds16(src + nb * 64, lds0 + nb * 1024);
    };
    auto store_h = [&](int s) {    // h_s (complete in LDS) -> aout[t][tile rows][d*128 ..], whole 512-byte rows
        const int t = d ? T_POS - 1 - s : s;
        float *base = p.aout + ((size_t)t * p.n_pad + (size_t)tile * 16) * (2 * HID) + d * HID;
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
            const int f = h2 * 256 + tid, row = f >> 5, c4 = f & 31;
            *(f32x4 *)(base + (size_t)row * (2 * HID) + c4 * 4) = *(const f32x4 *)&hbuf[s & 1][row][c4 * 4];
        }
    };

    fetch_zx(0);
    for (int s = 0; s < T_POS; ++s) {
        // everything older than this step's VMEM has landed: z(s) (issued a whole step ago) in particular
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // LDS reads first (they gate the MFMAs), then this step's VMEM issue under their latency
        f32x4 acc[8], afr[8];
#pragma unroll
        for (int nb = 0; nb < 8; ++nb) acc[nb] = *(const f32x4 *)&zlds[s & 1][w][nb][lane * 4];
        if (s > 0) {
            const float *hrow = &hbuf[(s - 1) & 1][li][lq * 32];
#pragma unroll
            for (int k4 = 0; k4 < 8; ++k4) afr[k4] = *(const f32x4 *)(hrow + k4 * 4);
        }
        if (s + 1 < T_POS) fetch_zx(s + 1);
        if (s > 0) {
            store_h(s - 1);
#pragma unroll
            for (int k4 = 0; k4 < 8; ++k4)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int nb = 0; nb < 8; ++nb) acc[nb] = mfma16(afr[k4][j], Bw[nb][k4 * 4 + j], acc[nb]);
        }
        // gates; acc holds exp2 arguments (pre-scaled columns): e^-i, e^2g, e^-f, e^-o
        constexpr float K2 = 2.0f * 1.44269504088896340736f;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float ri = fast_rcp(1.0f + __builtin_amdgcn_exp2f(acc[0 + hh][r]));   // sig(i)
                const float rg = fast_rcp(1.0f + __builtin_amdgcn_exp2f(acc[2 + hh][r]));   // 1/(1+e^2g)
                const float rf = fast_rcp(1.0f + __builtin_amdgcn_exp2f(acc[4 + hh][r]));   // sig(f)
                const float ro = fast_rcp(1.0f + __builtin_amdgcn_exp2f(acc[6 + hh][r]));   // sig(o)
                const float kg = fmaf(rg, -2.0f * K2, K2);                                    // K2 tanh(g)
                const float c = fmaf(rf, cst[hh * 4 + r], ri * kg);                           // c' = sig(f) c' + K2 sig(i) tanh(g)
                cst[hh * 4 + r] = c;
                const float rc = fast_rcp(1.0f + __builtin_amdgcn_exp2f(c));                 // 1/(1+e^2c)
                hbuf[s & 1][lq * 4 + r][w * 32 + hh * 16 + li] = fmaf(rc, -2.0f * ro, ro);   // sig(o) tanh(c)
            }
        __syncthreads();
    }
    store_h(T_POS - 1);
}

// ---- LSTM1 with the input projection fused in ------------------------------------------------------
// Layer 1's x-part has K = 32 only: as a separate GEMM it is bound by writing (and re-reading) the
// 135 KB/candidate x-projection, not by its 14 us of MFMA work.  Here the recurrent wave also holds its
// [32 x 128] slice of Wx (64 more registers), reads x_t straight from the caller's [n][33][32] tensor
// (one 128-byte line per candidate and position, fetched a step ahead) and runs 64 extra MFMAs per
// step; the accumulators start from the (gate-scaled) bias.  No zx buffer, no DMA, one kernel less.
struct Lstm1Args {
    const float *x;     // [n_pad][33][32]  (rows >= n are zero)
    const float *wxp;   // packed x-part  [2][4][8][2][64][4]  (dir, wave, nb, kk/4, lane, kk%4): Wx[lq*8 + kk][col], gate-scaled
    const float *whp;   // packed h-part  [2][4][8][8][64][4]  as LstmArgs::whp
    const float *bias;  // [2][512] gate-scaled
    unsigned short *aout3;  // [2][33][n_pad][256] fp16: the output as its 2-way fp16 split, the form the
                            // LSTM2 projection GEMM consumes (gemm_split.hip.h); nothing else reads layer 1's output
    int n_pad;
    int ntiles;
};

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void lstm1_fused_kernel(Lstm1Args p) {
    __shared__ __attribute__((aligned(16))) float hbuf[2][16][H_LDS_ROW];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lq = lane >> 4;
    const int d = blockIdx.x & 1;
    const int tile = blockIdx.x >> 1;

    float Bw[8][32], Bx[8][8], bv[8];
    {
        const f32x4 *wp = (const f32x4 *)p.whp + (size_t)(d * 4 + w) * (8 * 8 * 64) + lane;
        const f32x4 *xp = (const f32x4 *)p.wxp + (size_t)(d * 4 + w) * (8 * 2 * 64) + lane;
#pragma unroll
        for (int nb = 0; nb < 8; ++nb) {
#pragma unroll
            for (int k4 = 0; k4 < 8; ++k4) {
                const f32x4 v = wp[(nb * 8 + k4) * 64];
                Bw[nb][k4 * 4 + 0] = v[0]; Bw[nb][k4 * 4 + 1] = v[1]; Bw[nb][k4 * 4 + 2] = v[2]; Bw[nb][k4 * 4 + 3] = v[3];
            }
#pragma unroll
            for (int k4 = 0; k4 < 2; ++k4) {
                const f32x4 v = xp[(nb * 2 + k4) * 64];
                Bx[nb][k4 * 4 + 0] = v[0]; Bx[nb][k4 * 4 + 1] = v[1]; Bx[nb][k4 * 4 + 2] = v[2]; Bx[nb][k4 * 4 + 3] = v[3];
            }
            bv[nb] = p.bias[d * GATES + (nb >> 1) * HID + 32 * w + 16 * (nb & 1) + li];
        }
    }
    float cst[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) cst[e] = 0.0f;

    // this lane's 8 features (lq*8 .. lq*8+7) of candidate li at position t(s)
    const float *xrow = p.x + ((size_t)tile * 16 + li) * (T_POS * F_IN) + lq * 8;
    auto load_x = [&](f32x4 (&xf)[2], int s) {
        const int t = d ? T_POS - 1 - s : s;
        xf[0] = *(const f32x4 *)(xrow + t * F_IN);
        xf[1] = *(const f32x4 *)(xrow + t * F_IN + 4);
    };
    const size_t plane = (size_t)T_POS * p.n_pad * (2 * HID);
    auto store_h = [&](int s) {   // h_s -> two fp16 planes, 8 bytes (4 units) per thread and plane, 256-byte row segments
        const int t = d ? T_POS - 1 - s : s;
        unsigned short *base = p.aout3 + ((size_t)t * p.n_pad + (size_t)tile * 16) * (2 * HID) + d * HID;
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
            const int f = h2 * 256 + tid, row = f >> 5, c4 = f & 31;
            const f32x4 h = *(const f32x4 *)&hbuf[s & 1][row][c4 * 4];
            f16x4 hi, lo;
#pragma unroll
            for (int j = 0; j < 4; ++j) { _Float16 a, b; split2(h[j], a, b); hi[j] = a; lo[j] = b; }
            *(f16x4 *)(base + (size_t)row * (2 * HID) + c4 * 4) = hi;
            *(f16x4 *)(base + plane + (size_t)row * (2 * HID) + c4 * 4) = lo;
        }
    };

    f32x4 xcur[2], xnext[2];
    load_x(xcur, 0);
    for (int s = 0; s < T_POS; ++s) {
        f32x4 acc[8], afr[8];
        if (s > 0) {   // A fragments of h_{s-1} first: their LDS latency hides under the 64 x-part MFMAs
            const float *hrow = &hbuf[(s - 1) & 1][li][lq * 32];
#pragma unroll
            for (int k4 = 0; k4 < 8; ++k4) afr[k4] = *(const f32x4 *)(hrow + k4 * 4);
        }
        load_x(xnext, s + 1 < T_POS ? s + 1 : s);
#pragma unroll
        for (int nb = 0; nb < 8; ++nb) acc[nb] = (f32x4){bv[nb], bv[nb], bv[nb], bv[nb]};
#pragma unroll
        for (int k4 = 0; k4 < 2; ++k4)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int nb = 0; nb < 8; ++nb) acc[nb] = mfma16(xcur[k4][j], Bx[nb][k4 * 4 + j], acc[nb]);
        if (s > 0) {
            store_h(s - 1);
#pragma unroll
            for (int k4 = 0; k4 < 8; ++k4)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int nb = 0; nb < 8; ++nb) acc[nb] = mfma16(afr[k4][j], Bw[nb][k4 * 4 + j], acc[nb]);
        }
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
                hbuf[s & 1][lq * 4 + r][w * 32 + hh * 16 + li] = fmaf(rc, -2.0f * ro, ro);
            }
        xcur[0] = xnext[0];
        xcur[1] = xnext[1];
        __syncthreads();
    }
    store_h(T_POS - 1);
}

// ====================================================================================================
// Split-precision recurrent kernels: h_{t-1}.Wh (and x_t.Wx in layer 1) on the fp16 matrix cores
// ====================================================================================================
// Same structure as above (one workgroup = one 16-candidate tile of one direction, wave w owns hidden units
// 32w..32w+31 of all four gates, weights resident in registers, h through double-buffered LDS, one barrier per
// step), but every product runs as the 2-way fp16 split of common.hip.h:  a*b ~= a1*b1 + a1*b2 + a2*b1  on
// v_mfma_f32_16x16x32_f16.  Wh as two fp16 planes occupies exactly the 256 registers its fp32 form did, and
// a step costs 8 blocks x 4 k-steps x 3 terms = 96 MFMAs of ~17 cycles instead of 256 of 32: the matrix pipe
// drops from ~80 % to ~40 % of the step, the rest being the gate math and the LDS round trip of h.
// h lives in LDS as its two fp16 planes (the gate code splits it once, 4 VALU instructions per element);
// layer 1 copies those planes to HBM unchanged -- they are the A operand of the LSTM2 projection GEMM --
// layer 2 hands the fp32 sum p1 + p2 (exactly the h its own recurrence used) to the L3/L4 kernel.
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
