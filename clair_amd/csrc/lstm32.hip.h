// The two BiLSTM layers (both directions each), recurrence kept on chip for all 33 steps.
//
// Reference semantics: CudnnCompatibleLSTMCell(128) under stack_bidirectional_dynamic_rnn
// (clair/model.py:299-312, 423-451): per step z = [x_t, h_{t-1}].W + b, gates (i, c~, f, o),
// c_t = sig(f) c_{t-1} + sig(i) tanh(c~), h_t = sig(o) tanh(c_t); the backward direction walks
// t = 32..0; zero initial state.
//
// Mapping to the CU.  One 256-thread workgroup (one wave per SIMD) owns one 32-candidate tile of one
// direction for all 33 steps.  Wh of a direction (128 x 512; 256 KiB as two fp16 planes) is larger than the
// 160 KiB LDS, so it lives in REGISTERS: wave w holds the slice for hidden units 32w..32w+31 of all four
// gates -- in the 256 accumulation VGPRs, which hold nothing else; all working state is in the 256
// architectural VGPRs.  c_t never leaves registers; h_t is exchanged between the four waves through a
// double-buffered LDS tile, one barrier per step.
//
// Products run as the 2-way fp16 split of common.hip.h  (a*b ~= a1*b1 + a1*b2 + a2*b1)  on
// v_mfma_f32_32x32x16_f16 with the WEIGHTS as the A operand (rows = gate rows) and the activations as B
// (columns = candidates), i.e. the transposed product z^T = W^T h^T.  Measured facts that shape the kernel
// (tools/ubench/mfma_filler.hip, mfma_regfile.hip, lstm32_probe.hip; profiles/r01_microbench.txt):
//   * a wave hides about four VALU/transcendental instructions (at most three transcendental) in the
//     32-cycle shadow of every v_mfma_f32_32x32x16_f16 it issues -- the 16x16x32 form hides 1-2 in 17 cycles,
//     the fp32 MFMA none -- provided none of them waits on a result younger than the previous MFMA;
//   * a single dependent accumulator chain of that MFMA still issues every 32 cycles, wherever A, B and the
//     accumulator live (VGPR or AGPR);
//   * an LDS-DMA piece costs 100+ issue cycles inside such a stream, a plain global_load_dwordx4 does not.
// So the 128 gate rows of a wave are cut into four blocks of 32 rows = 8 hidden units x 4 gates, ordered so
// that one lane's 16 accumulator registers of a block are the four gates of four units of ONE candidate
// (row 8a + 4h' + c: element a = reg/4, gate c = reg%4, h' = lane/32, unit 8b + 4h' + a).  The gate
// non-linearities of block b-1 (about 100 VALU instructions per lane) are threaded by hand through the 24
// MFMAs of block b as a static schedule (L32_GAP below): three blocks out of four cost no time.
//
//   * h lives in LDS as its two fp16 planes (the gate code splits it once); layer 1 copies those planes to
//     HBM unchanged -- they are the B operand of the LSTM2 projection GEMM (gemm_split.hip.h) -- layer 2 hands
//     the fp32 sum p1 + p2, exactly the h its own recurrence used, to the L3/L4 kernel.  Copy-out happens
//     from LDS as whole rows inside the MFMA stream of the NEXT step's first block.
//   * layer 1 computes its input projection itself (K = 32: two k-steps of 16; x_t read straight from the
//     caller's [n][33][32] tensor one step ahead, split on the fly; Wx1 fragments and bias quads in LDS).
//   * layer 2 gets its x-projection (written by gemm_split.hip.h in exactly this kernel's accumulator
//     layout, bias included) straight into registers one whole step ahead: the 16 values of a block are the
//     C operand of that block's first MFMA.
//
// Gate pre-scaling: the host multiplies every gate row of Wx, Wh and the bias by the constant its
// activation needs in front of v_exp_f32 (2^x): -log2(e) for the sigmoid gates i, f, o and 2*log2(e) for the
// tanh gate c~ (engine.hip: gate_scale), and the cell state is carried as c' = 2*log2(e)*c.
#pragma once
#include "common.hip.h"

namespace clair {

constexpr int L32_TILE = 32;      // candidates per workgroup
// Layer 1's input is raw pileup counts (0..250 by CreateTensor's depth cap, 32767 at the int16 boundary) against weights
// of ~0.1 and below: the fp16 low plane of such a weight is subnormal (3e-8 ABSOLUTE, common.hip.h), and 250 x 3e-8 per term
// is 10-100x the float32 rounding of the same product -- measured as 1e-5 on the layer output for the 300x Illumina
// profile and 1e-4 on the probabilities once the recurrence amplifies it (tools/parity_sweep.py).  So the x-part runs as
// (x 2^-S)(Wx 2^S): the weight image keeps its 22 bits down to |w| ~ 2e-4, integer counts up to 32767 stay EXACT in the two
// planes (hi: 11 bits, lo: the remainder, a multiple of 2^-S >= 2^-14), and the product is unchanged.
constexpr int L32_X_SHIFT = 8;
constexpr int HP_ROW = HID + 8;   // fp16 units per LDS row of one h plane: 272 B, conflict-free ds_read_b128 over 32 rows
constexpr int L32_HBUF_BYTES = 2 * 2 * L32_TILE * HP_ROW * 2;                 // 34 816
constexpr int L32_XT_ROW = F_IN + 8;   // fp16 per row of one plane of the staged input tile (80 B pitch)
constexpr int L32_LDS_FIRST = L32_HBUF_BYTES + 4 * 4 * 2 * 2 * 64 * 16 + 4 * 4 * 4 * 2 * 16 + 2 * 2 * L32_TILE * L32_XT_ROW * 2;   // + Wx1 fragments (64 KiB) + bias quads (2 KiB) + two input tiles of two planes (10 KiB)
constexpr int L32_LDS_SECOND = L32_HBUF_BYTES;

struct Lstm32Args {
    const float *x;             // FIRST: [n_pad][33][32] network input
    const unsigned short *wxs;  // FIRST: [2 dir][4 wave][4 b][2 kk][2 plane][64 lane][8] fp16  A fragments of Wx1^T, gate-scaled
    const float *biasq;         // FIRST: [2 dir][4 wave][4 b][4 a][2 h'][4 c]  gate-scaled bias as accumulator quads
    const float *zx;            // !FIRST: [2 dir][n_pad/32][33][4 wave][4 b][4 a][64 lane][4 c]  x-projection, bias included;
                                // tile-major: a workgroup streams one contiguous 2.1 MB piece (with t outermost every step
                                // opened a new page for every CU: ~1100 cycles of translation latency per step, lstm32_probe)
    const unsigned short *whs;  // [2 dir][4 wave][4 b][8 kk][2 plane][64 lane][8] fp16  A fragments of Wh^T, gate-scaled
    unsigned short *aout2;      // FIRST: [2 plane][33][n_pad][256] fp16 planes of the layer output
    float *aout;                // !FIRST: [32 groups of 8 features][33][n_pad][8] fp32 (feature = direction * 128 + unit)
    int n_pad;
    int ntiles;                 // n_pad / 32
    int dir_only;               // -1: workgroup id = 2*tile + direction; 0 / 1: this direction only, workgroup id = tile
};

// The same MFMA as inline asm, accumulating in VGPRs, the resident weight operand pinned to the AGPR half of
// the register file ("a").  hipcc pads no hazards around an asm statement (cdna_hip_programming.md 5.7):
//   * A/B operands are never VALU-written right before (LDS loads; layer 1's converted features get an s_nop 1);
//   * accumulator chains (D of one MFMA = C of the next) need no wait states;
//   * the first reader of a finished accumulator is kept 12 wait states away by construction (see the step).
__device__ __forceinline__ void mfma32_av(f32x16 &acc, const f16x8 &w, const f16x8 &b) {
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "a"(w), "v"(b));
}
__device__ __forceinline__ void mfma32_av_first(f32x16 &acc, const f16x8 &w, const f16x8 &b, const f32x16 &c) {   // D = A.B + C, D != C
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %3" : "=&v"(acc) : "a"(w), "v"(b), "v"(c));
}
__device__ __forceinline__ void mfma32_vv(f32x16 &acc, const f16x8 &a, const f16x8 &b) {
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma32_vv_first(f32x16 &acc, const f16x8 &a, const f16x8 &b, const f32x16 &c) {
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %3" : "=&v"(acc) : "v"(a), "v"(b), "v"(c));
}

constexpr float GATE_K2 = 2.0f * 1.44269504088896340736f;
constexpr float GATE_BIG = 0x1p63f;   // clamp of 2^(2 log2(e) g): beyond it tanh(g) = 1 in float32 anyway, and (e - 1) / ((1 + e)(...)) stays inf-free

// ---- Gate arithmetic of one block (4 hidden units of one candidate per lane), shared by lstm32_body and lstm32_pair_kernel ----
// The block's accumulator Z holds v_exp_f32 arguments (pre-scaled rows, see the header): 2^Z = e^-i, e^2g, e^-f, e^-o.  Round 5
// ("arithmetic v2"): EIGHT transcendentals per hidden unit and step instead of ten -- each product of a
// sigmoid and a tanh takes ONE reciprocal:
//     K2 sig(i) tanh(g) = K2 (eg - 1) / ((1 + ei)(1 + eg))          c' <- c' / (1 + ef) + that        (c' = K2 c, K2 = 2 log2 e)
//     h = sig(o) tanh(c) = (ec - 1) / ((1 + eo)(1 + ec)),  ec = 2^c'
// What keeps inf and NaN out: eg is clamped to 2^63 (tanh is 1 in float32 long before; without it inf / inf), every other
// exponential may be inf -- it only ever sits in a denominator whose numerator is finite: (1 + ei) = inf gives rcp = 0 and
// K2 (eg - 1) * 0 = 0, the limit; ec <= 2^96 because |c| <= 33 (one unit per step at most).  A denominator that overflows
// although the value is representable (ei > 2^65 with eg at its clamp) returns 0 for something below 2^-65.
// Static schedule of 23 "gaps" (gap G is issued right after MFMA G of the block that follows): inside a gap the instructions
// are independent, every operand is at least one gap old, at most three are transcendental, the accumulator is read in gaps
// 1-6 only, and every gap costs at most 24 issue cycles (8 per transcendental, 4 per other instruction; tools/ubench/
// mfma_gap_mix.hip).  Registers: eg ei ef eo tt ng hh (4 each) + hp lp (2 each); eg / ei / ng are reused along the way:
//   E  2^Z             MN eg = min(eg, 2^63)     NG ng = K2 eg - K2        A  x += 1
//   P  eg = (1+eg)(1+ei)    R rcp                T  tt = ng / P            C  c' = c' rcp(1+ef) + tt
//   X  ei = 2^c'       AC eg = ei + 1            NC ng = ei - 1            Q  eg = (1+ec)(1+eo)       H  h = ng rcp(eg)
//   HP / D / LP: the fp16 split of h, two elements per instruction (v_cvt_pk_f16_f32; the residual h - float(hi) is one
//   v_fma_mix_f32 that converts the selected half on the fly).  The last gap stores the lane's four h values, 8 bytes per plane.
// Placement: a sched_barrier on both sides of every MFMA keeps an op from rising above the MFMA that opens its gap; pinning its
// OUTPUT (an empty asm volatile, ordered with the asm MFMAs) keeps it from sinking below the MFMA that closes it.
// Not two elements per instruction: v_pk_add/mul/fma_f32 wait for the matrix pipe (a block went 940 -> 1200 cycles, round 4).
#define CG_PIN(x) asm volatile("" : "+v"(x));
#define CG_E(R, C, e) { R[e] = __builtin_amdgcn_exp2f(Z[4 * (e) + (C)]); CG_PIN(R[e]) }
#define CG_MN(e) asm volatile("v_min_f32 %0, %1, %0" : "+v"(eg[e]) : "s"(GATE_BIG));
#define CG_NG(e) { ng[e] = fmaf(eg[e], GATE_K2, -GATE_K2); CG_PIN(ng[e]) }
#define CG_A(R, e) { R[e] += 1.0f; CG_PIN(R[e]) }
#define CG_P(e) { eg[e] = eg[e] * ei[e]; CG_PIN(eg[e]) }
#define CG_R(R, e) { R[e] = fast_rcp(R[e]); CG_PIN(R[e]) }
#define CG_T(e) { tt[e] = ng[e] * eg[e]; CG_PIN(tt[e]) }
#define CG_C(e) { C_[e] = fmaf(ef[e], C_[e], tt[e]); CG_PIN(C_[e]) }
#define CG_X(e) { ei[e] = __builtin_amdgcn_exp2f(C_[e]); CG_PIN(ei[e]) }
#define CG_AC(e) { eg[e] = ei[e] + 1.0f; CG_PIN(eg[e]) }
#define CG_NC(e) { ng[e] = ei[e] - 1.0f; CG_PIN(ng[e]) }
#define CG_Q(e) { eg[e] = eg[e] * eo[e]; CG_PIN(eg[e]) }
#define CG_H(e) { hh[e] = ng[e] * eg[e]; CG_PIN(hh[e]) }
#define CG_HP(q) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hp[q]) : "v"(hh[2 * (q)]), "v"(hh[2 * (q) + 1]));
#define CG_D(e) { if ((e) & 1) asm volatile("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(tt[e]) : "v"(hh[e]), "v"(hp[(e) >> 1])); \
                  else asm volatile("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(tt[e]) : "v"(hh[e]), "v"(hp[(e) >> 1])); }
#define CG_LP(q) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(lp[q]) : "v"(tt[2 * (q)]), "v"(tt[2 * (q) + 1]));
// in scope at the point of use: Z (the block's accumulator), C_ (its four cell states), eg ei ef eo tt ng hh hp lp; STORE = the two LDS stores
#define CLAIR_GATE_GAP(G, STORE)                                                                                  \
        switch (G) {                                                                                              \
            case 1: CG_E(eg, 1, 0) CG_E(eg, 1, 1) CG_E(eg, 1, 2) break;                                           \
            case 2: CG_E(eg, 1, 3) CG_E(ei, 0, 0) CG_E(ei, 0, 1) break;                                           \
            case 3: CG_E(ei, 0, 2) CG_E(ei, 0, 3) CG_E(ef, 2, 0) break;                                           \
            case 4: CG_E(ef, 2, 1) CG_E(ef, 2, 2) CG_E(ef, 2, 3) break;                                           \
            case 5: CG_E(eo, 3, 0) CG_E(eo, 3, 1) CG_E(eo, 3, 2) break;                                           \
            case 6: CG_E(eo, 3, 3) CG_MN(0) CG_MN(1) CG_MN(2) CG_MN(3) break;                                     \
            case 7: CG_A(ei, 0) CG_A(ei, 1) CG_A(ei, 2) CG_A(ei, 3) CG_NG(0) CG_A(eg, 0) break;                   \
            case 8: CG_NG(1) CG_A(eg, 1) CG_NG(2) CG_A(eg, 2) CG_NG(3) CG_A(eg, 3) break;                         \
            case 9: CG_P(0) CG_P(1) CG_P(2) CG_P(3) CG_A(ef, 0) CG_A(ef, 1) break;                                \
            case 10: CG_A(ef, 2) CG_A(ef, 3) CG_R(eg, 0) CG_R(eg, 1) break;                                       \
            case 11: CG_R(eg, 2) CG_R(eg, 3) CG_R(ef, 0) break;                                                   \
            case 12: CG_R(ef, 1) CG_R(ef, 2) CG_R(ef, 3) break;                                                   \
            case 13: CG_T(0) CG_T(1) CG_T(2) CG_T(3) CG_A(eo, 0) CG_A(eo, 1) break;                               \
            case 14: CG_C(0) CG_C(1) CG_C(2) CG_C(3) CG_A(eo, 2) CG_A(eo, 3) break;                               \
            case 15: CG_X(0) CG_X(1) CG_X(2) break;                                                               \
            case 16: CG_X(3) CG_AC(0) CG_AC(1) CG_AC(2) CG_NC(0) break;                                           \
            case 17: CG_AC(3) CG_Q(0) CG_Q(1) CG_Q(2) CG_NC(1) CG_NC(2) break;                                    \
            case 18: CG_Q(3) CG_NC(3) CG_R(eg, 0) CG_R(eg, 1) break;                                              \
            case 19: CG_R(eg, 2) CG_R(eg, 3) CG_H(0) CG_H(1) break;                                               \
            case 20: CG_H(2) CG_H(3) CG_HP(0) break;                                                              \
            case 21: CG_HP(1) CG_D(0) CG_D(1) break;                                                              \
            case 22: CG_D(2) CG_D(3) CG_LP(0) break;                                                              \
            default: CG_LP(1)   /* gap 23 */                                                                       \
                     STORE                                                                                        \
                     break;                                                                                       \
        }

// Hand-off words of the fused layer-2 kernel (lstm2_fused.hip.h): the projection workgroups publish "this (direction, tile, t)
// block is written" per producing wave, the recurrent workgroups wait for them a step ahead of their seed loads.
struct FuseArgs {
    unsigned *flags;        // [2 dir][n_pad/32][33][8 producer waves]  = ticket of the forward pass that wrote the block
    unsigned ticket;        // this forward pass (never 0; the words start zeroed)
    unsigned *claims;       // [workgroups of the launch]  ticket of the pass whose workgroup took this logical id
    unsigned *error;        // raised when a logical id is claimed twice or a wait runs out: the pass's results are not to be used
};

template <bool FIRST, bool FUSED>
__device__ __forceinline__ void lstm32_body(const Lstm32Args &p, const int d, const int tile, const FuseArgs &fz) {
    __shared__ __attribute__((aligned(16))) unsigned char lds_raw[FIRST ? L32_LDS_FIRST : L32_LDS_SECOND];
    _Float16 (*hbuf)[2][L32_TILE][HP_ROW] = (_Float16 (*)[2][L32_TILE][HP_ROW])lds_raw;   // [step parity][plane][cand][unit]
    _Float16 *wxl = (_Float16 *)(lds_raw + L32_HBUF_BYTES);             // FIRST: [wave][b][kk][plane][lane][8]
    float *bql = (float *)(lds_raw + L32_HBUF_BYTES + (FIRST ? 4 * 4 * 2 * 2 * 64 * 16 : 0));   // FIRST: [wave][b][a][h'][4]
    _Float16 *xt = (_Float16 *)(bql + (FIRST ? 4 * 4 * 4 * 2 * 4 : 0));   // FIRST: [step parity][plane][32 cand][L32_XT_ROW] input tile, already split

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cand = lane & 31, hq = lane >> 5;

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
        // asm-defined AGPR values (after ALL loads are in flight): hipcc can neither re-load them from memory inside
        // the step loop (it does, given the chance) nor park them in architectural VGPRs
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int kk = 0; kk < 8; ++kk)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) asm volatile("" : "+a"(Aw[b][kk][pl]));
    }
    if (FIRST) {   // this wave's Wx1 fragments and bias quads -> LDS (read back by this wave only)
        const f32x4 *src = (const f32x4 *)p.wxs + (size_t)(d * 4 + w) * (4 * 2 * 2 * 64) + lane;
        f32x4 *dst = (f32x4 *)wxl + (size_t)w * (4 * 2 * 2 * 64) + lane;
#pragma unroll
        for (int i = 0; i < 16; ++i) dst[i * 64] = src[i * 64];
        if (lane < 32) ((f32x4 *)bql)[w * 32 + lane] = ((const f32x4 *)p.biasq)[(d * 4 + w) * 32 + lane];
    }
    float cst[4][4];   // c' = 2 log2(e) c of (block b, element a): unit 32w + 8b + 4h' + a of candidate lane%32
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int a = 0; a < 4; ++a) cst[b][a] = 0.0f;

    // ---- layer 1 input: x_t of the 32 candidates (4 KiB) goes through LDS -- ONE coalesced 16-byte load per thread and step
    //      (row tid>>3, chunk tid&7), a step ahead; the loading thread splits its four values into the two fp16 planes (eight VALU
    //      instructions in block 0's free gaps) and every wave then reads its B fragments (features 16*kk + 8*h' .. +7 of
    //      candidate lane%32) from the tile at the step head, next to the h fragments.  (Per-lane loads straight from [n][33][32]
    //      were four half-used-segment loads per wave and step; splitting after the tile made every wave convert the same 16
    //      values per lane at the exposed step head: 6 us of the kernel.)
    const float *xg = FIRST ? p.x + ((size_t)tile * L32_TILE + (tid >> 3)) * (T_POS * F_IN) + (tid & 7) * 4 : nullptr;
    auto load_x = [&](int s) -> f32x4 {
        const int sc = s < T_POS ? s : T_POS - 1;
        return *(const f32x4 *)(xg + (d ? T_POS - 1 - sc : sc) * F_IN);
    };
    unsigned xp_hi[2], xp_lo[2];
    float xres[4];
    auto split_x = [&](f32x4 &v, int part) {   // same arithmetic as split2 on x 2^-L32_X_SHIFT: hi = fp16(x), lo = fp16(x - hi)
        if (part < 0) {
            constexpr float k = 1.0f / (float)(1 << L32_X_SHIFT);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] *= k; asm volatile("" : "+v"(v[e])); }
        } else if (part == 0) {
            asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(xp_hi[0]) : "v"(v[0]), "v"(v[1]));
            asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(xp_hi[1]) : "v"(v[2]), "v"(v[3]));
        } else if (part == 1) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (e & 1) asm volatile("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(xres[e]) : "v"(v[e]), "v"(xp_hi[e >> 1]));
                else asm volatile("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(xres[e]) : "v"(v[e]), "v"(xp_hi[e >> 1]));
            }
        } else {
            asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(xp_lo[0]) : "v"(xres[0]), "v"(xres[1]));
            asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(xp_lo[1]) : "v"(xres[2]), "v"(xres[3]));
        }
    };
    auto stage_x = [&](int s) {
        _Float16 *dst = &xt[(((s & 1) * 2 + 0) * L32_TILE + (tid >> 3)) * L32_XT_ROW + (tid & 7) * 4];
        *(uint2 *)dst = make_uint2(xp_hi[0], xp_hi[1]);
        *(uint2 *)(dst + L32_TILE * L32_XT_ROW) = make_uint2(xp_lo[0], xp_lo[1]);
    };
    // ---- accumulator seeds (the C operand of a block's first MFMA)
    // layer 2: block b of step s of the x-projection, 4 x 16 bytes per lane; layer 1: the block's bias quads from LDS
    const float *zx0 = FIRST ? nullptr : p.zx + ((((size_t)d * p.ntiles + tile) * T_POS * 4 + w) * 4) * 1024 + lane * 4;
    auto load_seed = [&](f32x16 &z, int s, int b) {
        if (FIRST) {
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const f32x4 v = *(const f32x4 *)&bql[((w * 4 + b) * 4 + a) * 8 + hq * 4];
                z[4 * a + 0] = v[0]; z[4 * a + 1] = v[1]; z[4 * a + 2] = v[2]; z[4 * a + 3] = v[3];
            }
        } else {
            const int sc = s < T_POS ? s : T_POS - 1;   // the prefetch past the last step re-reads the last one
            const int t = d ? T_POS - 1 - sc : sc;
            const float *src = zx0 + ((size_t)t * 16 + b) * 1024;
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const f32x4 v = __builtin_nontemporal_load((const f32x4 *)(src + a * 256));   // read once: keep it out of the caches' way
                z[4 * a + 0] = v[0]; z[4 * a + 1] = v[1]; z[4 * a + 2] = v[2]; z[4 * a + 3] = v[3];
            }
        }
    };
    // ---- fused layer 2: wave w's seeds of step s come from producer waves 2w and 2w+1 of the projection workgroup that owns this
    //      direction's gate rows 128 (w >> 1) .. +127 (gemm_split.hip.h: slice = gtile*4 + wave).  Their two ticket words are read
    //      one step before they are needed (the words are written through to memory, ~2 us away) and polled only if still behind.
    const unsigned long long *fl_base = FUSED ? (const unsigned long long *)fz.flags + ((size_t)(d * p.ntiles + tile) * T_POS) * 4 + w : nullptr;
    unsigned long long fl_next = 0;
    auto flag_fetch = [&](int s) {
        if (!FUSED || s >= T_POS) return;
        fl_next = __hip_atomic_load(fl_base + (size_t)(d ? T_POS - 1 - s : s) * 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    auto flag_wait = [&](int s) {
        if (!FUSED || s >= T_POS) return;
        const unsigned long long want = ((unsigned long long)fz.ticket << 32) | fz.ticket;
        const unsigned long long *fp = fl_base + (size_t)(d ? T_POS - 1 - s : s) * 4;
        unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)fl_next), hi = __builtin_amdgcn_readfirstlane((unsigned)(fl_next >> 32));
        if ((((unsigned long long)hi << 32) | lo) != want) {
            // Bounded: 50 ms at 100 MHz is three orders of magnitude beyond any honest wait.  A queue that was descheduled for longer
            // (several processes on one GPU) runs it out; the engine then re-runs the pass on the two-launch path (engine.hip:
            // recover_fused), it does not fail.
            const long long deadline = wall_clock64() + 5000000;
            while ((((unsigned long long)hi << 32) | lo) != want) {
                __builtin_amdgcn_s_sleep(2);
                const unsigned long long v = __hip_atomic_load(fp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                lo = __builtin_amdgcn_readfirstlane((unsigned)v); hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
                if (wall_clock64() > deadline) { if (lane == 0) *fz.error = 1u; break; }
            }
        }
        // acquire side of the hand-off: nothing this CU may hold of the block (vector L1) survives the sighting of its ticket.  Free
        // here: the only vector-memory operations in flight are older than a step.  The release side is the producer's in-order
        // retirement of its stores into the L2 this wave reads through, then its ticket (gemm_split.hip.h: publish); an agent-scope
        // release there would write back the XCD's whole L2 per ticket (measured: 29 -> 68 us), which is why the hand-off is
        // confined to ONE L2 and the placement that guarantees it is checked per launch (lstm2_fused.hip.h: claims).
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    };
    // ---- h_s (both planes complete in LDS) -> HBM, as four pieces per thread laid out so that every wave-level store is one
    //      contiguous run per row (a thread-per-row-chunk map made each store touch 64 quarter-filled 64-byte segments and cost
    //      ~200 issue cycles; this way a store covers two whole 512-byte fp32 rows / four 256-byte fp16 rows).  Split into an LDS
    //      read, conversion micro-steps and the stores so that the pieces can sit in different MFMA shadows.
    //      layer 1: piece j = 16-byte chunk g = 256 j + tid of [plane][32 rows][16 chunks of 8 units]
    //      layer 2: piece j = 16-byte chunk g = 256 j + tid of [16 channel groups][32 candidates][2 halves of 4 units] (fp32 sum of the two
    //      planes): the output is stored channel-group-major, [32 groups of 8 features][33 t][n_pad][8], because its only reader
    //      (l3l4_kernel) stages [t][64 candidates][one group] tiles: with candidate rows of 256 features that tile was 2 112 separate
    //      32-byte quarter-lines per workgroup and its 66 LDS-DMA instructions took 9 700 cycles to ISSUE (a quarter of the
    //      kernel, tools/gpu/l34_stamps.py); group-major it is 66 contiguous KiB
    f16x8 cp[FIRST ? 4 : 1];
    f16x4 c4[FIRST ? 1 : 4][2];
    f32x4 co[FIRST ? 1 : 4];
    auto copy_read = [&](int s) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int g = j * 256 + tid;
            if (FIRST) cp[j] = *(const f16x8 *)&hbuf[s & 1][g >> 9][(g >> 4) & 31][(g & 15) * 8];
            else {   // chunk g: channel group g >> 6 (8 units), candidate (g & 63) >> 1, half g & 1
                c4[j][0] = *(const f16x4 *)&hbuf[s & 1][0][(g & 63) >> 1][(g >> 6) * 8 + (g & 1) * 4];
                c4[j][1] = *(const f16x4 *)&hbuf[s & 1][1][(g & 63) >> 1][(g >> 6) * 8 + (g & 1) * 4];
            }
        }
    };
    auto copy_cvt = [&](int i) {   // layer 2: micro-step i = 0..7 of the fp16-planes -> fp32 conversion: two units each
        if (FIRST) return;
        const int j = i >> 1, q = (i & 1) * 2;
        co[j][q] = (float)c4[j][0][q] + (float)c4[j][1][q];
        co[j][q + 1] = (float)c4[j][0][q + 1] + (float)c4[j][1][q + 1];
    };
    auto copy_write = [&](int s, int j) {
        const int t = d ? T_POS - 1 - s : s;
        const int g = j * 256 + tid;
        const size_t row0 = ((size_t)t * p.n_pad + (size_t)tile * L32_TILE) * (2 * HID) + d * HID;
        if (FIRST) {
            const size_t plane = (size_t)T_POS * p.n_pad * (2 * HID);
            *(f16x8 *)(p.aout2 + (g >> 9) * plane + row0 + (size_t)((g >> 4) & 31) * (2 * HID) + (g & 15) * 8) = cp[j];
        } else {   // a2 is channel-group-major for its only reader (dense.hip.h): [32 groups][33 t][n_pad][8]; a wave store = one group's 32 x 8 block, 1 KiB.
                   // Written once, read once by another kernel: non-temporal on both sides (round 6, with l3l4's nt LDS-DMA: +0.5 % at batch 1 024,
                   // +1.0 % at 4 096, profiles/r06_ab_cache_hints*.txt).  NOT a1 above: the projection re-reads it four times per XCD and wants the caches
                   // (its stores non-temporal: the projection alone 77 -> 85 us).
            __builtin_nontemporal_store(co[j], (f32x4 *)(p.aout + ((((size_t)(d * 16 + (g >> 6)) * T_POS + t) * p.n_pad + (size_t)tile * L32_TILE) * 8) + (g & 63) * 4));
        }
    };

    f32x16 acc[2];        // layer 2: block b accumulates in acc[b & 1]
    f32x16 xacc[4];       // layer 1: block b accumulates in xacc[b], which already holds bias + x-part when the step begins
    f32x16 zq[4];         // layer 2 seeds: [b] = block b of the next step it is needed in
    f32x4 xreg;           // layer 1: this thread's 16 bytes of x three steps ahead
    f16x8 xh[2], xl[2];   // layer 1: the NEXT step's B fragments (k-step kk), hi / lo plane
#define L32_ACC(b) (*(FIRST ? &xacc[(b)] : &acc[(b) & 1]))
    f16x8 hf[8][2];       // B fragments of h_{s-1}: [kk][plane]
    f16x8 wxa[2][2][2];   // layer 1: A fragments of Wx1 for block b in [b & 1][kk][plane], fetched from LDS a block ahead
    auto load_wx = [&](f16x8 (&dst)[2][2], int b) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) dst[kk][pl] = *(const f16x8 *)&wxl[((((size_t)w * 4 + b) * 2 + kk) * 2 + pl) * 512 + lane * 8];
    };

    // h_{-1} = 0: step 0 runs the same code as every other step (its h-part MFMAs add zero)
    for (int i = tid; i < 2 * L32_TILE * HP_ROW / 8; i += 256) ((f32x4 *)&hbuf[1][0][0][0])[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (FIRST) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            xreg = load_x(t);
            split_x(xreg, -1);
            split_x(xreg, 0);
            split_x(xreg, 1);
            split_x(xreg, 2);
            stage_x(t);
        }
        xreg = load_x(2);
    } else {
        flag_fetch(0);
        flag_wait(0);
        flag_fetch(1);
#pragma unroll
        for (int b = 0; b < 4; ++b) load_seed(zq[b], 0, b);
    }
    __syncthreads();   // zeros, Wx1 fragments, bias quads and the first two input tiles visible

    // Gate math of block PB in gap G of the block that follows it: the shared schedule above (CLAIR_GATE_GAP)
#define L32_GAP(G, PB)                                                                                            \
    {                                                                                                             \
        const f32x16 &Z = L32_ACC(PB);                                                                            \
        float (&C_)[4] = cst[PB];                                                                                 \
        CLAIR_GATE_GAP(G, *(uint2 *)&hbuf[s & 1][0][cand][w * 32 + (PB) * 8 + hq * 4] = make_uint2(hp[0], hp[1]);  \
                          *(uint2 *)&hbuf[s & 1][1][cand][w * 32 + (PB) * 8 + hq * 4] = make_uint2(lp[0], lp[1]);) \
    }
    // What goes into the gap after MFMA number M (0-based, NM per block) of block B: the previous block's gate
    // schedule from gap 1 on (the previous block's last MFMA needs 12 wait states before its result is read), and in block 0 the copy-out of h_{s-1}.
#define L32_AFTER_MFMA(M, NM, B)                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                            \
    CLAIR_DBG_FENCE();                                                                                            \
    if ((B) == 0 && (M) % 3 == 0 && (M) / 3 + 2 < 8) {   /* the fragments of k-step M/3 + 2 */                    \
        hf[(M) / 3 + 2][0] = *(const f16x8 *)&hbuf[(s + 1) & 1][0][cand][((M) / 3 + 2) * 16 + hq * 8];           \
        hf[(M) / 3 + 2][1] = *(const f16x8 *)&hbuf[(s + 1) & 1][1][cand][((M) / 3 + 2) * 16 + hq * 8];           \
    }                                                                                                             \
    if (FUSED && (B) == 0 && (M) == 2) { flag_wait(s + 1); flag_fetch(s + 2); }                                   \
    if (!FIRST && (M) == 3) load_seed(zq[B], s + 1, B);   /* after the keep-alive below: the refill can land in the very registers it replaces */ \
    if ((B) > 0 && (M) >= 1 && (M) <= 23) L32_GAP(M, ((B) > 0 ? (B) - 1 : 0))                                                     \
    if (!FIRST && (M) == 2) asm volatile("" :: "v"(zold));   /* the first MFMA's C registers stay untouched until here */ \
    if (FIRST && (B) > 0 && (M) == 8) load_seed(xacc[(B) - 1], 0, (B) - 1);   /* the gates above read their accumulators in gaps 1-6: block B-1's restarts from its bias */ \
    if ((B) == 0) {   /* at s = 0 this copies the (uninitialised) other h buffer to row t(0); step 1 overwrites it */ \
        if ((M) == 1) copy_read(s_prev);                                                                          \
        if ((M) >= 4 && (M) < 12) copy_cvt((M) - 4);   /* two units per MFMA shadow */                            \
        if ((M) >= 12 && (M) < 20 && ((M) & 1) == 0) copy_write(s_prev, ((M) - 12) >> 1);                          \
    }                                                                                                             \
    if (FIRST && (B) == 0) {   /* x_{s+2} (loaded a step ago): split, stage into the tile this step does not read */ \
        if ((M) >= 19 && (M) <= 22) split_x(xreg, (M) - 20);                                                      \
        if ((M) == 23) stage_x(s + 2);                                                                            \
    }                                                                                                             \
    if (FIRST && (B) == 1 && (M) == 0) xreg = load_x(s + 3);                                                      \
    if (FIRST && (B) == 3) {   /* operands of the x-part that follows block 3: x_{s+1} fragments, Wx1 fragments of blocks 0 and 1 */ \
        if ((M) == 12) read_xfrag(s + 1);                                                                         \
        if ((M) == 14) load_wx(wxa[0], 0);                                                                        \
        if ((M) == 16) load_wx(wxa[1], 1);                                                                        \
    }                                                                                                             \
    __builtin_amdgcn_sched_barrier(0);

#define L32_BLOCK(b)                                                                                              \
    {                                                                                                             \
            constexpr int NM = 24;                                                                               \
            const f32x16 zold = zq[b];                                                                           \
_Pragma("unroll")                                                                                                \
            for (int m = 0; m < 24; ++m) {                                                                       \
                const int kk = m / 3, term = m % 3;                                                              \
                if (!FIRST && m == 0) mfma32_av_first(acc[b & 1], Aw[b][kk][1], hf[kk][0], zold);                \
                else mfma32_av(L32_ACC(b), Aw[b][kk][term == 0 ? 1 : 0], hf[kk][term == 1 ? 1 : 0]);             \
                L32_AFTER_MFMA(m, NM, b)                                                                         \
            }                                                                                                    \
    }

    // Layer 1: the x-part of the NEXT step (K = 32 = two k-steps per block, 24 MFMAs; Wx1 fragments from LDS) does not depend on
    // h, so it runs after block 3 and the last block's gate math hides behind it exactly as the other blocks' gates hide behind
    // the following block.  Block 3's accumulator is read by gaps 1-6 and restarts from its bias in gap 8; its own x-part is the
    // last six MFMAs.
#define L32_XTAIL(GATES)                                                                                          \
    _Pragma("unroll") for (int q = 0; q < 24; ++q) {                                                              \
        const int xb = q / 6, kk = (q % 6) / 3, term = q % 3;                                                     \
        mfma32_vv(xacc[xb], wxa[xb & 1][kk][term == 0 ? 1 : 0], term == 1 ? xl[kk] : xh[kk]);                     \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        CLAIR_DBG_FENCE();                                                                                        \
        if ((GATES) && q >= 1) L32_GAP(q, 3)                                                                      \
        if ((GATES) && q == 8) load_seed(xacc[3], 0, 3);                                                          \
        if (q == 5) load_wx(wxa[0], 2);                                                                           \
        if (q == 11) load_wx(wxa[1], 3);                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
    }
    auto read_xfrag = [&](int s) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const _Float16 *xp = &xt[(((s & 1) * 2 + 0) * L32_TILE + cand) * L32_XT_ROW + kk * 16 + hq * 8];
            xh[kk] = *(const f16x8 *)xp;
            xl[kk] = *(const f16x8 *)(xp + L32_TILE * L32_XT_ROW);
        }
    };
    unsigned hp[2], lp[2];   // packed fp16 pairs of h: hi plane, lo plane
    float eg[4], ei[4], ef[4], eo[4], tt[4], ng[4], hh[4];
    if (FIRST) {   // bias + x-part of step 0
#pragma unroll
        for (int b = 0; b < 4; ++b) load_seed(xacc[b], 0, b);
        load_wx(wxa[0], 0);
        load_wx(wxa[1], 1);
        read_xfrag(0);
        {
            const int s = 0;   // (named by the gap macro; no gap runs here)
            (void)s;
            L32_XTAIL(0)
        }
        __syncthreads();   // step 0 re-stages the tile just read
    }

    for (int s = 0; s < T_POS; ++s) {
        // h_{s-1} fragments: the first two k-steps now, the others two k-steps ahead of their MFMAs inside block 0 (L32_AFTER_MFMA).  All sixteen
        // reads in front of the block were 530 cycles per step during which no MFMA could issue: four waves x 16 KiB through the CU's 128 B/clk
        // of LDS (tools/gpu/lstm_stamps.py, profiles/r04_lstm_stamps.txt).
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) hf[kk][pl] = *(const f16x8 *)&hbuf[(s + 1) & 1][pl][cand][kk * 16 + hq * 8];
        const int s_prev = s > 0 ? s - 1 : 0;
        // x-part first (layer 1: K = 32 = two k-steps, Wx1 fragments from LDS), then the h-part (K = 128 = eight k-steps);
        // terms per k-step: w_lo.h_hi, w_hi.h_lo, w_hi.h_hi.  The C operand of a block's first MFMA (D != C there) is kept alive
        // two MFMAs longer by L32_AFTER_MFMA: hipcc knows nothing about the asm MFMA still reading it and would hand the
        // registers to the next VALU result.
        L32_BLOCK(0)
        L32_BLOCK(1)
        L32_BLOCK(2)
        L32_BLOCK(3)
        if (FIRST) {
            L32_XTAIL(1)
        } else {
            // layer 2: the last block's gates have no MFMAs left to hide behind (12 wait states after its last MFMA)
            asm volatile("s_nop 11" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 1; g <= 23; ++g) L32_GAP(g, 3)
        }
        __syncthreads();
    }
#undef L32_BLOCK
#undef L32_XTAIL
#undef L32_ACC
#undef L32_AFTER_MFMA
#undef L32_GAP
    copy_read(T_POS - 1);
#pragma unroll
    for (int i = 0; i < 8; ++i) copy_cvt(i);
#pragma unroll
    for (int j = 0; j < 4; ++j) copy_write(T_POS - 1, j);
}

template <bool FIRST>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void lstm32_kernel(Lstm32Args p) {
    const int d = p.dir_only < 0 ? (blockIdx.x & 1) : p.dir_only;
    const int tile = p.dir_only < 0 ? (blockIdx.x >> 1) : blockIdx.x;
    lstm32_body<FIRST, false>(p, d, tile, FuseArgs{nullptr, 0u, nullptr, nullptr});
}

}  // namespace clair
