// Shared device helpers for the Clair forward-pass kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace clair {

constexpr int T_POS = 33;      // positions (shared/param.py:9 -> 2*16+1)
constexpr int F_IN = 32;       // features per position (8 rows x 4 channels)
constexpr int HID = 128;       // LSTM units per direction (clair/model.py:92-93)
constexpr int GATES = 512;     // 4*HID, column order i | c~ | f | o
constexpr int L3_UNITS = 30;   // clair/model.py:81
constexpr int L3_OUT = 7680;   // 30*256, flat index u*256+c (clair/model.py:474-478)
constexpr int L4_UNITS = 192;  // clair/model.py:82
constexpr int L5_UNITS = 96;   // clair/model.py:84-91
constexpr int OUT_FLOATS = 90; // 21 + 3 + 33 + 33
constexpr int L4_SPLITS = 8;   // split-K factor of the 7680->192 GEMM: one partial per 32 LSTM2 features = four channel groups of 8 (dense.hip.h)

// exp via v_exp_f32 (2^x); relative error ~1 ulp, enough for the 2e-6 probability tolerance.
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

// clair/selu.py:26-30 : scale * where(x >= 0, x, alpha * elu(x))
// elu needs expm1, not exp - 1: near 0 the subtraction leaves 6e-8 ABSOLUTE, i.e. 1e-5 relative on an activation of
// 0.01 -- visible as 1e-5 on the probabilities once a layer with small outputs feeds one with large weights
// (tools/parity_sweep.py, cell "L4 kernel x0.01").  On (-0.125, 0] the degree-6 Taylor polynomial is exact to 9e-8
// relative; below, exp - 1 is (<= 5e-7 relative).
__device__ __forceinline__ float selu_scaled(float x, float k) {   // k * selu(x); k folds a following power-of-two activation scale
    constexpr float alpha = 1.6732632423543772848170429916717f;
    constexpr float scale = 1.0507009873554804934193349852946f;
    float p = fmaf(x, 1.0f / 720.0f, 1.0f / 120.0f);
    p = fmaf(p, x, 1.0f / 24.0f);
    p = fmaf(p, x, 1.0f / 6.0f);
    p = fmaf(p, x, 0.5f);
    p = fmaf(p, x, 1.0f);
    const float em1 = x > -0.125f ? p * x : fast_exp(x) - 1.0f;
    return x >= 0.0f ? x * (scale * k) : em1 * (alpha * scale * k);
}
__device__ __forceinline__ float selu_f(float x) { return selu_scaled(x, 1.0f); }

// Two values at once on the packed fp32 pipe (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: one issue slot per PAIR): ~10 slots per value
// instead of ~14.  Same function, bit for bit: the positive and the negative branch are computed on max(x, 0) and min(x, 0) and added --
// one of the two terms is exactly zero (the polynomial branch gives p(0) * 0 = 0 for x >= 0), so the sum is the other term unrounded.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 selu_scaled2(f32x2 x, float k) {
    constexpr float alpha = 1.6732632423543772848170429916717f;
    constexpr float scale = 1.0507009873554804934193349852946f;
    const f32x2 xm = {fminf(x[0], 0.0f), fminf(x[1], 0.0f)};
    const f32x2 xp = x - xm;        // max(x, 0) exactly (one of the two is x, the other 0): one packed subtraction instead of two v_max
    f32x2 p = xm * (1.0f / 720.0f) + (1.0f / 120.0f);
    p = p * xm + (1.0f / 24.0f);
    p = p * xm + (1.0f / 6.0f);
    p = p * xm + 0.5f;
    p = p * xm + 1.0f;
    const f32x2 small = p * xm;
    const f32x2 t = xm * 1.44269504088896340736f;
    const f32x2 big = (f32x2){__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])} - 1.0f;
    const f32x2 em1 = {xm[0] > -0.125f ? small[0] : big[0], xm[1] > -0.125f ? small[1] : big[1]};
    return em1 * (alpha * scale * k) + xp * (scale * k);
}

// ---- 2-way fp16 split of fp32 values (gemm_split.hip.h, lstm32.hip.h, dense.hip.h) ------------------------------------
// x ~= x1 + x2 with x1 = fp16(x), x2 = fp16(x - x1): 22 significand bits (relative error <= 2^-22, absolute
// error <= 3e-8 once x2 falls into the fp16 subnormal range), round-to-nearest-even both times.  Products of
// two fp16 values are exact in fp32, so  a*b ~= a1*b1 + a1*b2 + a2*b1  on v_mfma_f32_16x16x32_f16 (16x the
// rate of the fp32 MFMA) reproduces the fp32 product to ~3e-7 relative: end to end the probabilities stay as
// close to a float64 evaluation as with fp32 MFMAs (4.6e-7..8e-7 vs 3.5e-7..4.6e-7, tools/split_emulation.py).
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));   // one MFMA A/B operand (4 VGPRs)
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void split2(float x, _Float16 &hi, _Float16 &lo) {
    hi = (_Float16)x;
    // ONE definition of hi.  Under hipcc's default -ffp-contract=fast the compiler may otherwise materialise hi twice --
    // once as v_cvt_f16_f32 of the rounded fp32 value, once fused with the producing multiply (v_fma_mixlo_f16, single
    // rounding) -- and the two disagree by one fp16 ulp about once in 30 000 values: the stored hi plane and the
    // residual then belong to different splits (found as 1e-5 instead of 2e-6 error on L4, tools/debug_taps.py).
    asm("" : "+v"(hi));
    lo = (_Float16)(x - (float)hi);
}

// The same split for four values at once, as packed pairs ready for an 8-byte LDS store: v_cvt_pk_f16_f32 rounds two values per
// instruction (round to nearest even, like the scalar conversion), v_fma_mix_f32 forms x - float(hi) with the fp16 operand
// converted on the fly (exact), a second packed conversion rounds the residuals: 8 instructions instead of ~20.
__device__ __forceinline__ void split2_pk4(const float (&x)[4], uint2 &hi, uint2 &lo) {
    float r[4];
    asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hi.x) : "v"(x[0]), "v"(x[1]));
    asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hi.y) : "v"(x[2]), "v"(x[3]));
    asm volatile("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(r[0]) : "v"(x[0]), "v"(hi.x));
    asm volatile("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r[1]) : "v"(x[1]), "v"(hi.x));
    asm volatile("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(r[2]) : "v"(x[2]), "v"(hi.y));
    asm volatile("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r[3]) : "v"(x[3]), "v"(hi.y));
    asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(lo.x) : "v"(r[0]), "v"(r[1]));
    asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(lo.y) : "v"(r[2]), "v"(r[3]));
}

__device__ __forceinline__ f32x16 mfma32h(f16x8 a, f16x8 b, f32x16 c) {
    // v_mfma_f32_32x32x16_f16: lane (i = l%32, q = l/32) supplies A[i][8q..8q+7] / B[8q..8q+7][i];
    // C/D: column l%32, rows 8*(reg/4) + 4*(l/32) + reg%4
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

typedef __attribute__((address_space(3))) void *lptr_t;

// Hand-counted waits.  The production build waits for exactly the vector-memory operations a point depends on (CLAIR_VMWAIT(n):
// "at most n still outstanding").  The CHECK build (-DCLAIR_WAIT_ALL, clair_amd/build.py: build_waitall -> libclair_amd_waitall.so)
// waits for everything at every such point AND drains memory, LDS and the matrix pipe after every hand-placed MFMA
// (CLAIR_DBG_FENCE: 48 wait states cover the 8-pass MFMA plus every documented read-after-MFMA hazard), so that none of the
// timing arguments in gemm_split.hip.h / lstm32*.hip.h is load-bearing in it.  tools/gpu/waitall_compare.py runs both builds on
// the same candidates and compares their outputs bit for bit: a latent ordering hazard in the production build shows up as a
// difference.  This is a test instrument, not a second code path: nothing ships or runs it outside that tool.
#ifdef CLAIR_WAIT_ALL
#define CLAIR_VMWAIT(n) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory")
#define CLAIR_DBG_FENCE() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory")
#else
#define CLAIR_VMWAIT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define CLAIR_DBG_FENCE()
#endif

// One LDS-DMA piece: 64 lanes x 16 B from per-lane global addresses to LDS [lds_base, lds_base + 1 KiB).
// Inline asm on purpose: hipcc then neither counts it nor fences later ds_reads of the same __shared__
// array behind it with vmcnt(0).  M0 carries the LDS base and is compiler-reserved, so it is
// saved/restored inside the statement.
__device__ __forceinline__ void glds16(const f32x4 *gsrc, unsigned lds_base) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_base) : "memory");
}

// glds16_s (below) with the nt bit: for data that is read exactly once (l3l4's a2 tiles) -- the line is not kept in the L2 at the expense of
// what other kernels in flight re-read (round 6: +0.6 % of the pipeline at batch 1 024, profiles/r06_ab_cache_hints2.txt).
__device__ __forceinline__ void glds16_s_nt(unsigned lane_off, const void *sbase, unsigned lds_base) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(lane_off), "s"(sbase), "s"(lds_base) : "memory");
}

// The same with a wave-uniform 64-bit base in SGPRs and a per-lane 32-bit byte offset: a persistent kernel keeps its lane
// offsets in registers for its whole life and only re-bases (scalar arithmetic) from piece to piece.
__device__ __forceinline__ void glds16_s(unsigned lane_off, const void *sbase, unsigned lds_base) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(lane_off), "s"(sbase), "s"(lds_base) : "memory");
}

}  // namespace clair
