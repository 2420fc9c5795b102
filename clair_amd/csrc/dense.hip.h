// Position-mixing "slice dense" layer (L3) and the classifier tail (L4 reduce, L5 x4, heads, softmax).
#pragma once
#include "common.hip.h"

namespace clair {

// ---- L3: for every LSTM2 feature c, dense(33 -> 30) over the position axis + selu -------------
// clair/model.py:225-244 (slice_dense_layer), :464-479: l3[n][u*256 + c] =
//   selu( sum_t a2[t][n][c] * W3[c][t][u] + b3[c][u] ).
// One thread per feature c, L3_CAND candidates per workgroup so every weight load (coalesced over
// c from the host-packed W3p[t][u][c]) feeds L3_CAND FMAs.
constexpr int L3_CAND = 4;

struct L3Args {
    const float *a2;   // [33][n_pad][256]
    const float *w3p;  // [33][30][256]
    const float *b3p;  // [30][256]
    float *l3;         // [n_pad][7680]
    int n_pad;
};

__global__ __launch_bounds__(256) void l3_kernel(L3Args p) {
    const int c = threadIdx.x;
    const int n0 = blockIdx.x * L3_CAND;
    float acc[L3_CAND][L3_UNITS];
#pragma unroll
    for (int u = 0; u < L3_UNITS; ++u) {
        const float b = p.b3p[u * 256 + c];
#pragma unroll
        for (int j = 0; j < L3_CAND; ++j) acc[j][u] = b;
    }
    for (int t = 0; t < T_POS; ++t) {
        float a[L3_CAND];
#pragma unroll
        for (int j = 0; j < L3_CAND; ++j) a[j] = p.a2[((size_t)t * p.n_pad + n0 + j) * 256 + c];
        const float *wt = p.w3p + (size_t)t * L3_UNITS * 256 + c;
#pragma unroll
        for (int u = 0; u < L3_UNITS; ++u) {
            const float wv = wt[u * 256];
#pragma unroll
            for (int j = 0; j < L3_CAND; ++j) acc[j][u] = fmaf(a[j], wv, acc[j][u]);
        }
    }
#pragma unroll
    for (int j = 0; j < L3_CAND; ++j)
#pragma unroll
        for (int u = 0; u < L3_UNITS; ++u) p.l3[(size_t)(n0 + j) * L3_OUT + u * 256 + c] = selu_f(acc[j][u]);
}

// ---- tail: L4 split-K reduce + selu, L5_1..4 + selu, heads + selu + softmax ---------------------
// clair/model.py:482-488 (L4), :507-569 (L5_k), :582-620 (heads: selu on the logits, then softmax).
// Output rows are packed gt21(21) | genotype(3) | len1(33) | len2(33).
constexpr int TAIL_CAND = 8;

struct TailArgs {
    const float *l4part;  // [L4_SPLITS][n_pad][192]
    const float *b4;      // [192]
    const float *w5p;     // [192][384]   column k5*96 + j
    const float *b5p;     // [384]
    const float *whp;     // [96][90]     column = packed output index; rows = units of that head's L5 branch
    const float *bhp;     // [90]
    float *out;           // [n][90]
    int n_pad;
    int n;                // valid candidates
};

__device__ __forceinline__ int head_of_output(int o) { return o < 21 ? 0 : (o < 24 ? 1 : (o < 57 ? 2 : 3)); }

__global__ __launch_bounds__(256) void tail_kernel(TailArgs p) {
    __shared__ float l4s[TAIL_CAND][L4_UNITS];
    __shared__ float l5s[TAIL_CAND][4 * L5_UNITS];
    __shared__ float lgs[TAIL_CAND][OUT_FLOATS + 6];
    const int tid = threadIdx.x;
    const int n0 = blockIdx.x * TAIL_CAND;

    for (int idx = tid; idx < TAIL_CAND * L4_UNITS; idx += 256) {
        const int m = idx / L4_UNITS, j = idx - m * L4_UNITS;
        float s = p.b4[j];
#pragma unroll
        for (int sp = 0; sp < L4_SPLITS; ++sp) s += p.l4part[((size_t)sp * p.n_pad + n0 + m) * L4_UNITS + j];
        l4s[m][j] = selu_f(s);
    }
    __syncthreads();

    for (int col = tid; col < 4 * L5_UNITS; col += 256) {
        float acc[TAIL_CAND];
        const float b = p.b5p[col];
#pragma unroll
        for (int m = 0; m < TAIL_CAND; ++m) acc[m] = b;
        for (int k = 0; k < L4_UNITS; ++k) {
            const float wv = p.w5p[k * (4 * L5_UNITS) + col];
#pragma unroll
            for (int m = 0; m < TAIL_CAND; ++m) acc[m] = fmaf(l4s[m][k], wv, acc[m]);
        }
#pragma unroll
        for (int m = 0; m < TAIL_CAND; ++m) l5s[m][col] = selu_f(acc[m]);
    }
    __syncthreads();

    if (tid < OUT_FLOATS) {
        const int k5 = head_of_output(tid);
        float acc[TAIL_CAND];
        const float b = p.bhp[tid];
#pragma unroll
        for (int m = 0; m < TAIL_CAND; ++m) acc[m] = b;
        for (int k = 0; k < L5_UNITS; ++k) {
            const float wv = p.whp[k * OUT_FLOATS + tid];
#pragma unroll
            for (int m = 0; m < TAIL_CAND; ++m) acc[m] = fmaf(l5s[m][k5 * L5_UNITS + k], wv, acc[m]);
        }
#pragma unroll
        for (int m = 0; m < TAIL_CAND; ++m) lgs[m][tid] = selu_f(acc[m]);  // selu on logits: model.py:586
    }
    __syncthreads();

    if (tid < TAIL_CAND * 4) {
        const int m = tid >> 2, k5 = tid & 3;
        const int off = k5 == 0 ? 0 : (k5 == 1 ? 21 : (k5 == 2 ? 24 : 57));
        const int cnt = k5 == 0 ? 21 : (k5 == 1 ? 3 : 33);
        if (n0 + m < p.n) {
            float mx = -INFINITY;
            for (int j = 0; j < cnt; ++j) mx = fmaxf(mx, lgs[m][off + j]);
            float sum = 0.0f;
            for (int j = 0; j < cnt; ++j) {
                const float e = __expf(lgs[m][off + j] - mx);
                lgs[m][off + j] = e;
                sum += e;
            }
            float *o = p.out + (size_t)(n0 + m) * OUT_FLOATS + off;
            for (int j = 0; j < cnt; ++j) o[j] = lgs[m][off + j] / sum;  // true division, as tf.nn.softmax
        }
    }
}

}  // namespace clair
