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
//
// One workgroup per 16-candidate tile; wave k owns branch k end to end (L5_k -> head k -> softmax),
// both products on v_mfma_f32_16x16x4_f32 with the weights streamed from L2 as pre-packed B
// fragments (engine.hip: pack_tail_weights).  K is visited as k = (lane>>4)*(K/4) + kk so a lane's A
// operands for four MFMAs are one ds_read_b128 of the LDS activation tile.
constexpr int TAIL_TILE = 16;
constexpr int L4S_ROW = L4_UNITS + 4;  // padded LDS rows (16 B aligned, conflict-free b128 reads)
constexpr int L5S_ROW = L5_UNITS + 4;

struct TailArgs {
    const float *l4part;  // [L4_SPLITS][n_pad][192]
    const float *b4;      // [192]
    const float *w5f;     // [4][12][6][64][4]  B fragments of L5_k: W5[k5][lq*48 + k4*4 + j][nb*16 + li]
    const float *b5;      // [4][96]
    const float *whf;     // [4][6][3][64][4]   B fragments of head k: Wh[lq*24 + k4*4 + j][nb*16 + li], 0-padded
    const float *bhf;     // [4][48]            head biases, 0-padded
    float *out;           // [n][90]
    int n_pad;
    int n;                // valid candidates
};

__global__ __launch_bounds__(256) void tail_kernel(TailArgs p) {
    __shared__ __attribute__((aligned(16))) float l4s[TAIL_TILE][L4S_ROW];
    __shared__ __attribute__((aligned(16))) float l5s[4][TAIL_TILE][L5S_ROW];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lq = lane >> 4;
    const int n0 = blockIdx.x * TAIL_TILE;

    // L4: fixed-order reduction of the split-K partials, bias, selu
    for (int f = tid; f < TAIL_TILE * (L4_UNITS / 4); f += 256) {
        const int m = f / (L4_UNITS / 4), j4 = f - m * (L4_UNITS / 4);
        f32x4 s = *(const f32x4 *)(p.b4 + j4 * 4);
#pragma unroll
        for (int sp = 0; sp < L4_SPLITS; ++sp)
            s += *(const f32x4 *)(p.l4part + ((size_t)sp * p.n_pad + n0 + m) * L4_UNITS + j4 * 4);
        f32x4 o = {selu_f(s[0]), selu_f(s[1]), selu_f(s[2]), selu_f(s[3])};
        *(f32x4 *)&l4s[m][j4 * 4] = o;
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
