/*
 * Blocked CPU port (plain C, float32, AVX2 through GCC vector extensions, OpenMP) of Clair's call_var forward pass:
 * the timed CPU baseline of bench.py.
 *
 * TEST INFRASTRUCTURE ONLY, like clair_oracle.c next to it: never linked or loaded by the product path (clair_amd/).
 * clair_oracle.c is the CHECKER (one candidate at a time, the reference's operation order written out); this file is
 * the same graph arranged the way a CPU wants it, so that the "GPU vs CPU" line of the benchmark compares against a
 * port that uses the cores properly (the checker streams all 5.9 MB of W4 per candidate and saturates
 * at 3 k candidates/s on any number of threads).  It is validated against the checker in tests/test_oracle.py at the
 * same 1e-5 tolerance as the HIP path (different summation order; exp through a polynomial).
 *
 * Structure: every OpenMP task owns a block of PB = 48 candidates and runs the whole network on it; every matrix
 * product is  C[48 x N] += A[48 x K] . W[K x N]  on a 6 x 16 register tile (12 accumulators + 2 weight vectors + 1
 * broadcast in the 16 ymm registers), weights re-packed once per call into [N/16][K][16] panels.  Reference sites as in
 * clair_oracle.c: clair/model.py:299-312, 423-451 (BiLSTM x2), :225-244, 464-479 (slice dense + flatten u*256+c),
 * :482-488 (L4), :507-569 (L5_k), :582-620 (heads: selu, softmax); clair/selu.py:26-30.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define T 33
#define FIN 32
#define H 128
#define G4 512
#define L3U 30
#define L3P 32 /* L3 units padded to whole 16-column panels */
#define L4U 192
#define L5U 96
#define PB 48 /* candidates per block: 8 row tiles of 6 */
#define KC 1920 /* K chunk of the 7680 -> 192 product: the A chunk (48 x 1920 floats) stays in L2 */

typedef float v8 __attribute__((vector_size(32), aligned(4)));

static const int HEAD_SIZE[4] = {21, 3, 33, 33};
static const int HEAD_PAD[4] = {32, 16, 48, 48};

/* C[6 x 16] (+)= A[6 x K] . Bp[K x 16];  A element (r, k) at A[r * lda + k * ka] */
static inline void mk6x16(int K, const float *A, long lda, long ka, const float *Bp, float *C, long ldc, int accumulate) {
    v8 c00, c01, c10, c11, c20, c21, c30, c31, c40, c41, c50, c51;
    if (accumulate) {
        c00 = *(const v8 *)(C + 0 * ldc); c01 = *(const v8 *)(C + 0 * ldc + 8);
        c10 = *(const v8 *)(C + 1 * ldc); c11 = *(const v8 *)(C + 1 * ldc + 8);
        c20 = *(const v8 *)(C + 2 * ldc); c21 = *(const v8 *)(C + 2 * ldc + 8);
        c30 = *(const v8 *)(C + 3 * ldc); c31 = *(const v8 *)(C + 3 * ldc + 8);
        c40 = *(const v8 *)(C + 4 * ldc); c41 = *(const v8 *)(C + 4 * ldc + 8);
        c50 = *(const v8 *)(C + 5 * ldc); c51 = *(const v8 *)(C + 5 * ldc + 8);
    } else {
        const v8 z = {0, 0, 0, 0, 0, 0, 0, 0};
        c00 = c01 = c10 = c11 = c20 = c21 = c30 = c31 = c40 = c41 = c50 = c51 = z;
    }
    const float *a0 = A, *a1 = A + lda, *a2 = A + 2 * lda, *a3 = A + 3 * lda, *a4 = A + 4 * lda, *a5 = A + 5 * lda;
    for (int k = 0; k < K; ++k) {
        const v8 b0 = *(const v8 *)(Bp + 16 * k), b1 = *(const v8 *)(Bp + 16 * k + 8);
        const long o = k * ka;
        v8 a;
#define ROW(ap, ca, cb) a = (v8){ap[o], ap[o], ap[o], ap[o], ap[o], ap[o], ap[o], ap[o]}; ca += a * b0; cb += a * b1;
        ROW(a0, c00, c01) ROW(a1, c10, c11) ROW(a2, c20, c21) ROW(a3, c30, c31) ROW(a4, c40, c41) ROW(a5, c50, c51)
#undef ROW
    }
    *(v8 *)(C + 0 * ldc) = c00; *(v8 *)(C + 0 * ldc + 8) = c01;
    *(v8 *)(C + 1 * ldc) = c10; *(v8 *)(C + 1 * ldc + 8) = c11;
    *(v8 *)(C + 2 * ldc) = c20; *(v8 *)(C + 2 * ldc + 8) = c21;
    *(v8 *)(C + 3 * ldc) = c30; *(v8 *)(C + 3 * ldc + 8) = c31;
    *(v8 *)(C + 4 * ldc) = c40; *(v8 *)(C + 4 * ldc + 8) = c41;
    *(v8 *)(C + 5 * ldc) = c50; *(v8 *)(C + 5 * ldc + 8) = c51;
}

/* C[PB x 16*np] (+)= A[PB x K] . panels */
static void gemm_pb(int K, const float *A, long lda, long ka, const float *Bp, int np, float *C, long ldc, int accumulate) {
    for (int j = 0; j < np; ++j)
        for (int i = 0; i < PB; i += 6)
            mk6x16(K, A + i * lda, lda, ka, Bp + (size_t)j * K * 16, C + i * ldc + 16 * j, ldc, accumulate);
}

/* W[K x N] row-major (row stride ldw) -> [ceil(N/16)][K][16], zero padded */
static float *pack_panels(const float *W, int K, int N, long ldw) {
    const int np = (N + 15) / 16;
    float *p = aligned_alloc(64, ((size_t)np * K * 16 * sizeof(float) + 63) / 64 * 64);
    if (!p) return 0;
    for (int j = 0; j < np; ++j)
        for (int k = 0; k < K; ++k)
            for (int c = 0; c < 16; ++c) {
                const int col = 16 * j + c;
                p[((size_t)j * K + k) * 16 + c] = col < N ? W[(size_t)k * ldw + col] : 0.0f;
            }
    return p;
}

/* exp for the gates and selu: Cephes-style range reduction + degree-5 polynomial, <= 2 ulp; written so that the
 * compiler vectorises the surrounding loops (no libm call, no branches) */
static inline float exp_f(float x) {
    x = x < -87.0f ? -87.0f : (x > 88.0f ? 88.0f : x);
    const float t = x * 1.44269504088896341f;
    const float fn = (t + 12582912.0f) - 12582912.0f; /* round to nearest integer */
    float r = x - fn * 0.693359375f;
    r = r - fn * -2.12194440e-4f;
    float p = 1.9875691500e-4f;
    p = p * r + 1.3981999507e-3f;
    p = p * r + 8.3334519073e-3f;
    p = p * r + 4.1665795894e-2f;
    p = p * r + 1.6666665459e-1f;
    p = p * r + 5.0000001201e-1f;
    p = p * r * r + r + 1.0f;
    union { int32_t i; float f; } u;
    u.i = ((int32_t)fn + 127) << 23;
    return p * u.f;
}
static inline float sigmoid_f(float x) { return 1.0f / (1.0f + exp_f(-x)); }
static inline float tanh_f(float x) { return 1.0f - 2.0f / (1.0f + exp_f(2.0f * x)); }
static inline float selu_f(float x) { /* clair/selu.py:26-30 */
    const float alpha = 1.6732632423543772848170429916717f, scale = 1.0507009873554804934193349852946f;
    float p = x * (1.0f / 5040.0f) + (1.0f / 720.0f); /* expm1 on (-0.25, 0] by its series, exp - 1 below */
    p = p * x + (1.0f / 120.0f);
    p = p * x + (1.0f / 24.0f);
    p = p * x + (1.0f / 6.0f);
    p = p * x + 0.5f;
    p = p * x + 1.0f;
    const float em1 = x > -0.25f ? p * x : exp_f(x) - 1.0f;
    return scale * (x >= 0.0f ? x : alpha * em1);
}

typedef struct {
    float *lstm[4][2]; /* [layer*2 + dir][x-part | h-part] panels */
    const float *bias[4];
    float *w3;         /* [256][2 panels][33][16] */
    float *w4;         /* [12][7680][16] */
    float *w5[4], *wh[4];
} Packed;

/* one direction of one layer on a block: in element (q, t, k) at in[q * in_q + t * in_t + k]; out[t][q][256] */
static void lstm_dir(const Packed *P, int ld, const float *in, long in_q, long in_t, int D, int reverse, float *out, int dir,
                     float *z /* [PB][512] */, float *h /* [PB][128] */, float *c /* [PB][128] */) {
    const float *bias = P->bias[ld];
    memset(h, 0, sizeof(float) * PB * H);
    memset(c, 0, sizeof(float) * PB * H);
    for (int s = 0; s < T; ++s) {
        const int t = reverse ? T - 1 - s : s;
        for (int q = 0; q < PB; ++q) memcpy(z + q * G4, bias, G4 * sizeof(float));
        gemm_pb(D, in + t * in_t, in_q, 1, P->lstm[ld][0], G4 / 16, z, G4, 1);
        gemm_pb(H, h, H, 1, P->lstm[ld][1], G4 / 16, z, G4, 1);
        for (int q = 0; q < PB; ++q) {
            const float *zq = z + q * G4;
            float *hq = h + q * H, *cq = c + q * H, *o = out + ((size_t)t * PB + q) * 256 + dir * H;
            for (int j = 0; j < H; ++j) { /* TF 1.13 LSTMBlockCell: i, c~, f, o */
                const float ig = sigmoid_f(zq[j]), gg = tanh_f(zq[H + j]), fg = sigmoid_f(zq[2 * H + j]), og = sigmoid_f(zq[3 * H + j]);
                const float cn = fg * cq[j] + ig * gg;
                cq[j] = cn;
                hq[j] = og * tanh_f(cn);
                o[j] = hq[j];
            }
        }
    }
}

int clair_cpu_port_forward(const float *const *w, const float *x, int n, float *gt21, float *gt, float *len1, float *len2,
                           int threads) {
    if (n < 0 || !w || (!x && n)) return 1;
    float *outs[4] = {gt21, gt, len1, len2};
    Packed P;
    memset(&P, 0, sizeof P);
    int fail = 0;
    for (int ld = 0; ld < 4; ++ld) {
        const int D = ld < 2 ? FIN : 256;
        P.lstm[ld][0] = pack_panels(w[2 * ld], D, G4, G4);
        P.lstm[ld][1] = pack_panels(w[2 * ld] + (size_t)D * G4, H, G4, G4);
        P.bias[ld] = w[2 * ld + 1];
        fail |= !P.lstm[ld][0] || !P.lstm[ld][1];
    }
    P.w3 = aligned_alloc(64, (size_t)256 * 2 * T * 16 * sizeof(float));
    fail |= !P.w3;
    if (P.w3)
        for (int c = 0; c < 256; ++c)
            for (int j = 0; j < 2; ++j)
                for (int t = 0; t < T; ++t)
                    for (int u = 0; u < 16; ++u)
                        P.w3[(((size_t)c * 2 + j) * T + t) * 16 + u] = 16 * j + u < L3U ? w[8][((size_t)c * T + t) * L3U + 16 * j + u] : 0.0f;
    P.w4 = pack_panels(w[10], L3U * 256, L4U, L4U);
    fail |= !P.w4;
    for (int k = 0; k < 4; ++k) {
        P.w5[k] = pack_panels(w[12] + (size_t)k * L4U * L5U, L4U, L5U, L5U);
        P.wh[k] = pack_panels(w[14 + 2 * k], L5U, HEAD_SIZE[k], HEAD_SIZE[k]);
        fail |= !P.w5[k] || !P.wh[k];
    }
    const int nblocks = (n + PB - 1) / PB;
#ifdef _OPENMP
    omp_set_num_threads(threads > 0 ? threads : omp_get_num_procs());
#endif
    if (!fail) {
#pragma omp parallel
        {
            /* one workspace per OpenMP thread, kept for the life of the process: 5 MB per thread allocated and released on
             * every call is 1.3 GB of page faults per call on a 256-thread host, all serialised in the kernel */
            static __thread float *ws = 0;
            const size_t ws_floats = (size_t)PB * T * FIN + 2 * (size_t)T * PB * 256 + (size_t)PB * G4 + 2 * (size_t)PB * H +
                                     (size_t)PB * L3U * 256 + (size_t)PB * L3P + (size_t)PB * L4U + (size_t)PB * L5U + (size_t)PB * 48;
            if (!ws) ws = aligned_alloc(64, (ws_floats * sizeof(float) + 63) / 64 * 64);
            float *xb = ws, *a1 = 0, *a2 = 0, *z = 0, *h = 0, *c = 0, *l3 = 0, *t3 = 0, *l4 = 0, *l5 = 0, *lg = 0;
            if (ws) {
                a1 = xb + (size_t)PB * T * FIN;
                a2 = a1 + (size_t)T * PB * 256;
                z = a2 + (size_t)T * PB * 256;
                h = z + (size_t)PB * G4;
                c = h + (size_t)PB * H;
                l3 = c + (size_t)PB * H;
                t3 = l3 + (size_t)PB * L3U * 256;
                l4 = t3 + (size_t)PB * L3P;
                l5 = l4 + (size_t)PB * L4U;
                lg = l5 + (size_t)PB * L5U;
            }
            if (!xb || !a1 || !a2 || !z || !h || !c || !l3 || !t3 || !l4 || !l5 || !lg) {
#pragma omp atomic write
                fail = 1;
            } else {
#pragma omp for schedule(dynamic, 1)
                for (int blk = 0; blk < nblocks; ++blk) {
                    const int n0 = blk * PB, nb = n - n0 < PB ? n - n0 : PB;
                    memcpy(xb, x + (size_t)n0 * T * FIN, sizeof(float) * nb * T * FIN);
                    if (nb < PB) memset(xb + (size_t)nb * T * FIN, 0, sizeof(float) * (PB - nb) * T * FIN);
                    lstm_dir(&P, 0, xb, T * FIN, FIN, FIN, 0, a1, 0, z, h, c);
                    lstm_dir(&P, 1, xb, T * FIN, FIN, FIN, 1, a1, 1, z, h, c);
                    lstm_dir(&P, 2, a1, 256, (long)PB * 256, 256, 0, a2, 0, z, h, c);
                    lstm_dir(&P, 3, a1, 256, (long)PB * 256, 256, 1, a2, 1, z, h, c);
                    /* L3: per channel c a [PB x 33] . [33 x 30] product, A element (q, t) = a2[t][q][c]; flat index u*256 + c */
                    for (int ch = 0; ch < 256; ++ch) {
                        gemm_pb(T, a2 + ch, 256, (long)PB * 256, P.w3 + (size_t)ch * 2 * T * 16, 2, t3, L3P, 0);
                        const float *b3 = w[9] + ch * L3U;
                        for (int q = 0; q < PB; ++q)
                            for (int u = 0; u < L3U; ++u) l3[(size_t)q * (L3U * 256) + u * 256 + ch] = selu_f(t3[q * L3P + u] + b3[u]);
                    }
                    /* L4: K = 7680 in chunks so that the A chunk is re-read from L2 by the 12 column panels */
                    for (int q = 0; q < PB; ++q) memcpy(l4 + q * L4U, w[11], L4U * sizeof(float));
                    for (int k0 = 0; k0 < L3U * 256; k0 += KC)
                        for (int j = 0; j < L4U / 16; ++j)
                            for (int i = 0; i < PB; i += 6)
                                mk6x16(KC, l3 + (size_t)i * (L3U * 256) + k0, L3U * 256, 1, P.w4 + ((size_t)j * (L3U * 256) + k0) * 16,
                                       l4 + i * L4U + 16 * j, L4U, 1);
                    for (int i = 0; i < PB * L4U; ++i) l4[i] = selu_f(l4[i]);
                    for (int k5 = 0; k5 < 4; ++k5) {
                        const float *b5 = w[13] + k5 * L5U, *bh = w[15 + 2 * k5];
                        const int m = HEAD_SIZE[k5], mp = HEAD_PAD[k5];
                        gemm_pb(L4U, l4, L4U, 1, P.w5[k5], L5U / 16, l5, L5U, 0);
                        for (int q = 0; q < PB; ++q)
                            for (int j = 0; j < L5U; ++j) l5[q * L5U + j] = selu_f(l5[q * L5U + j] + b5[j]);
                        gemm_pb(L5U, l5, L5U, 1, P.wh[k5], mp / 16, lg, 48, 0);
                        for (int q = 0; q < nb; ++q) {
                            float *row = lg + q * 48, mx = -INFINITY, sum = 0.0f;
                            for (int j = 0; j < m; ++j) {
                                row[j] = selu_f(row[j] + bh[j]); /* selu on the logits: model.py:586 */
                                if (row[j] > mx) mx = row[j];
                            }
                            for (int j = 0; j < m; ++j) {
                                row[j] = expf(row[j] - mx);
                                sum += row[j];
                            }
                            float *o = outs[k5] + (size_t)(n0 + q) * m;
                            for (int j = 0; j < m; ++j) o[j] = row[j] / sum;
                        }
                    }
                }
            }
        }
    }
    for (int ld = 0; ld < 4; ++ld) { free(P.lstm[ld][0]); free(P.lstm[ld][1]); }
    free(P.w3); free(P.w4);
    for (int k = 0; k < 4; ++k) { free(P.w5[k]); free(P.wh[k]); }
    return fail;
}
