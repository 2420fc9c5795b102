/*
 * CPU restatement (plain C, float32) of Clair's call_var forward pass.
 *
 * TEST INFRASTRUCTURE ONLY: used by tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg as the checker / the timed CPU port.  Never linked or loaded by the
 * product path (clair_amd/).
 *
 * PARITY UNPINNED: the reference's arithmetic for this path lives in TensorFlow 1.13.2
 * (README.md:127), which is absent from /root/reference and from this image; the
 * reference holds no golden vectors.  This file restates the published semantics of the
 * TF ops at the reference's call sites (same citations as oracle/model_np.py, against
 * which it is checked in tests/test_oracle.py):
 *
 *   clair/model.py:403-418  input reshape/transposition      -> rows of 32 features per position
 *   clair/model.py:299-312  CudnnCompatibleLSTMCell x stack_bidirectional_dynamic_rnn
 *   clair/model.py:423-451  LSTM1, LSTM2
 *   clair/model.py:225-244, 464-479  slice dense L3 + flatten (u*256+c)
 *   clair/model.py:482-488  L4;  :507-569  L5_1..4;  :582-620  heads + softmax
 *   clair/selu.py:26-30     selu
 *
 * Weight pointers follow the tensor-id order of include/clair_amd.h.
 */
#include <math.h>
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
#define L4U 192
#define L5U 96
#define BS 8 /* candidates per inner block */

static const int HEAD_SIZE[4] = {21, 3, 33, 33};
static const int HEAD_OFF[4] = {0, 21, 24, 57};

static inline float selu_f(float x) { /* clair/selu.py:26-30 */
    const float alpha = 1.6732632423543772848170429916717f;
    const float scale = 1.0507009873554804934193349852946f;
    return scale * (x >= 0.0f ? x : alpha * expm1f(x));
}
static inline float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }

/* one direction of one BiLSTM layer for a block of nb<=BS candidates.
 * in:  [T][nb][D]   out: [T][nb][256] (writes columns dir*128..dir*128+127)
 * TF 1.13 LSTMBlockCell: z=[x,h].W+b ; (i, ci, f, o) ; cs = tanh(ci)*sig(i) + cs_prev*sig(f) ; h = tanh(cs)*sig(o) */
static void lstm_dir(const float *in, int D, int nb, const float *W, const float *b, int reverse,
                     float *out, int dir, float *z /* [BS][512] */) {
    float h[BS][H], c[BS][H];
    memset(h, 0, sizeof h);
    memset(c, 0, sizeof c);
    for (int s = 0; s < T; ++s) {
        int t = reverse ? T - 1 - s : s;
        for (int q = 0; q < nb; ++q) memcpy(z + q * G4, b, G4 * sizeof(float));
        for (int k = 0; k < D; ++k) {
            const float *wr = W + (size_t)k * G4;
            for (int q = 0; q < nb; ++q) {
                float a = in[((size_t)t * nb + q) * D + k];
                float *zq = z + q * G4;
                for (int j = 0; j < G4; ++j) zq[j] += a * wr[j];
            }
        }
        for (int k = 0; k < H; ++k) {
            const float *wr = W + (size_t)(D + k) * G4;
            for (int q = 0; q < nb; ++q) {
                float a = h[q][k];
                float *zq = z + q * G4;
                for (int j = 0; j < G4; ++j) zq[j] += a * wr[j];
            }
        }
        for (int q = 0; q < nb; ++q) {
            const float *zq = z + q * G4;
            float *o = out + ((size_t)t * nb + q) * 256 + dir * H;
            for (int j = 0; j < H; ++j) {
                float ig = sigmoid_f(zq[j]), gg = tanhf(zq[H + j]);
                float fg = sigmoid_f(zq[2 * H + j]), og = sigmoid_f(zq[3 * H + j]);
                float cn = fg * c[q][j] + ig * gg;
                c[q][j] = cn;
                h[q][j] = og * tanhf(cn);
                o[j] = h[q][j];
            }
        }
    }
}

/* weights[22] in include/clair_amd.h tensor-id order; x [n][33][32]; outputs [n][21],[n][3],[n][33],[n][33].
 * If a1_out / a2_out / l3_out / l4_out are non-NULL they receive intermediates
 * ([n][33][256], [n][33][256], [n][7680], [n][192]) for layer-wise parity tests. */
int clair_oracle_forward_ex(const float *const *w, const float *x, int n, float *gt21, float *gt,
                            float *len1, float *len2, float *a1_out, float *a2_out, float *l3_out,
                            float *l4_out, int threads) {
    if (n < 0 || !w || (!x && n)) return 1;
    float *outs[4] = {gt21, gt, len1, len2};
    int nblocks = (n + BS - 1) / BS;
#ifdef _OPENMP
    omp_set_num_threads(threads > 0 ? threads : omp_get_num_procs());
#endif
    int fail = 0;
#pragma omp parallel
    {
        float *s = malloc(sizeof(float) * T * BS * FIN);
        float *a1 = malloc(sizeof(float) * T * BS * 256);
        float *a2 = malloc(sizeof(float) * T * BS * 256);
        float *z = malloc(sizeof(float) * BS * G4);
        float *l3 = malloc(sizeof(float) * BS * L3U * 256);
        if (!s || !a1 || !a2 || !z || !l3) {
#pragma omp atomic write
            fail = 1;
        } else {
#pragma omp for schedule(dynamic, 1)
            for (int blk = 0; blk < nblocks; ++blk) {
                int n0 = blk * BS, nb = n - n0 < BS ? n - n0 : BS;
                /* [n][33][32] -> time-major [T][nb][32] (model.py:416-418) */
                for (int t = 0; t < T; ++t)
                    for (int q = 0; q < nb; ++q)
                        memcpy(s + ((size_t)t * nb + q) * FIN, x + ((size_t)(n0 + q) * T + t) * FIN,
                               FIN * sizeof(float));
                lstm_dir(s, FIN, nb, w[0], w[1], 0, a1, 0, z);
                lstm_dir(s, FIN, nb, w[2], w[3], 1, a1, 1, z);
                lstm_dir(a1, 256, nb, w[4], w[5], 0, a2, 0, z);
                lstm_dir(a1, 256, nb, w[6], w[7], 1, a2, 1, z);
                for (int q = 0; q < nb; ++q) {
                    float *l3q = l3 + (size_t)q * L3U * 256;
                    /* L3: l3[u*256+c] = selu(sum_t a2[t][q][c]*W3[c][t][u] + b3[c][u]) */
                    for (int c = 0; c < 256; ++c) {
                        float acc[L3U];
                        for (int u = 0; u < L3U; ++u) acc[u] = w[9][c * L3U + u];
                        for (int t = 0; t < T; ++t) {
                            float a = a2[((size_t)t * nb + q) * 256 + c];
                            const float *wr = w[8] + ((size_t)c * T + t) * L3U;
                            for (int u = 0; u < L3U; ++u) acc[u] += a * wr[u];
                        }
                        for (int u = 0; u < L3U; ++u) l3q[u * 256 + c] = selu_f(acc[u]);
                    }
                    float l4[L4U], l5[L5U], lg[33];
                    for (int j = 0; j < L4U; ++j) l4[j] = w[11][j];
                    for (int k = 0; k < L3U * 256; ++k) {
                        float a = l3q[k];
                        const float *wr = w[10] + (size_t)k * L4U;
                        for (int j = 0; j < L4U; ++j) l4[j] += a * wr[j];
                    }
                    for (int j = 0; j < L4U; ++j) l4[j] = selu_f(l4[j]);
                    for (int k5 = 0; k5 < 4; ++k5) {
                        const float *W5 = w[12] + (size_t)k5 * L4U * L5U, *b5 = w[13] + k5 * L5U;
                        for (int j = 0; j < L5U; ++j) l5[j] = b5[j];
                        for (int k = 0; k < L4U; ++k)
                            for (int j = 0; j < L5U; ++j) l5[j] += l4[k] * W5[k * L5U + j];
                        for (int j = 0; j < L5U; ++j) l5[j] = selu_f(l5[j]);
                        int m = HEAD_SIZE[k5];
                        const float *Wh = w[14 + 2 * k5], *bh = w[15 + 2 * k5];
                        for (int j = 0; j < m; ++j) lg[j] = bh[j];
                        for (int k = 0; k < L5U; ++k)
                            for (int j = 0; j < m; ++j) lg[j] += l5[k] * Wh[k * m + j];
                        float mx = -INFINITY, sum = 0.0f;
                        for (int j = 0; j < m; ++j) {
                            lg[j] = selu_f(lg[j]); /* selu on the logits: model.py:586 */
                            if (lg[j] > mx) mx = lg[j];
                        }
                        for (int j = 0; j < m; ++j) {
                            lg[j] = expf(lg[j] - mx);
                            sum += lg[j];
                        }
                        float *o = outs[k5] + (size_t)(n0 + q) * m;
                        for (int j = 0; j < m; ++j) o[j] = lg[j] / sum;
                    }
                    if (l4_out) memcpy(l4_out + (size_t)(n0 + q) * L4U, l4, sizeof l4);
                    if (l3_out) memcpy(l3_out + (size_t)(n0 + q) * L3U * 256, l3q, sizeof(float) * L3U * 256);
                    for (int t = 0; t < T; ++t) {
                        if (a1_out)
                            memcpy(a1_out + ((size_t)(n0 + q) * T + t) * 256, a1 + ((size_t)t * nb + q) * 256,
                                   256 * sizeof(float));
                        if (a2_out)
                            memcpy(a2_out + ((size_t)(n0 + q) * T + t) * 256, a2 + ((size_t)t * nb + q) * 256,
                                   256 * sizeof(float));
                    }
                }
            }
        }
        free(s); free(a1); free(a2); free(z); free(l3);
    }
    (void)HEAD_OFF;
    return fail;
}

int clair_oracle_forward(const float *const *w, const float *x, int n, float *gt21, float *gt, float *len1,
                         float *len2, int threads) {
    return clair_oracle_forward_ex(w, x, n, gt21, gt, len1, len2, 0, 0, 0, 0, threads);
}

int clair_oracle_max_threads(void) {
#ifdef _OPENMP
    return omp_get_num_procs();
#else
    return 1;
#endif
}
