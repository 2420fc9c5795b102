/* Body of the CPU restatement, compiled twice by clair_oracle.c: REAL = float (the checker: every value and every
 * operation in float32, as the reference's tf.float32 graph) and REAL = double (an evaluation of the same graph in
 * float64 from the same float32 weights and inputs: the yardstick for how much float32 rounding itself moves the
 * outputs of a given weight set).  TEST INFRASTRUCTURE ONLY -- see clair_oracle.c. */
static inline REAL NAME(selu)(REAL x) { /* clair/selu.py:26-30 */
    const REAL alpha = (REAL)1.6732632423543772848170429916717;
    const REAL scale = (REAL)1.0507009873554804934193349852946;
    return scale * (x >= (REAL)0 ? x : alpha * M_EXPM1(x));
}
static inline REAL NAME(sigmoid)(REAL x) { return (REAL)1 / ((REAL)1 + M_EXP(-x)); }

/* one direction of one BiLSTM layer for a block of nb<=BS candidates.
 * in:  [T][nb][D]   out: [T][nb][256] (writes columns dir*128..dir*128+127)
 * TF 1.13 LSTMBlockCell: z=[x,h].W+b ; (i, ci, f, o) ; cs = tanh(ci)*sig(i) + cs_prev*sig(f) ; h = tanh(cs)*sig(o) */
static void NAME(lstm_dir)(const REAL *in, int D, int nb, const float *W, const float *b, int reverse,
                     REAL *out, int dir, REAL *z /* [BS][512] */) {
    REAL h[BS][H], c[BS][H];
    memset(h, 0, sizeof h);
    memset(c, 0, sizeof c);
    for (int s = 0; s < T; ++s) {
        int t = reverse ? T - 1 - s : s;
        for (int q = 0; q < nb; ++q)
            for (int j = 0; j < G4; ++j) z[q * G4 + j] = (REAL)b[j];
        for (int k = 0; k < D; ++k) {
            const float *wr = W + (size_t)k * G4;
            for (int q = 0; q < nb; ++q) {
                REAL a = in[((size_t)t * nb + q) * D + k];
                REAL *zq = z + q * G4;
                for (int j = 0; j < G4; ++j) zq[j] += a * (REAL)wr[j];
            }
        }
        for (int k = 0; k < H; ++k) {
            const float *wr = W + (size_t)(D + k) * G4;
            for (int q = 0; q < nb; ++q) {
                REAL a = h[q][k];
                REAL *zq = z + q * G4;
                for (int j = 0; j < G4; ++j) zq[j] += a * (REAL)wr[j];
            }
        }
        for (int q = 0; q < nb; ++q) {
            const REAL *zq = z + q * G4;
            REAL *o = out + ((size_t)t * nb + q) * 256 + dir * H;
            for (int j = 0; j < H; ++j) {
                REAL ig = NAME(sigmoid)(zq[j]), gg = M_TANH(zq[H + j]);
                REAL fg = NAME(sigmoid)(zq[2 * H + j]), og = NAME(sigmoid)(zq[3 * H + j]);
                REAL cn = fg * c[q][j] + ig * gg;
                c[q][j] = cn;
                h[q][j] = og * M_TANH(cn);
                o[j] = h[q][j];
            }
        }
    }
}

/* weights[22] in include/clair_amd.h tensor-id order; x [n][33][32]; outputs [n][21],[n][3],[n][33],[n][33].
 * If a1_out / a2_out / l3_out / l4_out are non-NULL they receive intermediates
 * ([n][33][256], [n][33][256], [n][7680], [n][192]) for layer-wise parity tests. */
int NAME(clair_oracle_forward_ex)(const float *const *w, const float *x, int n, REAL *gt21, REAL *gt,
                            REAL *len1, REAL *len2, REAL *a1_out, REAL *a2_out, REAL *l3_out,
                            REAL *l4_out, int threads) {
    if (n < 0 || !w || (!x && n)) return 1;
    REAL *outs[4] = {gt21, gt, len1, len2};
    int nblocks = (n + BS - 1) / BS;
#ifdef _OPENMP
    omp_set_num_threads(threads > 0 ? threads : omp_get_num_procs());
#endif
    int fail = 0;
#pragma omp parallel
    {
        REAL *s = malloc(sizeof(REAL) * T * BS * FIN);
        REAL *a1 = malloc(sizeof(REAL) * T * BS * 256);
        REAL *a2 = malloc(sizeof(REAL) * T * BS * 256);
        REAL *z = malloc(sizeof(REAL) * BS * G4);
        REAL *l3 = malloc(sizeof(REAL) * BS * L3U * 256);
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
                        for (int f = 0; f < FIN; ++f)
                            s[((size_t)t * nb + q) * FIN + f] = (REAL)x[((size_t)(n0 + q) * T + t) * FIN + f];
                NAME(lstm_dir)(s, FIN, nb, w[0], w[1], 0, a1, 0, z);
                NAME(lstm_dir)(s, FIN, nb, w[2], w[3], 1, a1, 1, z);
                NAME(lstm_dir)(a1, 256, nb, w[4], w[5], 0, a2, 0, z);
                NAME(lstm_dir)(a1, 256, nb, w[6], w[7], 1, a2, 1, z);
                for (int q = 0; q < nb; ++q) {
                    REAL *l3q = l3 + (size_t)q * L3U * 256;
                    /* L3: l3[u*256+c] = selu(sum_t a2[t][q][c]*W3[c][t][u] + b3[c][u]) */
                    for (int c = 0; c < 256; ++c) {
                        REAL acc[L3U];
                        for (int u = 0; u < L3U; ++u) acc[u] = w[9][c * L3U + u];
                        for (int t = 0; t < T; ++t) {
                            REAL a = a2[((size_t)t * nb + q) * 256 + c];
                            const float *wr = w[8] + ((size_t)c * T + t) * L3U;
                            for (int u = 0; u < L3U; ++u) acc[u] += a * (REAL)wr[u];
                        }
                        for (int u = 0; u < L3U; ++u) l3q[u * 256 + c] = NAME(selu)(acc[u]);
                    }
                    REAL l4[L4U], l5[L5U], lg[33];
                    for (int j = 0; j < L4U; ++j) l4[j] = w[11][j];
                    for (int k = 0; k < L3U * 256; ++k) {
                        REAL a = l3q[k];
                        const float *wr = w[10] + (size_t)k * L4U;
                        for (int j = 0; j < L4U; ++j) l4[j] += a * (REAL)wr[j];
                    }
                    for (int j = 0; j < L4U; ++j) l4[j] = NAME(selu)(l4[j]);
                    for (int k5 = 0; k5 < 4; ++k5) {
                        const float *W5 = w[12] + (size_t)k5 * L4U * L5U, *b5 = w[13] + k5 * L5U;
                        for (int j = 0; j < L5U; ++j) l5[j] = b5[j];
                        for (int k = 0; k < L4U; ++k)
                            for (int j = 0; j < L5U; ++j) l5[j] += l4[k] * (REAL)W5[k * L5U + j];
                        for (int j = 0; j < L5U; ++j) l5[j] = NAME(selu)(l5[j]);
                        int m = HEAD_SIZE[k5];
                        const float *Wh = w[14 + 2 * k5], *bh = w[15 + 2 * k5];
                        for (int j = 0; j < m; ++j) lg[j] = bh[j];
                        for (int k = 0; k < L5U; ++k)
                            for (int j = 0; j < m; ++j) lg[j] += l5[k] * (REAL)Wh[k * m + j];
                        REAL mx = -INFINITY, sum = (REAL)0;
                        for (int j = 0; j < m; ++j) {
                            lg[j] = NAME(selu)(lg[j]); /* selu on the logits: model.py:586 */
                            if (lg[j] > mx) mx = lg[j];
                        }
                        for (int j = 0; j < m; ++j) {
                            lg[j] = M_EXP(lg[j] - mx);
                            sum += lg[j];
                        }
                        REAL *o = outs[k5] + (size_t)(n0 + q) * m;
                        for (int j = 0; j < m; ++j) o[j] = lg[j] / sum;
                    }
                    if (l4_out) memcpy(l4_out + (size_t)(n0 + q) * L4U, l4, sizeof l4);
                    if (l3_out) memcpy(l3_out + (size_t)(n0 + q) * L3U * 256, l3q, sizeof(REAL) * L3U * 256);
                    for (int t = 0; t < T; ++t) {
                        if (a1_out)
                            memcpy(a1_out + ((size_t)(n0 + q) * T + t) * 256, a1 + ((size_t)t * nb + q) * 256,
                                   256 * sizeof(REAL));
                        if (a2_out)
                            memcpy(a2_out + ((size_t)(n0 + q) * T + t) * 256, a2 + ((size_t)t * nb + q) * 256,
                                   256 * sizeof(REAL));
                    }
                }
            }
        }
        free(s); free(a1); free(a2); free(z); free(l3);
    }
    return fail;
}

