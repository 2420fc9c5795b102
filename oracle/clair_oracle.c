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

#define REAL float
#define NAME(x) x
#define M_EXP expf
#define M_EXPM1 expm1f
#define M_TANH tanhf
#include "clair_oracle_impl.h"
#undef REAL
#undef NAME
#undef M_EXP
#undef M_EXPM1
#undef M_TANH

#define REAL double
#define NAME(x) x##_f64
#define M_EXP exp
#define M_EXPM1 expm1
#define M_TANH tanh
#include "clair_oracle_impl.h"
#undef REAL
#undef NAME
#undef M_EXP
#undef M_EXPM1
#undef M_TANH

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
