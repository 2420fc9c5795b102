"""CPU restatement (NumPy, float32) of Clair's call_var forward pass.

TEST INFRASTRUCTURE ONLY.  Nothing under ``clair_amd/`` may import this module;
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg use it, and only as the checker.

PARITY UNPINNED: the reference delegates all arithmetic of this path to
TensorFlow 1.13.2 (pinned in /root/reference/README.md:127), which is neither
vendored under /root/reference nor installable here, and the reference ships no
tests, golden vectors or checkpoints.  This file restates the *published*
semantics of the TF ops the reference calls, at the reference's own call sites:

  clair/model.py:403-418   reshape [n,33,8,4] -> [n,33,32], transpose to time-major
  clair/model.py:299-312   CudnnCompatibleLSTMCell(128) x stack_bidirectional_dynamic_rnn
                           (TF 1.13 LSTMBlockCell: gate order i, c~, f, o; forget_bias 0;
                            no peephole; no clipping; zero initial state)
  clair/model.py:423-451   LSTM1 (in 32) and LSTM2 (in 256), outputs concat(fw, bw)
  clair/model.py:225-244   slice_dense_layer: 256 x dense(33 -> 30) over the position axis
  clair/model.py:464-479   L3 + flatten to 7680 (flat index u*256 + c)
  clair/model.py:482-488   L4 dense 7680 -> 192, selu
  clair/model.py:507-569   L5_1..4 dense 192 -> 96, selu (dropout_selu is identity at inference,
                           clair/selu.py:72-74)
  clair/model.py:582-620   heads dense 96 -> 21/3/33/33 with selu, then softmax
  clair/selu.py:26-30      selu

It is cross-checked against an independent torch-CPU implementation
(tests/test_oracle.py, tools/make_nn_golden.py) and against a float64 evaluation
of the same graph.
"""
import numpy as np

SELU_ALPHA = 1.6732632423543772848170429916717
SELU_SCALE = 1.0507009873554804934193349852946

T = 33          # positions (2*flankingBaseNum+1, shared/param.py:9)
F_IN = 32       # matrixRow*matrixNum (shared/param.py:10-11)
H = 128         # LSTM units per direction (clair/model.py:92-93)
L3_UNITS = 30   # clair/model.py:81
L4_UNITS = 192  # clair/model.py:82
L5_UNITS = 96   # clair/model.py:84-91
HEAD_SIZES = (21, 3, 33, 33)  # clair/task/main.py:10-29


def selu(x):
    """clair/selu.py:26-30 -- scale * where(x >= 0, x, alpha * elu(x))."""
    dt = x.dtype.type
    neg = dt(SELU_ALPHA) * np.expm1(np.minimum(x, dt(0)))
    return dt(SELU_SCALE) * np.where(x >= 0, x, neg)


def sigmoid(x):
    dt = x.dtype.type
    return dt(1) / (dt(1) + np.exp(-x))


def lstm_direction(inp, kernel, bias, reverse):
    """One direction of one layer.  inp [T, n, D]; kernel [D+H, 4H]; bias [4H].

    TF 1.13 LSTMBlockCell (reached via CudnnCompatibleLSTMCell, clair/model.py:301):
      z = [x, h_prev] . W + b ; i, ci, f, o = split(z, 4)
      cs = tanh(ci) * sigmoid(i) + cs_prev * sigmoid(f) ; h = tanh(cs) * sigmoid(o)
    The backward cell sees the time-reversed sequence and its outputs are reversed
    back (bidirectional_dynamic_rnn with sequence_length=None).
    """
    dt = inp.dtype
    n = inp.shape[1]
    h = np.zeros((n, H), dtype=dt)
    c = np.zeros((n, H), dtype=dt)
    out = np.empty((T, n, H), dtype=dt)
    kernel = kernel.astype(dt)
    bias = bias.astype(dt)
    order = range(T - 1, -1, -1) if reverse else range(T)
    for t in order:
        z = np.concatenate([inp[t], h], axis=1) @ kernel + bias
        i, g, f, o = z[:, 0:H], z[:, H:2 * H], z[:, 2 * H:3 * H], z[:, 3 * H:4 * H]
        c = sigmoid(f) * c + sigmoid(i) * np.tanh(g)
        h = sigmoid(o) * np.tanh(c)
        out[t] = h
    return out


def bilstm(inp, w, prefix):
    fw = lstm_direction(inp, w[prefix + "_fw_kernel"], w[prefix + "_fw_bias"], False)
    bw = lstm_direction(inp, w[prefix + "_bw_kernel"], w[prefix + "_bw_bias"], True)
    return np.concatenate([fw, bw], axis=2)


def forward(w, x, dtype=np.float32, keep_intermediates=False):
    """x: [n,33,8,4] (already ch1..3 -= ch0, clair/utils.py:96-98).

    Returns [gt21 [n,21], genotype [n,3], len1 [n,33], len2 [n,33]] in `dtype`
    (and a dict of intermediates when keep_intermediates).
    """
    x = np.asarray(x, dtype=dtype)
    n = x.shape[0]
    s = x.reshape(n, T, F_IN).transpose(1, 0, 2)                 # model.py:403-418
    a1 = bilstm(s, w, "lstm1")                                   # [T,n,256]
    a2 = bilstm(a1, w, "lstm2")                                  # [T,n,256]
    a2b = a2.transpose(1, 0, 2)                                  # [n,T,256] model.py:461
    # L3: for channel c: selu(a2b[:, :, c] @ W3[c] + b3[c])  -> [n,30,256]
    w3 = w["l3_kernel"].astype(dtype)                            # [256,33,30]
    b3 = w["l3_bias"].astype(dtype)                              # [256,30]
    l3 = selu(np.einsum("ntc,ctu->nuc", a2b, w3).astype(dtype) + b3.T[None])
    v = l3.reshape(n, L3_UNITS * 2 * H)                          # flat u*256+c, model.py:474-478
    l4 = selu(v @ w["l4_kernel"].astype(dtype) + w["l4_bias"].astype(dtype))
    outs, l5s = [], []
    names = ("gt21", "genotype", "len1", "len2")
    for k in range(4):
        l5 = selu(l4 @ w["l5_kernel"][k].astype(dtype) + w["l5_bias"][k].astype(dtype))
        l5s.append(l5)
        logit = selu(l5 @ w["head_%s_kernel" % names[k]].astype(dtype)
                     + w["head_%s_bias" % names[k]].astype(dtype))
        e = np.exp(logit - logit.max(axis=1, keepdims=True))
        outs.append((e / e.sum(axis=1, keepdims=True)).astype(dtype))
    if keep_intermediates:
        return outs, dict(a1=a1, a2=a2, l3=l3, l4=l4, l5=np.stack(l5s))
    return outs
