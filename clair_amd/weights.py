"""Weight tensors of Clair's inference graph: name table, synthetic initialiser, container I/O.

The reference keeps its parameters in a tf.train.Saver checkpoint
(/root/reference/clair/model.py:712, 1016-1020).  This module defines the canonical
tensor table the HIP engine consumes (ids match include/clair_amd.h), a mapping from
the TF variable names of the reference graph to that table, a fixed-seed synthetic
initialiser following the reference's initialisers (clair/model.py:394-398 and the
TF variable-scope default for the LSTM kernels), and a small ``.npz`` container.
"""
import os
from collections import OrderedDict

import numpy as np

T = 33
F_IN = 32
H = 128
L3_UNITS = 30
L4_UNITS = 192
L5_UNITS = 96
HEAD_NAMES = ("gt21", "genotype", "len1", "len2")
HEAD_SIZES = (21, 3, 33, 33)

# (key, shape) in C-ABI tensor-id order (include/clair_amd.h: enum clair_tensor_id)
TENSOR_TABLE = OrderedDict([
    ("lstm1_fw_kernel", (F_IN + H, 4 * H)), ("lstm1_fw_bias", (4 * H,)),
    ("lstm1_bw_kernel", (F_IN + H, 4 * H)), ("lstm1_bw_bias", (4 * H,)),
    ("lstm2_fw_kernel", (2 * H + H, 4 * H)), ("lstm2_fw_bias", (4 * H,)),
    ("lstm2_bw_kernel", (2 * H + H, 4 * H)), ("lstm2_bw_bias", (4 * H,)),
    ("l3_kernel", (2 * H, T, L3_UNITS)), ("l3_bias", (2 * H, L3_UNITS)),
    ("l4_kernel", (L3_UNITS * 2 * H, L4_UNITS)), ("l4_bias", (L4_UNITS,)),
    ("l5_kernel", (4, L4_UNITS, L5_UNITS)), ("l5_bias", (4, L5_UNITS)),
    ("head_gt21_kernel", (L5_UNITS, 21)), ("head_gt21_bias", (21,)),
    ("head_genotype_kernel", (L5_UNITS, 3)), ("head_genotype_bias", (3,)),
    ("head_len1_kernel", (L5_UNITS, 33)), ("head_len1_bias", (33,)),
    ("head_len2_kernel", (L5_UNITS, 33)), ("head_len2_bias", (33,)),
])
TENSOR_IDS = {k: i for i, k in enumerate(TENSOR_TABLE)}
N_PARAMS = sum(int(np.prod(s)) for s in TENSOR_TABLE.values())  # 2 377 818


def tf_variable_names():
    """TF-1.13 variable name -> (key, index-into-leading-axis or None).

    Names follow the scopes opened in clair/model.py: "LSTM1"/"LSTM2"
    (adaptive_LSTM_layer, :299-312), "L3/Unit_i" (slice_dense_layer, :238-243), "L4",
    "L5_k" (:482-569), "Prediction/Y_*_logits" (:581-620).  The LSTM part is what
    stack_bidirectional_dynamic_rnn + CudnnCompatibleLSTMCell create in TF 1.13.
    """
    m = {}
    for layer in (1, 2):
        for d in ("fw", "bw"):
            base = ("LSTM%d/stack_bidirectional_rnn/cell_0/bidirectional_rnn/%s/"
                    "cudnn_compatible_lstm_cell/" % (layer, d))
            m[base + "kernel"] = ("lstm%d_%s_kernel" % (layer, d), None)
            m[base + "bias"] = ("lstm%d_%s_bias" % (layer, d), None)
    for c in range(2 * H):
        m["L3/Unit_%d/kernel" % c] = ("l3_kernel", c)
        m["L3/Unit_%d/bias" % c] = ("l3_bias", c)
    m["L4/kernel"] = ("l4_kernel", None)
    m["L4/bias"] = ("l4_bias", None)
    for k in range(4):
        m["L5_%d/kernel" % (k + 1)] = ("l5_kernel", k)
        m["L5_%d/bias" % (k + 1)] = ("l5_bias", k)
    for head, tfname in zip(HEAD_NAMES, ("Y_base_change_logits", "Y_genotype_logits",
                                         "Y_indel_length_logits_1", "Y_indel_length_logits_2")):
        m["Prediction/%s/kernel" % tfname] = ("head_%s_kernel" % head, None)
        m["Prediction/%s/bias" % tfname] = ("head_%s_bias" % head, None)
    return m


def _truncated_normal(rng, shape, stddev):
    """tf.truncated_normal: resample values beyond 2 sigma."""
    out = rng.standard_normal(shape)
    bad = np.abs(out) > 2.0
    while bad.any():
        out[bad] = rng.standard_normal(int(bad.sum()))
        bad = np.abs(out) > 2.0
    return (out * stddev).astype(np.float32)


def synthetic_weights(seed=20250928, head_gain=1.0, lstm_bias_scale=0.0, lstm_gain=1.0, forget_bias=0.0,
                      input_gain=1.0, l4_gain=1.0, small_fraction=0.0, small_scale=1e-3):
    """Random-init weights of the reference architecture with a fixed seed.

    Dense kernels: variance_scaling_initializer(factor=1.0, mode='FAN_IN')
    (clair/model.py:394-398) = truncated normal with stddev sqrt(1.3/fan_in); dense
    biases zero.  LSTM kernels: no initialiser is passed on the CPU branch
    (clair/model.py:300-311), i.e. the variable-scope default Glorot-uniform; biases zero.
    ``head_gain`` scales the four head kernels so that the softmaxes become peaky and
    every branch of the VCF decode is exercised by synthetic data.

    The remaining switches shape the weights like a TRAINED model rather than a fresh one, for the parity sweep
    (tests/test_parity_gpu.py): ``lstm_gain`` scales the four LSTM kernels (saturating gates), ``forget_bias`` is added
    to the forget-gate biases (columns 256..383), ``input_gain`` scales the rows of the LSTM1 kernels that multiply the
    pileup counts (a net trained on counts of 50..250 keeps them small), ``l4_gain`` scales the 7680->192 kernel and
    divides the L5 kernels by the same factor (same function up to the selu in between, different magnitudes), and
    ``small_fraction`` of every kernel's entries are multiplied by ``small_scale`` (entries of 1e-4 magnitude: their
    fp16 low planes are subnormal).
    """
    rng = np.random.default_rng(seed)
    w = OrderedDict()
    for key, shape in TENSOR_TABLE.items():
        if key.endswith("_bias"):
            if key.startswith("lstm") and lstm_bias_scale:
                w[key] = (rng.standard_normal(shape) * lstm_bias_scale).astype(np.float32)
            else:
                w[key] = np.zeros(shape, dtype=np.float32)
        elif key.startswith("lstm"):
            limit = np.sqrt(6.0 / (shape[0] + shape[1]))
            w[key] = rng.uniform(-limit, limit, size=shape).astype(np.float32)
        else:
            fan_in = shape[-2]
            w[key] = _truncated_normal(rng, shape, np.sqrt(1.3 / fan_in))
            if key.startswith("head_"):
                w[key] *= np.float32(head_gain)
    for key in TENSOR_TABLE:
        if key.startswith("lstm") and key.endswith("_kernel"):
            w[key] *= np.float32(lstm_gain)
            if key.startswith("lstm1"):
                w[key][:F_IN] *= np.float32(input_gain)
        elif key.startswith("lstm") and forget_bias:
            w[key][2 * H:3 * H] += np.float32(forget_bias)
    if l4_gain != 1.0:
        w["l4_kernel"] *= np.float32(l4_gain)
        w["l5_kernel"] *= np.float32(1.0 / l4_gain)
    if small_fraction:
        srng = np.random.default_rng(seed + 977)
        for key in TENSOR_TABLE:
            if key.endswith("_kernel"):
                mask = srng.random(w[key].shape) < small_fraction
                w[key][mask] *= np.float32(small_scale)
    return w


def check_weights(w):
    for key, shape in TENSOR_TABLE.items():
        if key not in w:
            raise ValueError("missing weight tensor %s" % key)
        if tuple(w[key].shape) != tuple(shape):
            raise ValueError("weight tensor %s has shape %s, expected %s"
                             % (key, tuple(w[key].shape), tuple(shape)))


def save_weights(path, w):
    check_weights(w)
    if not path.endswith(".npz"):
        path = path + ".npz"
    np.savez(path, **{k: np.ascontiguousarray(v, dtype=np.float32) for k, v in w.items()})
    return path


def load_weights(path):
    """Load a weight container.

    ``path`` is what the reference passes to ``restore_parameters``: a checkpoint
    *prefix* (clair/model.py:1016-1020, clair/callVarBam.py:72).  Accepted, in order:
    ``path`` itself if it is an .npz file, ``path + '.npz'``, or a TF-bundle checkpoint
    (``path + '.index'`` / ``'.data-00000-of-00001'``) read by clair_amd.tf_bundle.
    """
    candidates = [path, path + ".npz"]
    for p in candidates:
        if os.path.isfile(p) and p.endswith(".npz"):
            with np.load(p) as z:
                w = OrderedDict((k, np.ascontiguousarray(z[k], dtype=np.float32))
                                for k in TENSOR_TABLE)
            check_weights(w)
            return w
    if os.path.isfile(path + ".index"):
        from clair_amd import tf_bundle
        return tf_bundle.load_checkpoint(path)
    raise FileNotFoundError("[ERROR] no weight container found at %s(.npz|.index)" % path)
