#!/usr/bin/env python3
"""Mint the reference-pinned golden vectors of the network arithmetic -- to be run by someone who HAS TensorFlow 1.13.

Why this exists.  The forward pass the HIP engine replaces is executed, in the reference, by TensorFlow 1.13.2
(/root/reference/README.md:127; graph: /root/reference/clair/model.py:299-312, 400-622; activation: clair/selu.py:26-30).
TensorFlow is not in the build image and cannot be installed there, so every numeric parity claim of this repository is made
against oracle/clair_oracle.c -- a restatement with the TF-1.13 semantics written out from the documentation ("parity
unpinned").  This script closes that gap from the other side: on any machine with TF 1.13 it builds the
SAME graph with TF's own ops (tf.contrib.cudnn_rnn.CudnnCompatibleLSTMCell under tf.contrib.rnn.stack_bidirectional_dynamic_rnn,
tf.layers.dense, tf.nn.softmax), loads a fixed, recipe-defined set of weights into TF's variables BY THEIR TF NAMES, runs 64 fixed
candidates and writes every intermediate to tests/golden/nn_tf113_64.npz.  tests/test_oracle.py and tests/test_parity_gpu.py consume
that file when it is present (and skip, loudly, while it is not): the day it is committed the oracle is pinned to the real
reference arithmetic and parity stops being "unpinned".

    # Python 3.6/3.7, tensorflow==1.13.2, numpy<1.20; from the repository root (tools/pin/run.sh does all of it in a container):
    python tools/mint_tf_golden.py                       # -> tests/golden/nn_tf113_64.npz (~0.6 MB): fresh-init-like weights, ONT counts
                                                         #    tests/golden/nn_tf113_trained_64.npz: TRAINED-LIKE weights (LSTM kernels x4, forget
                                                         #    bias +1, head gain 6) on 300x Illumina counts -- saturated gates, an integrating cell
                                                         #    state: the regime fresh initialisation never reaches and a semantic slip cannot hide in
    python tools/mint_tf_golden.py --mini-checkpoint     # + tests/golden/tf113_mini.{index,data-00000-of-00001,json}: a tf.train.Saver
                                                         #   checkpoint of a 4-unit version of the graph (real bundle format, real names)
    python tools/mint_tf_golden.py --cudnn-checkpoint    # (tensorflow-gpu==1.13.x + a CUDA GPU) + tests/golden/tf113_cudnn.*: the two BiLSTM
                                                         #   layers as tf.contrib.cudnn_rnn.CudnnLSTM (the reference's GPU branch, clair/model.py:281-296)
                                                         #   saved through TF's own CudnnLSTMSaveable AND as the raw opaque buffers, with the
                                                         #   layers' outputs: pins clair_amd/tf_bundle.py's names and its opaque -> canonical conversion
    python tools/mint_tf_golden.py --list-checkpoint /path/to/model-000016      # variable names / shapes of a REAL Clair model

It is stand-alone on purpose (no import from clair_amd, nothing from /root/reference): the weights come from an integer hash
written out below, so they are the same bits under any NumPy; the candidates are the committed tests/golden/nn_forward_64.npz
counts.  Nothing of the reference's source is used -- only TensorFlow's public API, called the way clair/model.py calls it.
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RECIPE = "splitmix-uniform-v1"
T, F_IN, H, L3_UNITS, L4_UNITS, L5_UNITS = 33, 32, 128, 30, 192, 96
HEADS = (("gt21", "Y_base_change_logits", 21), ("genotype", "Y_genotype_logits", 3),
         ("len1", "Y_indel_length_logits_1", 33), ("len2", "Y_indel_length_logits_2", 33))


# ---- recipe weights: identical bits everywhere (pure uint64 arithmetic, no NumPy random streams) ----------------------------------
def _uniform(tag, count):
    """`count` float32 values in [-1, 1): splitmix64 of (tag, index), top 24 bits."""
    with np.errstate(over="ignore"):
        z = np.arange(count, dtype=np.uint64) + np.uint64(tag) * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return ((z >> np.uint64(40)).astype(np.float64) / float(1 << 23) - 1.0).astype(np.float32)


def tensor_shapes(h=H, l3=L3_UNITS, l4=L4_UNITS, l5=L5_UNITS):
    """(key, shape) in the order of include/clair_amd.h's tensor ids, for LSTM width h (128 in the real graph)."""
    shapes = [("lstm1_fw_kernel", (F_IN + h, 4 * h)), ("lstm1_fw_bias", (4 * h,)), ("lstm1_bw_kernel", (F_IN + h, 4 * h)), ("lstm1_bw_bias", (4 * h,)),
              ("lstm2_fw_kernel", (3 * h, 4 * h)), ("lstm2_fw_bias", (4 * h,)), ("lstm2_bw_kernel", (3 * h, 4 * h)), ("lstm2_bw_bias", (4 * h,)),
              ("l3_kernel", (2 * h, T, l3)), ("l3_bias", (2 * h, l3)), ("l4_kernel", (l3 * 2 * h, l4)), ("l4_bias", (l4,)),
              ("l5_kernel", (4, l4, l5)), ("l5_bias", (4, l5))]
    for key, _, n in HEADS:
        shapes += [("head_%s_kernel" % key, (l5, n)), ("head_%s_bias" % key, (n,))]
    return shapes


TRAINED_LSTM_GAIN, TRAINED_FORGET_BIAS, TRAINED_HEAD_GAIN = 4.0, 1.0, 6.0


def recipe_weights(h=H, l3=L3_UNITS, l4=L4_UNITS, l5=L5_UNITS, trained=False):
    """The golden weight set (trained=True: the TRAINED-LIKE variant, see below): uniform with the scale each initialiser of clair/model.py would give (dense: variance 1.3 / fan_in,
    LSTM kernels: 2 x Glorot), non-zero biases everywhere (a zero bias would hide a mis-wired one), head kernels x8 so that the
    softmaxes are peaky (mean top probability 0.55 .. 0.98 per head on the golden candidates).  LSTM outputs reach +-0.78 without
    saturating: no dead gate hides an error either; the float32 oracle sits 4e-6 from its float64 twin on this set.
    trained=True is what training leaves behind and fresh initialisation never shows: LSTM kernels x4 (gates that saturate), +1 on the
    forget-gate bias rows [2h, 3h) (a cell state that integrates over all 33 steps), head kernels x6 -- the cells of
    tools/parity_sweep.py where the recurrence amplifies float32 rounding most (on the 300x counts |a1| and |a2| reach 1.000 and the
    float32 oracle sits 7.9e-6 from float64 on the probabilities, twice the fresh set's distance), and where a forget bias that is
    not applied, or applied twice, changes every output instead of hiding in a zero bias."""
    w = {}
    for tag, (key, shape) in enumerate(tensor_shapes(h, l3, l4, l5), start=1):
        u = _uniform(tag, int(np.prod(shape))).reshape(shape)
        if key.endswith("_bias"):
            scale = 0.1
        elif key.startswith("lstm"):
            scale = 2.0 * np.sqrt(6.0 / (shape[0] + shape[1]))
        else:
            scale = np.sqrt(3.0 * 1.3 / shape[-2]) * ((TRAINED_HEAD_GAIN if trained else 8.0) if key.startswith("head_") else 1.0)
        w[key] = (u * np.float32(scale)).astype(np.float32)
        if key.startswith("lstm1") and key.endswith("_kernel"):
            w[key][:F_IN] *= np.float32(0.1)           # rows that multiply raw pileup counts of 0..250: what training would leave
        if trained and key.startswith("lstm"):
            if key.endswith("_kernel"):
                w[key] *= np.float32(TRAINED_LSTM_GAIN)
            else:
                w[key][2 * h:3 * h] += np.float32(TRAINED_FORGET_BIAS)     # gate order i, c~, f, o: the forget rows
    return w


INPUT_FIXTURES = {"ont": "nn_forward_64.npz", "illumina300": "nn_illumina300_64.npz"}
# variant -> (trained-like weights?, input profile, output file, recipe tag)
VARIANTS = {"fresh": (False, "ont", "nn_tf113_64.npz", RECIPE), "trained": (True, "illumina300", "nn_tf113_trained_64.npz", RECIPE + "-trained")}


def golden_input(profile="ont"):
    """64 candidates of a committed fixture as the network sees them: float32 [64,33,8,4] with channels 1..3 minus channel 0
    (clair/utils.py:96-98).  "ont": tests/golden/nn_forward_64.npz (ONT-like depth ~50); "illumina300": tests/golden/
    nn_illumina300_64.npz (300x coverage capped at 250 per position, clair/dataPrepScripts/CreateTensor.py:431 -- counts two orders
    of magnitude above the weights' scale; tools/make_nn_golden.py --illumina300)."""
    with np.load(os.path.join(ROOT, "tests", "golden", INPUT_FIXTURES[profile])) as z:
        x = z["raw"].astype(np.float32)
    x[:, :, :, 1:] -= x[:, :, :, 0:1]
    return x


def tf_variable_names(h=H):
    """TF variable name -> (key, index into the leading axis or None): the scopes clair/model.py opens (LSTM1 / LSTM2, L3/Unit_i, L4,
    L5_k, Prediction/Y_*), with what stack_bidirectional_dynamic_rnn and CudnnCompatibleLSTMCell add in TF 1.13."""
    m = {}
    for layer in (1, 2):
        for d in ("fw", "bw"):
            base = "LSTM%d/stack_bidirectional_rnn/cell_0/bidirectional_rnn/%s/cudnn_compatible_lstm_cell/" % (layer, d)
            m[base + "kernel"] = ("lstm%d_%s_kernel" % (layer, d), None)
            m[base + "bias"] = ("lstm%d_%s_bias" % (layer, d), None)
    for c in range(2 * h):
        m["L3/Unit_%d/kernel" % c] = ("l3_kernel", c)
        m["L3/Unit_%d/bias" % c] = ("l3_bias", c)
    m["L4/kernel"], m["L4/bias"] = ("l4_kernel", None), ("l4_bias", None)
    for k in range(4):
        m["L5_%d/kernel" % (k + 1)] = ("l5_kernel", k)
        m["L5_%d/bias" % (k + 1)] = ("l5_bias", k)
    for key, tfname, _ in HEADS:
        m["Prediction/%s/kernel" % tfname] = ("head_%s_kernel" % key, None)
        m["Prediction/%s/bias" % tfname] = ("head_%s_bias" % key, None)
    return m


# ---- the graph, with TensorFlow's own ops ------------------------------------------------------------------------------------------
def build_graph(tf, h=H, n3=L3_UNITS, n4=L4_UNITS, n5=L5_UNITS):
    """Inference graph of clair/model.py:400-622 on the CPU branch of adaptive_LSTM_layer (:299-312).  Returns (x placeholder, dict of
    tensors).  Dropouts are identity at inference (training=False, clair/selu.py:72-74) and are left out."""
    def selu(v):      # clair/selu.py:26-30
        return 1.0507009873554804934193349852946 * tf.where(v >= 0.0, v, 1.6732632423543772848170429916717 * tf.nn.elu(v))

    def bilstm(inp, name):
        with tf.variable_scope(name):
            out, _, _ = tf.contrib.rnn.stack_bidirectional_dynamic_rnn(
                [tf.contrib.cudnn_rnn.CudnnCompatibleLSTMCell(h)], [tf.contrib.cudnn_rnn.CudnnCompatibleLSTMCell(h)],
                inp, dtype=tf.float32, time_major=True)
        return out

    x = tf.placeholder(tf.float32, shape=(None, T, 8, 4), name="X_placeholder")
    flat = tf.reshape(x, (tf.shape(x)[0], T, F_IN))
    a1 = bilstm(tf.transpose(flat, perm=[1, 0, 2]), "LSTM1")                       # [33, n, 2h]
    a2 = bilstm(a1, "LSTM2")
    a2_nm = tf.transpose(a2, [1, 0, 2])                                             # [n, 33, 2h]
    with tf.variable_scope("L3"):                                                   # slice_dense_layer, model.py:225-244
        units = [tf.layers.dense(v, units=n3, name="Unit_%d" % i, activation=selu) for i, v in enumerate(tf.unstack(a2_nm, axis=2))]
        l3_t = tf.stack(units, axis=2)                                              # [n, l3, 2h]
    l3_flat = tf.reshape(l3_t, (tf.shape(l3_t)[0], n3 * 2 * h))                     # flat index u * 2h + c
    l4_t = tf.layers.dense(l3_flat, units=n4, name="L4", activation=selu)
    l5 = [tf.layers.dense(l4_t, units=n5, name="L5_%d" % (k + 1), activation=selu) for k in range(4)]
    outs, logits = [], []
    with tf.variable_scope("Prediction"):
        for k, (_, tfname, n) in enumerate(HEADS):
            lg = tf.layers.dense(l5[k], units=n, activation=selu, name=tfname)      # selu on the logits, then softmax (model.py:586)
            logits.append(lg)
            outs.append(tf.nn.softmax(lg))
    return x, dict(a1=a1, a2=a2, l3=l3_flat, l4=l4_t, l5=tf.stack(l5, axis=1), logits=logits, outs=outs)


def load_variables(tf, sess, w, h=H):
    names = tf_variable_names(h)
    seen = set()
    for v in tf.global_variables():
        name = v.name.split(":")[0]
        if name not in names:
            raise SystemExit("TensorFlow created a variable this script does not know: %s %s\n(the name table in clair_amd/weights.py "
                             "would not find it in a real checkpoint either -- report the full list below)\n%s"
                             % (name, v.shape, "\n".join("%s %s" % (u.name, u.shape) for u in tf.global_variables())))
        key, idx = names[name]
        value = w[key] if idx is None else w[key][idx]
        if tuple(v.shape.as_list()) != tuple(value.shape):
            raise SystemExit("variable %s has shape %s, the weight table says %s" % (name, v.shape, value.shape))
        v.load(value, sess)
        seen.add(name)
    missing = sorted(set(names) - seen)
    if missing:
        raise SystemExit("TensorFlow did not create: %s ..." % ", ".join(missing[:5]))


def mint(out_path, variant="fresh"):
    import tensorflow as tf
    if not tf.__version__.startswith("1.13"):
        sys.stderr.write("warning: TensorFlow %s, the reference pins 1.13.2 (README.md:127); the file records the version\n" % tf.__version__)
    trained, profile, _, recipe = VARIANTS[variant]
    x_np, w = golden_input(profile), recipe_weights(trained=trained)
    tf.reset_default_graph()
    x, t = build_graph(tf)
    cfg = tf.ConfigProto(intra_op_parallelism_threads=1, inter_op_parallelism_threads=1, device_count={"GPU": 0})
    with tf.Session(config=cfg) as sess:
        sess.run(tf.global_variables_initializer())
        load_variables(tf, sess, w)
        a1, a2, l3, l4, l5, logits, outs = sess.run([t["a1"], t["a2"], t["l3"], t["l4"], t["l5"], t["logits"], t["outs"]], feed_dict={x: x_np})
    np.savez_compressed(
        out_path, recipe=recipe, variant=variant, tf_version=tf.__version__, numpy_version=np.__version__,
        weights_checksum=np.float64(sum(float(np.abs(v.astype(np.float64)).sum()) for v in w.values())),
        gt21=outs[0], genotype=outs[1], len1=outs[2], len2=outs[3],
        logits_gt21=logits[0], logits_genotype=logits[1], logits_len1=logits[2], logits_len2=logits[3],
        a1_first4=a1[:, :4], a2_first4=a2[:, :4], l3_first4=l3[:4], l4=l4, l5=l5)
    print("%s written (%d bytes) with TensorFlow %s" % (out_path, os.path.getsize(out_path), tf.__version__))


def mini_checkpoint(prefix):
    """A tf.train.Saver checkpoint of the same graph at LSTM width 4 / dense widths 3, 5, 6 (a few KB): a REAL tensor bundle with the
    REAL variable names, for clair_amd/tf_bundle.py's reader test; plus <prefix>.json with every variable's name, shape and values."""
    import tensorflow as tf
    h, l3, l4, l5 = 4, 3, 5, 6
    w = recipe_weights(h, l3, l4, l5)
    tf.reset_default_graph()
    build_graph(tf, h, l3, l4, l5)
    step = tf.train.get_or_create_global_step()
    with tf.variable_scope("Training_Operation"):      # clair/model.py:716: the optimizer's slots live here in a real checkpoint
        tf.get_variable("beta1_power", initializer=np.float32(0.9))
    with tf.Session() as sess:
        sess.run(tf.global_variables_initializer())
        names = tf_variable_names(h)
        for v in tf.global_variables():
            name = v.name.split(":")[0]
            if name in names:
                key, idx = names[name]
                v.load(w[key] if idx is None else w[key][idx], sess)
        tf.train.Saver().save(sess, prefix, write_meta_graph=False)
        listing = {v.name.split(":")[0]: {"shape": v.shape.as_list(), "dtype": v.dtype.base_dtype.name,
                                          "values": np.asarray(sess.run(v)).ravel().tolist()} for v in tf.global_variables()}
    del step
    with open(prefix + ".json", "w") as f:
        json.dump({"tf_version": tf.__version__, "widths": [h, l3, l4, l5], "variables": listing}, f)
    print("%s.{index,data-00000-of-00001,json} written" % prefix)


def cudnn_checkpoint(prefix, n=8):
    """The reference's GPU branch (clair/model.py:281-296): each BiLSTM layer as ONE tf.contrib.cudnn_rnn.CudnnLSTM whose only variable
    is the flat "opaque_kernel".  Needs tensorflow-gpu 1.13 and a CUDA GPU (CudnnLSTM has no CPU kernel).  Writes
      <prefix>.{index,data-00000-of-00001}          tf.train.Saver().save: the layers' CudnnLSTMSaveable stores the CANONICAL per-direction
                                                     kernel / bias tensors -- under whatever names TF gives them (the point of the exercise)
      <prefix>_raw.{index,data-00000-of-00001}      the opaque buffers themselves (a Saver over a plain name -> variable dict: no saveable)
      <prefix>.json                                  the opaque values, and a1 / a2 of the first `n` illumina300 candidates
    tests/test_weights.py: clair_amd/tf_bundle.py must find the canonical tensors under its candidate names, its opaque -> canonical
    conversion of <prefix>_raw must equal what the saveable wrote, and the oracle's BiLSTM with those tensors must reproduce a1 / a2."""
    import tensorflow as tf
    if hasattr(tf, "test") and not tf.test.is_gpu_available(cuda_only=True):
        raise SystemExit("--cudnn-checkpoint needs tensorflow-gpu==1.13.x and a CUDA GPU: tf.contrib.cudnn_rnn.CudnnLSTM has no CPU kernel")
    x_np = golden_input("illumina300")[:n].reshape(n, T, F_IN).transpose(1, 0, 2)          # time-major, as clair/model.py:403-418 feeds it
    tf.reset_default_graph()
    x = tf.placeholder(tf.float32, shape=(T, None, F_IN), name="X_time_major")
    layers, prev = [], x
    for name in ("LSTM1", "LSTM2"):
        with tf.variable_scope(name):
            lstm = tf.contrib.cudnn_rnn.CudnnLSTM(num_layers=1, num_units=H, direction="bidirectional", dtype=tf.float32,
                                                  kernel_initializer=tf.random_uniform_initializer(-0.15, 0.15, seed=11),
                                                  bias_initializer=tf.random_uniform_initializer(-0.5, 0.5, seed=12))
            lstm.build(prev.get_shape())
            prev, _ = lstm(prev)
        layers.append(prev)
    opaque = [v for v in tf.global_variables() if v.name.split(":")[0].endswith("opaque_kernel")]
    if len(opaque) != 2:
        raise SystemExit("expected one opaque_kernel per layer, TensorFlow created: %s" % [v.name for v in tf.global_variables()])
    with tf.Session() as sess:
        sess.run(tf.global_variables_initializer())
        a1, a2 = sess.run(layers, feed_dict={x: x_np})
        tf.train.Saver().save(sess, prefix, write_meta_graph=False)
        tf.train.Saver(var_list={v.name.split(":")[0]: v for v in opaque}).save(sess, prefix + "_raw", write_meta_graph=False)
        listing = {v.name.split(":")[0]: np.asarray(sess.run(v)).ravel().tolist() for v in opaque}
    with open(prefix + ".json", "w") as f:
        json.dump({"tf_version": tf.__version__, "n": n, "opaque": listing, "a1": np.asarray(a1).ravel().tolist(), "a2": np.asarray(a2).ravel().tolist()}, f)
    print("%s{,_raw}.{index,data-00000-of-00001} and %s.json written" % (prefix, prefix))


def list_checkpoint(prefix):
    import tensorflow as tf
    reader = tf.train.NewCheckpointReader(prefix)
    shapes, dtypes = reader.get_variable_to_shape_map(), reader.get_variable_to_dtype_map()
    expected = tf_variable_names()
    for name in sorted(shapes):
        mark = "model  " if name in expected else "other  "
        print("%s%-110s %-16s %s" % (mark, name, shapes[name], dtypes[name].name))
    missing = sorted(set(expected) - set(shapes))
    print("# %d variables; %d of the %d names clair_amd/weights.py expects are present%s"
          % (len(shapes), len(expected) - len(missing), len(expected), "" if not missing else "; MISSING e.g. " + ", ".join(missing[:4])))


if __name__ == "__main__":
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--out-dir", default=os.path.join(ROOT, "tests", "golden"))
    ap.add_argument("--variant", default="both", choices=("fresh", "trained", "both"))
    ap.add_argument("--mini-checkpoint", action="store_true")
    ap.add_argument("--cudnn-checkpoint", action="store_true")
    ap.add_argument("--list-checkpoint", default=None, metavar="PREFIX")
    a = ap.parse_args()
    if a.list_checkpoint:
        list_checkpoint(a.list_checkpoint)
    else:
        for variant in (("fresh", "trained") if a.variant == "both" else (a.variant,)):
            mint(os.path.join(a.out_dir, VARIANTS[variant][2]), variant)
        if a.mini_checkpoint:
            mini_checkpoint(os.path.join(a.out_dir, "tf113_mini"))
        if a.cudnn_checkpoint:
            cudnn_checkpoint(os.path.join(a.out_dir, "tf113_cudnn"))
