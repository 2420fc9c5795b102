#!/usr/bin/env python3
"""Mint tests/golden/nn_forward_64.npz: a 64-candidate batch, the synthetic-weight seed, the four
output distributions and sampled intermediates of the forward pass.

The reference cannot produce these values (TensorFlow 1.13 is absent; no checkpoints, no golden
vectors in /root/reference -- "parity unpinned").  The values come from oracle/model_np.py
(float32 restatement of clair/model.py:400-622) and are accepted only if the independent torch-CPU
implementation (tools/torch_ref.py) and a float64 evaluation agree within 2e-6.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

from clair_amd import synth, weights  # noqa: E402
from oracle import model_np  # noqa: E402
import torch_ref  # noqa: E402

SEED_W, SEED_X, N = 4242, 99, 64


def main():
    w = weights.synthetic_weights(seed=SEED_W, head_gain=4.0, lstm_bias_scale=0.1)
    raw, infos = synth.synthetic_candidates(N, "ont", seed=SEED_X)
    x = synth.to_model_input(raw)
    outs, inter = model_np.forward(w, x, keep_intermediates=True)
    outs64 = model_np.forward(w, x, dtype=np.float64)
    outs_t, inter_t = torch_ref.forward(w, x)
    for a, b, c in zip(outs, outs64, outs_t):
        assert np.abs(a - b).max() < 2e-6 and np.abs(a - c).max() < 2e-6
    for k in ("a1", "a2", "l3", "l4"):
        assert np.abs(inter[k] - inter_t[k]).max() < 5e-6, k
    path = os.path.join(ROOT, "tests", "golden", "nn_forward_64.npz")
    np.savez_compressed(
        path, seed_w=SEED_W, head_gain=4.0, lstm_bias_scale=0.1, raw=raw.astype(np.int16),
        gt21=outs[0], genotype=outs[1], len1=outs[2], len2=outs[3],
        a1_first4=inter["a1"][:, :4], a2_first4=inter["a2"][:, :4], l3_first4=inter["l3"][:4], l4=inter["l4"])
    print(path, os.path.getsize(path))


def illumina300():
    """tests/golden/nn_illumina300_64.npz: the input of the TRAINED-LIKE mint variant (tools/mint_tf_golden.py: VARIANTS["trained"]) --
    64 synthetic candidates of the 300x Illumina profile (depth capped at 250 per position) as int16 counts, with this repository's
    float32 and float64 oracle outputs on the trained-like recipe weights beside them (so that a drift of the restatement itself
    shows without TensorFlow).  The recurrence amplifies float32 rounding here: the float32 / float64 distance is recorded in the file."""
    import mint_tf_golden as m
    from oracle import c_oracle
    raw, _ = synth.synthetic_candidates(N, "illumina", seed=301)
    x = synth.to_model_input(raw)
    w = m.recipe_weights(trained=True)
    o32, inter = c_oracle.forward(w, x, keep_intermediates=True)
    o64 = c_oracle.forward(w, x, dtype=np.float64)
    np_outs = model_np.forward(w, x)
    dist = max(float(np.abs(a - b).max()) for a, b in zip(o32, o64))
    assert max(float(np.abs(a - b).max()) for a, b in zip(o32, np_outs)) < max(2e-6, 2 * dist)          # the NumPy restatement agrees with the C one
    path = os.path.join(ROOT, "tests", "golden", "nn_illumina300_64.npz")
    np.savez_compressed(path, raw=raw.astype(np.int16), recipe=m.RECIPE + "-trained", f32_f64_distance=dist,
                        gt21=o32[0], genotype=o32[1], len1=o32[2], len2=o32[3],
                        gt21_f64=o64[0], genotype_f64=o64[1], len1_f64=o64[2], len2_f64=o64[3],
                        a2_absmax=float(np.abs(inter["a2"]).max()), a1_absmax=float(np.abs(inter["a1"]).max()))
    print(path, os.path.getsize(path), "float32 vs float64 oracle: %.3g; |a1| max %.3f, |a2| max %.3f; mean top probability per head %s"
          % (dist, float(np.abs(inter["a1"]).max()), float(np.abs(inter["a2"]).max()), [round(float(o.max(axis=1).mean()), 3) for o in o32]))


if __name__ == "__main__":
    if "--illumina300" in sys.argv:
        illumina300()
    else:
        main()
