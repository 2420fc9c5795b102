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


if __name__ == "__main__":
    main()
