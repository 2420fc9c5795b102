#!/usr/bin/env python3
"""Parity sweep on the GPU box: HIP forward pass vs the float32 oracle vs a float64 evaluation of the same graph, over
trained-like weight shapes and extreme inputs (VERDICT r01 "harden parity").  Prints one row per cell:

  cell | max|hip - o32| | max|hip - o64| | max|o32 - o64| | worst a1 / a2 tap error vs o64

`o32` = oracle/clair_oracle.c in float32 (the checker), `o64` = the same code in float64 (what float32 rounding itself
costs on that weight set).  The cells are shared with tests/test_parity_gpu.py::test_weight_and_input_sweep.
Usage:  python tools/parity_sweep.py [--n 256] [--json gpurun_out/parity_sweep.json]
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from clair_amd import _capi, synth, weights  # noqa: E402

BASE = dict(seed=20250928, head_gain=4.0)
WEIGHT_CELLS = [
    ("fresh init (bench weights)", dict()),
    ("lstm kernels x4", dict(lstm_gain=4.0)),
    ("lstm kernels x8", dict(lstm_gain=8.0)),
    ("forget bias +1, lstm biases N(0,0.1)", dict(forget_bias=1.0, lstm_bias_scale=0.1)),
    ("lstm x4 + forget bias +1", dict(lstm_gain=4.0, forget_bias=1.0, lstm_bias_scale=0.1)),
    ("count rows of LSTM1 x0.01", dict(input_gain=0.01)),
    ("count rows x0.01, lstm x4, forget +1", dict(input_gain=0.01, lstm_gain=4.0, forget_bias=1.0)),
    ("L4 kernel x0.01 (1e-4 magnitude), L5 x100", dict(l4_gain=0.01)),
    ("half of all kernel entries x1e-3 (1e-4 magnitude)", dict(small_fraction=0.5, small_scale=1e-3)),
    ("head gain 1", dict(head_gain=1.0)),
    ("head gain 8", dict(head_gain=8.0)),
    ("head gain 12", dict(head_gain=12.0)),
]
# input cells: raw int16 counts through clair_submit_counts
COUNT_LEVELS = (0, 250, 2047, 32767)


def count_batches(n, seed=5):
    """One batch of raw counts per level: level 0 = all zeros; otherwise Poisson-like pileups rescaled so that the largest
    count of every candidate is exactly `level` (channel 0 <= level, channels 1..3 around it)."""
    out = []
    raw, _ = synth.synthetic_candidates(n, "illumina", seed=seed)
    raw = raw.astype(np.float64)
    peak = raw.reshape(n, -1).max(axis=1).reshape(n, 1, 1, 1)
    for level in COUNT_LEVELS:
        c = np.rint(raw * (level / peak)).astype(np.int64) if level else np.zeros_like(raw, dtype=np.int64)
        assert c.max() == level and c.min() >= 0
        out.append((level, c.astype(np.int16)))
    return out


def errors(eng, w, x, counts=None, taps=True):
    from oracle import c_oracle
    if counts is not None:
        eng.submit_counts(0, counts)
        got = eng.wait(0)
    else:
        got = eng.predict(x)
    o32, i32 = c_oracle.forward(w, x, keep_intermediates=True)
    o64, i64 = c_oracle.forward(w, x, keep_intermediates=True, dtype=np.float64)
    r = {
        "hip_vs_o32": max(float(np.abs(g - t).max()) for g, t in zip(got, o32)),
        "hip_vs_o64": max(float(np.abs(g - t).max()) for g, t in zip(got, o64)),
        "o32_vs_o64": max(float(np.abs(a - b).max()) for a, b in zip(o32, o64)),
        "finite": bool(all(np.isfinite(g).all() for g in got)),
    }
    if taps:
        n = x.shape[0]
        n_pad = (n + 31) // 32 * 32
        for name, which in (("a1", 1), ("a2", 2)):
            tap = eng.debug_read(0, which, (33, n_pad, 256)).transpose(1, 0, 2)[:n]
            r["%s_hip_vs_o64" % name] = float(np.abs(tap - i64[name]).max())
            r["%s_o32_vs_o64" % name] = float(np.abs(i32[name] - i64[name]).max())
    return r


def run(n=256, log=print):
    rows = []
    eng = _capi.Engine(device=0, max_batch=max(n, 32), n_slots=1)
    try:
        for label, kw in WEIGHT_CELLS:
            w = weights.synthetic_weights(**dict(BASE, **kw))
            eng.load_weights(w)
            for platform in ("ont", "illumina"):
                x, _ = synth.synthetic_input(n, platform, seed=4000 + n)
                r = errors(eng, w, x)
                r.update(cell="%s | %s" % (label, platform))
                rows.append(r)
                log(fmt(r))
        for label, kw in (("fresh init", dict()), ("count rows x0.01, lstm x4, forget +1", dict(input_gain=0.01, lstm_gain=4.0, forget_bias=1.0))):
            w = weights.synthetic_weights(**dict(BASE, **kw))
            eng.load_weights(w)
            for level, counts in count_batches(n):
                x = counts.astype(np.float32)
                x[..., 1:] -= x[..., 0:1]
                r = errors(eng, w, x, counts=counts)
                r.update(cell="%s | counts up to %d via clair_submit_counts" % (label, level))
                rows.append(r)
                log(fmt(r))
    finally:
        eng.close()
    return rows


def fmt(r):
    return "%-78s hip-o32 %.2e  hip-o64 %.2e  o32-o64 %.2e  a1 %.1e/%.1e  a2 %.1e/%.1e%s" % (
        r["cell"], r["hip_vs_o32"], r["hip_vs_o64"], r["o32_vs_o64"], r.get("a1_hip_vs_o64", 0), r.get("a1_o32_vs_o64", 0),
        r.get("a2_hip_vs_o64", 0), r.get("a2_o32_vs_o64", 0), "" if r["finite"] else "  NON-FINITE")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=256)
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    rows = run(a.n)
    if a.json:
        with open(a.json, "w") as f:
            json.dump(rows, f, indent=1)
