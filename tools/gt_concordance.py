#!/usr/bin/env python3
"""VCF GT concordance at scale on the GPU box: decode the HIP probabilities and the float32 oracle's probabilities of the
same synthetic candidates with the same decoder (clair/call_var.py:733-762: arg-max over float32 products with
exact-equality membership tests) and count the rows whose CHROM/POS/REF/ALT/GT differ.  Every flip is then re-examined with
the float64 evaluation: a flip whose two candidate calls are within float32 noise of each other in float64 is a tie the
reference itself would break differently from run to run (multithreaded Eigen, no fixed reduction order).

Usage:  python tools/gt_concordance.py [--n 200000] [--platforms ont,pacbio_ccs,illumina] [--json out.json]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from clair_amd import _capi, call_var as cvar, synth, weights  # noqa: E402


def key(row):
    f = row.split("\t")
    return (f[0], f[1], f[3], f[4], f[-1].split(":")[0])


def concordance(eng, w, platform, n, seed, batch=4096, chunk=32768, log=print):
    from oracle import c_oracle
    dec = cvar.VariantDecoder(cvar.OutputConfig(True, False, False, False, False, None))
    flips, rows_total, worst = [], 0, 0.0
    t0 = time.time()
    for c0 in range(0, n, chunk):
        m = min(chunk, n - c0)
        raw, infos = synth.synthetic_candidates(m, platform, seed=seed + c0, start=100000 + 7 * c0)
        x = synth.to_model_input(raw)
        got = [np.empty((m, k), np.float32) for k in (21, 3, 33, 33)]
        for i in range(0, m, batch):
            for g, o in zip(got, eng.predict(x[i:i + batch])):
                g[i:i + o.shape[0]] = o
        want = c_oracle.forward(w, x)
        chunk_worst = max(float(np.abs(g - t).max()) for g, t in zip(got, want))
        if chunk_worst > 1e-5:      # beyond the tolerance: which side moved?  (re-run the candidate's batch, re-evaluate the oracle on it alone)
            k = int(np.argmax([float(np.abs(g - t).max()) for g, t in zip(got, want)]))
            i = int(np.argmax(np.abs(got[k] - want[k]).max(axis=1)))
            b0 = i // batch * batch
            again = eng.predict(x[b0:b0 + batch])
            alone32 = c_oracle.forward(w, x[i:i + 1])
            alone64 = c_oracle.forward(w, x[i:i + 1], dtype=np.float64)
            log("  !! %s chunk at %d: candidate %d output %d: |hip - oracle| %.2e;  hip again: |again - first| %.2e;  oracle alone: |alone32 - chunk32| %.2e, "
                "|hip - alone32| %.2e, |hip - alone64| %.2e" % (platform, c0, i, k, chunk_worst, float(np.abs(again[k][i - b0] - got[k][i]).max()),
                float(np.abs(alone32[k][0] - want[k][i]).max()), float(np.abs(got[k][i] - alone32[k][0]).max()), float(np.abs(got[k][i] - alone64[k][0]).max())))
        worst = max(worst, chunk_worst)
        rows_g = dec.decode_batch(x, infos, got)
        rows_w = dec.decode_batch(x, infos, want)
        assert len(rows_g) == len(rows_w)
        rows_total += len(rows_w)
        for j, (a, b) in enumerate(zip(rows_g, rows_w)):
            if key(a) != key(b):
                pos = int(a.split("\t")[1])
                idx = [i for i, inf in enumerate(infos) if int(inf[1]) == pos][0]
                o64 = c_oracle.forward(w, x[idx:idx + 1], dtype=np.float64)
                row64 = dec.decode_batch(x[idx:idx + 1], infos[idx:idx + 1], [o.astype(np.float32) for o in o64])
                flips.append({"hip": a, "oracle32": b, "oracle64_rounded": row64[0] if row64 else None,
                              "max_abs_dp": max(float(np.abs(g[idx] - t[idx]).max()) for g, t in zip(got, want))})
        log("%s: %d / %d candidates, %d rows, %d flips, max |dp| %.2e, %.0f s" % (platform, c0 + m, n, rows_total, len(flips), worst, time.time() - t0))
    return {"platform": platform, "candidates": n, "vcf_rows": rows_total, "gt_flips": len(flips), "max_abs_dp": worst, "flips": flips}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=200000)
    ap.add_argument("--platforms", default="ont,pacbio_ccs,illumina")
    ap.add_argument("--json", default=None)
    ap.add_argument("--head-gain", type=float, default=4.0)
    a = ap.parse_args()
    w = weights.synthetic_weights(seed=20250928, head_gain=a.head_gain)
    eng = _capi.Engine(device=0, max_batch=4096, n_slots=1)
    eng.load_weights(w)
    res = [concordance(eng, w, p, a.n, 777) for p in a.platforms.split(",")]
    eng.close()
    for r in res:
        print(json.dumps({k: v for k, v in r.items() if k != "flips"}))
        for f in r["flips"]:
            print("  FLIP", json.dumps(f))
    if a.json:
        with open(a.json, "w") as f:
            json.dump(res, f, indent=1)
