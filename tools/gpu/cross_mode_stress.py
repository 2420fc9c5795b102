"""The same 262 144 candidates through every execution mode (slots x batch size -> kernel selection), repeatedly; every output
compared BIT for bit with the first pass of the first mode.  Per-candidate arithmetic does not depend on the batch a candidate sits in,
so any difference is a fault (a race, a stale hand-off, hardware)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from clair_amd import _capi, synth, weights  # noqa: E402

w = weights.synthetic_weights(seed=20250928, head_gain=4.0)
N = 262144
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
x, _ = synth.synthetic_input(N, "ont", seed=4242)
ref = None
total = 0
t00 = time.time()
for slots, batch in ((3, 1024), (3, 4096), (1, 1024), (2, 2048), (1, 4096), (3, 8192), (2, 1024), (3, 512)):
    eng = _capi.Engine(device=0, max_batch=batch, n_slots=slots)
    try:
        eng.load_weights(w)
        xd, od = eng.dataset_alloc(N)
        eng.dataset_upload(xd, 0, x)
        bad = 0
        t0 = time.time()
        for r in range(reps):
            for b in range(N // batch):
                eng.run_resident(b % slots, xd, od, b * batch, batch)
            eng.sync()
            out = eng.dataset_download(od, 0, N)
            if ref is None:
                ref = out.copy()
            diff = np.argwhere((out != ref).any(axis=1)).ravel()
            if len(diff):
                bad += 1
                print("   slots %d batch %d rep %d: %d candidates differ, first %s, max |d| %.2e" % (slots, batch, r, len(diff), diff[:8].tolist(),
                      float(np.abs(out[diff] - ref[diff]).max())), flush=True)
            total += N
        wgs = eng.kernel_workgroups(batch)
        print("slots %d batch %4d (proj2 wgs %d, lstm2 wgs %d): %d / %d passes differ, %.1f s" % (slots, batch, wgs["proj2"], wgs["lstm2"], bad, reps, time.time() - t0), flush=True)
        eng.dataset_free(xd, od)
    finally:
        eng.close()
print("%.1f M candidate evaluations, %.0f s" % (total / 1e6, time.time() - t00))
