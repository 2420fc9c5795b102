"""A co-runner / self-check for the fused layer-2 launch under contention: one slot, batch 1024 (the fused one-launch path), the same
eight batches over and over for <seconds>; every pass is bit-compared with the handle's first pass and, at the end, with the two-launch
path on a second handle (CLAIR_AMD_LSTM2_FUSED=0).  Prints fused launches, recoveries (engine.hip: recover_fused) and differing rows.
usage: fused_corunner.py [seconds=60] [tag]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from clair_amd import _capi, synth, weights  # noqa: E402

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
tag = sys.argv[2] if len(sys.argv) > 2 else "fused"
w = weights.synthetic_weights(seed=20250928, head_gain=4.0)
xs = [synth.synthetic_input(1024, ("ont", "pacbio_ccs", "illumina")[i % 3], seed=700 + i)[0] for i in range(8)]
os.environ.pop("CLAIR_AMD_LSTM2_FUSED", None)
eng = _capi.Engine(device=0, max_batch=1024, n_slots=1)
eng.load_weights(w)
first = [np.concatenate(eng.predict(x), axis=1) for x in xs]
t0, passes, bad = time.time(), 0, 0
while time.time() - t0 < seconds:
    for x, ref in zip(xs, first):
        got = np.concatenate(eng.predict(x), axis=1)
        bad += int((got != ref).any(axis=1).sum())
    passes += 1
fused, recovered = eng.counter("fused_launches"), eng.counter("fused_recoveries")
eng.close()
os.environ["CLAIR_AMD_LSTM2_FUSED"] = "0"
two = _capi.Engine(device=0, max_batch=1024, n_slots=1)
two.load_weights(w)
bad2 = sum(int((np.concatenate(two.predict(x), axis=1) != ref).any(axis=1).sum()) for x, ref in zip(xs, first))
two.close()
print("%s: %d passes x 8 batches of 1024 in %.0f s; fused launches %d, recoveries %d; rows differing from the first pass %d, from the two-launch path %d"
      % (tag, passes, time.time() - t0, fused, recovered, bad, bad2), flush=True)
sys.exit(1 if bad or bad2 else 0)
