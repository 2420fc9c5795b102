"""PCIe-inclusive rate of the host-buffer boundary (clair_submit / clair_wait with NumPy arrays on both sides), 3 slots in flight."""
import sys
import time
import numpy as np
sys.path.insert(0, ".")
from clair_amd import _capi, synth, weights

eng = _capi.Engine(device=0, max_batch=1024, n_slots=3)
eng.load_weights(weights.synthetic_weights(seed=20250928, head_gain=4.0))
xs = [synth.synthetic_input(1024, "ont", seed=s)[0] for s in range(3)]
for rounds in (5, 200):
    t0 = time.perf_counter()
    for r in range(rounds):
        for s in range(3):
            if r:
                eng.wait(s)
            eng.submit(s, xs[s])
    for s in range(3):
        eng.wait(s)
    dt = time.perf_counter() - t0
print("host buffers in / host arrays out, batch 1024, 3 slots: %.0f candidates/s (%.1f GB/s H2D)" % (rounds * 3 * 1024 / dt, rounds * 3 * 1024 * 4224 / dt / 1e9))
