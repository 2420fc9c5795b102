"""One short run of the host-array boundary (submit / wait, page-locked float32 input) followed by the same number of resident passes,
for a rocprofv3 kernel + memory-copy trace (tools/boundary_timeline.py reads it).  usage: boundary_trace.py <slots> <batches> [counts]"""
import sys
import time
import numpy as np
sys.path.insert(0, ".")
from clair_amd import _capi, synth, weights

n_slots, batches = int(sys.argv[1]), int(sys.argv[2])
counts = len(sys.argv) > 3 and sys.argv[3] == "counts"
eng = _capi.Engine(device=0, max_batch=1024, n_slots=n_slots)
eng.load_weights(weights.synthetic_weights(seed=20250928, head_gain=4.0))
x = synth.synthetic_input(1024, "ont", seed=1)[0]
c = x.copy(); c[..., 1:] += c[..., 0:1]; c = c.astype(np.int16)
bufs = [eng.slot_input(s) for s in range(n_slots)]
for b in bufs:
    np.copyto(b, x)


def boundary(k):
    t0 = time.perf_counter()
    for i in range(k):
        s = i % n_slots
        if i >= n_slots:
            eng.wait(s)
        if counts:
            eng.submit_counts(s, c)
        else:
            eng.submit(s, bufs[s])
    for s in range(min(n_slots, k)):
        eng.wait(s)
    return k * 1024 / (time.perf_counter() - t0)


def resident(k):
    t0 = time.perf_counter()
    for i in range(k):
        eng.run_resident(i % min(n_slots, 3), xd, od, (i % 8) * 1024, 1024)
    eng.sync()
    return k * 1024 / (time.perf_counter() - t0)


xd, od = eng.dataset_alloc(8 * 1024)
for b in range(8):
    eng.dataset_upload(xd, b * 1024, x)
boundary(24); resident(24)
for leg in (boundary, resident, boundary, resident):
    time.sleep(0.05)                   # the idle gap tools/boundary_timeline.py splits the trace at
    print("%s: %.2f M candidates/s" % (leg.__name__, leg(batches) / 1e6))
