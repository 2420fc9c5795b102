"""A/B of the engine's boundary knobs on ONE box in ONE process: engines are created in turn (the knobs are read from the environment at
clair_engine_create), each runs the submit/wait loop in the three hand-over modes, the whole round is repeated and the medians are
printed next to the resident rate measured between the rounds.  usage: boundary_ab.py [rounds] [batches per loop]"""
import os
import sys
import time
import numpy as np
sys.path.insert(0, ".")
from clair_amd import _capi, synth, weights

ROUNDS = int(sys.argv[1]) if len(sys.argv) > 1 else 5
BATCHES = int(sys.argv[2]) if len(sys.argv) > 2 else 480
CONFIGS = [
    ("in sdma  2 workers 6 slots", dict(CLAIR_AMD_COPY_STREAMS="in", CLAIR_AMD_D2H="sdma", CLAIR_AMD_STAGING_THREADS="2"), 6),
    ("in kern  2 workers 6 slots", dict(CLAIR_AMD_COPY_STREAMS="in", CLAIR_AMD_D2H="kernel", CLAIR_AMD_STAGING_THREADS="2"), 6),
    ("in sdma  1 worker  6 slots", dict(CLAIR_AMD_COPY_STREAMS="in", CLAIR_AMD_D2H="sdma", CLAIR_AMD_STAGING_THREADS="1"), 6),
    ("in sdma  0 workers 6 slots", dict(CLAIR_AMD_COPY_STREAMS="in", CLAIR_AMD_D2H="sdma", CLAIR_AMD_STAGING_THREADS="0"), 6),
    ("in kern  0 workers 6 slots", dict(CLAIR_AMD_COPY_STREAMS="in", CLAIR_AMD_D2H="kernel", CLAIR_AMD_STAGING_THREADS="0"), 6),
    ("in sdma  2 workers 5 slots", dict(CLAIR_AMD_COPY_STREAMS="in", CLAIR_AMD_D2H="sdma", CLAIR_AMD_STAGING_THREADS="2"), 5),
    ("in sdma  2 workers 8 slots", dict(CLAIR_AMD_COPY_STREAMS="in", CLAIR_AMD_D2H="sdma", CLAIR_AMD_STAGING_THREADS="2"), 8),
    ("in sdma  3 workers 9 slots", dict(CLAIR_AMD_COPY_STREAMS="in", CLAIR_AMD_D2H="sdma", CLAIR_AMD_STAGING_THREADS="3"), 9),
    ("lane (round-3 layout) 3 slots", dict(CLAIR_AMD_COPY_STREAMS="lane", CLAIR_AMD_D2H="sdma", CLAIR_AMD_STAGING_THREADS="0"), 3),
]
KNOBS = sorted({k for _, env, _ in CONFIGS for k in env})
W = weights.synthetic_weights(seed=20250928, head_gain=4.0)
X = [synth.synthetic_input(1024, "ont", seed=s)[0] for s in range(9)]
C = []
for x in X:
    c = x.copy(); c[..., 1:] += c[..., 0:1]; C.append(c.astype(np.int16))


def boundary(eng, n_slots, mode, bufs, k):
    t0 = time.perf_counter()
    for i in range(k):
        s = i % n_slots
        if i >= n_slots:
            eng.wait(s)
        if mode == "pageable":
            eng.submit(s, X[s])
        elif mode == "int16":
            eng.submit_counts(s, C[s])
        else:
            eng.submit(s, bufs[s])
    for s in range(min(n_slots, k)):
        eng.wait(s)
    return k * 1024 / (time.perf_counter() - t0)


def resident(k):
    for key in KNOBS:
        os.environ.pop(key, None)
    eng = _capi.Engine(device=0, max_batch=1024, n_slots=3)
    eng.load_weights(W)
    xd, od = eng.dataset_alloc(8 * 1024)
    for b in range(8):
        eng.dataset_upload(xd, b * 1024, X[0])
    for rep in range(2):
        t0 = time.perf_counter()
        for i in range(k):
            eng.run_resident(i % 3, xd, od, (i % 8) * 1024, 1024)
        eng.sync()
        rate = k * 1024 / (time.perf_counter() - t0)
    eng.dataset_free(xd, od)
    eng.close()
    return rate


res = {(name, m): [] for name, _, _ in CONFIGS for m in ("pageable", "pinned", "int16")}
ref = []
for r in range(ROUNDS):
    ref.append(resident(BATCHES))
    for name, env, n_slots in CONFIGS:
        for key in KNOBS:
            os.environ.pop(key, None)
        os.environ.update(env)
        eng = _capi.Engine(device=0, max_batch=1024, n_slots=n_slots)
        eng.load_weights(W)
        bufs = [eng.slot_input(s) for s in range(n_slots)]
        for s in range(n_slots):
            np.copyto(bufs[s], X[s])
        for m in ("pageable", "pinned", "int16"):
            boundary(eng, n_slots, m, bufs, 4 * n_slots)
            res[(name, m)].append(boundary(eng, n_slots, m, bufs, BATCHES))
        eng.close()
    print("round %d done" % r, flush=True)
ref.append(resident(BATCHES))
med = lambda v: sorted(v)[len(v) // 2]  # noqa: E731
print("resident (3 lanes, inputs in HBM): median %.2f M candidates/s  (%s)" % (med(ref) / 1e6, " ".join("%.2f" % (v / 1e6) for v in ref)))
print("%-32s %28s %28s %28s" % ("configuration", "pageable float32", "page-locked float32", "pageable int16 counts"))
for name, _, _ in CONFIGS:
    print("%-32s" % name + "".join("   %5.2f (%5.2f .. %5.2f) M/s" % (med(res[(name, m)]) / 1e6, min(res[(name, m)]) / 1e6, max(res[(name, m)]) / 1e6)
                                    for m in ("pageable", "pinned", "int16")))
