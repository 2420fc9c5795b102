"""Where l3l4_kernel's time goes: per-wave s_memtime stamps at every phase boundary (CLAIR_AMD_L34_STAMPS=1), one batch alone on the chip.
Runs the PROBE build of the engine (clair_amd/build.py: build_probe).  Usage: python tools/gpu/l34_stamps.py [batch=1024]"""
import os
import sys

import numpy as np

os.environ["CLAIR_AMD_L34_STAMPS"] = "1"
sys.path.insert(0, ".")
from clair_amd import _capi, build, synth, weights  # noqa: E402

if not os.path.isfile(build.PROBE_OUT) or os.path.getmtime(build.PROBE_OUT) < os.path.getmtime(build.OUT):
    build.build_probe()

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
w = weights.synthetic_weights(seed=20250928, head_gain=4.0)
x, _ = synth.synthetic_input(n, "ont", seed=5)
eng = _capi.Engine(device=0, max_batch=n, n_slots=1, lib_path=build.PROBE_OUT)
eng.load_weights(w)
names = ["entry", "dma issued", "dma landed", "barrier", "L3 mfma", "barrier", "selu+split", "barrier", "L4 mfma", "barrier", "exchange+store"]
for rep in range(3):
    eng.predict(x)
    wgs = (n + 63) // 64 * 32
    raw = eng.debug_read(0, 5, (wgs * 4 * 16 * 2,)).view(np.uint64).reshape(wgs, 4, 16)[:, :, :11].astype(np.int64)
    zeros = int((raw[:, :, 0] == 0).sum())
    t0 = raw[:, :, 0][raw[:, :, 0] > 0].min()
    d = np.diff(raw, axis=2)
    if zeros:
        print("   (%d waves left no entry stamp)" % zeros)
    print("rep %d: %d workgroups; kernel span %d ticks (first entry -> last exit); entry spread %d; per-wave phase ticks (median / p90 / max):"
          % (rep, wgs, raw[:, :, 10].max() - t0, raw[:, :, 0].max() - t0))
    for i in range(10):
        v = d[:, :, i].ravel()
        print("   %-16s %8d %8d %8d" % (names[i + 1], np.median(v), np.percentile(v, 90), v.max()))
    life = (raw[:, :, 10] - raw[:, :, 0]).ravel()
    print("   %-16s %8d %8d %8d" % ("wave lifetime", np.median(life), np.percentile(life, 90), life.max()))
    for kind, sel in (("kh=0 waves", [0, 1]), ("kh=1 waves", [2, 3])):
        v = d[:, sel, 7].ravel()
        print("   L4 mfma, %s: median %d" % (kind, np.median(v)))
eng.close()
