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
# stamps (dense.hip.h): producers 0 entry, 1 first tile landed, then per unit g: 2+3g L3 done, 3+3g selu+split done (before B1), 4+3g after B2; 14 exit
#                       consumers 0 entry, 1 before Bp, then per unit g: 2+3g after B1, 3+3g after B2, 4+3g L4 done; 14 partials stored
for rep in range(3):
    eng.predict(x)
    wgs = (n + 63) // 64 * 8
    raw = eng.debug_read(0, 5, (wgs * 8 * 16 * 2,)).view(np.uint64).reshape(wgs, 8, 16)[:, :, :15].astype(np.int64)
    P, C = raw[:, :4], raw[:, 4:]
    med = lambda v: int(np.median(v))  # noqa: E731
    # s_memtime counts per XCD (the counters of different XCDs are not aligned): only differences inside one wave mean anything.
    # A wave's lifetime against the kernel's HIP-event duration puts about 2 000 ticks in a microsecond.
    print("rep %d: %d workgroups; median s_memtime ticks per wave:" % (rep, wgs))
    print("   producers: first tile in flight %d" % med(P[:, :, 1] - P[:, :, 0]))
    for g in range(4):
        start = P[:, :, 1] if g == 0 else P[:, :, 4 + 3 * (g - 1)]
        print("   unit %d  producers: L3 %5d  rendezvous+DMA issue+selu+split %5d  wait B1 + store + tile landed + B2 %5d   |   consumers: wait for B2 %5d  L4 %5d"
              % (g, med(P[:, :, 2 + 3 * g] - start), med(P[:, :, 3 + 3 * g] - P[:, :, 2 + 3 * g]), med(P[:, :, 4 + 3 * g] - P[:, :, 3 + 3 * g]),
                 med(C[:, :, 3 + 3 * g] - (C[:, :, 1] if g == 0 else C[:, :, 4 + 3 * (g - 1)])), med(C[:, :, 4 + 3 * g] - C[:, :, 3 + 3 * g])))
    print("   consumers: K-half exchange + partial stores %d;  wave lifetime median %d, max %d"
          % (med(C[:, :, 14] - C[:, :, 13]), med(raw[:, :, 14] - raw[:, :, 0]), int((raw[:, :, 14] - raw[:, :, 0]).max())))
eng.close()
