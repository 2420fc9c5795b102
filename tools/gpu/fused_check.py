"""Fused layer-2 kernel (CLAIR_AMD_LSTM2_FUSED=1) against the two-launch path: outputs and the LSTM2 tap must be bit-identical."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from clair_amd import _capi, synth, weights  # noqa: E402

w = weights.synthetic_weights(seed=20250928, head_gain=4.0)
ok = True
for n in [int(v) for v in (sys.argv[1:] or ["64", "1024", "200", "96", "2048"])]:
    x, _ = synth.synthetic_input(n, "ont", seed=900 + n)
    res = {}
    for mode in ("0", "1"):
        os.environ["CLAIR_AMD_LSTM2_FUSED"] = mode
        os.environ["CLAIR_AMD_LSTM2_PAIR"] = "0"
        eng = _capi.Engine(device=0, max_batch=max(n, 64), n_slots=1)
        try:
            eng.load_weights(w)
            t0 = time.perf_counter()
            outs = eng.predict(x)
            outs2 = eng.predict(x)     # a second pass: tickets move on, stale words of the first pass must not satisfy anyone early
            dt = time.perf_counter() - t0
            n_pad = (n + 31) // 32 * 32
            res[mode] = (outs, outs2, eng.debug_read(0, 2, (33, n_pad, 256))[:, :n])
        finally:
            eng.close()
    same = all(np.array_equal(a, b) for a, b in zip(res["0"][0], res["1"][0])) and all(np.array_equal(a, b) for a, b in zip(res["0"][0], res["1"][1])) \
        and np.array_equal(res["0"][2], res["1"][2])
    d = max(float(np.abs(a - b).max()) for a, b in zip(res["0"][0], res["1"][0]))
    print("n=%d: fused == unfused: %s (max |diff| %.3g, a2 max diff %.3g)" % (n, same, d, float(np.abs(res["0"][2] - res["1"][2]).max())), flush=True)
    ok &= same
sys.exit(0 if ok else 1)
