"""One-slot predict() at large batches vs the float32 oracle, by kernel selection."""
import os
import sys

import numpy as np

sys.path.insert(0, ".")
from clair_amd import _capi, synth, weights  # noqa: E402
from oracle import c_oracle  # noqa: E402

w = weights.synthetic_weights(seed=20250928, head_gain=4.0)
for n in (2048, 4096, 8192):
    x, _ = synth.synthetic_input(n, "ont", seed=5 + n)
    want = c_oracle.forward(w, x)
    for pair, fused, slots in (("0", "0", 1), ("1", "0", 1), ("", "", 1), ("", "", 3)):
        for k, v in (("CLAIR_AMD_LSTM2_PAIR", pair), ("CLAIR_AMD_LSTM2_FUSED", fused)):
            if v:
                os.environ[k] = v
            else:
                os.environ.pop(k, None)
        eng = _capi.Engine(device=0, max_batch=n, n_slots=slots)
        try:
            eng.load_weights(w)
            got = eng.predict(x)
            d = [float(np.abs(g - t).max()) for g, t in zip(got, want)]
            worst = int(np.argmax(np.abs(got[0] - want[0]).max(axis=1)))
            print("n=%d pair=%r fused=%r slots=%d: max |dp| %s  worst candidate %d (tile %d)" % (n, pair, fused, slots, ["%.2e" % v for v in d], worst, worst // 32), flush=True)
        finally:
            eng.close()
