"""Run-to-run determinism under repetition: the same batch through predict() many times, every result compared bit for bit with the
first; by batch size and kernel selection.  An intermittent race shows up as a non-zero count."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from clair_amd import _capi, synth, weights  # noqa: E402

w = weights.synthetic_weights(seed=20250928, head_gain=4.0)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
configs = [(4096, "1", "0", 1), (4096, "0", "0", 1), (1024, "0", "1", 1), (1024, "0", "0", 1), (2048, "1", "0", 1), (8192, "1", "0", 1)]
for n, pair, fused, slots in configs:
    os.environ["CLAIR_AMD_LSTM2_PAIR"] = pair
    os.environ["CLAIR_AMD_LSTM2_FUSED"] = fused
    xs = [synth.synthetic_input(n, "ont", seed=11 + n + 1000 * j)[0] for j in range(4)]      # four distinct batches, round-robin
    eng = _capi.Engine(device=0, max_batch=n, n_slots=slots)
    try:
        eng.load_weights(w)
        firsts, a1s, a2s = [], [], []
        for x in xs:
            firsts.append(eng.predict(x))
            a2s.append(eng.debug_read(0, 2, (33, n, 256)).copy())
            a1s.append(eng.debug_read(0, 1, (33, n, 256)).copy())
        bad = bad_a1 = bad_a2 = 0
        worst = 0.0
        t0 = time.time()
        for r in range(reps):
            x, first, a1, a2 = xs[r % 4], firsts[r % 4], a1s[r % 4], a2s[r % 4]
            got = eng.predict(x)
            same = all(np.array_equal(g, f) for g, f in zip(got, first))
            if not same:
                bad += 1
                worst = max(worst, max(float(np.abs(g - f).max()) for g, f in zip(got, first)))
                b1 = eng.debug_read(0, 1, (33, n, 256))
                b2 = eng.debug_read(0, 2, (33, n, 256))
                bad_a1 += int(not np.array_equal(a1, b1))
                bad_a2 += int(not np.array_equal(a2, b2))
                if bad <= 3:
                    d2 = np.argwhere(b2 != a2)
                    d1 = np.argwhere(b1 != a1)
                    print("   rep %d: outputs differ (max %.2e); a1 differs at %d places %s; a2 at %d places %s"
                          % (r, worst, len(d1), d1[:3].tolist(), len(d2), d2[:3].tolist()), flush=True)
        print("n=%d pair=%s fused=%s: %d / %d repetitions differ from the first (a1 %d, a2 %d), worst |dp| %.2e, %.0f s"
              % (n, pair, fused, bad, reps, bad_a1, bad_a2, worst, time.time() - t0), flush=True)
    finally:
        eng.close()
