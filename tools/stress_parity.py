"""Stress parity: many concurrent submits on several pipeline slots with random batch sizes and platforms, every output compared
with the oracle.  Looks for timing-dependent faults (the asm-MFMA kernels handle their hazards by construction, not by hipcc)."""
import sys
import numpy as np
sys.path.insert(0, ".")
from clair_amd import _capi, synth, weights
from oracle import c_oracle

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
w = weights.synthetic_weights(seed=20250928, head_gain=4.0)
eng = _capi.Engine(device=0, max_batch=1024, n_slots=3)
eng.load_weights(w)
rng = np.random.default_rng(1)
worst, total = 0.0, 0
for r in range(rounds):
    xs = []
    for slot in range(3):
        n = int(rng.integers(1, 1025)) if r % 4 else 1024
        x, _ = synth.synthetic_input(n, ("ont", "pacbio_ccs", "illumina")[int(rng.integers(0, 3))], seed=1000 * r + slot)
        xs.append(x)
        eng.submit(slot, x)
    for slot in range(3):
        got = eng.wait(slot)
        want = c_oracle.forward(w, xs[slot])
        err = max(float(np.abs(g - t).max()) for g, t in zip(got, want))
        assert all(np.isfinite(g).all() for g in got), "non-finite output in round %d slot %d" % (r, slot)
        assert err <= 1e-5, "round %d slot %d n=%d: max abs err %g" % (r, slot, len(xs[slot]), err)
        worst, total = max(worst, err), total + len(xs[slot])
print("stress parity: %d candidates in %d rounds x 3 slots, worst |dp| = %.3g" % (total, rounds, worst))
eng.close()
