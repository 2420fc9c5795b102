"""PCIe-inclusive rate of the host-buffer boundary (clair_submit / clair_wait, NumPy arrays on both sides), 3 slots in flight.

Three ways of handing a batch over:  pageable NumPy array (what a drop-in `predict(batchX)` gets);  the slot's page-locked
input buffer (clair_slot_input) filled by a memcpy from a pageable array (a producer that cannot be changed);  the same buffer
already holding the batch (a producer that writes there directly, as clair_host_parse_tensors / the pileup hand-off can);
raw int16 counts (clair_submit_counts: the subtraction of channel 0 and the conversion run on the device)."""
import sys
import time
import numpy as np
sys.path.insert(0, ".")
from clair_amd import _capi, synth, weights

eng = _capi.Engine(device=0, max_batch=1024, n_slots=3)
eng.load_weights(weights.synthetic_weights(seed=20250928, head_gain=4.0))
xs = [synth.synthetic_input(1024, "ont", seed=s)[0] for s in range(3)]
bufs = [eng.slot_input(s) for s in range(3)]
ref = [eng.predict(x) for x in xs]
cs = []
for x in xs:            # the raw counts the synthetic tensors stand for: channel 0 as is, channels 1..3 + channel 0
    c = x.copy()
    c[..., 1:] += c[..., 0:1]
    cs.append(c.astype(np.int16))


def loop(mode, rounds):
    outs = [None] * 3
    t0 = time.perf_counter()
    for r in range(rounds):
        for s in range(3):
            if r:
                outs[s] = eng.wait(s)
            if mode == "pageable":
                eng.submit(s, xs[s])
            elif mode == "int16 counts":
                eng.submit_counts(s, cs[s])
            else:
                if mode == "pinned+memcpy" or r == 0:
                    np.copyto(bufs[s], xs[s])
                eng.submit(s, bufs[s])
    for s in range(3):
        outs[s] = eng.wait(s)
    dt = time.perf_counter() - t0
    for s in range(3):
        assert all(np.array_equal(a, b) for a, b in zip(outs[s], ref[s])), mode
    return rounds * 3 * 1024 / dt


for mode in ("pageable", "pinned+memcpy", "pinned", "int16 counts"):
    loop(mode, 5)
    rate = loop(mode, 200)
    print("%-14s batch 1024, 3 slots: %9.0f candidates/s (%.1f GB/s H2D)"
          % (mode, rate, rate * (2112 if mode == "int16 counts" else 4224) / 1e9))
