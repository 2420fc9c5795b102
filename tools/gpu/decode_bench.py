"""The device decode kernel (decode.hip.h) by itself and inside the two-slot pipeline call_var runs: HIP-event time per launch and
what that is per candidate.  Usage: python tools/gpu/decode_bench.py [batch=1024] [steps=400]
(under `rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE`
tools/pmc_summary.py counters prints the per-launch counters: the kernel is latency bound -- one wave per candidate, ~600 instructions)"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from clair_amd import _capi, _hostapi, synth, weights  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
w = weights.synthetic_weights(seed=20250928, head_gain=4.0)
raw, infos = synth.synthetic_candidates(batch, "ont", seed=5)
counts = raw.astype(np.int16)
centre = _hostapi.centre_bytes(infos)
eng = _capi.Engine(device=0, max_batch=batch, n_slots=2)
eng.load_weights(w)
for i in range(16):
    eng.submit_calls(i % 2, counts, centre, counts=True)
    eng.wait(i % 2)
# (a) one slot at a time: every kernel alone
eng.timing_enable(True)
eng.timing_reset()
for i in range(32):
    eng.submit_calls(0, counts, centre, counts=True)
    eng.wait(0)
alone = eng.kernel_times()
# (b) two slots in flight, as call_var drives the engine
eng.timing_reset()
t0 = time.perf_counter()
eng.submit_calls(0, counts, centre, counts=True)
for i in range(1, steps):
    eng.submit_calls(i % 2, counts, centre, counts=True)
    eng.wait((i - 1) % 2)
eng.wait((steps - 1) % 2)
dt = time.perf_counter() - t0
flight = eng.kernel_times()
eng.timing_enable(False)
fmt = lambda t: {k: round(ms / cnt * 1e3, 1) for k, (ms, cnt) in t.items() if cnt}
print("batch %d: kernels alone (us per launch): %s" % (batch, fmt(alone)))
print("batch %d: two slots in flight (us per launch): %s" % (batch, fmt(flight)))
d = flight["decode"][0] / flight["decode"][1] * 1e3
print("decode kernel: %.1f us per %d candidates = %.1f ns per candidate = %.0f M candidates/s if it ran alone; 1 179 products + the arg-max per candidate: "
      "%.1f G products/s.  The submit/wait loop: %.0f candidates/s including the host link (2 114 B in, 32 B out per candidate)"
      % (d, batch, d * 1e3 / batch, batch / d, 1179 * batch / d / 1e3, steps * batch / dt))
eng.close()
