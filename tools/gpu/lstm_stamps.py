"""Where a step of the recurrent kernels goes: s_memtime stamps inside the step loop (exp/libclair_probe_lstm.so, tools/gpu/lstm_probe_build.py).
One batch alone on the chip, one slot, two launches for layer 2.  A step's MFMAs: 120 (layer 1) / 96 (layer 2) x 32 matrix-pipe cycles.
usage: lstm_stamps.py [batch=1024] [probe library = exp/libclair_probe_lstm.so]"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, ".")
from clair_amd import _capi, synth, weights  # noqa: E402

lib_path = os.path.abspath(sys.argv[2] if len(sys.argv) > 2 else "exp/libclair_probe_lstm.so")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
print("# %s, batch %d" % (os.path.basename(lib_path), n))
os.environ["CLAIR_AMD_LSTM2_FUSED"] = "0"
os.environ["CLAIR_AMD_LSTM2_PAIR"] = "0"
w = weights.synthetic_weights(seed=20250928, head_gain=4.0)
x, _ = synth.synthetic_input(n, "ont", seed=5)
eng = _capi.Engine(device=0, max_batch=n, n_slots=1, lib_path=lib_path)
eng.load_weights(w)
lib = ctypes.CDLL(lib_path)
for rep in range(3):
    eng.predict(x)
buf = np.zeros(2 * 512 * 4 * 8 * 8, np.uint64)
assert lib.clair_probe_lstm_stamps(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_longlong(buf.size)) == 0
s = buf.reshape(2, 512, 4, 8, 8).astype(np.int64)[:, :n // 16]       # [layer][workgroup][wave][step][point]
names = ["block 0 (+ the previous step's last gates in its gaps, layer 2: the seeds)", "block 1", "block 2", "block 3", "exposed tail (layer 1: next x-part; layer 2: block 3's gates)", "barrier"]
for layer in (0, 1):
    t = s[layer]
    ok = (t[..., 0] > 0) & (t[..., 6] > 0)
    print("layer %d (%d MFMAs per wave and step = %d matrix-pipe cycles): median s_memtime ticks per wave" % (layer + 1, 120 if layer == 0 else 96, (120 if layer == 0 else 96) * 32))
    for k in range(6):
        print("   %-82s %6d" % (names[k], np.median((t[..., k + 1] - t[..., k])[ok])))
    step = (t[:, :, 1:, 0] - t[:, :, :-1, 0])[ok[:, :, 1:] & ok[:, :, :-1]]
    head = (t[:, :, 1:, 0] - t[:, :, :-1, 6])[ok[:, :, 1:] & ok[:, :, :-1]]
    print("   %-82s %6d" % ("from the barrier to the next step's h fragments issued", np.median(head)))
    print("   %-82s %6d" % ("whole step", np.median(step)))
eng.close()
