"""Where the projection GEMM's time goes: s_memtime stamps at four points of every phase of a workgroup's first four tiles (exp/libclair_probe_gemm.so,
a sed/patch copy of the production sources under exp/probe_gemm: stamps compiled in, results not checked).  One batch alone on the chip, one slot
(256 workgroups) and the four-lane geometry (96).  usage: gemm_stamps.py [batch=1024]"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, ".")
from clair_amd import _capi, synth, weights  # noqa: E402

lib_path = os.path.abspath("exp/libclair_probe_gemm.so")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
w = weights.synthetic_weights(seed=20250928, head_gain=4.0)
x, _ = synth.synthetic_input(n, "ont", seed=5)
for groups in (8, 3):
    os.environ["CLAIR_AMD_PROJ2_GROUPS"] = str(groups)
    os.environ["CLAIR_AMD_LSTM2_FUSED"] = "0"
    eng = _capi.Engine(device=0, max_batch=n, n_slots=1, lib_path=lib_path)
    eng.load_weights(w)
    lib = ctypes.CDLL(lib_path)
    for rep in range(3):
        eng.predict(x)
    wgs = 32 * groups
    buf = np.zeros(512 * 4 * 64, np.uint64)
    assert lib.clair_probe_gemm_stamps(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_longlong(buf.size)) == 0
    s = buf.reshape(512, 4, 4, 4, 4)[:wgs].astype(np.int64)          # [wg][wave][tile][phase][point]
    ok = s[..., 0] > 0
    print("%d workgroups (%d groups per XCD): median s_memtime ticks per wave (~2 000 to the microsecond; a phase's 48 MFMAs are 1 536 matrix-pipe cycles)" % (wgs, groups))
    for it in range(1, 4):
        for ph in range(4):
            a, b, c, d = (s[:, :, it, ph, k] for k in range(4))
            nxt = s[:, :, it, ph + 1, 0] if ph < 3 else (s[:, :, it + 1, 0, 0] if it < 3 else None)
            sel = ok[:, :, it, ph]
            line = "   tile %d phase %d: first slab (24 MFMAs + 8 LDS reads) %5d | wait for the next phase's DMA %5d | barrier %5d" % (
                it, ph, np.median((b - a)[sel]), np.median((c - b)[sel]), np.median((d - c)[sel]))
            if nxt is not None:
                sel2 = sel & (nxt > 0)
                line += " | second slab (24 MFMAs + 8 LDS reads%s + 4 DMA) %5d | whole phase %5d" % (", 8 stores" if ph < 2 else "", np.median((nxt - d)[sel2]), np.median((nxt - a)[sel2]))
            print(line)
    eng.close()
