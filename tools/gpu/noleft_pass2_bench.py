"""--stop_consider_left_edge: pass 2 per operation (default) against per base (CLAIR_AMD_FE_PASS2=base) on the front-end bench's alignments:
time of clair_frontend_build_windows_ex(consider_left_edge=0), windows compared.  usage: noleft_pass2_bench.py [ref_len=2000000] [noisy_every=10]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tools")
import fast_reads  # noqa: E402
from clair_amd import _capi  # noqa: E402

ref_len = int(sys.argv[1]) if len(sys.argv) > 1 else 2000000
noisy = int(sys.argv[2]) if len(sys.argv) > 2 else 10
case = fast_reads.make(ref_len=ref_len, depth=50, noisy_every=noisy, seed=5, read_len=(2000, 9000))
sam = case["sam"]
print("inputs: %.1f MB of SAM text over %d bases" % (len(sam) / 1e6, ref_len), flush=True)
out = {}
for mode in ("base", "op", "base", "op"):
    if mode == "base":
        os.environ["CLAIR_AMD_FE_PASS2"] = "base"
    else:
        os.environ.pop("CLAIR_AMD_FE_PASS2", None)
    f = _capi.Frontend(0, case["ref"], 0, -64, ref_len + 64)
    f.text_options(case["ctg"])
    step = 64 << 20
    at = 0
    while at < len(sam):
        cut = len(sam) if at + step >= len(sam) else sam.rindex(b"\n", at, at + step) + 1
        f.add_text(sam[at:cut])
        at = cut
    n_cand = f.find_candidates(min_coverage=4, threshold=0.125)
    t0 = time.perf_counter()
    n = f.build_windows(min_coverage=0, drop_non_iupac_centre=False, consider_left_edge=False)
    dt = time.perf_counter() - t0
    centres, _ = f.window_info(0, n)
    counts = f.window_counts(0, n)
    tuples = f.window_tuples()[1]
    print("pass 2 per %-4s: %d candidates, %d windows, build_windows %.2f ms (prefix sums, pass 2, flags, compaction, assembly)" % (mode, n_cand, n, dt * 1e3), flush=True)
    if mode in out:
        continue
    out[mode] = (centres, counts, tuples)
    f.close()
same = all(np.array_equal(a, b) for a, b in zip(out["base"], out["op"]))
print("windows, counts and tuples per window identical between the two: %s" % same)
