"""Production build against the wait-all CHECK build (clair_amd/build.py: build_waitall; csrc/common.hip.h: CLAIR_WAIT_ALL), bit for bit.

The check build waits for every outstanding memory operation at every hand-counted wait and drains memory, LDS and the matrix
pipe after every hand-placed MFMA, so none of the timing arguments the production kernels make is load-bearing in it.  Both builds
evaluate the same candidates -- CHUNKS distinct chunks of 262 144 (a base set of synthetic ONT / CCS / Illumina candidates with fresh
small random counts added per chunk, so every chunk is new data), each chunk PASSES times through every production mode (slots x
batch size -> every kernel selection, one to three batches in flight) -- and every production output row is compared with the
check build's row of the same candidate.  Per-candidate arithmetic does not depend on the batch, the slot or the build, so ANY
difference is a fault: an ordering hazard (it will differ between the builds), a race (it will also differ between passes), or
hardware.

Usage: python tools/gpu/waitall_compare.py [chunks=24] [passes=2]        (24 x 2 x 8 modes x 262 144 = 100.7 M production evaluations)
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from clair_amd import _capi, build, synth, weights  # noqa: E402
sys.path.insert(0, "tools")
from gt_concordance import box_info, gpu_state  # noqa: E402

chunks = int(sys.argv[1]) if len(sys.argv) > 1 else 24
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 2
N = 262144
MODES = ((3, 1024), (1, 4096), (2, 1024), (3, 4096), (1, 1024), (2, 2048), (3, 8192), (3, 512))

if not os.path.isfile(build.WAITALL_OUT) or os.path.getmtime(build.WAITALL_OUT) < os.path.getmtime(build.OUT):
    build.build_waitall()          # stale or absent: the check build must come from the same sources as the production one
box = box_info(full=False)
print("box: %s %s" % (box["host"], " ".join(box["unique_ids"])), flush=True)
w = weights.synthetic_weights(seed=20250928, head_gain=4.0)
base = {p: synth.synthetic_candidates(N // 8, p, seed=31 + i)[0].astype(np.int16) for i, p in enumerate(("ont", "pacbio_ccs", "illumina"))}
rng = np.random.default_rng(20250928)

check = _capi.Engine(device=0, max_batch=4096, n_slots=1, lib_path=build.WAITALL_OUT)
check.load_weights(w)
prods = []
for slots, batch in MODES:
    e = _capi.Engine(device=0, max_batch=batch, n_slots=slots)
    e.load_weights(w)
    prods.append((slots, batch, e))
xd, od = check.dataset_alloc(N)
datasets = [e.dataset_alloc(N) for _, _, e in prods]

total = bad_total = 0
t00 = time.time()
for c in range(chunks):
    # fresh data: the base candidates of one platform, tiled 8x, plus new random counts 0..2 on a third of the cells (raw counts, before ch1..3 -= ch0)
    plat = ("ont", "pacbio_ccs", "illumina")[c % 3]
    raw = np.tile(base[plat], (8, 1, 1, 1))
    noise = rng.integers(0, 3, size=raw.shape, dtype=np.int16) * (rng.integers(0, 3, size=raw.shape, dtype=np.int8) == 0)
    x = synth.to_model_input(raw + noise)
    check.dataset_upload(xd, 0, x)
    for b in range(N // 4096):
        check.run_resident(0, xd, od, b * 4096, 4096)
    check.sync()
    ref = check.dataset_download(od, 0, N)
    for (slots, batch, e), (pxd, pod) in zip(prods, datasets):
        e.dataset_upload(pxd, 0, x)
        for r in range(passes):
            for b in range(N // batch):
                e.run_resident(b % slots, pxd, pod, b * batch, batch)
            e.sync()
            out = e.dataset_download(pod, 0, N)
            diff = np.flatnonzero((out != ref).any(axis=1))
            total += N
            if len(diff):
                bad_total += len(diff)
                print("   chunk %d (%s) slots %d batch %d pass %d: %d candidates differ from the check build, first %s, max |d| %.3e"
                      % (c, plat, slots, batch, r, len(diff), diff[:8].tolist(), float(np.abs(out[diff] - ref[diff]).max())), flush=True)
    print("chunk %d (%s): %.1f M production evaluations so far, %d rows differ, %.0f s  [%s]" % (c, plat, total / 1e6, bad_total, time.time() - t00, gpu_state()), flush=True)
print("RESULT box %s: %.1f M production candidate evaluations (%d distinct candidates, %d modes x %d passes) against the wait-all build: %d rows differ"
      % (" ".join(box["unique_ids"]), total / 1e6, chunks * N, len(MODES), passes, bad_total))
