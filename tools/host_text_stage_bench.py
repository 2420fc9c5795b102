"""Text tensor records -> (X, infos) batches: time per 4096-candidate batch of utils.tensor_generator_from (plain file), no GPU."""
import cProfile
import io
import os
import pstats
import sys
import time

sys.path.insert(0, ".")
from clair_amd import synth, utils  # noqa: E402

n = 40960
os.makedirs("gpurun_out", exist_ok=True)
path = "gpurun_out/hs_text.txt"
raw, infos = synth.synthetic_candidates(n, "ont", seed=77)
with open(path, "w") as f:
    for line in synth.tensor_records(raw, infos):
        f.write(line if line.endswith("\n") else line + "\n")
size = os.path.getsize(path)
err = sys.stderr
sys.stderr = io.StringIO()
for rep in range(2):
    t0 = time.perf_counter()
    k = sum(len(b[1]) for b in utils.tensor_generator_from(path, 4096))
    dt = time.perf_counter() - t0
pr = cProfile.Profile()
pr.enable()
sum(len(b[1]) for b in utils.tensor_generator_from(path, 4096))
pr.disable()
sys.stderr = err
print("text ingest: %d records, %.1f MB in %.3f s = %.0f records/s, %.2f ms per 4096-batch, %.0f MB/s" % (k, size / 1e6, dt, k / dt, dt / (n / 4096) * 1e3, size / dt / 1e6))
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(10)
print(s.getvalue()[-1600:])
os.remove(path)
