"""call_var end to end from binary tensor records, by batch size (one GPU): python tools/e2e_binary_sweep.py [n]"""
import os
import subprocess
import sys
import time

sys.path.insert(0, ".")
from clair_amd import synth, tensor_binary, weights

n = int(sys.argv[1]) if len(sys.argv) > 1 else 400000
os.makedirs("gpurun_out", exist_ok=True)
prefix = "gpurun_out/e2e_model"
weights.save_weights(prefix, weights.synthetic_weights(seed=20250928, head_gain=4.0))
raw, infos = synth.synthetic_candidates(n, "ont", seed=77)
binary = "gpurun_out/e2e_%d.bin" % n
with open(binary, "wb") as f:
    f.write(tensor_binary.MAGIC)
    for k in range(0, n, 8192):
        f.write(tensor_binary.pack_records(infos[k][0], [int(i[1]) for i in infos[k:k + 8192]], [i[2] for i in infos[k:k + 8192]], raw[k:k + 8192]))
for rep in range(2):
    for bs in (1024, 2048, 4096, 8192):
        t0 = time.perf_counter()
        p = subprocess.run([sys.executable, "-m", "clair_amd.call_var", "--chkpnt_fn", prefix, "--tensor_fn", binary, "--call_fn", "gpurun_out/e2e.vcf",
                            "--sampleName", "S", "--showRef", "--batch_size", str(bs)], stderr=subprocess.PIPE, text=True)
        dt = time.perf_counter() - t0
        inner = [l for l in p.stderr.splitlines() if "Total time elapsed" in l]
        print("batch %5d: %d candidates in %.2f s = %.0f candidates/s end to end; %s" % (bs, n, dt, n / dt, inner[-1].strip() if inner else ""), flush=True)
for f in (binary, "gpurun_out/e2e.vcf", prefix + ".npz"):
    if os.path.exists(f):
        os.remove(f)
