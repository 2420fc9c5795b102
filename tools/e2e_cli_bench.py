"""End-to-end CLI throughput: tensor text (gz) -> python -m clair_amd.call_var -> VCF, on one GPU; synthetic weights/candidates."""
import gzip
import os
import subprocess
import sys
import time

sys.path.insert(0, ".")
from clair_amd import synth, tensor_binary, weights

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
os.makedirs("gpurun_out", exist_ok=True)
prefix = "gpurun_out/e2e_model"
weights.save_weights(prefix, weights.synthetic_weights(seed=20250928, head_gain=4.0))
path = "gpurun_out/e2e_%d.txt.gz" % n
raw, infos = synth.synthetic_candidates(n, "ont", seed=77)
with gzip.open(path, "wt", compresslevel=1) as f:
    for line in synth.tensor_records(raw, infos):
        f.write(line if line.endswith("\n") else line + "\n")
plain = path[:-3]
with gzip.open(path, "rb") as src, open(plain, "wb") as dst:      # `gzip -fdc` passes uncompressed text through unchanged
    dst.write(src.read())
binary = path[:-7] + ".bin"
with open(binary, "wb") as f:                                      # the same candidates as fixed 2 192-byte records (create_tensor --binary)
    f.write(tensor_binary.MAGIC)
    for k in range(0, n, 8192):
        f.write(tensor_binary.pack_records(infos[k][0], [int(i[1]) for i in infos[k:k + 8192]], [i[2] for i in infos[k:k + 8192]], raw[k:k + 8192]))
kind = {path: "gz text", plain: "plain text", binary: "binary records"}
for extra, source in (([], path), (["--batch_size", "4096"], path), (["--batch_size", "4096"], plain), (["--batch_size", "4096"], binary),
                      (["--batch_size", "1024"], binary)):
    t0 = time.perf_counter()
    subprocess.check_call([sys.executable, "-m", "clair_amd.call_var", "--chkpnt_fn", prefix, "--tensor_fn", source,
                           "--call_fn", "gpurun_out/e2e.vcf", "--sampleName", "S", "--showRef"] + extra,
                          stderr=subprocess.DEVNULL)
    dt = time.perf_counter() - t0
    rows = sum(1 for l in open("gpurun_out/e2e.vcf") if not l.startswith("#"))
    print("CLI end to end %s, %s: %d candidates -> %d VCF rows in %.2f s = %.0f candidates/s (process start-up included)"
          % (" ".join(extra) or "(batch 1000)", kind[source], n, rows, dt, n / dt))
for f in (plain, binary, path, "gpurun_out/e2e.vcf"):      # keep gpurun_out small: it is copied back from the GPU box
    os.remove(f)
