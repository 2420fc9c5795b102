"""call_var end to end from binary tensor records (one GPU): python tools/e2e_binary_sweep.py [n=2000000] [batch sizes ...]

n candidates = a base set of 200 000 synthetic ONT candidates written n / 200 000 times (positions shifted), so that the run is long
enough for the steady state to show beside the ~0.4 s of process start-up.  Every configuration runs with the decode on the device
(the default: call records come back from the GPU, include/clair_call.h) and with CLAIR_AMD_DEVICE_DECODE=0 (probabilities come back,
the native decoder resolves them on the host); the two VCFs must be byte-identical."""
import hashlib
import os
import subprocess
import sys
import time

sys.path.insert(0, ".")
from clair_amd import synth, tensor_binary, weights

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000000
sizes = [int(a) for a in sys.argv[2:]] or [1024, 4096]
os.makedirs("gpurun_out", exist_ok=True)
prefix = "/tmp/e2e_model"
weights.save_weights(prefix, weights.synthetic_weights(seed=20250928, head_gain=4.0))
base = min(n, 200000)
raw, infos = synth.synthetic_candidates(base, "ont", seed=77)
binary = "/tmp/e2e_%d.bin" % n
with open(binary, "wb") as f:
    f.write(tensor_binary.MAGIC)
    for rep in range((n + base - 1) // base):
        m = min(base, n - rep * base)
        for k in range(0, m, 8192):
            f.write(tensor_binary.pack_records(infos[k][0], [int(i[1]) + rep * 7 * base for i in infos[k:min(k + 8192, m)]],
                                               [i[2] for i in infos[k:min(k + 8192, m)]], raw[k:min(k + 8192, m)]))
print("%d candidates, %.2f GB of binary records" % (n, os.path.getsize(binary) / 1e9), flush=True)
digests = {}
for rep in range(2):
    for bs in sizes:
        for dev in ("1", "0"):
            out = "/tmp/e2e_%s.vcf" % dev
            t0 = time.perf_counter()
            p = subprocess.run([sys.executable, "-m", "clair_amd.call_var", "--chkpnt_fn", prefix, "--tensor_fn", binary, "--call_fn", out,
                                "--sampleName", "S", "--showRef", "--batch_size", str(bs)], stderr=subprocess.PIPE, text=True,
                               env=dict(os.environ, CLAIR_AMD_DEVICE_DECODE=dev))
            dt = time.perf_counter() - t0
            if p.returncode != 0:
                print(p.stderr[-2000:])
                sys.exit(1)
            inner = [l for l in p.stderr.splitlines() if "Total time elapsed" in l]
            secs = float(inner[-1].split("elapsed:")[1].split()[0]) if inner else float("nan")
            digests[(bs, dev)] = hashlib.sha256(open(out, "rb").read()).hexdigest()
            print("batch %5d, decode on the %s: %d candidates in %.2f s = %.0f candidates/s end to end; inside call_variants %.2f s = %.0f candidates/s"
                  % (bs, "device" if dev == "1" else "host  ", n, dt, n / dt, secs, n / secs), flush=True)
        assert digests[(bs, "1")] == digests[(bs, "0")], "the VCF differs between device and host decode"
print("VCFs byte-identical between device and host decode (sha256 %s)" % digests[(sizes[0], "1")][:16])
for f in (binary, "/tmp/e2e_0.vcf", "/tmp/e2e_1.vcf", prefix + ".npz"):
    if os.path.exists(f):
        os.remove(f)
