"""Host stages of call_var per 4096-candidate batch, without a GPU: binary records -> (X, infos), native decode (random probabilities), VCF write."""
import cProfile, io, pstats, sys, time, os
sys.path.insert(0, ".")
import numpy as np
from clair_amd import synth, tensor_binary, call_var as cv

n = 40960
raw, infos = synth.synthetic_candidates(n, "ont", seed=77)
path = "gpurun_out/hs_e2e.bin"
with open(path, "wb") as f:
    f.write(tensor_binary.MAGIC)
    f.write(tensor_binary.pack_records(infos[0][0], [int(i[1]) for i in infos], [i[2] for i in infos], raw))
rng = np.random.default_rng(1)
def probs(k):
    def sm(a):
        a = np.exp(a * 3); return (a / a.sum(1, keepdims=True)).astype(np.float32)
    return [sm(rng.standard_normal((k, 21))), sm(rng.standard_normal((k, 3))), sm(rng.standard_normal((k, 33))), sm(rng.standard_normal((k, 33)))]
dec = cv.VariantDecoder(cv.OutputConfig(True, False, False, False, False, None))
wr = cv.VcfWriter("gpurun_out/hs_e2e.vcf", "S", None, False)
Y = probs(4096)
def run():
    f = open(path, "rb"); f.read(8)
    t = {"read": 0.0, "decode": 0.0, "write": 0.0}
    t0 = time.perf_counter()
    for x, inf, counts in tensor_binary.read_batches(f, 4096):
        t1 = time.perf_counter(); t["read"] += t1 - t0
        rows = dec.decode_batch(x, inf, [y[:len(x)] for y in Y])
        t2 = time.perf_counter(); t["decode"] += t2 - t1
        wr.write_rows(rows)
        t0 = time.perf_counter(); t["write"] += t0 - t2
    print({k: round(v / (n / 4096) * 1e3, 2) for k, v in t.items()}, "ms per 4096-batch", file=sys.stderr)
sys.stderr = open("gpurun_out/hs_prof.err", "w")
run()
pr = cProfile.Profile(); pr.enable(); run(); pr.disable()
sys.stderr.flush()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(18); print(s.getvalue()[:3500])
print(open("gpurun_out/hs_prof.err").read().splitlines()[-1]); os.remove(path); os.remove("gpurun_out/hs_e2e.vcf")
