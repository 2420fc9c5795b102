#!/bin/sh
# Main-thread profile of call_var from binary records at two batch sizes.
cd "$(dirname "$0")/../.."
python - <<'PY'
import sys
sys.path.insert(0, ".")
from clair_amd import synth, tensor_binary, weights
n = 200000
weights.save_weights("gpurun_out/e2e_model", weights.synthetic_weights(seed=20250928, head_gain=4.0))
raw, infos = synth.synthetic_candidates(n, "ont", seed=77)
with open("gpurun_out/e2e_p.bin", "wb") as f:
    f.write(tensor_binary.MAGIC)
    for k in range(0, n, 8192):
        f.write(tensor_binary.pack_records(infos[k][0], [int(i[1]) for i in infos[k:k + 8192]], [i[2] for i in infos[k:k + 8192]], raw[k:k + 8192]))
PY
for bs in 1024 2048; do
  echo "== batch $bs"
  python -m cProfile -s tottime -m clair_amd.call_var --chkpnt_fn gpurun_out/e2e_model --tensor_fn gpurun_out/e2e_p.bin --call_fn gpurun_out/e2e.vcf --sampleName S --showRef --batch_size $bs 2>/dev/null | head -22 | cut -c1-150
done
rm -f gpurun_out/e2e_p.bin gpurun_out/e2e.vcf gpurun_out/e2e_model.npz
