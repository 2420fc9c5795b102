#!/bin/sh
# Fused layer-2 launch (CLAIR_AMD_LSTM2_FUSED=1) against the two-launch path, same box, alternating.
cd "$(dirname "$0")/../.."
for b in ${BATCHES:-1024}; do
for f in 0 1 0 1; do
  for g in ${GROUPS_LIST:-4}; do
  v=$(CLAIR_AMD_LSTM2_FUSED=$f CLAIR_AMD_PROJ2_GROUPS=$g timeout 200 python bench.py --steps $((2000 * 1024 / b)) --batch $b --warmup 8 --unique-batches 4 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], {k: round(v['ms_mean'], 4) for k, v in d['kernels_in_flight_ms'].items() if (v['ms_mean'] or 0) > 0.001}, 'alone', {k: v for k, v in d['kernels_alone_ms'].items() if v}, d['parity_max_abs_diff'] if 'parity_max_abs_diff' in d else '')")
  echo "batch $b fused $f groups $g: $v"
  done
done
done
