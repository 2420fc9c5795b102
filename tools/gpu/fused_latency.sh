#!/bin/sh
# One batch at a time (one slot -- the reference's own predict() pattern): fused layer-2 launch vs two launches, by producer groups.
cd "$(dirname "$0")/../.."
for b in ${BATCHES:-1024 512 256}; do
for cfg in "0 8" "1 4" "1 5" "1 6" "1 8" "0 8"; do
  set -- $cfg
  v=$(CLAIR_AMD_LSTM2_FUSED=$1 CLAIR_AMD_PROJ2_GROUPS=$2 timeout 200 python bench.py --steps 1500 --batch $b --streams 1 --warmup 8 --unique-batches 4 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], {k: round(v['ms_mean'], 4) for k, v in d['kernels_in_flight_ms'].items() if (v['ms_mean'] or 0) > 0.001})")
  echo "batch $b streams 1 fused $1 groups $2: $v"
done
done
