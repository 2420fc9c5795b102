#!/bin/sh
# Two slots with the fused layer-2 launch: does a shorter chain make two batches in flight enough?
cd "$(dirname "$0")/../.."
for cfg in "3 0 4" "2 0 4" "2 0 8" "2 1 4" "2 1 5" "2 1 6" "3 1 4"; do
  set -- $cfg
  for k in 2000 20; do
  v=$(CLAIR_AMD_LSTM2_FUSED=$2 CLAIR_AMD_PROJ2_GROUPS=$3 CLAIR_AMD_FUSED_GROUPS=$3 timeout 200 python bench.py --steps $k --streams $1 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], {k: round(v['ms_mean'], 4) for k, v in d['kernels_in_flight_ms'].items() if (v['ms_mean'] or 0) > 0.001})")
  echo "streams $1 fused $2 groups $3 steps $k: $v"
  done
done
