#!/bin/sh
# One stream, wide batches: every kernel runs alone on the whole chip (the phase-aligned extreme of the schedule).
cd "$(dirname "$0")/../.."
for cfg in "1 3072" "1 4096" "1 8192" "1 16384" "2 4096" "2 8192" "3 4096"; do
  set -- $cfg
  v=$(timeout 200 python bench.py --gpus 1 --steps $((400000 / $2 * 2)) --warmup 5 --streams $1 --batch $2 --unique-batches 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], {k: round(v['ms_mean'], 4) for k, v in d['kernels_in_flight_ms'].items() if (v['ms_mean'] or 0) > 0.001}, d['roofline'].get('workgroups'))")
  echo "streams $1 batch $2: $v"
done
