#!/bin/sh
# The driver's bench call is 20 timed steps after 5 warm-up steps: fill, drain and the 7/7/6 split over three streams weigh on it.
cd "$(dirname "$0")/../.."
for s in 3 3 3 4 5 6; do
  for k in 20 21 60; do
    v=$(timeout 100 python bench.py --gpus 1 --steps $k --warmup 5 --streams $s --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")
    echo "streams $s steps $k: $v"
  done
done
