#!/bin/sh
# Stream priorities per slot (exp/libclair_prio.so: CLAIR_PRIO=1 high/normal/low, 2 high/low/low): short and long runs.
cd "$(dirname "$0")/../.."
for rep in 1 2; do
for p in 0 1 2; do
  for k in 20 21 2000; do
    v=$(CLAIR_AMD_LIB=$PWD/exp/libclair_prio.so CLAIR_PRIO=$p timeout 100 python bench.py --gpus 1 --steps $k --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")
    echo "prio $p steps $k: $v"
  done
done
done
