#!/bin/sh
# What each kernel is worth in the pipeline: bench throughput with its launches skipped (tools/gpu/ablate_build.sh).
# ids: lstm1=2 proj2=4 lstm2=8 l3l4=32 tail=64
cd "$(dirname "$0")/../.."
for m in 0 64 32 96 4 8 12 2 0; do
  v=$(CLAIR_AMD_LIB=$PWD/exp/libclair_ablate.so CLAIR_ABLATE=$m timeout 200 python bench.py --steps 400 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], {k:v['ms_mean'] for k,v in d['kernels_in_flight_ms'].items() if v['ms_mean']})")
  echo "mask $m: $v"
done
