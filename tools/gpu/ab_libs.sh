#!/bin/sh
# A/B two builds of the engine on the same box, alternating: ab_libs.sh libA.so libB.so [bench args]
cd "$(dirname "$0")/../.."
A=$1; B=$2; shift 2
for i in 1 2 3; do
  for lib in $A $B; do
    v=$(CLAIR_AMD_LIB=$PWD/$lib timeout 200 python bench.py --steps 2000 --warmup 8 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], {k: round(v['ms_mean'], 4) for k, v in d['kernels_in_flight_ms'].items() if (v['ms_mean'] or 0) > 0.001}, 'alone', {k: v for k, v in d['kernels_alone_ms'].items() if v})")
    echo "$lib: $v"
  done
done
