#!/bin/sh
# bench.py over slot counts and projection-GEMM launch geometries
cd "$(dirname "$0")/../.."
run() { CLAIR_AMD_PROJ2_GROUPS=$1 timeout 200 python bench.py --streams $2 --steps 400 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('groups $1 streams $2', d['value'], {k:v['ms_mean'] for k,v in d['kernels_in_flight_ms'].items() if v['ms_mean']})"; }
for g in 4 5; do for st in 3 4 5 6; do run $g $st; done; done
