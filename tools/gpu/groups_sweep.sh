#!/bin/sh
# projection GEMM launch geometry: workgroup groups per XCD (grid = 32 x groups) by number of slots
cd "$(dirname "$0")/../.."
run() { CLAIR_AMD_PROJ2_GROUPS=$1 timeout 200 python bench.py --streams $2 --steps 400 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('groups $1 streams $2', d['value'], 'proj2 in-flight', d['kernels_in_flight_ms']['proj2']['ms_mean'], 'alone', d['kernels_alone_ms']['proj2'], 'parity', d['parity_max_abs_err'])"; }
for g in ${GROUPS_LIST:-3 4 6 8}; do run $g 3; done
for g in 6 8; do run $g 1; done
