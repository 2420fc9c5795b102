#!/bin/sh
cd "$(dirname "$0")/../.."
run() { GPU_MAX_HW_QUEUES=$4 CLAIR_AMD_LSTM2_PAIR=$1 CLAIR_AMD_PROJ2_GROUPS=$3 timeout 200 python bench.py --streams $2 --steps 400 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('pair $1 streams $2 groups $3 hwq $4', d['value'], {k:v['ms_mean'] for k,v in d['kernels_in_flight_ms'].items() if v['ms_mean']})"; }
for st in 4 5 6; do run 1 $st 4 8; done
for st in 4 5 6; do run 1 $st 2 8; done
for st in 4 5; do run 0 $st 2 8; done
