#!/bin/sh
# two-tile LSTM2 (lstm32_pair.hip.h) against the one-tile kernel, by batch size
cd "$(dirname "$0")/../.."
run() { CLAIR_AMD_LSTM2_PAIR=$1 timeout 300 python bench.py --batch $2 --platform $3 --steps $4 --streams 3 --unique-batches 4 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('pair $1 batch $2', d['value'], 'lstm2 in flight', d['kernels_in_flight_ms']['lstm2']['ms_mean'], 'alone', d['kernels_alone_ms']['lstm2'], d['parity_max_abs_err'])"; }
for b in 2048 4096 8192; do for pp in 0 1 0 1; do run $pp $b ont $((204800 / b)); done; done
