#!/bin/sh
# throughput and per-kernel times by batch size (BASELINE.json configs 1, 2, 4)
cd "$(dirname "$0")/../.."
run() { timeout 300 python bench.py --batch $1 --platform $2 --steps $3 --streams ${4:-3} --unique-batches 4 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('batch $1 $2 streams', d['config']['streams'], d['value'], d['ms_per_step'], {k:v['ms_mean'] for k,v in d['kernels_in_flight_ms'].items() if v['ms_mean']}, 'alone', {k:v for k,v in d['kernels_alone_ms'].items() if v}, d['parity_max_abs_err'])"; }
run 1024 ont 400
run 4096 pacbio_ccs 100
run 4096 pacbio_ccs 100 2
run 8192 illumina 50
run 8192 illumina 50 2
