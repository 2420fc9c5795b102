#!/bin/sh
# round 6, GPU call 2: the whole -m gpu suite on the tree (with the new tests: GT contract, tie set, RCCL deadline against a stand-in library that hangs,
# unsorted text, config-4 share), smoke(), the driver's own command line, and the ONT batch-1024 profile set (kernel stats, traffic, matrix-pipe utilisation).
cd "$(dirname "$0")/../.."
O=gpurun_out
mkdir -p $O
S=$(date +%s)
timeout 1500 python -m pytest tests -m gpu -q --durations=12 > $O/r06_s2_gputests.txt 2>&1; echo "pytest rc $? in $(( $(date +%s) - S )) s" >> $O/r06_s2_gputests.txt
tail -25 $O/r06_s2_gputests.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/r06_s2_smoke.txt 2>&1; echo "smoke rc $?" >> $O/r06_s2_smoke.txt; cat $O/r06_s2_smoke.txt
tools/gpu/profile_r06.sh ont_b1024 196 > $O/r06_s2_profile_ont.log 2>&1; tail -30 $O/r06_s2_profile_ont.log
S=$(date +%s)
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r06_driver_like_bench.json 2> $O/r06_driver_like_bench.err; echo "bench rc $? in $(( $(date +%s) - S )) s" > $O/r06_s2_bench_time.txt
cat $O/r06_s2_bench_time.txt; python -c "
import json; d=json.loads(open('$O/r06_driver_like_bench.json').read().strip().splitlines()[-1]); r=d['roofline']
print(d['value'], d['value_full_config'], d['value_sustained'], d['bench_wall_s'], d['gt_concordance_200k']['seconds'], r['frac'], r['kernels'][r['kernel'].split()[0]]['frac'], r['kernel_ms'], r['kernel_ms_rocprof'], r['traffic'], d['roofline_path'].get('fabric_tb_s'), d['roofline_path'].get('fabric_frac_of_achievable'))
print({k: {kk: vv for kk, vv in v.items() if kk != 'flips'} for k, v in d['gt_concordance_200k']['platforms'].items()})"
