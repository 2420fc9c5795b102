#!/bin/sh
# final tree: the -m gpu suite, smoke(), and the driver's own command line (with its wall time)
cd "$(dirname "$0")/../.."
O=gpurun_out
S=$(date +%s)
timeout 1500 python -m pytest tests -m gpu -q > $O/r05_s10_gputests.txt 2>&1; echo "pytest rc $? in $(( $(date +%s) - S )) s" >> $O/r05_s10_gputests.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/r05_s10_smoke.txt 2>&1; echo "smoke rc $?" >> $O/r05_s10_smoke.txt
S=$(date +%s)
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r05_driver_like_bench.json 2> $O/r05_driver_like_bench.err; echo "bench rc $? in $(( $(date +%s) - S )) s" > $O/r05_s10_bench_time.txt
tail -4 $O/r05_s10_gputests.txt; cat $O/r05_s10_smoke.txt $O/r05_s10_bench_time.txt; python -c "
import json; d=json.loads(open('$O/r05_driver_like_bench.json').read().strip().splitlines()[-1]); print(d['value'], d['value_full_config'], d['value_sustained'], d['bench_wall_s'], d['gt_concordance_200k']['seconds'], d['roofline']['frac'], d['roofline']['traffic'])"
