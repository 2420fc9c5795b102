#!/bin/sh
# inter-kernel gaps per lane and the chip's demand over time, from a kernel trace of the resident loop
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
rm -rf $O/prof_gaps
timeout 400 rocprofv3 --kernel-trace -d $O/prof_gaps -o bench -- python $R/bench.py --steps 400 --warmup 8 --no-cpu-baseline --gt-candidates 0 --sustained-seconds 0 --boundary-slots 0 --full-candidates 0 > $O/prof_gaps.json 2> $O/prof_gaps.log
python $R/tools/lane_gaps.py $O/prof_gaps/bench_results.db --passes 300 640 > $O/r05_lane_gaps.txt 2>&1
rm -rf $O/prof_gaps
cat $O/r05_lane_gaps.txt
