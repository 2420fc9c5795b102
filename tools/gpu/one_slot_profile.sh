#!/bin/sh
# rocprofv3 kernel trace of the one-slot configuration (predict() pattern; fused layer-2 launch by default): one_slot_profile.sh [bench args]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out
timeout 300 python $R/bench.py --steps 196 --streams 1 --no-cpu-baseline "$@" > $O/r02_one_slot_bench.json 2>/dev/null
rm -rf $O/prof_one_slot
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_one_slot -o bench -- python $R/bench.py --steps 196 --streams 1 --no-cpu-baseline "$@" > $O/prof_one_slot.log 2>&1
python $R/tools/rocpd_summary.py $O/prof_one_slot/bench_results.db --phases 256,196,32 --resource-usage $R/profiles/r02_kernel_resource_usage.txt > $O/r02_one_slot_kernel_stats.txt 2>&1
head -c 400 $O/r02_one_slot_bench.json; echo; tail -8 $O/r02_one_slot_kernel_stats.txt
