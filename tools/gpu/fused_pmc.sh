#!/bin/sh
# L2<->fabric traffic of the fused layer-2 launch: FETCH_SIZE / WRITE_SIZE passes (separate rocprofv3 runs, --kernel-trace only).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/pmc_fused_$c
  CLAIR_AMD_LSTM2_FUSED=${FUSED:-1} timeout 400 rocprofv3 --kernel-trace --pmc $c -d $O/pmc_fused_$c -o bench -- env BENCH_WARM_STEPS=0 python $R/bench.py --steps 8 --warmup 2 --streams ${STREAMS:-3} --no-cpu-baseline "$@" > $O/pmc_fused_$c.log 2>&1
done
python $R/tools/pmc_summary.py traffic $O/pmc_fused_FETCH_SIZE/bench_results.db $O/pmc_fused_WRITE_SIZE/bench_results.db --batch 1024
