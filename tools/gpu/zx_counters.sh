#!/bin/sh
# L2 behaviour of the zx round trip (LSTM2 x-projection written by gemm_split_kernel, read by the recurrent kernel), two launches and
# fused: zx_counters.sh -> gpurun_out/r03_zx_counters.txt.  Counter passes only (--kernel-trace --pmc, no hip/hsa trace domains).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out
: > $O/r03_zx_counters.txt
for fused in 0 1; do
  for set in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum" "TCC_EA0_WRREQ_64B_sum TCC_REQ_sum"; do
    tag=$(echo $set | tr ' ' '_')
    rm -rf $O/pmc_zx_${fused}_$tag
    timeout 400 rocprofv3 --kernel-trace --pmc $set -d $O/pmc_zx_${fused}_$tag -o bench -- env BENCH_WARM_STEPS=0 CLAIR_AMD_LSTM2_FUSED=$fused python $R/bench.py --steps 8 --warmup 2 --streams 1 --no-cpu-baseline > $O/pmc_zx_${fused}_$tag.log 2>&1
    echo "## CLAIR_AMD_LSTM2_FUSED=$fused  counters: $set  (one slot, batch 1024, per launch)" >> $O/r03_zx_counters.txt
    python $R/tools/pmc_summary.py counters $O/pmc_zx_${fused}_$tag/bench_results.db >> $O/r03_zx_counters.txt 2>&1
  done
done
cat $O/r03_zx_counters.txt
