#!/bin/sh
# Only the matrix-pipe utilisation pass of profile_r02.sh: mfma_pass.sh <tag> <bench.py arguments ...>
TAG=$1; shift 1
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out
BATCH=1024; for a in "$@"; do [ "$prev" = "--batch" ] && BATCH=$a; prev=$a; done
rm -rf $O/pmc_r02_${TAG}_sq
timeout 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d $O/pmc_r02_${TAG}_sq -o bench -- env BENCH_WARM_STEPS=0 CLAIR_AMD_LSTM2_FUSED=${FUSED:-0} python $R/bench.py --steps 8 --warmup 2 --streams 1 --no-cpu-baseline "$@" > $O/pmc_r02_${TAG}_sq.log 2>&1
python $R/tools/pmc_summary.py mfma $O/pmc_r02_${TAG}_sq/bench_results.db --batch $BATCH --groups 8 > $O/r02_${TAG}_pmc_mfma_util${SUFFIX}.txt 2>&1
cat $O/r02_${TAG}_pmc_mfma_util${SUFFIX}.txt
