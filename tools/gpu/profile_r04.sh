#!/bin/sh
# Round-4 profile of one benchmark configuration.  usage: profile_r04.sh <tag> <steps> <bench.py arguments ...>
#   -> gpurun_out/r04_<tag>_{bench.json, kernel_stats.txt, pmc_hbm_traffic.txt, pmc_mfma_util.txt}   (copied into profiles/ afterwards)
# Counter passes are separate rocprofv3 runs with --kernel-trace only (never with the hip/hsa trace domains).  The matrix-pipe pass runs one
# slot so that every kernel has the chip to itself; it keeps the two layer-2 launches apart (CLAIR_AMD_LSTM2_FUSED=0), the kernels the
# three-slot pipeline runs.
TAG=$1; STEPS=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out
[ -f $O/pmc_traffic.json ] || cp $R/profiles/pmc_traffic.json $O/pmc_traffic.json   # merged per batch size, copied back into profiles/ afterwards
BATCH=1024; for a in "$@"; do [ "$prev" = "--batch" ] && BATCH=$a; prev=$a; done
rm -rf $O/prof_r04_${TAG}
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_r04_${TAG} -o bench -- python $R/bench.py --steps $STEPS --no-cpu-baseline "$@" > $O/prof_r04_${TAG}.json 2> $O/prof_r04_${TAG}.log
W=$(python -c "print(max(0, 256 - 8))")
python $R/tools/rocpd_summary.py $O/prof_r04_${TAG}/bench_results.db --phases-from $O/prof_r04_${TAG}.json --resource-usage $R/profiles/r04_kernel_resource_usage.txt > $O/r04_${TAG}_kernel_stats.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/pmc_r04_${TAG}_$c
  timeout 400 rocprofv3 --kernel-trace --pmc $c -d $O/pmc_r04_${TAG}_$c -o bench -- env BENCH_WARM_STEPS=0 python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline "$@" > $O/pmc_r04_${TAG}_$c.log 2>&1
done
python $R/tools/pmc_summary.py traffic $O/pmc_r04_${TAG}_FETCH_SIZE/bench_results.db $O/pmc_r04_${TAG}_WRITE_SIZE/bench_results.db --batch $BATCH --json $O/pmc_traffic.json --source profiles/r04_${TAG}_pmc_hbm_traffic.txt > $O/r04_${TAG}_pmc_hbm_traffic.txt 2>&1
rm -rf $O/pmc_r04_${TAG}_sq
timeout 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d $O/pmc_r04_${TAG}_sq -o bench -- env BENCH_WARM_STEPS=0 CLAIR_AMD_LSTM2_FUSED=0 python $R/bench.py --steps 8 --warmup 2 --streams 1 --no-cpu-baseline "$@" > $O/pmc_r04_${TAG}_sq.log 2>&1
python $R/tools/pmc_summary.py mfma $O/pmc_r04_${TAG}_sq/bench_results.db --batch $BATCH --groups 8 > $O/r04_${TAG}_pmc_mfma_util.txt 2>&1
rm -rf $O/prof_r04_${TAG} $O/pmc_r04_${TAG}_FETCH_SIZE $O/pmc_r04_${TAG}_WRITE_SIZE $O/pmc_r04_${TAG}_sq     # the tables above are what is kept (gpurun_out/ comes back only below 64 MiB)
# the bench line LAST: its `traffic` fields are reported only from a table measured on exactly these kernel sources, i.e. the passes above
cp $O/pmc_traffic.json $R/profiles/pmc_traffic.json
timeout 400 python $R/bench.py --steps $STEPS "$@" > $O/r04_${TAG}_bench.json 2> $O/r04_${TAG}_bench.err
cd $R
head -c 600 $O/r04_${TAG}_bench.json; echo; tail -12 $O/r04_${TAG}_kernel_stats.txt; cat $O/r04_${TAG}_pmc_hbm_traffic.txt $O/r04_${TAG}_pmc_mfma_util.txt
