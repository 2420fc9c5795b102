# HBM traffic of the default (3-slot) configuration: two PMC passes, kernels serialised by the profiler
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/pmc3_$c -o bench -- env BENCH_WARM_STEPS=0 python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline > $R/gpurun_out/pmc3_$c.log 2>&1
done
cd $R
python tools/pmc_traffic.py gpurun_out/pmc3_FETCH_SIZE/bench_results.db gpurun_out/pmc3_WRITE_SIZE/bench_results.db | tee gpurun_out/pmc3_traffic.txt
