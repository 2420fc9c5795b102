cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout 300 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r1h -o bench -- python $R/bench.py --steps 680 > $R/gpurun_out/prof_r1h.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r1h_s1 -o bench -- python $R/bench.py --steps 680 --streams 1 > $R/gpurun_out/prof_r1h_s1.log 2>&1
cd $R
ls gpurun_out/prof_r1h gpurun_out/prof_r1h_s1
python tools/rocpd_summary.py gpurun_out/prof_r1h/bench_results.db --phases 256,680,32 > gpurun_out/prof_r1h_summary.txt 2>&1
python tools/rocpd_summary.py gpurun_out/prof_r1h_s1/bench_results.db --phases 256,680,32 > gpurun_out/prof_r1h_s1_summary.txt 2>&1
tail -8 gpurun_out/prof_r1h_summary.txt; tail -7 gpurun_out/prof_r1h_s1_summary.txt
