#!/bin/sh
# Round 5, third GPU call: the round's profile set on the final build (three configurations + the driver's command line), l3l4 phase
# clocks, the round-4 arithmetic against the final build on the same box, one instrumented concordance / wait-all sample.
cd "$(dirname "$0")/../.."
O=gpurun_out
mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -q --durations=40 ) > $O/r05_s3_gputests.txt 2>&1; echo "pytest rc $?" >> $O/r05_s3_gputests.txt
nproc >> $O/r05_s3_gputests.txt; cat /sys/fs/cgroup/cpu.max >> $O/r05_s3_gputests.txt 2>&1; python -c "import os; print(len(os.sched_getaffinity(0)))" >> $O/r05_s3_gputests.txt
tools/gpu/profile_r05.sh ont_b1024 196 > $O/r05_profile_ont.log 2>&1
tools/gpu/profile_r05.sh ccs_b4096 48 --platform pacbio_ccs --batch 4096 > $O/r05_profile_ccs.log 2>&1
tools/gpu/profile_r05.sh illumina_b8192 24 --platform illumina --batch 8192 > $O/r05_profile_illumina.log 2>&1
timeout 300 python tools/gpu/l34_stamps.py 1024 > $O/r05_l34_stamps.txt 2>&1
timeout 900 tools/gpu/ab_multi.sh -r 3 r04arith=build_ab/libclair_amd_base.so l34dma4=build_ab/libclair_amd_l34dma4.so final=- > $O/r05_ab_final.txt 2>&1
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r05_driver_like_bench.json 2> $O/r05_driver_like_bench.err
timeout 600 python tools/gpu/waitall_compare.py 6 1 > $O/r05_waitall.txt 2>&1
grep -A45 "slowest" $O/r05_s3_gputests.txt | head -60; tail -8 $O/r05_s3_gputests.txt; tail -4 $O/r05_profile_ont.log | cut -c1-300; cat $O/r05_ab_final.txt; tail -8 $O/r05_l34_stamps.txt; tail -3 $O/r05_waitall.txt; cat $O/r05_ont_b1024_pmc_ea_requests.txt | head -40
