#!/bin/sh
# Round-4 profile of the device front end.  usage: profile_frontend.sh <ref_len> <noisy_every>
#   -> gpurun_out/r04_frontend_{bench.txt, kernels.txt}   (copied into profiles/ afterwards)
# Three rocprofv3 runs of tools/gpu/frontend_bench.py: kernel trace, FETCH_SIZE, WRITE_SIZE (counter passes with --kernel-trace only).
LEN=${1:-2000000}; NOISY=${2:-10}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out
timeout 600 python $R/tools/gpu/frontend_bench.py $LEN $NOISY > $O/r04_frontend_bench.txt 2>&1
rm -rf $O/prof_fe $O/pmc_fe_FETCH_SIZE $O/pmc_fe_WRITE_SIZE
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_fe -o fe -- python $R/tools/gpu/frontend_bench.py $LEN $NOISY > $O/prof_fe.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c -d $O/pmc_fe_$c -o fe -- python $R/tools/gpu/frontend_bench.py $LEN $NOISY > $O/pmc_fe_$c.log 2>&1
done
EL=$(grep -o "[0-9]* elements" $O/r04_frontend_bench.txt | head -1 | cut -d" " -f1)
WI=$(grep -o "[0-9]* windows," $O/r04_frontend_bench.txt | head -1 | cut -d" " -f1)
python $R/tools/frontend_profile_summary.py $O/prof_fe/fe_results.db $O/pmc_fe_FETCH_SIZE/fe_results.db $O/pmc_fe_WRITE_SIZE/fe_results.db \
  --elements ${EL:-0} --positions $((LEN + 128)) --windows ${WI:-0} > $O/r04_frontend_kernels.txt 2>&1
cd $R
cat $O/r04_frontend_bench.txt $O/r04_frontend_kernels.txt
rm -rf $O/prof_fe $O/pmc_fe_FETCH_SIZE $O/pmc_fe_WRITE_SIZE
