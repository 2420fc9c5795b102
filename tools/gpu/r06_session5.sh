#!/bin/sh
# round 6, GPU call 5: the final tree (a2 handled non-temporally on both sides) -- the -m gpu suite, smoke(), the three profile sets, the driver's
# command line, and the round's net effect against the round-5 build on the same box.
cd "$(dirname "$0")/../.."
O=gpurun_out
mkdir -p $O
S=$(date +%s)
timeout 1500 python -m pytest tests -m gpu -q --durations=10 > $O/r06_s5_gputests.txt 2>&1; echo "pytest rc $? in $(( $(date +%s) - S )) s" >> $O/r06_s5_gputests.txt
tail -18 $O/r06_s5_gputests.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/r06_s5_smoke.txt 2>&1; echo "smoke rc $?" >> $O/r06_s5_smoke.txt; cat $O/r06_s5_smoke.txt
tools/gpu/profile_r06.sh ont_b1024 196 > $O/r06_s5_profile_ont.log 2>&1; tail -3 $O/r06_s5_profile_ont.log | cut -c1-200
tools/gpu/profile_r06.sh ccs_b4096 48 --platform pacbio_ccs --batch 4096 > $O/r06_s5_profile_ccs.log 2>&1; tail -3 $O/r06_s5_profile_ccs.log | cut -c1-200
tools/gpu/profile_r06.sh illumina_b8192 24 --platform illumina --batch 8192 > $O/r06_s5_profile_illumina.log 2>&1; tail -3 $O/r06_s5_profile_illumina.log | cut -c1-200
S=$(date +%s)
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r06_driver_like_bench.json 2> $O/r06_driver_like_bench.err; echo "bench rc $? in $(( $(date +%s) - S )) s" > $O/r06_s5_bench_time.txt
cat $O/r06_s5_bench_time.txt; python -c "
import json; d=json.loads(open('$O/r06_driver_like_bench.json').read().strip().splitlines()[-1]); r=d['roofline']
print(d['value'], d['value_full_config'], d['value_sustained'], d['bench_wall_s'], d['gt_concordance_200k']['seconds'], r['frac'], r['frac_rocprof'], r['kernel_ms'], r['kernel_ms_rocprof'], r['traffic'], d['roofline_path'].get('fabric_tb_s'))"
{
echo "# the round's net effect on the hot path: round-5 build against the final tree, same box, alternating (tools/gpu/r06_session5.sh)"
echo "## ONT batch 1024"
tools/gpu/ab_multi.sh -r 3 r05=build_ab/libclair_amd_r05.so r06=-
echo "## CCS batch 4096"
tools/gpu/ab_multi.sh -r 2 -a "--platform pacbio_ccs --batch 4096 --steps 48 --warmup 4 --sustained-seconds 2" r05=build_ab/libclair_amd_r05.so r06=-
echo "## Illumina batch 8192"
tools/gpu/ab_multi.sh -r 2 -a "--platform illumina --batch 8192 --steps 24 --warmup 4 --sustained-seconds 2" r05=build_ab/libclair_amd_r05.so r06=-
} > $O/r06_ab_final.txt 2>&1
cat $O/r06_ab_final.txt
