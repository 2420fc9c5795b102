#!/bin/sh
# Round 5, first GPU call: the v2 gate arithmetic against the round-4 build (same box, alternating), the whole GPU test suite on it,
# one wait-all re-sample, the Infinity-Cache probe, the counter list, and the driver's own command line.
cd "$(dirname "$0")/../.."
O=gpurun_out
mkdir -p $O
(rocprofv3 -L 2>&1 | grep -i -E "^\s*(gpu|Name)|TCC_EA|DRAM|MALL|UMC|HBM|TCC_HIT|TCC_MISS|TCC_REQ\b|BUBBLE|TCC_.*WRITEBACK" | head -200) > $O/r05_counters.txt 2>&1
timeout 120 tools/ubench/mall_probe 0.4 16 > $O/r05_mall_probe.txt 2>&1
timeout 600 python -m pytest tests -m gpu -x -q > $O/r05_s1_gputests.txt 2>&1; echo "pytest rc $?" >> $O/r05_s1_gputests.txt
timeout 900 tools/gpu/ab_multi.sh -r 3 base=build_ab/libclair_amd_base.so v2=- v2pair=-,CLAIR_AMD_LSTM2_PAIR=1 > $O/r05_ab_arith_v2.txt 2>&1
timeout 300 tools/gpu/ab_multi.sh -r 1 -a "--steps 400 --warmup 8 --sustained-seconds 1 --streams 1" s1_two=- s1_fused=-,CLAIR_AMD_LSTM2_FUSED=1 >> $O/r05_ab_fused_default.txt 2>&1
timeout 300 tools/gpu/ab_multi.sh -r 1 -a "--steps 400 --warmup 8 --sustained-seconds 1 --streams 2" s2_two=- s2_fused=-,CLAIR_AMD_LSTM2_FUSED=1 >> $O/r05_ab_fused_default.txt 2>&1
timeout 400 python tools/gpu/waitall_compare.py 4 1 > $O/r05_waitall.txt 2>&1
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r05_driver_like_bench.json 2> $O/r05_driver_like_bench.err
tail -3 $O/r05_s1_gputests.txt; cat $O/r05_ab_arith_v2.txt; cat $O/r05_ab_fused_default.txt; tail -4 $O/r05_waitall.txt; head -40 $O/r05_mall_probe.txt; wc -c $O/r05_driver_like_bench.json
