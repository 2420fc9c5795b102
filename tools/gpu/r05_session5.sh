#!/bin/sh
# Round 5, last GPU call: the -m gpu suite once more (durations), CPU binding on a real sysfs, end-to-end rates on the final build.
cd "$(dirname "$0")/../.."
O=gpurun_out
S=$(date +%s)
timeout 1500 python -m pytest tests -m gpu -q --durations=8 > $O/r05_s5_gputests.txt 2>&1; echo "pytest rc $? in $(( $(date +%s) - S )) s" >> $O/r05_s5_gputests.txt
CLAIR_AMD_BIND=1 timeout 300 python bench.py --steps 100 --no-cpu-baseline --gt-candidates 0 --sustained-seconds 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('CLAIR_AMD_BIND=1:', d['per_rank'][0]['affinity'], 'value_sustained', d['value_sustained'], 'boundary', d['value_boundary'], d['value_boundary_int16'])" > $O/r05_bind_one_rank.txt 2>&1
ls /sys/bus/pci/devices/*/numa_node 2>/dev/null | head -3 >> $O/r05_bind_one_rank.txt; nproc >> $O/r05_bind_one_rank.txt
timeout 900 python tools/e2e_binary_sweep.py 2000000 4096,1024 > $O/r05_e2e_binary.txt 2>&1
timeout 600 python tools/e2e_bam_bench.py 10000000 > $O/r05_e2e_bam.txt 2>&1
tail -14 $O/r05_s5_gputests.txt; cat $O/r05_bind_one_rank.txt; tail -12 $O/r05_e2e_binary.txt; tail -12 $O/r05_e2e_bam.txt
