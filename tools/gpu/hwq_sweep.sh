cd $GRAFT_REPO_ROOT
run() { timeout 200 python bench.py --streams $1 --steps 400 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('hwq $GPU_MAX_HW_QUEUES streams $1', d['value'], {k:v['ms_mean'] for k,v in d['kernels_in_flight_ms'].items() if v['ms_mean']})"; }
for q in 8 16; do export GPU_MAX_HW_QUEUES=$q; for st in 3 4 5 6 8; do run $st; done; done
