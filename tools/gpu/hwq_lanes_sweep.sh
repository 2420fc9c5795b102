cd $GRAFT_REPO_ROOT
for Q in 4 6 8; do for L in 3 4 5 6; do
    v=$(GPU_MAX_HW_QUEUES=$Q CLAIR_AMD_LANES=$L timeout 300 python bench.py --streams $L --boundary-slots $((2 * L)) --steps 1000 --warmup 8 --no-cpu-baseline --full-candidates 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('resident', d['value'], d['ms_per_step'], 'boundary f32', d.get('value_boundary'), 'int16', d.get('value_boundary_int16'))")
    echo "GPU_MAX_HW_QUEUES=$Q, $L lanes: $v"
done; done
