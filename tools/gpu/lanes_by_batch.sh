cd $GRAFT_REPO_ROOT
for i in 1 2; do for cfg in "1024 ont 1000" "4096 pacbio_ccs 300" "8192 illumina 150"; do set -- $cfg; for L in 3 4; do
    v=$(timeout 300 python bench.py --batch $1 --platform $2 --streams $L --steps $3 --warmup 4 --unique-batches 4 --no-cpu-baseline --boundary-slots 0 --full-candidates 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], {k: round(v['ms_mean'], 4) for k, v in d['kernels_in_flight_ms'].items() if (v['ms_mean'] or 0) > 0.001})")
    echo "batch $1 $2, $L lanes: $v"
done; done; done
