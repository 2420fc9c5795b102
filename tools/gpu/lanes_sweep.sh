#!/bin/sh
# Forward passes in flight (compute lanes) against throughput, resident and through the host-array boundary (slots = 2 x lanes), alternating.
# usage: lanes_sweep.sh [lanes ...]   -> stdout
cd "$(dirname "$0")/../.."
LANES=${@:-"3 4 5 6"}
for i in 1 2; do
  for L in $LANES; do
    v=$(CLAIR_AMD_LANES=$L timeout 300 python bench.py --streams $L --boundary-slots $((2 * L)) --steps 1000 --warmup 8 --no-cpu-baseline --full-candidates 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('resident', d['value'], d['ms_per_step'], 'boundary f32', d.get('value_boundary'), 'int16', d.get('value_boundary_int16'), {k: round(v['ms_mean'], 4) for k, v in d['kernels_in_flight_ms'].items() if (v['ms_mean'] or 0) > 0.001}, d['gpu_state']['value'])")
    echo "$L lanes: $v"
  done
done
