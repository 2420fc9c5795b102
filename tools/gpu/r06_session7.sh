#!/bin/sh
# round 6, GPU call 7: soak -- the resident loop for 120 s (clock, power and rate over time), the boundary path for 60 s through call_var's own driver
# (binary records -> VCF, device decode) repeated, and the cross-mode stress; everything bit-compared where it can be.
cd "$(dirname "$0")/../.."
O=gpurun_out
mkdir -p $O
{
echo "# soak: python bench.py --steps 196 --sustained-seconds 120 (ONT batch 1024, four lanes): the three resident rates, clock / power of the sustained leg"
timeout 600 python bench.py --steps 196 --warmup 8 --sustained-seconds 120 --no-cpu-baseline --gt-candidates 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); g=d['gpu_state']['value_sustained']
print('value %.0f full_config %.0f sustained %.0f over %.1f s (%d steps)  power %s W  sclk %s MHz  parity %.2e  boundary %s / %s' % (d['value'], d['value_full_config'], d['value_sustained'], d['sustained']['seconds'], d['sustained']['steps'], g['power_w'], g['sclk_mhz'], d['parity_max_abs_err'], d['value_boundary'], d['value_boundary_int16']))"
echo "# cross-mode stress (tools/gpu/cross_mode_stress.py 40): the same batches through eight slot / batch-size modes, bit-compared"
timeout 900 python tools/gpu/cross_mode_stress.py 40 2>&1 | tail -6
echo "# determinism (tools/gpu/determinism_stress.py 2000)"
timeout 900 python tools/gpu/determinism_stress.py 2000 2>&1 | tail -8
} > $O/r06_soak.txt 2>&1
cat $O/r06_soak.txt
