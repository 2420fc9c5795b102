#!/bin/sh
# is the forward pass bound by the L2 <-> fabric path (it moves 3.8 TB/s of intermediates, 61 % of the 6.29 TB/s a
# streaming copy achieves) or by the board's power cap?  One table:
#   (a) the resident loop (sustained leg, 3 s) against a co-running streaming copy at rising intensity (tools/ubench/hbm_hog.hip, `hbm`:
#       the workgroup count sets its rate; the TB/s it achieved BESIDE the pipeline is printed by the hog itself), and
#   (b) against THE SAME instruction stream confined to the L1 / L2 (`l2`): the hog's CU slots, issue cycles and most of its watts without
#       its fabric traffic.  (a) - (b) at equal workgroup count is what the fabric traffic itself costs the pipeline.
# Usage: tools/gpu/fabric_sensitivity.sh [out file]   (GPU box; ~6 min)
cd "$(dirname "$0")/../.."
OUT=${1:-gpurun_out/r06_fabric_sensitivity.txt}
[ -x tools/ubench/hbm_hog ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/ubench/hbm_hog.hip -o tools/ubench/hbm_hog
BENCH="python bench.py --steps 200 --warmup 8 --sustained-seconds 3 --no-cpu-baseline --gt-candidates 0 --full-candidates 0 --boundary-slots 0"
line() {
  python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
g=d['gpu_state'].get('value_sustained') or {}
k={n: round(v['ms_mean']*1e3,1) for n,v in d['kernels_in_flight_ms'].items() if v['ms_mean']}
print('sustained %.0f cand/s  power %s W  sclk %s MHz  in-flight us %s' % (d['value_sustained'], g.get('power_w'), g.get('sclk_mhz'), k))"
}
{
echo "# fabric sensitivity, $(date -u +%Y-%m-%dT%H:%MZ), $(git rev-parse --short HEAD 2>/dev/null || echo tree) -- tools/gpu/fabric_sensitivity.sh"
echo "# hog alone (nothing else on the GPU), 8 s each:"
for b in 16 32 64 128 256; do tools/ubench/hbm_hog 8 $b hbm; done
tools/ubench/hbm_hog 8 256 l2
echo "# pipeline alone:"
for r in 1 2; do echo "alone: $($BENCH 2>/dev/null | line)"; done
for mode in hbm l2; do
  for b in 16 32 64 128 256; do
    tools/ubench/hbm_hog 45 $b $mode > /tmp/hog.log 2>&1 &
    HOG=$!
    sleep 2
    echo "$mode hog, $b workgroups: $($BENCH 2>/dev/null | line)"
    wait $HOG
    echo "    $(cat /tmp/hog.log)"
  done
done
echo "# pipeline alone again:"
echo "alone: $($BENCH 2>/dev/null | line)"
} > $OUT 2>&1
cat $OUT
