#!/bin/sh
# Energy per 1024-batch of every kernel of the forward pass, two ways:
#   alone        -- the bench loop (3 lanes) with every OTHER kernel's launches skipped: what the kernel draws when it has the chip;
#   in the mix   -- the full pipeline minus that kernel: E(all) - E(all but k) is what it adds to the 3-lane mix.
# Power and shader clock are bench.py's own `gpu_state` of the timed leg (sysfs, 1 ms period; tools/gpu_state_sampler.py);
# mJ per step = mean socket power x ms per step.  Results of ablated runs are garbage; power and time only.
# Needs exp/libclair_ablate.so (tools/gpu/ablate_build.sh).  ids: lstm1=2 proj2=4 lstm2=8 l3l4=32 tail=64 (sum 110)
# usage: energy_by_kernel.sh [batch=1024]   -> gpurun_out/r04_energy_by_kernel.txt
cd "$(dirname "$0")/../.."
B=${1:-1024}
O=gpurun_out/r04_energy_by_kernel.txt
mkdir -p gpurun_out
run() {   # name mask steps
  CLAIR_AMD_LIB=$PWD/exp/libclair_ablate.so CLAIR_ABLATE=$2 timeout 200 python bench.py --batch $B --steps $3 --warmup 8 --no-cpu-baseline --boundary-slots 0 --full-candidates 0 > /tmp/e_$1.json 2>/tmp/e_$1.err
  python - "$1" "$2" <<'PY'
import json, sys
name, mask = sys.argv[1], int(sys.argv[2])
try:
    d = json.loads(open('/tmp/e_%s.json' % name).read().strip().splitlines()[-1])
except Exception as ex:
    print("%-22s failed: %s" % (name, ex)); sys.exit(0)
g = (d.get("gpu_state") or {}).get("value") or {}
pw, mhz = g.get("power_w"), g.get("sclk_mhz")
ms = d["ms_per_step"]
print("%-22s mask %3d  %8.4f ms/step  %7.1f W  %8.2f mJ/step  sclk %s MHz  (%s)  in flight %s" % (
    name, mask, ms, pw or -1, (pw or 0) * ms, mhz, g.get("samples"),
    {k: round(v['ms_mean'], 4) for k, v in d.get('kernels_in_flight_ms', {}).items() if (v['ms_mean'] or 0) > 0.001}))
PY
}
{
echo "# energy per $B-candidate batch by kernel; $(date -u +%Y-%m-%dT%H:%MZ); $(rocm-smi --showproductname 2>/dev/null | grep -m1 -i 'card series' || true)"
echo "# idle: $(rocm-smi --showpower --showclocks 2>/dev/null | grep -E 'Power|sclk' | tr '\n' ' ')"
run all 0 40000
echo "# alone (every other kernel skipped)"
run lstm1_alone 108 160000
run proj2_alone 106 160000
run lstm2_alone 102 160000
run l3l4_alone 78 300000
run tail_alone 46 500000
echo "# in the mix (the pipeline without the kernel; its share = all - this)"
run without_lstm1 2 50000
run without_proj2 4 50000
run without_lstm2 8 50000
run without_l3l4 32 50000
run without_tail 64 45000
run without_l3l4_tail 96 55000
run all_again 0 40000
} > $O 2>&1
cat $O
