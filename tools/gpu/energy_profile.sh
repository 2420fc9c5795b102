#!/bin/sh
# Energy per launch of each kernel: the bench loop with every OTHER kernel's launches skipped (tools/gpu/ablate_build.sh), three
# streams, rocm-smi socket power sampled while it runs.  energy/launch = mean power x ms_per_step.  Results are garbage; power and time only.
# ids: lstm1=2 proj2=4 lstm2=8 l3l4=32 tail=64 (sum 110)
cd "$(dirname "$0")/../.."
B=${1:-1024}
echo "idle:"; rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk"
for cfg in "all 0 40000" "lstm1 108 160000" "proj2 106 160000" "lstm2 102 155000" "l3l4 78 250000" "tail 46 600000" "rec 100 90000" "lstm1+proj2 104 65000" "dense 14 200000"; do
  set -- $cfg
  (CLAIR_AMD_LIB=$PWD/exp/libclair_ablate.so CLAIR_ABLATE=$2 timeout 200 python bench.py --batch $B --steps $3 --warmup 8 --no-cpu-baseline > /tmp/e_$1.json 2>/dev/null) &
  BP=$!
  sleep ${WAIT:-6}
  P=""; S=""
  while kill -0 $BP 2>/dev/null; do
    o=$(rocm-smi --showpower --showclocks 2>/dev/null)
    P="$P $(echo "$o" | grep -E "Power" | grep -oE "[0-9]+\.[0-9]+" | head -1)"
    S="$S $(echo "$o" | grep -E "sclk" | grep -oE "\([0-9]+Mhz\)" | grep -oE "[0-9]+" | head -1)"
    sleep 0.3
  done
  wait $BP
  python - "$1" "$P" "$S" <<'PY'
import json, sys
name, p, s = sys.argv[1], [float(v) for v in sys.argv[2].split()], [int(v) for v in sys.argv[3].split()]
keep = [i for i in range(min(len(p), len(s)) - 1) if s[i] >= 1200 and s[i + 1] >= 1200]    # the GPU idles (sclk ~150 MHz) during the closing CPU parity check
p, s = [p[i] for i in keep] or [0.0], [s[i] for i in keep] or [0]
d = json.loads(open('/tmp/e_%s.json' % name).read().strip().splitlines()[-1])
pw = sum(p) / max(len(p), 1)
print("%-11s  %8.4f ms/step  %7.1f W (min %.0f max %.0f)  %8.2f mJ/step   sclk %4d MHz (%d samples)  in flight %s" % (name, d['ms_per_step'], pw, min(p), max(p), pw * d['ms_per_step'], sum(s) / max(len(s), 1), len(p),
      {k: round(v['ms_mean'], 4) for k, v in d.get('kernels_in_flight_ms', {}).items() if (v['ms_mean'] or 0) > 0.01}))
PY
done
