#!/bin/sh
# Where the joules of the recurrent kernels go: the kernels rebuilt WITHOUT one ingredient at a time, each timed alone on the chip like the rows of
# energy_by_kernel.sh (ablation switch CLAIR_ABLATE; power and clock from bench.py's gpu_state).  Results of these builds are garbage -- and the data
# they chew on is not the real data either, which moves switching power; read the table as a decomposition to +-10 %, not a measurement of a product.
#   base     the production kernels
#   nomfma   every v_mfma_f32_32x32x16_f16 of lstm32.hip.h removed (operands and accumulators stay allocated)
#   notrans  v_exp_f32 / v_rcp_f32 of the gate arithmetic replaced by moves (40 of a block's 88 gate instructions)
#   nogate   the whole gate schedule removed (L32_GAP empty: no transcendentals, no VALU, no h stores)
# usage (build container): lstm_energy_variants.sh build       (GPU box): lstm_energy_variants.sh   -> gpurun_out/r04_lstm_energy_variants.txt
cd "$(dirname "$0")/../.."
VARIANTS="base nomfma notrans nogate"
if [ "$1" = build ]; then
  for v in $VARIANTS; do
    d=exp/csrc_e_$v
    rm -rf $d && mkdir -p $d && cp clair_amd/csrc/* $d/
    case $v in
      nomfma)  sed -i 's|v_mfma_f32_32x32x16_f16 %0, %1, %2, %[03]|; mfma removed|' $d/lstm32.hip.h; grep -c "mfma removed" $d/lstm32.hip.h ;;
      notrans) sed -i 's|^#define L32_EXP2(x) __builtin_amdgcn_exp2f(x)|#define L32_EXP2(x) (x)|; s|^#define L32_RCP(x) fast_rcp(x)|#define L32_RCP(x) (x)|' $d/lstm32.hip.h; grep -c "define L32_EXP2(x) (x)\|define L32_RCP(x) (x)" $d/lstm32.hip.h ;;
      nogate)  python3 - $d/lstm32.hip.h <<'PY'
import sys
p = sys.argv[1]
s = open(p).read()
old = "    // What goes into the gap after MFMA number M"
assert s.count(old) == 1
s = s.replace(old, "#undef L32_GAP\n#define L32_GAP(G, PB) {}\n" + old)
open(p, "w").write(s)
print(1)
PY
      ;;
    esac
    sed -e 's|^int enqueue_forward(|static bool ablated(int id) { static const unsigned m = getenv("CLAIR_ABLATE") ? (unsigned)strtoul(getenv("CLAIR_ABLATE"), nullptr, 0) : 0u; return (m >> id) \& 1u; }\nint enqueue_forward(|' \
        -e 's|^\(        *\)hipLaunchKernelGGL(\(.*s\.stream, a);\)|\1if (!ablated(kt.id)) hipLaunchKernelGGL(\2|' $d/engine.hip > $d/engine_ablate.hip
    grep -c "ablated(kt.id)" $d/engine_ablate.hip
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -std=c++17 -shared -fPIC $d/engine_ablate.hip $d/comm.hip -o exp/libclair_e_$v.so -ldl || exit 1
  done
  exit 0
fi
O=gpurun_out/r04_lstm_energy_variants.txt
mkdir -p gpurun_out
run() {   # variant kernel-name mask
  CLAIR_AMD_LIB=$PWD/exp/libclair_e_$1.so CLAIR_ABLATE=$3 timeout 200 python bench.py --steps 160000 --warmup 8 --no-cpu-baseline --boundary-slots 0 --full-candidates 0 > /tmp/ev.json 2>/tmp/ev.err
  python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open('/tmp/ev.json').read().strip().splitlines()[-1])
except Exception as ex:
    print("%-8s %-6s failed: %s" % (sys.argv[1], sys.argv[2], ex)); sys.exit(0)
g = (d.get("gpu_state") or {}).get("value") or {}
pw, ms = g.get("power_w") or 0, d["ms_per_step"]
print("%-8s %-6s alone: %8.4f ms/step  %7.1f W  %7.2f mJ/step  sclk %s MHz" % (sys.argv[1], sys.argv[2], ms, pw, pw * ms, g.get("sclk_mhz")))
PY
}
{
echo "# recurrent kernels without one ingredient at a time, each alone on the chip (4 lanes of the same kernel in flight), per 1024-candidate batch; $(date -u +%Y-%m-%dT%H:%MZ)"
for v in $VARIANTS; do run $v lstm1 108; run $v lstm2 102; done
} > $O 2>&1
cat $O
