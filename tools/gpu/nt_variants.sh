#!/bin/sh
# Timing-only builds (build container -> build_ab/) of round 6's cache-hint probe: the inter-kernel intermediates that are written once by one kernel
# and read once by the next -- a1 (LSTM1 -> projection, fp16 planes) and a2 (LSTM2 -> l3l4, fp32) -- stored NON-TEMPORALLY like zx already is
# (zx with plain stores / loads costs 3-4 %: profiles/r06_nozx_probe.txt, prod_nt0 vs prod_nt1).  Results are unchanged (a hint, not arithmetic).
#   libclair_amd_nt_a1.so  libclair_amd_nt_a2.so  libclair_amd_nt_a1a2.so
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
mkdir -p $R/build_ab
for v in a1 a2 a1a2; do
  T=$(mktemp -d); mkdir -p $T/clair_amd $T/include; cp -r $R/clair_amd/csrc $T/clair_amd/csrc; cp $R/include/*.h $T/include/
  (cd $T/clair_amd && python3 - "$v" <<'PY'
import sys
v = sys.argv[1]
def patch(path, pairs):
    s = open(path).read()
    for old, new in pairs:
        assert old in s, (path, old)
        s = s.replace(old, new)
    open(path, "w").write(s)
l, q = [], []
if "a1" in v:
    l.append(("            *(f16x8 *)(p.aout2 + (g >> 9) * plane + row0 + (size_t)((g >> 4) & 31) * (2 * HID) + (g & 15) * 8) = cp[j];",
              "            __builtin_nontemporal_store(cp[j], (f16x8 *)(p.aout2 + (g >> 9) * plane + row0 + (size_t)((g >> 4) & 31) * (2 * HID) + (g & 15) * 8));"))
if "a2" in v:
    l.append(("            *(f32x4 *)(p.aout + ((((size_t)(d * 16 + (g >> 6)) * T_POS + t) * p.n_pad + (size_t)tile * L32_TILE) * 8) + (g & 63) * 4) = co[j];",
              "            __builtin_nontemporal_store(co[j], (f32x4 *)(p.aout + ((((size_t)(d * 16 + (g >> 6)) * T_POS + t) * p.n_pad + (size_t)tile * L32_TILE) * 8) + (g & 63) * 4));"))
    q.append(("        *(f32x4 *)(p.aout + ((((size_t)(d * 16 + (g >> 6)) * T_POS + t) * p.n_pad + (size_t)tile_of[tl] * L32_TILE) * 8) + (g & 63) * 4) = co[j];",
              "        __builtin_nontemporal_store(co[j], (f32x4 *)(p.aout + ((((size_t)(d * 16 + (g >> 6)) * T_POS + t) * p.n_pad + (size_t)tile_of[tl] * L32_TILE) * 8) + (g & 63) * 4));"))
patch("csrc/lstm32.hip.h", l)
if q:
    patch("csrc/lstm32_pair.hip.h", q)
PY
  )
  (cd $T/clair_amd && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -std=c++17 -shared -fPIC csrc/engine.hip csrc/comm.hip csrc/frontend.hip -o $R/build_ab/libclair_amd_nt_$v.so -ldl)
  rm -rf $T; echo build_ab/libclair_amd_nt_$v.so
done
