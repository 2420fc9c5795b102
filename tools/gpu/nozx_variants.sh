#!/bin/sh
# Timing-only builds (build container -> build_ab/, travels with gpurun) of the round-6 `nozx` probe: what the zx round trip through
# the fabric costs the CURRENT kernels.  Projection GEMM and LSTM2 address zx blocks (64 KB each) modulo ZX_FOLD per direction, i.e.
# through a ring of ZX_FOLD x 64 KB that the L2 absorbs when small; NT=0 also replaces the non-temporal stores / loads by plain ones
# (a write-back L2 then merges the repeated writes of a small ring: the write traffic vanishes too).  Results are garbage; timing only.
#   libclair_amd_zxfold<F>_nt<0|1>.so
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
mkdir -p $R/build_ab
[ -n "$VARIANTS" ] || VARIANTS="100000:1 100000:0 16:0 1:0 16:1"
for v in $VARIANTS; do
  F=${v%%:*}; NT=${v##*:}
  T=$(mktemp -d); mkdir -p $T/clair_amd $T/include; cp -r $R/clair_amd/csrc $T/clair_amd/csrc; cp $R/include/*.h $T/include/
  (cd $T/clair_amd && python3 - "$F" "$NT" <<'PY'
import sys
F, NT = sys.argv[1], sys.argv[2]
def patch(path, pairs):
    s = open(path).read()
    for old, new in pairs:
        assert old in s, (path, old)
        s = s.replace(old, new)
    open(path, "w").write(s)
g = [("        return ((size_t)tile * T_POS + t) * 16 * 1024;", "        return ((size_t)((tile * T_POS + t) %% %s)) * 16 * 1024;" % F)]
l = [("p.zx + ((((size_t)d * p.ntiles + tile) * T_POS * 4 + w) * 4) * 1024 + lane * 4", "p.zx + (((size_t)d * p.ntiles * T_POS * 4 + w) * 4) * 1024 + lane * 4"),
     ("            const float *src = zx0 + ((size_t)t * 16 + b) * 1024;", "            const float *src = zx0 + ((size_t)((tile * T_POS + t) %% %s) * 16 + b) * 1024;" % F)]
q = [("zx0[tl] = p.zx + ((((size_t)d * p.ntiles + tile_of[tl]) * T_POS * 4 + w) * 4) * 1024 + lane * 4;", "zx0[tl] = p.zx + (((size_t)d * p.ntiles * T_POS * 4 + w) * 4) * 1024 + lane * 4;"),
     ("        const float *src = zx0[tl] + ((size_t)t * 16 + b) * 1024;", "        const float *src = zx0[tl] + ((size_t)((tile_of[tl] * T_POS + t) %% %s) * 16 + b) * 1024;" % F)]
if NT == "0":
    g.append(("__builtin_nontemporal_store((f32x4){acc[mi][ni][4 * a], acc[mi][ni][4 * a + 1], acc[mi][ni][4 * a + 2], acc[mi][ni][4 * a + 3]}, dst);",
              "*dst = (f32x4){acc[mi][ni][4 * a], acc[mi][ni][4 * a + 1], acc[mi][ni][4 * a + 2], acc[mi][ni][4 * a + 3]};"))
    l.append(("__builtin_nontemporal_load((const f32x4 *)(src + a * 256));   // read once: keep it out of the caches' way", "*(const f32x4 *)(src + a * 256);"))
    q.append(("__builtin_nontemporal_load((const f32x4 *)(src + a * 256));", "*(const f32x4 *)(src + a * 256);"))
patch("csrc/gemm_split.hip.h", g); patch("csrc/lstm32.hip.h", l); patch("csrc/lstm32_pair.hip.h", q)
PY
  )
  (cd $T/clair_amd && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -std=c++17 -shared -fPIC csrc/engine.hip csrc/comm.hip csrc/frontend.hip -o $R/build_ab/libclair_amd_zxfold${F}_nt${NT}.so -ldl)
  rm -rf $T; echo build_ab/libclair_amd_zxfold${F}_nt${NT}.so
done
