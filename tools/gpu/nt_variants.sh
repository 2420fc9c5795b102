#!/bin/sh
# Timing-only builds (build container -> build_ab/) of round 6's cache-hint probe: the inter-kernel intermediates that are written once by one kernel
# and read once by the next, handled NON-TEMPORALLY like zx already is (zx with plain stores / loads costs 3-4 %: profiles/r06_nozx_probe.txt).
# Results are unchanged bits (a hint, not arithmetic).  Patches, combinable with '+':
#   a1w   LSTM1's stores of a1 (fp16 planes, read four times per XCD by the projection)        a2w   LSTM2's stores of a2 (fp32, read once by l3l4)
#   a2r   l3l4's LDS-DMA of a2 with the nt bit                                                 xr    LSTM1's loads of the network input (read once)
# usage: tools/gpu/nt_variants.sh "a2r a2w+a2r a2w+a2r+xr"   ->  build_ab/libclair_amd_nt_<name>.so
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
mkdir -p $R/build_ab
[ -n "$1" ] || set -- "a1w a2w a1w+a2w a2r a2w+a2r a2w+a2r+xr"
for v in $1; do
  T=$(mktemp -d); mkdir -p $T/clair_amd $T/include; cp -r $R/clair_amd/csrc $T/clair_amd/csrc; cp $R/include/*.h $T/include/
  (cd $T/clair_amd && python3 - "$v" <<'PY'
import sys
want = set(sys.argv[1].split("+"))
def patch(path, pairs):
    s = open(path).read()
    for old, new in pairs:
        assert old in s, (path, old)
        s = s.replace(old, new)
    open(path, "w").write(s)
l, q, c, d = [], [], [], []
if "a1w" in want:
    l.append(("            *(f16x8 *)(p.aout2 + (g >> 9) * plane + row0 + (size_t)((g >> 4) & 31) * (2 * HID) + (g & 15) * 8) = cp[j];",
              "            __builtin_nontemporal_store(cp[j], (f16x8 *)(p.aout2 + (g >> 9) * plane + row0 + (size_t)((g >> 4) & 31) * (2 * HID) + (g & 15) * 8));"))
if "a2w" in want:
    l.append(("            *(f32x4 *)(p.aout + ((((size_t)(d * 16 + (g >> 6)) * T_POS + t) * p.n_pad + (size_t)tile * L32_TILE) * 8) + (g & 63) * 4) = co[j];",
              "            __builtin_nontemporal_store(co[j], (f32x4 *)(p.aout + ((((size_t)(d * 16 + (g >> 6)) * T_POS + t) * p.n_pad + (size_t)tile * L32_TILE) * 8) + (g & 63) * 4));"))
    q.append(("        *(f32x4 *)(p.aout + ((((size_t)(d * 16 + (g >> 6)) * T_POS + t) * p.n_pad + (size_t)tile_of[tl] * L32_TILE) * 8) + (g & 63) * 4) = co[j];",
              "        __builtin_nontemporal_store(co[j], (f32x4 *)(p.aout + ((((size_t)(d * 16 + (g >> 6)) * T_POS + t) * p.n_pad + (size_t)tile_of[tl] * L32_TILE) * 8) + (g & 63) * 4));"))
if "xr" in want:
    l.append(("        return *(const f32x4 *)(xg + (d ? T_POS - 1 - sc : sc) * F_IN);", "        return __builtin_nontemporal_load((const f32x4 *)(xg + (d ? T_POS - 1 - sc : sc) * F_IN));"))
if "a2r" in want:
    c.append(("__device__ __forceinline__ void glds16_s(unsigned lane_off, const void *sbase, unsigned lds_base) {",
              "__device__ __forceinline__ void glds16_s_nt(unsigned lane_off, const void *sbase, unsigned lds_base) {\n    unsigned keep;\n"
              "    asm volatile(\"s_mov_b32 %0, m0\\n\\ts_mov_b32 m0, %3\\n\\ts_nop 0\\n\\tglobal_load_lds_dwordx4 %1, %2 nt\\n\\ts_mov_b32 m0, %0\"\n"
              "                 : \"=&s\"(keep) : \"v\"(lane_off), \"s\"(sbase), \"s\"(lds_base) : \"memory\");\n}\n"
              "__device__ __forceinline__ void glds16_s(unsigned lane_off, const void *sbase, unsigned lds_base) {"))
    d.append(("            glds16_s(dma_lane_off, sbase, lds_a2 + (q4 + 4 * i) * 1024);", "            glds16_s_nt(dma_lane_off, sbase, lds_a2 + (q4 + 4 * i) * 1024);"))
for path, pairs in (("csrc/lstm32.hip.h", l), ("csrc/lstm32_pair.hip.h", q), ("csrc/common.hip.h", c), ("csrc/dense.hip.h", d)):
    if pairs:
        patch(path, pairs)
PY
  )
  n=$(echo $v | tr -d '+')
  (cd $T/clair_amd && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -std=c++17 -shared -fPIC csrc/engine.hip csrc/comm.hip csrc/frontend.hip -o $R/build_ab/libclair_amd_nt_$n.so -ldl)
  rm -rf $T; echo build_ab/libclair_amd_nt_$n.so
done
