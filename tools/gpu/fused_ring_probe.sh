#!/bin/sh
# What a ring hand-off could buy: the fused layer-2 launch with zx addressed through ZX_RING slots per (direction, tile) instead of
# 33 (no back-pressure: with 3 producer groups the projection is the slower side and stays just ahead; results may be garbage,
# timing and traffic only).  Plain stores (a write HIT updates the L2 line; write misses do not allocate) and plain / nt loads.
set -e
cd "$(dirname "$0")/../.."
for cfg in ${CFGS:-"4 plain" "4 nt" "8 plain" "2 plain"}; do
  set -- $cfg
  mkdir -p exp/fr
  cp clair_amd/csrc/*.h clair_amd/csrc/*.hip exp/fr/
  sed -i -e 's|float \*dst = p.C + ((((size_t)(d \* p.ntiles + tile) \* T_POS + t) \* 16 + wb) \* 1024)|float *dst = p.C + ((((size_t)(d * p.ntiles + tile) * ZX_RING + ((d ? T_POS - 1 - t : t) % ZX_RING)) * 16 + wb) * 1024)|' \
         -e 's|__builtin_nontemporal_store(\(.*\), (f32x4 \*)dst);|*(f32x4 *)dst = \1;|' exp/fr/gemm_split.hip.h
  sed -i -e 's|p.zx + ((((size_t)d \* p.ntiles + tile) \* T_POS \* 4 + w) \* 4) \* 1024 + lane \* 4|p.zx + ((((size_t)d * p.ntiles + tile) * ZX_RING * 4 + w) * 4) * 1024 + lane * 4|' \
         -e 's|const float \*src = zx0 + ((size_t)t \* 16 + b) \* 1024;|const float *src = zx0 + ((size_t)(sc % ZX_RING) * 16 + b) * 1024;|' exp/fr/lstm32.hip.h
  if [ $2 = plain ]; then sed -i -e 's|__builtin_nontemporal_load((const f32x4 \*)(src + a \* 256))|*(const f32x4 *)(src + a * 256)|' exp/fr/lstm32.hip.h; fi
  test $(grep -c ZX_RING exp/fr/gemm_split.hip.h) = 1 && test $(grep -c ZX_RING exp/fr/lstm32.hip.h) = 2
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -std=c++17 -DZX_RING=$1 -shared -fPIC exp/fr/engine.hip exp/fr/comm.hip -o exp/libclair_fr.so -ldl 2>/dev/null
  echo "ring $1 slots, $2 loads, ${G:-3} producer groups"
  CLAIR_AMD_LIB=$PWD/exp/libclair_fr.so CLAIR_AMD_PROJ2_GROUPS=${G:-3} tools/gpu/fused_pmc.sh 2>&1 | grep fused_kernel
  CLAIR_AMD_LIB=$PWD/exp/libclair_fr.so GROUPS_LIST="${G:-3}" tools/gpu/fused_sweep.sh 2>&1 | grep "fused 1"
done
