#!/bin/sh
# Cache-policy variants of the fused layer-2 hand-off (sed copies in exp/): A = plain zx stores + non-temporal seed loads,
# B = plain stores + plain loads.  Prints the launch's L2<->fabric traffic and the pipeline rate.
set -e
cd "$(dirname "$0")/../.."
for v in A B; do
  mkdir -p exp/fv$v
  cp clair_amd/csrc/*.h clair_amd/csrc/*.hip exp/fv$v/
  sed -i -e 's|__builtin_nontemporal_store(\(.*\), (f32x4 \*)dst);|*(f32x4 *)dst = \1;|' exp/fv$v/gemm_split.hip.h
  if [ $v = B ]; then sed -i -e 's|__builtin_nontemporal_load((const f32x4 \*)(src + a \* 256))|*(const f32x4 *)(src + a * 256)|' exp/fv$v/lstm32.hip.h; fi
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -std=c++17 -shared -fPIC exp/fv$v/engine.hip exp/fv$v/comm.hip -o exp/libclair_fv$v.so -ldl 2>/dev/null
  echo "variant $v"
  CLAIR_AMD_LIB=$PWD/exp/libclair_fv$v.so CLAIR_AMD_PROJ2_GROUPS=${G:-4} tools/gpu/fused_pmc.sh 2>&1 | grep fused_kernel
  CLAIR_AMD_LIB=$PWD/exp/libclair_fv$v.so GROUPS_LIST="${G:-4}" tools/gpu/fused_sweep.sh 2>&1 | grep "fused 1" | head -1
done
