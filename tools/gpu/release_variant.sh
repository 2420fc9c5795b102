#!/bin/sh
# What a real agent-scope RELEASE on the fused launch's ticket stores costs: exp/libclair_release.so = the production
# sources with the two ticket stores of gemm_split.hip.h's publish() as __ATOMIC_RELEASE (buffer_wbl2 + wait, issued by the publishing wave's
# lane 0 only) instead of relaxed.  Timing instrument: the production hand-off stays relaxed inside ONE XCD's L2, guarded by the placement check.
# usage (build container): release_variant.sh build        (GPU box): release_variant.sh   -> one-slot fused pipeline, both builds, alternating
cd "$(dirname "$0")/../.."
if [ "$1" = build ]; then
  rm -rf exp/csrc_release && mkdir -p exp/csrc_release && cp clair_amd/csrc/* exp/csrc_release/
  sed -i 's|__hip_atomic_store(fw, fz.ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)|__hip_atomic_store(fw, fz.ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT)|; s|__hip_atomic_store(fw + T_POS \* 8, fz.ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)|__hip_atomic_store(fw + T_POS * 8, fz.ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT)|' exp/csrc_release/gemm_split.hip.h
  grep -c "__ATOMIC_RELEASE" exp/csrc_release/gemm_split.hip.h
  sed -i 's|#include "../../include/|#include "../../include/|' exp/csrc_release/*.hip
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -std=c++17 -shared -fPIC exp/csrc_release/engine.hip exp/csrc_release/comm.hip exp/csrc_release/frontend.hip -o exp/libclair_release.so -ldl
  exit $?
fi
for i in 1 2 3; do
  for lib in clair_amd/libclair_amd.so exp/libclair_release.so; do
    v=$(CLAIR_AMD_LIB=$PWD/$lib timeout 200 python bench.py --streams 1 --steps 600 --warmup 8 --no-cpu-baseline --boundary-slots 0 --full-candidates 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], {k: round(v['ms_mean'], 4) for k, v in d['kernels_in_flight_ms'].items() if (v['ms_mean'] or 0) > 0.001})")
    echo "one slot, fused layer 2, $lib: $v"
  done
done
