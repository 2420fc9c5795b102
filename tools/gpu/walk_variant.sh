#!/bin/sh
# l3l4 with TWO channel groups per workgroup (256 workgroups at batch 1024, split-K 16) against the production FOUR (128 workgroups, split-K 8),
# and four lanes against three: what the walk length and the number of passes in flight are worth to the pipeline.
# exp/libclair_walk2.so = the production sources with L4_SPLITS = 16 (sed; the kernels are written for either).
# usage (build container): walk_variant.sh build        (GPU box): walk_variant.sh
cd "$(dirname "$0")/../.."
if [ "$1" = build8 ]; then      # EIGHT groups per workgroup (64 workgroups at batch 1024, split-K 4): exp/libclair_walk8.so, timed with ab_libs.sh (profiles/r04_ab_walk8.txt)
  rm -rf exp/csrc_walk8 && mkdir -p exp/csrc_walk8 && cp clair_amd/csrc/* exp/csrc_walk8/
  sed -i 's|^constexpr int L4_SPLITS = 8; |constexpr int L4_SPLITS = 4; |' exp/csrc_walk8/common.hip.h
  grep -c "L4_SPLITS = 4" exp/csrc_walk8/common.hip.h
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -std=c++17 -shared -fPIC exp/csrc_walk8/engine.hip exp/csrc_walk8/comm.hip exp/csrc_walk8/frontend.hip -o exp/libclair_walk8.so -ldl
  exit $?
fi
if [ "$1" = build ]; then
  rm -rf exp/csrc_walk2 && mkdir -p exp/csrc_walk2 && cp clair_amd/csrc/* exp/csrc_walk2/
  sed -i 's|^constexpr int L4_SPLITS = 8; |constexpr int L4_SPLITS = 16;|' exp/csrc_walk2/common.hip.h
  grep -c "L4_SPLITS = 16" exp/csrc_walk2/common.hip.h
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -std=c++17 -shared -fPIC exp/csrc_walk2/engine.hip exp/csrc_walk2/comm.hip exp/csrc_walk2/frontend.hip -o exp/libclair_walk2.so -ldl
  exit $?
fi
one() {   # label, lib, lanes
  v=$(CLAIR_AMD_LIB=$PWD/$2 CLAIR_AMD_LANES=$3 timeout 200 python bench.py --streams $3 --steps 2000 --warmup 8 --no-cpu-baseline --boundary-slots 0 --full-candidates 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], {k: round(v['ms_mean'], 4) for k, v in d['kernels_in_flight_ms'].items() if (v['ms_mean'] or 0) > 0.001}, 'alone', {k: v for k, v in d['kernels_alone_ms'].items() if v})")
  echo "$1: $v"
}
for i in 1 2 3; do
  one "walk 4 (production), 3 lanes" clair_amd/libclair_amd.so 3
  one "walk 2,              3 lanes" exp/libclair_walk2.so 3
  one "walk 4 (production), 4 lanes" clair_amd/libclair_amd.so 4
  one "walk 2,              4 lanes" exp/libclair_walk2.so 4
done
