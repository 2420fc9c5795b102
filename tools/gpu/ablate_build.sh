#!/bin/sh
# Build exp/libclair_ablate.so: the production engine with one extra switch -- CLAIR_ABLATE=<bit mask over enum clair_kernel_id>
# skips those kernel launches (results are garbage; timing only).  Generated from the production source by sed so that the
# production files carry no ablation code.  Use:  CLAIR_AMD_LIB=$PWD/exp/libclair_ablate.so CLAIR_ABLATE=64 python bench.py ...
set -e
cd "$(dirname "$0")/../.."
mkdir -p exp
sed -e 's|^int enqueue_forward(|static bool ablated(int id) { static const unsigned m = getenv("CLAIR_ABLATE") ? (unsigned)strtoul(getenv("CLAIR_ABLATE"), nullptr, 0) : 0u; return (m >> id) \& 1u; }\nint enqueue_forward(|' \
    -e 's|^\(        *\)hipLaunchKernelGGL(\(.*s\.stream, a);\)|\1if (!ablated(kt.id)) hipLaunchKernelGGL(\2|' \
    -e 's|#include "\([a-z0-9_]*\.hip\.h\)"|#include "../clair_amd/csrc/\1"|' -e 's|#include "../../include/clair_amd.h"|#include "../include/clair_amd.h"|' \
    clair_amd/csrc/engine.hip > exp/engine_ablate.hip
grep -c "ablated(kt.id)" exp/engine_ablate.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -std=c++17 -shared -fPIC exp/engine_ablate.hip clair_amd/csrc/comm.hip -o exp/libclair_ablate.so -ldl
