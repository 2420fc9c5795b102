#!/bin/sh
# Builds (in the build container) the timing-only variants of round 5's probes from the tree's sources into build_ab/ (git-ignored *.so, travels with gpurun):
#   libclair_amd_lstm2_plus700.so  44 x s_nop 15 appended to every step of lstm32_kernel<false> (idle cycles, no work)     -> r05_session8.sh
#   libclair_amd_lstm1_noxlo.so    LSTM1's w_hi.x_lo MFMAs replaced by an s_nop of their issue cost (same results for counts <= 2048)
#   libclair_amd_anyorder.so       LSTM1 launched with hipExtAnyOrderLaunch when CLAIR_AMD_ANYORDER=1                         -> r05_session9.sh
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
mkdir -p $R/build_ab
build() {  # name, python patch
  T=$(mktemp -d); (cd $R && git archive HEAD clair_amd/csrc include | tar -x -C $T)
  (cd $T && python3 -c "$2")
  (cd $T && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -std=c++17 -shared -fPIC clair_amd/csrc/engine.hip clair_amd/csrc/comm.hip clair_amd/csrc/frontend.hip -o $R/build_ab/libclair_amd_$1.so -ldl)
  rm -rf $T; echo build_ab/libclair_amd_$1.so
}
build lstm2_plus700 '
p="clair_amd/csrc/lstm32.hip.h"; s=open(p).read()
old="            for (int g = 1; g <= 23; ++g) L32_GAP(g, 3)\n        }"
assert old in s
open(p,"w").write(s.replace(old, "            for (int g = 1; g <= 23; ++g) L32_GAP(g, 3)\n            _Pragma(\"unroll\") for (int z = 0; z < 44; ++z) asm volatile(\"s_nop 15\" ::: \"memory\");\n        }", 1))'
build lstm1_noxlo '
p="clair_amd/csrc/lstm32.hip.h"; s=open(p).read()
old="        mfma32_vv(xacc[xb], wxa[xb & 1][kk][term == 0 ? 1 : 0], term == 1 ? xl[kk] : xh[kk]);"
assert old in s
open(p,"w").write(s.replace(old, "        if (term == 1) asm volatile(\"s_nop 12\" ::: \"memory\"); else mfma32_vv(xacc[xb], wxa[xb & 1][kk][term == 0 ? 1 : 0], term == 1 ? xl[kk] : xh[kk]);", 1))'
build anyorder '
p="clair_amd/csrc/engine.hip"; s=open(p).read()
old="        hipLaunchKernelGGL((lstm32_kernel<true>), dim3(ntiles * 2), dim3(256), 0, s.stream, a);"
assert old in s
s=s.replace(old, "        static const bool anyorder = getenv(\"CLAIR_AMD_ANYORDER\") && getenv(\"CLAIR_AMD_ANYORDER\")[0] == 49;\n        if (anyorder && !(e->timing_mask)) hipExtLaunchKernelGGL((lstm32_kernel<true>), dim3(ntiles * 2), dim3(256), 0, s.stream, nullptr, nullptr, hipExtAnyOrderLaunch, a);\n        else hipLaunchKernelGGL((lstm32_kernel<true>), dim3(ntiles * 2), dim3(256), 0, s.stream, a);", 1)
s=s.replace("#include <hip/hip_runtime.h>", "#include <hip/hip_runtime.h>\n#include <hip/hip_ext.h>", 1)
open(p,"w").write(s)'
