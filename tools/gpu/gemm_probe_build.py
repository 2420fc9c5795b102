"""exp/libclair_probe_gemm.so: the production sources with s_memtime stamps at four points of every phase of gemm_split_kernel's first four tiles
(tools/gpu/gemm_stamps.py reads them).  Generated into the git-ignored exp/: the production sources carry no such switches.  Build container only."""
import os
import shutil
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src, dst = os.path.join(ROOT, "clair_amd", "csrc"), os.path.join(ROOT, "exp", "probe_gemm")
shutil.rmtree(dst, ignore_errors=True)
shutil.copytree(src, dst)


def patch(path, pairs):
    s = open(path).read()
    for old, new in pairs:
        assert s.count(old) == 1, old
        s = s.replace(old, new)
    open(path, "w").write(s)


patch(os.path.join(dst, "gemm_split.hip.h"), [
    ("constexpr int GS_RING = 4;                        // phases resident in LDS\n",
     "constexpr int GS_RING = 4;                        // phases resident in LDS\n"
     "__device__ unsigned long long gs_stamp_buf[512 * 4 * 64];   // PROBE: [workgroup][wave][tile 0..3][phase][4 points]\n"
     "#define GS_STAMP(it, ph, k) if (!FUSED && (it) < 4 && lane == 0) gs_stamp_buf[((((size_t)block * 4 + wave) * 4 + (it)) * 4 + (ph)) * 4 + (k)] = __builtin_amdgcn_s_memtime();\n"),
    ("            const int P = it * 4 + ph;\n", "            const int P = it * 4 + ph;\n            GS_STAMP(it, ph, 0)\n"),
    ("            CLAIR_VMWAIT(4);\n            __syncthreads();\n",
     "            GS_STAMP(it, ph, 1)\n            CLAIR_VMWAIT(4);\n            GS_STAMP(it, ph, 2)\n            __syncthreads();\n            GS_STAMP(it, ph, 3)\n"),
])
patch(os.path.join(dst, "engine.hip"), [
    ("int clair_abi_version(void) { return CLAIR_ABI_VERSION; }",
     "int clair_abi_version(void) { return CLAIR_ABI_VERSION; }\n"
     "int clair_probe_gemm_stamps(unsigned long long *host, long long count) {\n"
     "    return hipMemcpyFromSymbol(host, HIP_SYMBOL(clair::gs_stamp_buf), (size_t)count * sizeof(unsigned long long), 0, hipMemcpyDeviceToHost) == hipSuccess ? 0 : 1;\n}"),
])
out = os.path.join(ROOT, "exp", "libclair_probe_gemm.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fno-slp-vectorize", "-std=c++17", "-shared", "-fPIC"]
                      + [os.path.join(dst, f) for f in ("engine.hip", "comm.hip", "frontend.hip")] + ["-o", out, "-ldl"])
print(out)
