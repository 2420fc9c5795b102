"""exp/libclair_probe_lstm.so: the production sources with s_memtime stamps inside the recurrent kernels' step loop (after the h-fragment reads are
issued, after each of the four gate blocks, after the exposed tail, after the barrier), first eight steps of every wave (tools/gpu/lstm_stamps.py reads
them).  Generated into the git-ignored exp/.  Build container only."""
import os
import shutil
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src, dst = os.path.join(ROOT, "clair_amd", "csrc"), os.path.join(ROOT, "exp", "probe_lstm")
shutil.rmtree(dst, ignore_errors=True)
shutil.copytree(src, dst)


def patch(path, pairs):
    s = open(path).read()
    for old, new in pairs:
        assert s.count(old) == 1, old
        s = s.replace(old, new)
    open(path, "w").write(s)


patch(os.path.join(dst, "lstm32.hip.h"), [
    ("constexpr int L32_TILE = 32;      // candidates per workgroup\n",
     "constexpr int L32_TILE = 32;      // candidates per workgroup\n"
     "__device__ unsigned long long l32_stamp_buf[2][512 * 4 * 8 * 8];   // PROBE: [layer][workgroup][wave][step 0..7][point]\n"
     "#define L32_STAMP(k) if (!FUSED && s >= 8 && s < 16 && lane == 0) l32_stamp_buf[FIRST ? 0 : 1][((((size_t)(2 * tile + d) * 4 + w) * 8 + (s - 8)) * 8) + (k)] = __builtin_amdgcn_s_memtime();\n"),
    ("        const int s_prev = s > 0 ? s - 1 : 0;\n", "        const int s_prev = s > 0 ? s - 1 : 0;\n        L32_STAMP(0)\n"),
    ("        L32_BLOCK(0)\n        L32_BLOCK(1)\n        L32_BLOCK(2)\n        L32_BLOCK(3)\n",
     "        L32_BLOCK(0)\n        L32_STAMP(1)\n        L32_BLOCK(1)\n        L32_STAMP(2)\n        L32_BLOCK(2)\n        L32_STAMP(3)\n        L32_BLOCK(3)\n        L32_STAMP(4)\n"),
    ("            for (int g = 1; g <= 23; ++g) L32_GAP(g, 3)\n        }\n        __syncthreads();\n    }\n#undef L32_BLOCK",
     "            for (int g = 1; g <= 23; ++g) L32_GAP(g, 3)\n        }\n        L32_STAMP(5)\n        __syncthreads();\n        L32_STAMP(6)\n    }\n#undef L32_BLOCK"),
])
patch(os.path.join(dst, "engine.hip"), [
    ("int clair_abi_version(void) { return CLAIR_ABI_VERSION; }",
     "int clair_abi_version(void) { return CLAIR_ABI_VERSION; }\n"
     "int clair_probe_lstm_stamps(unsigned long long *host, long long count) {\n"
     "    return hipMemcpyFromSymbol(host, HIP_SYMBOL(clair::l32_stamp_buf), (size_t)count * sizeof(unsigned long long), 0, hipMemcpyDeviceToHost) == hipSuccess ? 0 : 1;\n}"),
])
out = os.path.join(ROOT, "exp", "libclair_probe_lstm.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fno-slp-vectorize", "-std=c++17", "-shared", "-fPIC"]
                      + [os.path.join(dst, f) for f in ("engine.hip", "comm.hip", "frontend.hip")] + ["-o", out, "-ldl"])
print(out)
