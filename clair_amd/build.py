"""Build the HIP extension in-tree: hipcc --offload-arch=gfx950 -> clair_amd/libclair_amd.so.

hipcc cross-compiles without a GPU, so this runs in the build container; the .so is
git-ignored but travels to the GPU box with the source snapshot.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRCS = [os.path.join(HERE, "csrc", f) for f in ("engine.hip", "comm.hip", "frontend.hip")]
OUT = os.path.join(HERE, "libclair_amd.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def csrc_digest(root=None):
    """sha256 over the kernel sources and the launch code (csrc/*.hip.h + engine.hip; names and bytes, sorted): what a
    measurement of "this build" is stamped with (profiles/pmc_traffic.json, bench.py)."""
    import hashlib
    root = root or os.path.join(HERE, "csrc")
    h = hashlib.sha256()
    for f in sorted(os.listdir(root)):
        if f.endswith(".hip.h") or f == "engine.hip":
            h.update(f.encode() + b"\0" + open(os.path.join(root, f), "rb").read() + b"\0")
    return h.hexdigest()[:16]


def needs_build():
    if not os.path.isfile(OUT):
        return True
    newest = max(os.path.getmtime(os.path.join(dp, f))
                 for dp, _, fs in os.walk(os.path.join(HERE, "csrc")) for f in fs)
    newest = max([newest] + [os.path.getmtime(os.path.join(HERE, "..", "include", h)) for h in ("clair_amd.h", "clair_call.h", "clair_reads.h")])
    return newest > os.path.getmtime(OUT)


HOST_SRCS = [os.path.join(HERE, "hostsrc", f) for f in ("host_io.cpp", "host_decode.cpp", "host_pileup.cpp", "host_sampack.cpp")]
HOST_OUT = os.path.join(HERE, "libclair_host.so")
CXX = os.environ.get("CXX", "g++")


def build_host(force=False):
    """Host-side helpers (include/clair_host.h): plain C++, no HIP."""
    # every header a host source includes: clair_call.h is the 32-byte record host_decode.cpp shares bit for bit with the device decode
    hdrs = [os.path.join(HERE, "..", "include", h) for h in ("clair_host.h", "clair_reads.h", "clair_call.h", "clair_amd.h")]
    if (force or not os.path.isfile(HOST_OUT)
            or max([os.path.getmtime(f) for f in HOST_SRCS + hdrs]) > os.path.getmtime(HOST_OUT)):
        # -ffp-contract=off: the decode restates float32 product chains bit for bit (no fused multiply-add)
        subprocess.check_call([CXX, "-O3", "-std=c++17", "-ffp-contract=off", "-pthread", "-shared", "-fPIC", "-Wall"] + HOST_SRCS + ["-o", HOST_OUT])
    return HOST_OUT


def build(force=False, verbose=False):
    build_host(force)
    if not force and not needs_build():
        return OUT
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-fno-slp-vectorize", "-std=c++17", "-shared", "-fPIC"] + SRCS + ["-o", OUT, "-ldl"]
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
    subprocess.check_call(cmd)
    return OUT


WAITALL_OUT = os.path.join(HERE, "libclair_amd_waitall.so")


def build_waitall():
    """The CHECK build of the same sources (-DCLAIR_WAIT_ALL, csrc/common.hip.h): every hand-counted wait becomes a full one and the
    memory, LDS and matrix pipes are drained after every hand-placed MFMA.  Only tools/gpu/waitall_compare.py loads it."""
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-fno-slp-vectorize", "-std=c++17", "-shared", "-fPIC", "-DCLAIR_WAIT_ALL"]
                          + SRCS + ["-o", WAITALL_OUT, "-ldl"])
    return WAITALL_OUT


PROBE_OUT = os.path.join(HERE, "libclair_amd_probe.so")


def build_probe():
    """The PROBE build (-DCLAIR_L34_STAMPS): l3l4_kernel stamps its phase boundaries with s_memtime (tools/gpu/l34_stamps.py).
    Timing instrument only; nothing ships or tests against it."""
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-fno-slp-vectorize", "-std=c++17", "-shared", "-fPIC", "-DCLAIR_L34_STAMPS"]
                          + SRCS + ["-o", PROBE_OUT, "-ldl"])
    return PROBE_OUT


if __name__ == "__main__":
    if "--probe" in sys.argv:
        print(build_probe())
    elif "--waitall" in sys.argv:
        print(build_waitall())
    else:
        print(build(force="--force" in sys.argv, verbose="-v" in sys.argv))
