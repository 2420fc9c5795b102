"""ctypes loader for oracle/libclair_oracle.so (TEST INFRASTRUCTURE ONLY, see clair_oracle.c)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
TENSOR_ORDER = (
    "lstm1_fw_kernel", "lstm1_fw_bias", "lstm1_bw_kernel", "lstm1_bw_bias",
    "lstm2_fw_kernel", "lstm2_fw_bias", "lstm2_bw_kernel", "lstm2_bw_bias",
    "l3_kernel", "l3_bias", "l4_kernel", "l4_bias", "l5_kernel", "l5_bias",
    "head_gt21_kernel", "head_gt21_bias", "head_genotype_kernel", "head_genotype_bias",
    "head_len1_kernel", "head_len1_bias", "head_len2_kernel", "head_len2_bias",
)


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "libclair_oracle.so"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libclair_oracle.so")
        if not os.path.isfile(path):
            build()
        _LIB = ctypes.CDLL(path)
        _LIB.clair_oracle_forward_ex.restype = ctypes.c_int
        _LIB.clair_oracle_forward_ex_f64.restype = ctypes.c_int
        _LIB.clair_oracle_max_threads.restype = ctypes.c_int
    return _LIB


_PORT = None


def port_lib():
    global _PORT
    if _PORT is None:
        path = os.path.join(_HERE, "libclair_cpu_port.so")
        if not os.path.isfile(path):
            subprocess.check_call(["make", "-s", "-C", _HERE, "libclair_cpu_port.so"])
        _PORT = ctypes.CDLL(path)
        _PORT.clair_cpu_port_forward.restype = ctypes.c_int
    return _PORT


def port_forward(w, x, threads=0):
    """The blocked CPU port (clair_cpu_port.c): same interface as forward(), float32 only, no intermediates."""
    L = port_lib()
    x = np.ascontiguousarray(x, dtype=np.float32)
    n = x.shape[0]
    keep = [np.ascontiguousarray(w[k], dtype=np.float32) for k in TENSOR_ORDER]
    ptrs = (ctypes.c_void_p * len(keep))(*[a.ctypes.data for a in keep])
    outs = [np.empty((n, m), dtype=np.float32) for m in (21, 3, 33, 33)]
    rc = L.clair_cpu_port_forward(ptrs, ctypes.c_void_p(x.ctypes.data), ctypes.c_int(n),
                                  *[ctypes.c_void_p(o.ctypes.data) for o in outs], ctypes.c_int(threads))
    if rc != 0:
        raise RuntimeError("clair_cpu_port_forward failed (rc=%d)" % rc)
    return outs


def usable_threads():
    """Hardware threads this process may actually keep busy: the host's count capped by the cgroup CPU quota (cpu.max:
    the GPU boxes expose 256 hardware threads under a quota of 16 CPUs -- 256 OpenMP threads then spend their time throttled)."""
    n = max_threads()
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = max(1, min(n, q // p))
        except (OSError, ValueError):
            pass
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    return n


def max_threads():
    return int(lib().clair_oracle_max_threads())


def forward(w, x, threads=0, keep_intermediates=False, dtype=np.float32):
    """x [n,33,8,4] float32 -> [gt21, genotype, len1, len2] (+ dict of intermediates).

    dtype=np.float64 evaluates the same graph in double precision from the same float32 weights and inputs
    (outputs and intermediates come back as float64): the yardstick for float32 rounding itself."""
    L = lib()
    dtype = np.dtype(dtype)
    fn = L.clair_oracle_forward_ex if dtype == np.float32 else L.clair_oracle_forward_ex_f64
    x = np.ascontiguousarray(x, dtype=np.float32)
    n = x.shape[0]
    keep = [np.ascontiguousarray(w[k], dtype=np.float32) for k in TENSOR_ORDER]
    ptrs = (ctypes.c_void_p * len(keep))(*[a.ctypes.data for a in keep])
    outs = [np.empty((n, m), dtype=dtype) for m in (21, 3, 33, 33)]
    inter = {}
    if keep_intermediates:
        inter = dict(a1=np.empty((n, 33, 256), dtype), a2=np.empty((n, 33, 256), dtype),
                     l3=np.empty((n, 7680), dtype), l4=np.empty((n, 192), dtype))
    p = lambda a: ctypes.c_void_p(a.ctypes.data)
    opt = lambda k: p(inter[k]) if keep_intermediates else ctypes.c_void_p(0)
    rc = fn(ptrs, p(x), ctypes.c_int(n), *[p(o) for o in outs],
                                   opt("a1"), opt("a2"), opt("l3"), opt("l4"), ctypes.c_int(threads))
    if rc != 0:
        raise RuntimeError("clair_oracle_forward failed (rc=%d)" % rc)
    return (outs, inter) if keep_intermediates else outs
