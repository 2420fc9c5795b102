"""ctypes binding of include/clair_host.h (libclair_host.so, built by clair_amd/build.py with g++: no GPU involved)."""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libclair_host.so")
SYMBOLS = ("clair_host_abi_version", "clair_host_last_error", "clair_host_threads", "clair_host_parse_tensors",
           "clair_host_decode_rows")
N_VALUES = 1056
_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise RuntimeError("%s not found: run `python -m clair_amd.build`" % LIB_PATH)
        lib = ctypes.CDLL(LIB_PATH)
        vp, i32, i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
        lib.clair_host_abi_version.restype = i32
        lib.clair_host_last_error.restype = ctypes.c_char_p
        lib.clair_host_parse_tensors.argtypes = [vp, i64, i32, i32, vp, vp, ctypes.POINTER(i32), ctypes.POINTER(i32),
                                                 ctypes.POINTER(i64)]
        lib.clair_host_decode_rows.argtypes = [vp, vp, vp, vp, vp, ctypes.c_char_p, vp, i32, i32, i32, i32, i32, i32, vp, i64,
                                               ctypes.POINTER(i64), ctypes.POINTER(i32)]
        if lib.clair_host_abi_version() != 1:
            raise RuntimeError("libclair_host.so has ABI version %d, expected 1" % lib.clair_host_abi_version())
        _lib = lib
    return _lib


def parse_tensors(chunk, final, max_rows, x_out, row0, offset=0):
    """Parse up to max_rows lines of `chunk` (bytes), starting at byte `offset`, into x_out[row0:] (float32 [*,1056],
    C-contiguous).  -> (rows_taken, infos of the kept rows as [[ctg, pos, seq], ...], bytes_consumed)"""
    lib = load()
    tok = np.empty((max(max_rows, 1), 6), dtype=np.int32)
    taken, kept, used = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int64(0)
    base = ctypes.cast(ctypes.c_char_p(chunk), ctypes.c_void_p).value      # bytes: passed by pointer, no copy
    rc = lib.clair_host_parse_tensors(base + offset, len(chunk) - offset, 1 if final else 0, max_rows,
                                      x_out[row0:].ctypes.data, tok.ctypes.data,
                                      ctypes.byref(taken), ctypes.byref(kept), ctypes.byref(used))
    if rc != 0:
        raise ValueError("malformed tensor record: " + lib.clair_host_last_error().decode())
    infos = []
    if kept.value:
        t = (tok[:kept.value].astype(np.int64) + np.array([offset, 0, offset, 0, offset, 0], dtype=np.int64)).tolist()
        for o in t:
            infos.append([chunk[o[0]:o[0] + o[1]].decode(), chunk[o[2]:o[2] + o[3]].decode(), chunk[o[4]:o[4] + o[5]].decode()])
    return taken.value, infos, used.value


def decode_rows(X, infos, Y, show_reference, haploid_precision, haploid_sensitive, qual_threshold, arith_numpy2):
    """clair_host_decode_rows over one batch -> list of VCF row strings (input order, skipped candidates left out)."""
    lib = load()
    n = len(infos)
    if n == 0:
        return []
    x = np.ascontiguousarray(X, dtype=np.float32).reshape(n, N_VALUES)
    gt21, genotype, len1, len2 = [np.ascontiguousarray(a, dtype=np.float32) for a in Y]
    parts = [s for info in infos for s in (info[0], str(info[1]), info[2])]
    lens = np.fromiter(map(len, parts), dtype=np.int32, count=3 * n)
    tok = np.empty((n, 6), dtype=np.int32)
    tok[:, 1::2] = lens.reshape(n, 3)
    starts = np.cumsum(lens, dtype=np.int64) - lens
    tok[:, 0::2] = starts.reshape(n, 3)
    meta = "".join(parts).encode("ascii")
    cap = 4096 + 256 * n + 2 * len(meta)
    out = ctypes.create_string_buffer(cap)
    out_len, n_rows = ctypes.c_int64(0), ctypes.c_int(0)
    rc = lib.clair_host_decode_rows(x.ctypes.data, gt21.ctypes.data, genotype.ctypes.data, len1.ctypes.data, len2.ctypes.data,
                                    meta, tok.ctypes.data, n, int(bool(show_reference)), int(bool(haploid_precision)),
                                    int(bool(haploid_sensitive)), -1 if qual_threshold is None else int(qual_threshold),
                                    int(bool(arith_numpy2)), out, cap, ctypes.byref(out_len), ctypes.byref(n_rows))
    if rc != 0:
        raise ValueError("native decode: " + lib.clair_host_last_error().decode())
    if out_len.value == 0:
        return []
    return out.raw[:out_len.value - 1].decode("ascii").split("\n")
