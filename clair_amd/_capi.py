"""ctypes binding of the C ABI in include/clair_amd.h (libclair_amd.so, built by build.py).

There is deliberately no fallback: if the shared library is missing or no HIP device is
present, construction of an engine raises.  The HIP path is the product.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CLAIR_AMD_LIB") or os.path.join(_HERE, "libclair_amd.so")   # override: debugging builds only

# every symbol include/clair_amd.h declares (tests/test_abi.py checks header <-> this list <-> .so)
SYMBOLS = (
    "clair_abi_version", "clair_device_count", "clair_device_pci_bus_id", "clair_last_error",
    "clair_engine_create", "clair_engine_destroy",
    "clair_set_tensor", "clair_finalize_weights",
    "clair_predict", "clair_submit", "clair_wait", "clair_slot_input", "clair_submit_counts", "clair_submit_ex", "clair_decode", "clair_pinned_alloc", "clair_pinned_free",
    "clair_dataset_alloc", "clair_dataset_free", "clair_dataset_upload", "clair_dataset_download",
    "clair_run_resident", "clair_sync",
    "clair_timing_enable", "clair_kernel_times", "clair_timing_reset", "clair_kernel_workgroups",
    "clair_debug_read", "clair_engine_counter",
    "clair_comm_preflight", "clair_comm_unique_id", "clair_comm_create", "clair_comm_create_timed", "clair_comm_destroy", "clair_comm_abort", "clair_comm_last_error", "clair_comm_barrier",
    "clair_comm_allreduce_f64", "clair_comm_broadcast", "clair_comm_allgather", "clair_comm_allgather_device",
    "clair_frontend_create", "clair_frontend_destroy", "clair_frontend_last_error", "clair_frontend_add_reads",
    "clair_frontend_find_candidates", "clair_frontend_set_candidates", "clair_frontend_get_candidates", "clair_frontend_build_windows",
    "clair_frontend_build_windows_ex", "clair_frontend_window_info", "clair_frontend_window_counts", "clair_frontend_counts_device", "clair_frontend_budget_inputs",
    "clair_frontend_stats", "clair_frontend_text_options", "clair_frontend_add_text", "clair_frontend_text_stats", "clair_frontend_slab_reads",
)
KERNEL_NAMES = ("proj1", "lstm1", "proj2", "lstm2", "l3", "l4", "tail", "decode")

_lib = None
_libs = {}


class EngineError(RuntimeError):
    pass


def older_ok_early(path):
    """An OLDER build of the same sources, named explicitly (A/B timing, tools/gpu/ab_libs.sh), may predate the newest entry points."""
    return path is not None or bool(os.environ.get("CLAIR_AMD_LIB"))


def load(path=None):
    """Load libclair_amd.so (CDLL: calls release the GIL, as TF's session.run does for the
    reference's predict thread, clair/call_var.py:1343).  `path` names another BUILD of the same sources (the wait-all check
    build of tools/gpu/waitall_compare.py); it never names a different implementation."""
    global _lib
    if path is None and _lib is not None:
        return _lib
    if path is not None and path in _libs:
        return _libs[path]
    lib_path = path or LIB_PATH
    if not os.path.isfile(lib_path):
        raise EngineError(
            "%s not found: build the HIP extension first (python -m clair_amd.build, or "
            "__graft_entry__.build()). There is no CPU fallback." % lib_path)
    lib = ctypes.CDLL(lib_path)
    c_int, c_i64, c_vp = ctypes.c_int, ctypes.c_int64, ctypes.c_void_p
    lib.clair_abi_version.restype = c_int
    lib.clair_device_count.restype = c_int
    if not older_ok_early(path) or hasattr(lib, "clair_device_pci_bus_id"):
        lib.clair_device_pci_bus_id.argtypes = [c_int, ctypes.c_char_p, c_int]
    lib.clair_last_error.restype = ctypes.c_char_p
    lib.clair_last_error.argtypes = [c_vp]
    lib.clair_engine_create.argtypes = [c_int, c_int, c_int, ctypes.POINTER(c_vp)]
    lib.clair_engine_destroy.argtypes = [c_vp]
    lib.clair_engine_destroy.restype = None
    lib.clair_set_tensor.argtypes = [c_vp, c_int, c_vp, c_i64]
    lib.clair_finalize_weights.argtypes = [c_vp]
    lib.clair_predict.argtypes = [c_vp, c_vp, c_int, c_vp, c_vp, c_vp, c_vp]
    lib.clair_submit.argtypes = [c_vp, c_int, c_vp, c_int, c_vp, c_vp, c_vp, c_vp]
    older_ok = path is not None or bool(os.environ.get("CLAIR_AMD_LIB"))
    if not older_ok or hasattr(lib, "clair_submit_ex"):     # an OLDER build named by `path` (A/B timing, tools/gpu/ab_libs.sh) may predate these
        lib.clair_submit_ex.argtypes = [c_vp, c_int, c_vp, c_int, c_i64, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]
        lib.clair_decode.argtypes = [c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_vp, c_vp]
        lib.clair_pinned_alloc.argtypes = [c_vp, c_i64, ctypes.POINTER(c_vp)]
        lib.clair_pinned_free.argtypes = [c_vp, c_vp]
    lib.clair_wait.argtypes = [c_vp, c_int]
    lib.clair_slot_input.argtypes = [c_vp, c_int, ctypes.POINTER(c_vp)]
    lib.clair_submit_counts.argtypes = [c_vp, c_int, c_vp, c_int, c_vp, c_vp, c_vp, c_vp]
    lib.clair_dataset_alloc.argtypes = [c_vp, c_i64, ctypes.POINTER(c_vp), ctypes.POINTER(c_vp)]
    lib.clair_dataset_free.argtypes = [c_vp, c_vp, c_vp]
    lib.clair_dataset_upload.argtypes = [c_vp, c_vp, c_i64, c_vp, c_i64]
    lib.clair_dataset_download.argtypes = [c_vp, c_vp, c_i64, c_vp, c_i64]
    lib.clair_run_resident.argtypes = [c_vp, c_int, c_vp, c_vp, c_i64, c_int]
    lib.clair_sync.argtypes = [c_vp]
    lib.clair_timing_enable.argtypes = [c_vp, c_int]
    lib.clair_kernel_times.argtypes = [c_vp, c_vp, c_vp]
    lib.clair_timing_reset.argtypes = [c_vp]
    lib.clair_kernel_workgroups.argtypes = [c_vp, c_int, c_vp]
    lib.clair_debug_read.argtypes = [c_vp, c_int, c_int, c_vp, c_i64]
    lib.clair_engine_counter.argtypes = [c_vp, c_int, ctypes.POINTER(c_i64)]
    lib.clair_comm_preflight.argtypes = [c_int]
    lib.clair_comm_unique_id.argtypes = [c_vp]
    lib.clair_comm_create.argtypes = [c_int, c_int, c_int, c_vp, ctypes.POINTER(c_vp)]
    if not older_ok or hasattr(lib, "clair_comm_create_timed"):
        lib.clair_comm_create_timed.argtypes = [c_int, c_int, c_int, c_vp, c_int, ctypes.POINTER(c_vp)]
    lib.clair_comm_destroy.argtypes = [c_vp]
    lib.clair_comm_destroy.restype = None
    if not older_ok or hasattr(lib, "clair_comm_abort"):
        lib.clair_comm_abort.argtypes = [c_vp]
        lib.clair_comm_abort.restype = None
    lib.clair_comm_last_error.argtypes = [c_vp]
    lib.clair_comm_last_error.restype = ctypes.c_char_p
    lib.clair_comm_barrier.argtypes = [c_vp]
    lib.clair_comm_allreduce_f64.argtypes = [c_vp, c_vp, c_int, c_int]
    lib.clair_comm_broadcast.argtypes = [c_vp, c_vp, c_i64, c_int]
    lib.clair_comm_allgather.argtypes = [c_vp, c_vp, c_vp, c_i64]
    lib.clair_comm_allgather_device.argtypes = [c_vp, c_vp, c_vp, c_i64]
    if not older_ok or hasattr(lib, "clair_frontend_create"):
        lib.clair_frontend_create.argtypes = [c_int, ctypes.c_char_p, c_i64, c_i64, c_i64, c_i64, ctypes.POINTER(c_vp)]
        lib.clair_frontend_destroy.argtypes = [c_vp]
        lib.clair_frontend_destroy.restype = None
        lib.clair_frontend_last_error.argtypes = [c_vp]
        lib.clair_frontend_last_error.restype = ctypes.c_char_p
        lib.clair_frontend_add_reads.argtypes = [c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_i64]
        lib.clair_frontend_find_candidates.argtypes = [c_vp, ctypes.c_double, ctypes.c_double, c_i64, c_i64, c_vp, c_vp, c_i64, ctypes.POINTER(c_i64)]
        lib.clair_frontend_set_candidates.argtypes = [c_vp, c_vp, c_i64, ctypes.POINTER(c_i64)]
        lib.clair_frontend_get_candidates.argtypes = [c_vp, c_vp]
        lib.clair_frontend_build_windows.argtypes = [c_vp, c_int, c_int, ctypes.POINTER(c_i64)]
        lib.clair_frontend_build_windows_ex.argtypes = [c_vp, c_int, c_int, c_int, ctypes.POINTER(c_i64)]
        lib.clair_frontend_window_info.argtypes = [c_vp, c_i64, c_i64, c_vp, c_vp]
        lib.clair_frontend_window_counts.argtypes = [c_vp, c_i64, c_i64, c_vp]
        lib.clair_frontend_counts_device.argtypes = [c_vp, c_i64]
        lib.clair_frontend_counts_device.restype = c_vp
        lib.clair_frontend_budget_inputs.argtypes = [c_vp, c_i64, c_vp, c_vp, c_vp]
        lib.clair_frontend_stats.argtypes = [c_vp, c_vp]
        lib.clair_frontend_text_options.argtypes = [c_vp, ctypes.c_char_p, c_int, c_int, c_int, c_i64, c_i64]
        lib.clair_frontend_add_text.argtypes = [c_vp, c_vp, c_i64]
        lib.clair_frontend_text_stats.argtypes = [c_vp, c_vp]
        lib.clair_frontend_slab_reads.argtypes = [c_vp, c_i64, c_vp, c_i64, ctypes.POINTER(c_i64)]
    for name in SYMBOLS:
        if older_ok and not hasattr(lib, name):
            continue
        fn = getattr(lib, name)
        if name not in ("clair_last_error", "clair_engine_destroy", "clair_comm_last_error", "clair_comm_destroy",
                        "clair_frontend_last_error", "clair_frontend_destroy", "clair_frontend_counts_device"):
            fn.restype = c_int
    if path is None:
        _lib = lib
    else:
        _libs[path] = lib
    return lib


def _ptr(a):
    return ctypes.c_void_p(a.ctypes.data)


class Engine(object):
    """Thin object wrapper over one clair_engine_t."""

    def __init__(self, device=0, max_batch=1024, n_slots=1, lib_path=None):
        self._lib = load(lib_path)
        self._h = ctypes.c_void_p()
        self.max_batch = int(max_batch)
        self.n_slots = int(n_slots)
        rc = self._lib.clair_engine_create(int(device), int(max_batch), int(n_slots), ctypes.byref(self._h))
        if rc != 0:
            msg = self._lib.clair_last_error(None).decode()
            self._h = ctypes.c_void_p()
            raise EngineError("clair_engine_create failed: %s" % msg)
        self._pending = {}

    def _check(self, rc, what):
        if rc != 0:
            raise EngineError("%s failed: %s" % (what, self._lib.clair_last_error(self._h).decode()))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.clair_engine_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- weights ---------------------------------------------------------------------------
    def load_weights(self, w):
        from clair_amd.weights import TENSOR_IDS, check_weights
        check_weights(w)
        for key, tid in TENSOR_IDS.items():
            a = np.ascontiguousarray(w[key], dtype=np.float32)
            self._check(self._lib.clair_set_tensor(self._h, tid, _ptr(a), a.size), "clair_set_tensor(%s)" % key)
        self._check(self._lib.clair_finalize_weights(self._h), "clair_finalize_weights")

    # -- predict ---------------------------------------------------------------------------
    @staticmethod
    def _prep_x(x):
        x = np.ascontiguousarray(x, dtype=np.float32)
        if x.ndim < 2 or x.size != x.shape[0] * 1056:
            raise ValueError("input must be [n,33,8,4] float32, got shape %s" % (x.shape,))
        return x

    @staticmethod
    def _alloc_out(n):
        return [np.empty((n, m), dtype=np.float32) for m in (21, 3, 33, 33)]

    def predict(self, x):
        x = self._prep_x(x)
        outs = self._alloc_out(x.shape[0])
        self._check(self._lib.clair_predict(self._h, _ptr(x), x.shape[0], *[_ptr(o) for o in outs]), "clair_predict")
        return outs

    def submit(self, slot, x):
        x = self._prep_x(x)
        outs = self._alloc_out(x.shape[0])
        self._check(self._lib.clair_submit(self._h, slot, _ptr(x), x.shape[0], *[_ptr(o) for o in outs]), "clair_submit")
        self._pending[slot] = (x, outs)  # keep the buffers alive until wait()

    def submit_counts(self, slot, counts):
        """submit() for raw pileup counts [n,33,8,4] int16 (channel 0 not yet subtracted): half the bytes on the host link,
        conversion on the device; pair with wait(slot)."""
        c = np.ascontiguousarray(counts, dtype=np.int16)
        if c.ndim != 4 or c.shape[1:] != (33, 8, 4):
            raise ValueError("counts must have shape [n,33,8,4], got %r" % (c.shape,))
        outs = self._alloc_out(c.shape[0])
        self._check(self._lib.clair_submit_counts(self._h, slot, _ptr(c), c.shape[0], *[_ptr(o) for o in outs]), "clair_submit_counts")
        self._pending[slot] = (c, outs)

    def submit_calls(self, slot, batch, centre, counts=False, with_probabilities=False):
        """clair_submit_ex: forward pass + decode on the device.  batch: [n,33,8,4] float32, or raw int16 counts with counts=True (the
        candidates may be a strided view, e.g. the counts column of an array of binary tensor records: no copy is made here);
        centre: uint8 [n,2] (clair_amd._hostapi.centre_bytes).  wait(slot) then returns the call records (structured array,
        _hostapi.CALL_DTYPE), or (records, [gt21, genotype, len1, len2]) with with_probabilities=True."""
        from clair_amd._hostapi import CALL_DTYPE
        if isinstance(batch, DeviceWindows):       # windows the device front end left in HBM: the address goes through as it is
            n = len(batch)
            c = np.ascontiguousarray(centre, dtype=np.uint8)
            if c.shape != (n, 2):
                raise ValueError("centre must be uint8 [%d,2], got %r" % (n, c.shape))
            calls = np.zeros(n, dtype=CALL_DTYPE)
            outs = self._alloc_out(n) if with_probabilities else None
            ptrs = [_ptr(o) for o in outs] if outs else [None] * 4
            self._check(self._lib.clair_submit_ex(self._h, int(slot), ctypes.c_void_p(batch.address), 1, 0, n, _ptr(c), _ptr(calls), *ptrs), "clair_submit_ex")
            self._pending[slot] = ((batch, c), (calls, outs) if outs else calls)
            return
        dtype = np.int16 if counts else np.float32
        x = np.asarray(batch)
        if x.ndim != 4 or x.shape[1:] != (33, 8, 4):
            raise ValueError("batch must have shape [n,33,8,4], got %r" % (x.shape,))
        n = x.shape[0]
        inner_dense = x.dtype == dtype and n > 0 and x[0].flags.c_contiguous and x.strides[0] >= x[0].nbytes
        if not inner_dense:
            x = np.ascontiguousarray(x, dtype=dtype)
        stride = 0 if x.flags.c_contiguous else int(x.strides[0])
        c = np.ascontiguousarray(centre, dtype=np.uint8)
        if c.shape != (n, 2):
            raise ValueError("centre must be uint8 [%d,2], got %r" % (n, c.shape))
        calls = np.zeros(n, dtype=CALL_DTYPE)
        outs = self._alloc_out(n) if with_probabilities else None
        ptrs = [_ptr(o) for o in outs] if outs else [None] * 4
        self._check(self._lib.clair_submit_ex(self._h, int(slot), _ptr(x), int(bool(counts)), stride, n, _ptr(c), _ptr(calls), *ptrs), "clair_submit_ex")
        self._pending[slot] = ((x, c), (calls, outs) if outs else calls)

    def pinned_buffer(self, nbytes):
        """A page-locked uint8 array of nbytes (clair_pinned_alloc): data placed in it -- e.g. read from a file with readinto -- goes to
        the GPU from where it lies when a view INTO it is handed to submit_calls.  It lives as long as the engine."""
        p = ctypes.c_void_p()
        self._check(self._lib.clair_pinned_alloc(self._h, int(nbytes), ctypes.byref(p)), "clair_pinned_alloc")
        return np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_uint8)), shape=(int(nbytes),))

    def decode(self, x, Y, centre, slot=0):
        """clair_decode: the device decode on given probabilities Y = [gt21, genotype, len1, len2] -> call records."""
        from clair_amd._hostapi import CALL_DTYPE
        x = self._prep_x(x)
        n = x.shape[0]
        ys = [np.ascontiguousarray(a, dtype=np.float32) for a in Y]
        c = np.ascontiguousarray(centre, dtype=np.uint8)
        if c.shape != (n, 2) or [a.shape for a in ys] != [(n, 21), (n, 3), (n, 33), (n, 33)]:
            raise ValueError("decode: shapes do not match %d candidates" % n)
        calls = np.zeros(n, dtype=CALL_DTYPE)
        self._check(self._lib.clair_decode(self._h, int(slot), _ptr(x), *[_ptr(a) for a in ys], n, _ptr(c), _ptr(calls)), "clair_decode")
        return calls

    def slot_input(self, slot):
        """The slot's page-locked input buffer as a NumPy array [max_batch,33,8,4] float32: fill rows [0,n) and submit
        `buf[:n]` for a direct DMA transfer instead of the staged copy pageable arrays get."""
        p = ctypes.c_void_p()
        self._check(self._lib.clair_slot_input(self._h, int(slot), ctypes.byref(p)), "clair_slot_input")
        n = self.max_batch * 1056
        return np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_float)), shape=(n,)).reshape(self.max_batch, 33, 8, 4)

    def wait(self, slot):
        self._check(self._lib.clair_wait(self._h, slot), "clair_wait")
        _, outs = self._pending.pop(slot)
        return outs

    # -- resident data sets ------------------------------------------------------------------
    def dataset_alloc(self, n):
        xd, od = ctypes.c_void_p(), ctypes.c_void_p()
        self._check(self._lib.clair_dataset_alloc(self._h, int(n), ctypes.byref(xd), ctypes.byref(od)), "clair_dataset_alloc")
        return xd, od

    def dataset_free(self, xd, od):
        self._check(self._lib.clair_dataset_free(self._h, xd, od), "clair_dataset_free")

    def dataset_upload(self, xd, first, x):
        x = self._prep_x(x)
        self._check(self._lib.clair_dataset_upload(self._h, xd, int(first), _ptr(x), x.shape[0]), "clair_dataset_upload")

    def dataset_download(self, od, first, n):
        out = np.empty((n, 90), dtype=np.float32)
        self._check(self._lib.clair_dataset_download(self._h, od, int(first), _ptr(out), int(n)), "clair_dataset_download")
        return out

    def run_resident(self, slot, xd, od, first, n):
        self._check(self._lib.clair_run_resident(self._h, int(slot), xd, od, int(first), int(n)), "clair_run_resident")

    def sync(self):
        self._check(self._lib.clair_sync(self._h), "clair_sync")

    # -- measurement ---------------------------------------------------------------------------
    def timing_enable(self, on=True, only=None):
        """HIP-event timing of every kernel (on=True), of none (False), or of the kernels named in `only` (KERNEL_NAMES)."""
        mask = int(bool(on))
        if only:
            mask = 0
            for k in only:
                mask |= 1 << KERNEL_NAMES.index(k)
        self._check(self._lib.clair_timing_enable(self._h, mask), "clair_timing_enable")

    def timing_reset(self):
        self._check(self._lib.clair_timing_reset(self._h), "clair_timing_reset")

    def kernel_times(self):
        ms = np.zeros(len(KERNEL_NAMES), dtype=np.float64)
        cnt = np.zeros(len(KERNEL_NAMES), dtype=np.int64)
        self._check(self._lib.clair_kernel_times(self._h, _ptr(ms), _ptr(cnt)), "clair_kernel_times")
        return {k: (float(ms[i]), int(cnt[i])) for i, k in enumerate(KERNEL_NAMES)}

    def kernel_workgroups(self, n):
        """Workgroups each kernel is launched with for a batch of n candidates (its share of the 256 CUs)."""
        wg = np.zeros(len(KERNEL_NAMES), dtype=np.int32)
        self._check(self._lib.clair_kernel_workgroups(self._h, int(n), _ptr(wg)), "clair_kernel_workgroups")
        return {k: int(wg[i]) for i, k in enumerate(KERNEL_NAMES)}

    def counter(self, name):
        """Event counters of the handle: "fused_launches", "fused_recoveries" (include/clair_amd.h: clair_engine_counter)."""
        v = ctypes.c_int64()
        self._check(self._lib.clair_engine_counter(self._h, ("fused_launches", "fused_recoveries").index(name), ctypes.byref(v)), "clair_engine_counter")
        return int(v.value)

    def debug_read(self, slot, which, shape):
        out = np.empty(shape, dtype=np.float32)
        self._check(self._lib.clair_debug_read(self._h, int(slot), int(which), _ptr(out), out.size), "clair_debug_read")
        return out


def split_outputs(packed):
    """[n,90] packed rows -> [gt21, genotype, len1, len2] (copies, C-contiguous)."""
    return [np.ascontiguousarray(packed[:, a:b]) for a, b in ((0, 21), (21, 24), (24, 57), (57, 90))]


class MalformedText(EngineError):
    pass


class DeviceWindows(object):
    """n pileup windows [33][8][4] int16 in device memory (clair_frontend_counts_device): what Engine.submit_calls takes in place of a
    host array.  Keeps its Frontend alive; host() copies the counts back (the decode needs them only when a BAM is consulted)."""

    def __init__(self, frontend, first, n):
        self.frontend, self.first, self.n = frontend, int(first), int(n)
        self.address = frontend.counts_address(first)

    def __len__(self):
        return self.n

    def host(self):
        return self.frontend.window_counts(self.first, self.n)


class Frontend(object):
    """Thin object wrapper over one clair_frontend_t (include/clair_amd.h, "front end on the device")."""

    def __init__(self, device, reference_sequence, reference_start_0_based, span_lo, span_hi, lib_path=None):
        self._lib = load(lib_path)
        ref = reference_sequence.encode("latin-1") if isinstance(reference_sequence, str) else bytes(reference_sequence)
        self._h = ctypes.c_void_p()
        rc = self._lib.clair_frontend_create(int(device), ref, len(ref), int(reference_start_0_based), int(span_lo), int(span_hi), ctypes.byref(self._h))
        if rc != 0:
            msg = self._lib.clair_frontend_last_error(None).decode()
            self._h = ctypes.c_void_p()
            raise EngineError("clair_frontend_create failed: %s" % msg)
        self.slab_reads = []                # host copies of each slab's read records: the budget replay walks them

    def _check(self, rc, what):
        if rc != 0:
            raise EngineError("%s failed: %s" % (what, self._lib.clair_frontend_last_error(self._h).decode()))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.clair_frontend_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def add_slab(self, packer):
        """Send the slab a clair_amd._hostapi.SamPacker is holding to the device and start the next one."""
        (r, o, e, q), st = packer.slab_pointers()
        if st["reads"] and st["ops"]:
            from clair_amd._hostapi import READ_DTYPE
            self._check(self._lib.clair_frontend_add_reads(self._h, r, st["reads"], o, st["ops"], e, q, st["seq_bytes"]), "clair_frontend_add_reads")
            buf = (ctypes.c_char * (st["reads"] * READ_DTYPE.itemsize)).from_address(r)
            self.slab_reads.append(np.frombuffer(buf, dtype=READ_DTYPE).copy())
        packer.reset()

    def add_arrays(self, reads, ops, op_elem, seq):
        """The same from NumPy arrays (tests)."""
        from clair_amd._hostapi import OP_DTYPE, READ_DTYPE
        reads = np.ascontiguousarray(reads, dtype=READ_DTYPE)
        ops = np.ascontiguousarray(ops, dtype=OP_DTYPE)
        op_elem = np.ascontiguousarray(op_elem, dtype=np.uint32)
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        if len(reads) and len(ops):
            self._check(self._lib.clair_frontend_add_reads(self._h, _ptr(reads), len(reads), _ptr(ops), len(ops), _ptr(op_elem), _ptr(seq), len(seq)),
                        "clair_frontend_add_reads")
            self.slab_reads.append(reads.copy())

    def text_options(self, ctg_name, dcov=250, evc_min_mq=0, pile_min_mq=0, pile_region=None):
        """What clair_amd._hostapi.SamPacker takes: from here on add_text() does the packing on the device."""
        a, b = (-1, -1) if pile_region is None else (int(pile_region[0]), int(pile_region[1]))
        self._check(self._lib.clair_frontend_text_options(self._h, ctg_name.encode(), int(dcov), int(evc_min_mq), int(pile_min_mq), a, b),
                    "clair_frontend_text_options")

    def add_text(self, sam, length=None):
        """`samtools view` text, whole lines: bytes, or (address, length) of a buffer -- e.g. a page-locked one of the engine the pipe was
        read into.  Raises MalformedText when a line is not an alignment line (the host packer on the same text says which and why)."""
        if length is None:
            address, length = ctypes.cast(ctypes.c_char_p(sam), ctypes.c_void_p).value, len(sam)
        else:
            address = int(sam)
        before = self.stats()["slabs"]
        rc = self._lib.clair_frontend_add_text(self._h, address, int(length))
        if rc == 2:
            raise MalformedText(self._lib.clair_frontend_last_error(self._h).decode())
        self._check(rc, "clair_frontend_add_text")
        if self.stats()["slabs"] > before:
            from clair_amd._hostapi import READ_DTYPE
            n = ctypes.c_int64(0)
            self._check(self._lib.clair_frontend_slab_reads(self._h, before, None, 0, ctypes.byref(n)), "clair_frontend_slab_reads")
            reads = np.empty(n.value, dtype=READ_DTYPE)
            self._check(self._lib.clair_frontend_slab_reads(self._h, before, _ptr(reads), n.value, ctypes.byref(n)), "clair_frontend_slab_reads")
            self.slab_reads.append(reads)

    def text_stats(self):
        v = (ctypes.c_int64 * 4)()
        self._check(self._lib.clair_frontend_text_stats(self._h, v), "clair_frontend_text_stats")
        return dict(zip(("lines", "evc_reads", "pile_reads", "anomalies"), [int(x) for x in v]))

    def find_candidates(self, min_coverage=4, threshold=0.125, ctg_start=None, ctg_end=None, bed=None):
        have_range = ctg_start is not None and ctg_end is not None
        if bed is None:
            bs = be = np.zeros(0, dtype=np.int64)
            n_bed = -1
        else:
            bs = np.ascontiguousarray([b[0] for b in bed], dtype=np.int64)
            be = np.ascontiguousarray([b[1] for b in bed], dtype=np.int64)
            n_bed = len(bs)
        n = ctypes.c_int64(0)
        self._check(self._lib.clair_frontend_find_candidates(self._h, float(min_coverage), float(threshold), int(ctg_start) if have_range else -1,
                                                             int(ctg_end) if have_range else -1, _ptr(bs), _ptr(be), n_bed, ctypes.byref(n)),
                    "clair_frontend_find_candidates")
        return int(n.value)

    def set_candidates(self, positions):
        p = np.ascontiguousarray(positions, dtype=np.int64)
        n = ctypes.c_int64(0)
        self._check(self._lib.clair_frontend_set_candidates(self._h, _ptr(p), len(p), ctypes.byref(n)), "clair_frontend_set_candidates")
        return int(n.value)

    def candidates(self):
        out = np.empty(max(self.stats()["candidates"], 0), dtype=np.int64)
        self._check(self._lib.clair_frontend_get_candidates(self._h, _ptr(out)), "clair_frontend_get_candidates")
        return out

    def build_windows(self, min_coverage=0, drop_non_iupac_centre=True, consider_left_edge=True):
        n = ctypes.c_int64(0)
        self._check(self._lib.clair_frontend_build_windows_ex(self._h, int(min_coverage), int(bool(drop_non_iupac_centre)), int(bool(consider_left_edge)),
                                                              ctypes.byref(n)), "clair_frontend_build_windows")
        return int(n.value)

    def window_info(self, first, n):
        """-> (centres int64 [n], refseq uint8 [n,34] NUL-padded)"""
        centres = np.empty(n, dtype=np.int64)
        seqs = np.zeros((n, 34), dtype=np.uint8)
        self._check(self._lib.clair_frontend_window_info(self._h, int(first), int(n), _ptr(centres), _ptr(seqs)), "clair_frontend_window_info")
        return centres, seqs

    def window_counts(self, first, n):
        counts = np.empty((n, 33, 8, 4), dtype=np.int16)
        self._check(self._lib.clair_frontend_window_counts(self._h, int(first), int(n), _ptr(counts)), "clair_frontend_window_counts")
        return counts

    def counts_address(self, first):
        a = self._lib.clair_frontend_counts_device(self._h, int(first))
        if not a:
            raise EngineError("clair_frontend_counts_device: no windows yet")
        return int(a)

    def stats(self):
        v = (ctypes.c_int64 * 6)()
        self._check(self._lib.clair_frontend_stats(self._h, v), "clair_frontend_stats")
        return dict(zip(("anomalies", "slabs", "reads", "elements", "candidates", "windows"), [int(x) for x in v]))

    def budget_binds(self, available_slots=5000000):
        """Replay CreateTensor's count of free tuple slots over everything added (clair_host_tuple_budget_binds)."""
        from clair_amd import _hostapi
        n_cand = self.stats()["candidates"]
        centres = np.empty(max(n_cand, 0), dtype=np.int64)
        window_tuples = np.empty(max(n_cand, 0), dtype=np.uint64)
        self._check(self._lib.clair_frontend_budget_inputs(self._h, 0, None, _ptr(centres), _ptr(window_tuples)), "clair_frontend_budget_inputs")
        state = np.array([available_slots, 0], dtype=np.int64)
        for k, reads in enumerate(self.slab_reads):
            tuples = np.empty(len(reads), dtype=np.uint64)
            self._check(self._lib.clair_frontend_budget_inputs(self._h, k, _ptr(tuples), None, None), "clair_frontend_budget_inputs")
            if _hostapi.tuple_budget_binds(reads, tuples, centres, window_tuples, state):
                return True
        return False

    def read_tuples(self, slab):
        tuples = np.empty(len(self.slab_reads[slab]), dtype=np.uint64)
        self._check(self._lib.clair_frontend_budget_inputs(self._h, int(slab), _ptr(tuples), None, None), "clair_frontend_budget_inputs")
        return tuples

    def window_tuples(self):
        n_cand = max(self.stats()["candidates"], 0)
        centres = np.empty(n_cand, dtype=np.int64)
        window_tuples = np.empty(n_cand, dtype=np.uint64)
        self._check(self._lib.clair_frontend_budget_inputs(self._h, 0, None, _ptr(centres), _ptr(window_tuples)), "clair_frontend_budget_inputs")
        return centres, window_tuples
