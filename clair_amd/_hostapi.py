"""ctypes binding of include/clair_host.h (libclair_host.so, built by clair_amd/build.py with g++: no GPU involved)."""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libclair_host.so")
SYMBOLS = ("clair_host_abi_version", "clair_host_last_error", "clair_host_threads", "clair_host_crc32c", "clair_host_parse_tensors",
           "clair_host_counts_to_input_i16", "clair_host_counts_to_input_i32",
           "clair_host_decode_rows", "clair_host_decode_rows_ex", "clair_host_resolve_calls", "clair_host_format_calls", "clair_host_format_calls_records", "clair_host_centre_bytes",
           "clair_host_pileup_create", "clair_host_pileup_destroy", "clair_host_pileup_feed", "clair_host_pileup_finish",
           "clair_host_pileup_pending", "clair_host_pileup_take", "clair_host_pileup_take_text", "clair_host_pileup_stats",
           "clair_host_pileup_set_order", "clair_host_pyset_order",
           "clair_host_evc_create", "clair_host_evc_destroy", "clair_host_evc_feed", "clair_host_evc_finish",
           "clair_host_evc_pending", "clair_host_evc_reads", "clair_host_evc_take", "clair_host_evc_take_text",
           "clair_host_sampack_create", "clair_host_sampack_destroy", "clair_host_sampack_feed", "clair_host_sampack_stats",
           "clair_host_sampack_slab", "clair_host_sampack_reset", "clair_host_tuple_budget_binds")
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
        lib.clair_host_crc32c.restype = ctypes.c_uint32
        lib.clair_host_crc32c.argtypes = [ctypes.c_char_p, i64]
        lib.clair_host_parse_tensors.argtypes = [vp, i64, i32, i32, vp, vp, ctypes.POINTER(i32), ctypes.POINTER(i32),
                                                 ctypes.POINTER(i64)]
        lib.clair_host_decode_rows.argtypes = [vp, vp, vp, vp, vp, ctypes.c_char_p, vp, i32, i32, i32, i32, i32, i32, vp, i64,
                                               ctypes.POINTER(i64), ctypes.POINTER(i32)]
        lib.clair_host_decode_rows_ex.argtypes = lib.clair_host_decode_rows.argtypes + [vp]
        lib.clair_host_resolve_calls.argtypes = [vp, vp, vp, vp, vp, vp, i32, vp]
        lib.clair_host_format_calls.argtypes = [vp, ctypes.c_char_p, vp, i32, i32, i32, i32, i32, i32, vp, i64, ctypes.POINTER(i64), ctypes.POINTER(i32), vp]
        lib.clair_host_centre_bytes.argtypes = [ctypes.c_char_p, vp, i32, vp]
        lib.clair_host_format_calls_records.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, i64, ctypes.POINTER(i64),
                                                        ctypes.POINTER(i32), vp]
        lib.clair_host_pileup_create.argtypes = [ctypes.c_char_p, i64, i64, vp, i64, i32, i32, i32, i32, i64, i32, ctypes.POINTER(vp)]
        lib.clair_host_pileup_destroy.argtypes = [vp]
        lib.clair_host_pileup_destroy.restype = None
        lib.clair_host_pileup_set_order.argtypes = [vp, i32]
        lib.clair_host_pyset_order.argtypes = [vp, i64, vp, i64, ctypes.POINTER(i64)]
        lib.clair_host_pileup_feed.argtypes = [vp, vp, i64, i32, ctypes.POINTER(i64)]
        lib.clair_host_pileup_finish.argtypes = [vp]
        lib.clair_host_pileup_pending.argtypes = [vp]
        lib.clair_host_pileup_pending.restype = i64
        lib.clair_host_pileup_take.argtypes = [vp, i64, vp, vp, vp, ctypes.POINTER(i64)]
        lib.clair_host_pileup_take_text.argtypes = [vp, ctypes.c_char_p, vp, i64, ctypes.POINTER(i64), ctypes.POINTER(i64)]
        lib.clair_host_pileup_stats.argtypes = [vp, vp]
        lib.clair_host_evc_create.argtypes = [ctypes.c_char_p, ctypes.c_char_p, i64, i64, i64, i64, vp, vp, i64, ctypes.c_double,
                                              ctypes.c_double, i32, ctypes.POINTER(vp)]
        lib.clair_host_evc_destroy.argtypes = [vp]
        lib.clair_host_evc_destroy.restype = None
        lib.clair_host_evc_feed.argtypes = [vp, vp, i64, i32, ctypes.POINTER(i64)]
        lib.clair_host_evc_finish.argtypes = [vp]
        lib.clair_host_evc_pending.argtypes = [vp]
        lib.clair_host_evc_pending.restype = i64
        lib.clair_host_evc_reads.argtypes = [vp]
        lib.clair_host_evc_reads.restype = i64
        lib.clair_host_evc_take.argtypes = [vp, i64, vp, ctypes.POINTER(i64)]
        lib.clair_host_evc_take_text.argtypes = [vp, vp, i64, ctypes.POINTER(i64), ctypes.POINTER(i64)]
        lib.clair_host_counts_to_input_i16.argtypes = [vp, i64, vp]
        lib.clair_host_counts_to_input_i32.argtypes = [vp, i64, vp]
        lib.clair_host_sampack_create.argtypes = [ctypes.c_char_p, i32, i32, i32, i64, i64, ctypes.POINTER(vp)]
        lib.clair_host_sampack_destroy.argtypes = [vp]
        lib.clair_host_sampack_destroy.restype = None
        lib.clair_host_sampack_feed.argtypes = [vp, vp, i64, i32, ctypes.POINTER(i64)]
        lib.clair_host_sampack_stats.argtypes = [vp, vp]
        lib.clair_host_sampack_slab.argtypes = [vp, ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(vp)]
        lib.clair_host_sampack_reset.argtypes = [vp]
        lib.clair_host_tuple_budget_binds.argtypes = [vp, vp, i64, vp, vp, i64, vp, ctypes.POINTER(i32)]
        if lib.clair_host_abi_version() != 6:
            raise RuntimeError("libclair_host.so has ABI version %d, expected 6: run `python -m clair_amd.build`"
                               % lib.clair_host_abi_version())
        _lib = lib
    return _lib


def crc32c(data):
    """CRC32C (Castagnoli) of a bytes object."""
    return int(load().clair_host_crc32c(data, len(data)))


def counts_to_input(counts):
    """Raw pileup counts [n,33,8,4] (int16 or int32) -> float32 network input with channels 1..3 -= channel 0 (utils.py:96-98)."""
    lib = load()
    c = np.ascontiguousarray(counts)
    if c.dtype not in (np.int16, np.int32):
        c = c.astype(np.int32)
    x = np.empty(c.shape, dtype=np.float32)
    fn = lib.clair_host_counts_to_input_i16 if c.dtype == np.int16 else lib.clair_host_counts_to_input_i32
    if fn(c.ctypes.data, c.size // 4, x.ctypes.data) != 0:
        raise ValueError("counts_to_input: " + lib.clair_host_last_error().decode())
    return x


class MetaInfoTable(object):
    """The [[ctg, pos, seq], ...] list of a batch of text records, kept as the bytes the parser found them in: `meta` holds the
    three fields of every candidate back to back, tok[i] = (offset, length) x 3 into it -- the form clair_host_decode_rows takes, so
    a batch reaches the native decoder without a Python string per field.  List-like: len, iteration, indexing and comparison give
    the lists of three strings."""

    def __init__(self, meta, tok):
        self.meta, self.tok = meta, tok
        self._rows = None

    @classmethod
    def from_chunk(cls, chunk, tok):
        """Compact the fields tok points at inside `chunk` (a parse buffer of megabytes) into their own small buffer."""
        k = len(tok)
        lens = tok[:, 1::2].astype(np.int64).ravel()
        starts = tok[:, 0::2].astype(np.int64).ravel()
        before = np.cumsum(lens) - lens
        index = np.repeat(starts - before, lens) + np.arange(int(lens.sum()), dtype=np.int64)
        meta = np.frombuffer(chunk, dtype=np.uint8)[index].tobytes()
        out = np.empty((k, 6), dtype=np.int32)
        out[:, 0::2] = before.reshape(k, 3)
        out[:, 1::2] = lens.reshape(k, 3)
        return cls(meta, out)

    @classmethod
    def concat(cls, tables):
        tables = [t for t in tables if len(t)]
        if len(tables) == 1:
            return tables[0]
        if not tables:
            return cls(b"", np.zeros((0, 6), dtype=np.int32))
        shift, toks = 0, []
        for t in tables:
            tk = t.tok.copy()
            tk[:, 0::2] += shift
            toks.append(tk)
            shift += len(t.meta)
        return cls(b"".join(t.meta for t in tables), np.concatenate(toks))

    def __len__(self):
        return len(self.tok)

    def rows(self):
        if self._rows is None:
            m = self.meta
            self._rows = [[m[o[0]:o[0] + o[1]].decode(), m[o[2]:o[2] + o[3]].decode(), m[o[4]:o[4] + o[5]].decode()] for o in self.tok.tolist()]
        return self._rows

    def __getitem__(self, i):
        return self.rows()[i]

    def __iter__(self):
        return iter(self.rows())

    def __eq__(self, other):
        return self.rows() == (other.rows() if hasattr(other, "rows") else other)

    def native_meta(self):
        return self.meta, self.tok


def parse_tensors(chunk, final, max_rows, x_out, row0, offset=0):
    """Parse up to max_rows lines of `chunk` (bytes), starting at byte `offset`, into x_out[row0:] (float32 [*,1056],
    C-contiguous).  -> (rows_taken, infos of the kept rows as a MetaInfoTable ([[ctg, pos, seq], ...]), bytes_consumed)"""
    lib = load()
    tok = np.empty((max(max_rows, 1), 6), dtype=np.int32)
    taken, kept, used = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int64(0)
    base = ctypes.cast(ctypes.c_char_p(chunk), ctypes.c_void_p).value      # bytes: passed by pointer, no copy
    rc = lib.clair_host_parse_tensors(base + offset, len(chunk) - offset, 1 if final else 0, max_rows,
                                      x_out[row0:].ctypes.data, tok.ctypes.data,
                                      ctypes.byref(taken), ctypes.byref(kept), ctypes.byref(used))
    if rc != 0:
        raise ValueError("malformed tensor record: " + lib.clair_host_last_error().decode())
    t = tok[:kept.value].astype(np.int64)
    t[:, 0::2] += offset
    return taken.value, MetaInfoTable.from_chunk(chunk, t), used.value


def decode_rows(X, infos, Y, show_reference, haploid_precision, haploid_sensitive, qual_threshold, arith_numpy2, with_status=False):
    """clair_host_decode_rows over one batch -> list of VCF row strings (input order, skipped candidates left out).
    with_status: also the per-candidate status bytes of clair_host_decode_rows_ex (bit 0: produced a row, bit 1: passed a
    point where the reference would consult the BAM) -> (rows, status uint8 [n])."""
    lib = load()
    n = len(infos)
    if n == 0:
        return ([], np.zeros(0, np.uint8)) if with_status else []
    x = np.ascontiguousarray(X, dtype=np.float32).reshape(n, N_VALUES)
    gt21, genotype, len1, len2 = [np.ascontiguousarray(a, dtype=np.float32) for a in Y]
    meta, tok = _meta_of(infos)
    cap = 4096 + 256 * n + 2 * len(meta)
    out = ctypes.create_string_buffer(cap)
    out_len, n_rows = ctypes.c_int64(0), ctypes.c_int(0)
    status = np.zeros(n, dtype=np.uint8)
    rc = lib.clair_host_decode_rows_ex(x.ctypes.data, gt21.ctypes.data, genotype.ctypes.data, len1.ctypes.data, len2.ctypes.data,
                                       meta, tok.ctypes.data, n, int(bool(show_reference)), int(bool(haploid_precision)),
                                       int(bool(haploid_sensitive)), -1 if qual_threshold is None else int(qual_threshold),
                                       int(bool(arith_numpy2)), out, cap, ctypes.byref(out_len), ctypes.byref(n_rows),
                                       status.ctypes.data)
    if rc != 0:
        raise ValueError("native decode: " + lib.clair_host_last_error().decode())
    rows = out.raw[:out_len.value - 1].decode("ascii").split("\n") if out_len.value else []
    return (rows, status) if with_status else rows


CALL_DTYPE = np.dtype([("status", "u1"), ("family", "u1"), ("index", "<u2"), ("flags", "<u2"), ("gt", "u1"), ("gi", "u1"),
                       ("alt_b0", "u1"), ("alt_b1", "u1"), ("ins_avail", "u1"), ("reserved0", "u1"), ("ins_code", "<u4"),
                       ("depth", "<f4"), ("support", "<f4"), ("p_call", "<f4"), ("rounds", "<u4")])      # include/clair_call.h
assert CALL_DTYPE.itemsize == 32


def _meta_of(infos):
    """(meta bytes, tok int32 [n,6]) of a batch's [[ctg, pos, seq], ...] -- the layout the native decode functions take."""
    n = len(infos)
    if hasattr(infos, "native_meta"):          # tensor_binary.InfoTable / MetaInfoTable: the record columns as they are
        return infos.native_meta()
    parts = [s for info in infos for s in (info[0], str(info[1]), info[2])]
    lens = np.fromiter(map(len, parts), dtype=np.int32, count=3 * n)
    tok = np.empty((n, 6), dtype=np.int32)
    tok[:, 1::2] = lens.reshape(n, 3)
    starts = np.cumsum(lens, dtype=np.int64) - lens
    tok[:, 0::2] = starts.reshape(n, 3)
    return "".join(parts).encode("ascii"), tok


def centre_bytes(infos):
    """uint8 [n,2]: the centre character of every candidate's reference window and min(window length, 255) -- what the call
    resolution needs of the candidate's text (clair_host_resolve_calls, clair_submit_ex)."""
    if hasattr(infos, "centre_bytes"):          # tensor_binary.InfoTable: straight from the record columns
        return infos.centre_bytes()
    lib = load()
    n = len(infos)
    meta, tok = _meta_of(infos)
    out = np.zeros((n, 2), dtype=np.uint8)
    if n and lib.clair_host_centre_bytes(meta, tok.ctypes.data, n, out.ctypes.data) != 0:
        raise ValueError("native decode: " + lib.clair_host_last_error().decode())
    return out


def resolve_calls(X, Y, centre):
    """clair_host_resolve_calls: the arithmetic half of the decode on the CPU -> structured array of call records (CALL_DTYPE)."""
    lib = load()
    n = len(centre)
    calls = np.zeros(n, dtype=CALL_DTYPE)
    if n == 0:
        return calls
    x = np.ascontiguousarray(X, dtype=np.float32).reshape(n, N_VALUES)
    gt21, genotype, len1, len2 = [np.ascontiguousarray(a, dtype=np.float32) for a in Y]
    c = np.ascontiguousarray(centre, dtype=np.uint8)
    if lib.clair_host_resolve_calls(x.ctypes.data, gt21.ctypes.data, genotype.ctypes.data, len1.ctypes.data, len2.ctypes.data,
                                    c.ctypes.data, n, calls.ctypes.data) != 0:
        raise ValueError("native decode: " + lib.clair_host_last_error().decode())
    return calls


def format_calls(calls, infos, show_reference, haploid_precision, haploid_sensitive, qual_threshold, arith_numpy2, with_status=False, as_text=False):
    """clair_host_format_calls: call records (from the GPU decode kernel or resolve_calls) + the batch's text -> VCF rows (a list of
    strings; as_text=True: the rows as one bytes object, each '\\n'-terminated, ready for the file).  A batch of binary tensor records
    (tensor_binary.InfoTable) hands its columns over as they are (clair_host_format_calls_records)."""
    lib = load()
    n = len(infos)
    if n == 0:
        empty = b"" if as_text else []
        return (empty, np.zeros(0, np.uint8)) if with_status else empty
    calls = np.ascontiguousarray(calls, dtype=CALL_DTYPE)
    if len(calls) != n:
        raise ValueError("%d call records for %d candidates" % (len(calls), n))
    flags = (int(bool(show_reference)), int(bool(haploid_precision)), int(bool(haploid_sensitive)),
             -1 if qual_threshold is None else int(qual_threshold), int(bool(arith_numpy2)))
    out_len, n_rows = ctypes.c_int64(0), ctypes.c_int(0)
    status = np.zeros(n, dtype=np.uint8)
    if hasattr(infos, "record_columns"):
        ctg, ctg_len, pos, seq, seq_len = infos.record_columns()
        cap = 4096 + 320 * n
        out = ctypes.create_string_buffer(cap)
        rc = lib.clair_host_format_calls_records(calls.ctypes.data, ctg.ctypes.data, ctg_len.ctypes.data, pos.ctypes.data, seq.ctypes.data,
                                                 seq_len.ctypes.data, n, *flags, out, cap, ctypes.byref(out_len), ctypes.byref(n_rows),
                                                 status.ctypes.data)
    else:
        meta, tok = _meta_of(infos)
        cap = 4096 + 256 * n + 2 * len(meta)
        out = ctypes.create_string_buffer(cap)
        rc = lib.clair_host_format_calls(calls.ctypes.data, meta, tok.ctypes.data, n, *flags, out, cap, ctypes.byref(out_len), ctypes.byref(n_rows),
                                         status.ctypes.data)
    if rc != 0:
        raise ValueError("native decode: " + lib.clair_host_last_error().decode())
    if as_text:
        rows = out.raw[:out_len.value]
    else:
        rows = out.raw[:out_len.value - 1].decode("ascii").split("\n") if out_len.value else []
    return (rows, status) if with_status else rows


def pyset_order(ops):
    """Iteration order of CPython's set after a history of operations (key >= 0: add, -(key + 1): remove), by the native restatement."""
    lib = load()
    ops = np.ascontiguousarray(ops, dtype=np.int64)
    keys = np.empty(max(len(ops), 1), np.int64)
    n = ctypes.c_int64(0)
    if lib.clair_host_pyset_order(ops.ctypes.data, len(ops), keys.ctypes.data, len(keys), ctypes.byref(n)) != 0:
        raise ValueError(lib.clair_host_last_error().decode())
    return keys[:n.value].tolist()


class PileupBuilder(object):
    """clair_host_pileup_*: the native twin of clair_amd.create_tensor.PileupBuilderPy (same constructor, same records)."""

    def __init__(self, ctg_name, reference_sequence, reference_start_0_based, candidates, consider_left_edge=True,
                 dcov=250, min_coverage=0, min_mq=0, available_slots=5000000, force_general_path=False, set_order="ascending"):
        if set_order not in ("ascending", "cpython"):
            raise ValueError("set_order: 'ascending' or 'cpython'")
        self._lib = load()
        self.ctg = ctg_name
        ref = reference_sequence.encode("latin-1") if isinstance(reference_sequence, str) else bytes(reference_sequence)
        cands = np.ascontiguousarray(candidates, dtype=np.int64)
        h = ctypes.c_void_p()
        rc = self._lib.clair_host_pileup_create(ref, len(ref), int(reference_start_0_based), cands.ctypes.data, len(cands),
                                                int(bool(consider_left_edge)), int(dcov), int(min_coverage), int(min_mq),
                                                int(available_slots), int(bool(force_general_path)), ctypes.byref(h))
        if rc != 0:
            raise ValueError("pileup: " + self._lib.clair_host_last_error().decode())
        self._h = h
        self._text = None
        if set_order == "cpython" and self._lib.clair_host_pileup_set_order(self._h, 1) != 0:
            raise ValueError("pileup: " + self._lib.clair_host_last_error().decode())

    def close(self):
        if getattr(self, "_h", None):
            self._lib.clair_host_pileup_destroy(self._h)
            self._h = None

    __del__ = close

    def feed(self, sam, final=False):
        """Consume complete lines of `sam` (bytes or str); returns the unconsumed tail (same type)."""
        data = sam.encode("latin-1") if isinstance(sam, str) else sam
        used = ctypes.c_int64(0)
        base = ctypes.cast(ctypes.c_char_p(data), ctypes.c_void_p).value
        rc = self._lib.clair_host_pileup_feed(self._h, base, len(data), 1 if final else 0, ctypes.byref(used))
        if rc != 0:
            from .create_tensor import PileupError
            raise PileupError(self._lib.clair_host_last_error().decode())
        return sam[used.value:]

    def finish(self):
        self._lib.clair_host_pileup_finish(self._h)

    def pending(self):
        return int(self._lib.clair_host_pileup_pending(self._h))

    def take_arrays(self, max_rows=None):
        """-> (centres int64 [n], refseq list of str, counts int32 [n,33,8,4])"""
        n = self.pending() if max_rows is None else min(self.pending(), int(max_rows))
        centres = np.empty(n, dtype=np.int64)
        seqs = np.zeros((n, 34), dtype=np.uint8)
        counts = np.empty((n, 33, 8, 4), dtype=np.int32)
        taken = ctypes.c_int64(0)
        if n:
            self._lib.clair_host_pileup_take(self._h, n, centres.ctypes.data, seqs.ctypes.data, counts.ctypes.data, ctypes.byref(taken))
        raw = seqs.tobytes()
        return centres, [raw[i * 34:i * 34 + 34].split(b"\0", 1)[0].decode("latin-1") for i in range(n)], counts

    def take_columns(self, max_rows=None):
        """take_arrays without a Python string per window: -> (centres int64 [n], refseq bytes uint8 [n,34] NUL-terminated,
        counts int32 [n,33,8,4])"""
        n = self.pending() if max_rows is None else min(self.pending(), int(max_rows))
        centres = np.empty(n, dtype=np.int64)
        seqs = np.zeros((n, 34), dtype=np.uint8)
        counts = np.empty((n, 33, 8, 4), dtype=np.int32)
        taken = ctypes.c_int64(0)
        if n:
            self._lib.clair_host_pileup_take(self._h, n, centres.ctypes.data, seqs.ctypes.data, counts.ctypes.data, ctypes.byref(taken))
        return centres, seqs, counts

    def take(self):
        centres, seqs, counts = self.take_arrays()
        return [(int(c), s, counts[i]) for i, (c, s) in enumerate(zip(centres, seqs))]

    def take_text(self, cap=1 << 24):
        """Finished windows as text records (bytes), as many as fit in `cap` bytes; b"" when none is pending."""
        if self._text is None or len(self._text) < cap:
            self._text = ctypes.create_string_buffer(cap)
        n, taken = ctypes.c_int64(0), ctypes.c_int64(0)
        self._lib.clair_host_pileup_take_text(self._h, self.ctg.encode(), self._text, cap, ctypes.byref(n), ctypes.byref(taken))
        return self._text.raw[:n.value]

    def stats(self):
        st = np.zeros(4, dtype=np.int64)
        self._lib.clair_host_pileup_stats(self._h, st.ctypes.data)
        return {"reads": int(st[0]), "open_windows": int(st[1]), "slots_left": int(st[2]), "sorted_path": bool(st[3])}

    def text_from_sam(self, handle, chunk_bytes=1 << 22):
        """Feed a SAM stream (binary or text file object); yield the finished records as text chunks (str)."""
        tail = None
        while True:
            chunk = handle.read(chunk_bytes)
            if not chunk:
                break
            tail = self.feed(chunk if tail is None else tail + chunk)
            while self.pending():
                yield self.take_text().decode("latin-1")
        if tail:
            self.feed(tail, final=True)
        self.finish()
        while self.pending():
            yield self.take_text().decode("latin-1")


class CandidateFinder(object):
    """clair_host_evc_*: the native twin of clair_amd.extract_variant_candidates.CandidateFinderPy."""

    def __init__(self, ctg_name, reference_sequence, reference_start_0_based, ctg_start=None, ctg_end=None, bed=None,
                 min_coverage=4, threshold=0.125, min_mq=0):
        self._lib = load()
        ref = reference_sequence.encode("latin-1") if isinstance(reference_sequence, str) else bytes(reference_sequence)
        have_range = ctg_start is not None and ctg_end is not None
        if bed is None:
            bs = be = np.zeros(0, dtype=np.int64)
            n_bed = -1
        else:
            bs = np.ascontiguousarray([b[0] for b in bed], dtype=np.int64)
            be = np.ascontiguousarray([b[1] for b in bed], dtype=np.int64)
            n_bed = len(bs)
        h = ctypes.c_void_p()
        rc = self._lib.clair_host_evc_create(ctg_name.encode(), ref, len(ref), int(reference_start_0_based),
                                             int(ctg_start) if have_range else -1, int(ctg_end) if have_range else -1,
                                             bs.ctypes.data, be.ctypes.data, n_bed, float(min_coverage), float(threshold),
                                             int(min_mq), ctypes.byref(h))
        if rc != 0:
            raise ValueError("candidates: " + self._lib.clair_host_last_error().decode())
        self._h = h
        self._text = None

    def close(self):
        if getattr(self, "_h", None):
            self._lib.clair_host_evc_destroy(self._h)
            self._h = None

    __del__ = close

    @property
    def reads(self):
        return int(self._lib.clair_host_evc_reads(self._h))

    def feed(self, sam, final=False):
        data = sam.encode("latin-1") if isinstance(sam, str) else sam
        used = ctypes.c_int64(0)
        base = ctypes.cast(ctypes.c_char_p(data), ctypes.c_void_p).value
        rc = self._lib.clair_host_evc_feed(self._h, base, len(data), 1 if final else 0, ctypes.byref(used))
        if rc != 0:
            from .create_tensor import PileupError
            raise PileupError(self._lib.clair_host_last_error().decode())
        return sam[used.value:]

    def finish(self):
        self._lib.clair_host_evc_finish(self._h)

    def pending(self):
        return int(self._lib.clair_host_evc_pending(self._h))

    def take_positions(self):
        n = self.pending()
        pos = np.empty(n, dtype=np.int64)
        taken = ctypes.c_int64(0)
        if n:
            self._lib.clair_host_evc_take(self._h, n, pos.ctypes.data, ctypes.byref(taken))
        return pos

    def take_text(self, cap=1 << 22):
        if self._text is None or len(self._text) < cap:
            self._text = ctypes.create_string_buffer(cap)
        n, taken = ctypes.c_int64(0), ctypes.c_int64(0)
        self._lib.clair_host_evc_take_text(self._h, self._text, cap, ctypes.byref(n), ctypes.byref(taken))
        return self._text.raw[:n.value]

    def text_from_sam(self, handle, chunk_bytes=1 << 22):
        tail = None
        while True:
            chunk = handle.read(chunk_bytes)
            if not chunk:
                break
            tail = self.feed(chunk if tail is None else tail + chunk)
            while self.pending():
                yield self.take_text().decode("latin-1")
        if tail:
            self.feed(tail, final=True)
        self.finish()
        while self.pending():
            yield self.take_text().decode("latin-1")


READ_DTYPE = np.dtype([("pos0", "<i8"), ("seq0", "<u4"), ("seq_len", "<u4"), ("op0", "<u4"), ("n_ops", "<u4"), ("flags", "<u4"), ("reserved", "<u4")])
OP_DTYPE = np.dtype([("read", "<u4"), ("code_len", "<u4"), ("ref_off", "<i4"), ("q_off", "<u4")])
READ_REVERSE, READ_EVC, READ_PILE, READ_FLUSH = 1, 2, 4, 8


class SamPacker(object):
    """clair_host_sampack_*: `samtools view` text -> slabs of packed alignments (include/clair_reads.h) for the device front end."""

    def __init__(self, ctg_name, dcov=250, evc_min_mq=0, pile_min_mq=0, pile_region=None):
        self._lib = load()
        h = ctypes.c_void_p()
        a, b = (-1, -1) if pile_region is None else (int(pile_region[0]), int(pile_region[1]))
        if self._lib.clair_host_sampack_create(ctg_name.encode(), int(dcov), int(evc_min_mq), int(pile_min_mq), a, b, ctypes.byref(h)) != 0:
            raise ValueError("sampack: " + self._lib.clair_host_last_error().decode())
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._lib.clair_host_sampack_destroy(self._h)
            self._h = None

    __del__ = close

    def feed(self, sam, final=False):
        """Consume complete lines of `sam` (bytes); returns the unconsumed tail."""
        used = ctypes.c_int64(0)
        base = ctypes.cast(ctypes.c_char_p(sam), ctypes.c_void_p).value
        if self._lib.clair_host_sampack_feed(self._h, base, len(sam), 1 if final else 0, ctypes.byref(used)) != 0:
            from .create_tensor import PileupError
            raise PileupError(self._lib.clair_host_last_error().decode())
        return sam[used.value:]

    def stats(self):
        v = (ctypes.c_int64 * 8)()
        self._lib.clair_host_sampack_stats(self._h, v)
        return dict(zip(("reads", "ops", "elements", "seq_bytes", "anomalies", "lines", "evc_reads", "pile_reads"), [int(x) for x in v]))

    def slab_pointers(self):
        """-> (reads, ops, op_elem, seq) addresses of the slab being filled + its stats; valid until the next feed() / reset()."""
        p = [ctypes.c_void_p() for _ in range(4)]
        self._lib.clair_host_sampack_slab(self._h, *[ctypes.byref(x) for x in p])
        return [x.value or 0 for x in p], self.stats()

    def slab_arrays(self):
        """Copies of the slab as NumPy arrays (tests, the budget replay): reads READ_DTYPE, ops OP_DTYPE, op_elem uint32, seq uint8."""
        (r, o, e, q), st = self.slab_pointers()

        def view(addr, n, dtype):
            if n == 0:
                return np.zeros(0, dtype)
            return np.frombuffer((ctypes.c_char * (n * np.dtype(dtype).itemsize)).from_address(addr), dtype=dtype).copy()
        return view(r, st["reads"], READ_DTYPE), view(o, st["ops"], OP_DTYPE), view(e, st["ops"] + 1, np.uint32), view(q, st["seq_bytes"], np.uint8)

    def reset(self):
        self._lib.clair_host_sampack_reset(self._h)


def tuple_budget_binds(reads, tuples, centres, window_tuples, state):
    """clair_host_tuple_budget_binds over one slab; state = int64[2] carried between slabs ([free slots, first unreleased centre])."""
    reads = np.ascontiguousarray(reads, dtype=READ_DTYPE)
    tuples = np.ascontiguousarray(tuples, dtype=np.uint64)
    centres = np.ascontiguousarray(centres, dtype=np.int64)
    window_tuples = np.ascontiguousarray(window_tuples, dtype=np.uint64)
    binds = ctypes.c_int(0)
    if load().clair_host_tuple_budget_binds(reads.ctypes.data, tuples.ctypes.data, len(reads), centres.ctypes.data, window_tuples.ctypes.data,
                                            len(centres), state.ctypes.data, ctypes.byref(binds)) != 0:
        raise ValueError(load().clair_host_last_error().decode())
    return bool(binds.value)
