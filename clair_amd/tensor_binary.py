"""Binary form of the tensor records: fixed 2 192-byte records instead of ~2.3 KB of decimal text.

The reference's producers and consumers exchange `ctg pos refseq v0 ... v1055` text lines (dataPrepScripts/CreateTensor.py:60-65
-> clair/utils.py:72-109).  This is the same content as little-endian structs, for the cases where the records still cross a
process or file boundary (`create_tensor --binary | call_var`, tensor files kept on disk); `call_var` recognises it by its
first eight bytes, through `gzip -fdc` like the text.  Counts are the RAW pileup counts (before utils.py:96-98 subtracts channel
0): the reader performs the subtraction, and hands the int16 counts on so that the GPU boundary can take them as they are.

    file   := MAGIC record*
    record := pos i64 | seq_len u8 | seq 33 bytes | ctg_len u8 | ctg 37 bytes | counts int16[33][8][4]
"""
import os
import sys

import numpy as np

MAGIC = b"CLAIRT\x01\n"
MAX_CTG = 37
RECORD = np.dtype([("pos", "<i8"), ("seq_len", "u1"), ("seq", "S33"), ("ctg_len", "u1"), ("ctg", "S%d" % MAX_CTG),
                   ("counts", "<i2", (33, 8, 4))])
assert RECORD.itemsize == 2192
IUPAC = frozenset(b"ACGTURYSWKMBDHVN")


def pack_records(ctg_name, centres, seqs, counts):
    """-> bytes of len(centres) records.  counts: integer array [n,33,8,4]; values outside int16 cannot be stored."""
    ctg = ctg_name.encode()
    if len(ctg) > MAX_CTG:
        raise ValueError("contig name %r is longer than %d bytes: use the text records" % (ctg_name, MAX_CTG))
    counts = np.asarray(counts)
    if counts.size and (int(counts.max()) > 32767 or int(counts.min()) < -32768):
        raise ValueError("a pileup count does not fit int16: use the text records")
    rec = np.zeros(len(centres), dtype=RECORD)
    rec["pos"] = centres
    rec["seq_len"] = [len(s) for s in seqs]
    rec["seq"] = [s.encode() for s in seqs]
    rec["ctg_len"] = len(ctg)
    rec["ctg"] = ctg
    rec["counts"] = counts
    return rec.tobytes()


class InfoTable(object):
    """The [[ctg, pos, seq], ...] list of one batch, kept as the record columns it came from.  Behaves like that list (len,
    iteration, integer and slice indexing give the same lists of three strings) and lets the native decoder take the columns as
    they are (`native_meta`) instead of 12 000 Python strings per batch that it would only join again."""

    def __init__(self, ctg, ctg_len, pos, seq, seq_len):
        self.ctg, self.ctg_len, self.pos, self.seq, self.seq_len = ctg, ctg_len, pos, seq, seq_len
        self._rows = None
        self._meta = None

    def __len__(self):
        return len(self.pos)

    def rows(self):
        if self._rows is None:
            self._rows = [[c[:cl].decode(), str(p), s[:sl].decode()] for c, cl, p, s, sl in
                          zip(self.ctg.tolist(), self.ctg_len.tolist(), self.pos.tolist(), self.seq.tolist(), self.seq_len.tolist())]
        return self._rows

    def __getitem__(self, i):
        return self.rows()[i]

    def __iter__(self):
        return iter(self.rows())

    def __eq__(self, other):
        return self.rows() == (other.rows() if isinstance(other, InfoTable) else other)

    def native_meta(self):
        """-> (meta bytes, tok int32 [n,6]) in the form clair_host_decode_rows takes: per candidate (offset, length) of contig,
        decimal position and reference sequence inside `meta`, here fixed-width rows of 37 + 20 + 33 bytes."""
        if self._meta is not None:
            return self._meta
        n = len(self)
        width = MAX_CTG + 20 + 33
        meta = np.zeros((n, width), dtype=np.uint8)
        pos = self.pos.astype(np.int64)
        if n and int(pos.min()) < 0:                  # never produced by this package; formatted the slow way
            ptxt = np.char.encode(pos.astype("U20"), "ascii").astype("S20")
            meta[:, MAX_CTG:MAX_CTG + 20] = np.frombuffer(ptxt.tobytes(), dtype=np.uint8).reshape(n, 20)
            plen, pstart = np.char.str_len(ptxt), np.zeros(n, dtype=np.int64)
        else:                                         # decimal digits right-aligned in the 20-byte field, 19 divisions for the whole batch
            rest, digits = pos.copy(), meta[:, MAX_CTG:MAX_CTG + 20]
            for k in range(19, -1, -1):
                rest, d = np.divmod(rest, 10)
                digits[:, k] = d + 48
            plen = np.maximum(1, 20 - (digits != 48).argmax(axis=1)) if n else np.zeros(0, np.int64)
            plen = np.where(pos == 0, 1, plen)
            pstart = 20 - plen
        meta[:, :MAX_CTG] = np.frombuffer(np.ascontiguousarray(self.ctg).tobytes(), dtype=np.uint8).reshape(n, MAX_CTG)
        meta[:, MAX_CTG + 20:] = np.frombuffer(np.ascontiguousarray(self.seq).tobytes(), dtype=np.uint8).reshape(n, 33)
        tok = np.empty((n, 6), dtype=np.int32)
        base = np.arange(n, dtype=np.int64) * width
        tok[:, 0] = base
        tok[:, 1] = self.ctg_len
        tok[:, 2] = base + MAX_CTG + pstart
        tok[:, 3] = plen
        tok[:, 4] = base + MAX_CTG + 20
        tok[:, 5] = self.seq_len
        self._meta = (meta.tobytes(), tok)
        return self._meta

    def record_columns(self):
        """(contig S37 column, its lengths, positions int64, refseq S33 column, its lengths) as contiguous arrays: what
        clair_host_format_calls_records takes."""
        return (np.ascontiguousarray(self.ctg), np.ascontiguousarray(self.ctg_len, dtype=np.uint8), np.ascontiguousarray(self.pos, dtype=np.int64),
                np.ascontiguousarray(self.seq), np.ascontiguousarray(self.seq_len, dtype=np.uint8))

    def centre_bytes(self):
        """uint8 [n,2]: the centre character of each reference window and its length -- what the device decode takes of the text
        (include/clair_amd.h: clair_submit_ex).  Every kept record has more than 16 characters (read_batches filters on it)."""
        n = len(self)
        out = np.empty((n, 2), dtype=np.uint8)
        out[:, 0] = np.frombuffer(np.ascontiguousarray(self.seq).tobytes(), dtype=np.uint8).reshape(n, 33)[:, 16] if n else 0
        out[:, 1] = self.seq_len
        return out


_IUPAC_TABLE = np.zeros(256, dtype=bool)
_IUPAC_TABLE[list(IUPAC)] = True


class BufferPool(object):
    """Record buffers a reader fills and a consumer hands back: uint8 arrays of batch_size x 2 192 bytes, page-locked when they come
    from the GPU engine (clair_amd._capi.Engine.pinned_buffer), so that a batch read into one goes to the device from where it lies.
    get() blocks until a buffer is free, or raises Closed once the consumer has given up."""

    class Closed(Exception):
        pass

    def __init__(self, arrays):
        import queue
        self._q = queue.Queue()
        self._closed = False
        for a in arrays:
            self._q.put(a)

    def get(self):
        import queue
        while True:
            try:
                return self._q.get(timeout=0.1)
            except queue.Empty:
                if self._closed:
                    raise BufferPool.Closed()

    def put(self, a):
        self._q.put(a)

    def close(self):
        self._closed = True


def _fill(stream, view):
    """Read into `view` (a writable memoryview) until it is full or the stream ends; -> bytes read."""
    have, want = 0, len(view)
    readinto = getattr(stream, "readinto", None)
    while have < want:
        if readinto is not None:
            k = readinto(view[have:])
            if not k:
                break
        else:
            data = stream.read(want - have)
            if not data:
                break
            k = len(data)
            view[have:have + k] = data
        have += k
    return have


def _records_of(buf, have, want):
    """The batch that `have` bytes at the head of `buf` hold -> (infos, counts view, records kept) or None for an empty one."""
    take = have // RECORD.itemsize * RECORD.itemsize
    if take < have:
        raise ValueError("truncated binary tensor record (%d trailing bytes)" % (have - take))
    if take == 0:
        return None
    rec = buf[:take].view(RECORD)
    seq_bytes = np.frombuffer(np.ascontiguousarray(rec["seq"]).tobytes(), dtype=np.uint8).reshape(len(rec), 33)
    keep = (rec["seq_len"] > 16) & _IUPAC_TABLE[seq_bytes[:, 16]]
    if not keep.all():
        rec = rec[keep]                           # (a copy outside the buffer: that batch goes through the staging path)
    if len(rec) == 0:
        return (None, None, 0)
    return InfoTable(rec["ctg"].copy(), rec["ctg_len"].copy(), rec["pos"].copy(), rec["seq"].copy(), rec["seq_len"].copy()), rec["counts"], len(rec)


def _regular_file(stream):
    import stat
    try:
        fd = stream.fileno()
        return fd if stat.S_ISREG(os.fstat(fd).st_mode) and hasattr(os, "preadv") else None
    except (AttributeError, OSError, ValueError):
        return None


def read_batches_into(stream, batch_size, pool, readers=3):
    """read_batches(with_input=False) that reads every batch of records straight into a buffer of `pool` (BufferPool) -- no copy of the
    batch is made on the host at all -- and yields (None, infos, counts view, buffer); the consumer gives the buffer back
    (pool.put) once the GPU has taken the batch.  Same batching rules, same progress lines.

    A regular file is read by `readers` threads at once, batch k at its own offset (records are fixed-size): one thread moves 9 MB out of
    the page cache in ~0.7 ms, which alone caps a 4096-batch pipeline at 6 M candidates/s (round 4).  Batches are yielded in file order."""
    processed = 0
    want = batch_size * RECORD.itemsize
    fd = _regular_file(stream) if readers > 1 else None
    if fd is None:
        while True:
            try:
                buf = pool.get()
            except BufferPool.Closed:
                return
            have = _fill(stream, memoryview(buf)[:want])
            got = _records_of(buf, have, want)
            if got is None:
                pool.put(buf)
                return
            infos, counts, n = got
            processed += n
            print("Processed %d tensors" % processed, file=sys.stderr)
            if n == 0:
                pool.put(buf)
            else:
                yield None, infos, counts, buf
            if have < want:
                return
        return
    from concurrent.futures import ThreadPoolExecutor
    start = stream.tell()
    size = os.fstat(fd).st_size
    n_batches = max(0, (size - start + want - 1) // want)

    def load(k):
        buf = pool.get()                          # (raises BufferPool.Closed when the consumer has given up)
        view, have, off = memoryview(buf)[:want], 0, start + k * want
        while have < want:
            got = os.preadv(fd, [view[have:]], off + have)
            if got <= 0:
                break
            have += got
        try:
            return buf, have, _records_of(buf, have, want)
        except Exception:
            pool.put(buf)
            raise

    with ThreadPoolExecutor(max_workers=readers) as ex:
        ahead, k = [], 0
        try:
            while k < n_batches or ahead:
                while k < n_batches and len(ahead) < readers:
                    ahead.append(ex.submit(load, k))
                    k += 1
                try:
                    buf, have, got = ahead.pop(0).result()
                except BufferPool.Closed:
                    return
                if got is None:
                    pool.put(buf)
                    return
                infos, counts, n = got
                processed += n
                print("Processed %d tensors" % processed, file=sys.stderr)
                if n == 0:
                    pool.put(buf)
                else:
                    yield None, infos, counts, buf
        finally:
            for f in ahead:                       # a consumer that stopped early: the buffers in flight go back
                try:
                    pool.put(f.result()[0])
                except Exception:
                    pass


def read_batches(stream, batch_size, first=b"", with_input=True):
    """Yield (X float32 [n,33,8,4], infos, counts int16 [n,33,8,4]) from a binary record stream positioned after MAGIC.
    Batching follows clair/utils.py:72-109: batch_size records are TAKEN per batch, those whose centre base is not an IUPAC
    code are dropped from it, empty batches are skipped, progress goes to stderr.  infos is an InfoTable (list-like).
    with_input=False: X is None and counts is the records' own column (a strided view, no copy) -- for a consumer that sends the
    raw counts to the GPU and decodes there (clair_amd.call_var with the device decode): the batch costs no pass over its 8.6 MB."""
    processed = 0
    want = batch_size * RECORD.itemsize
    carry = first
    eof = False
    while not eof or carry:
        pieces, have = [carry] if carry else [], len(carry)
        carry = b""
        while not eof and have < want:
            more = stream.read(want - have)
            if more:
                pieces.append(more)
                have += len(more)
            else:
                eof = True
        pending = pieces[0] if len(pieces) == 1 else b"".join(pieces)
        if len(pending) > want:                      # only a caller-supplied `first` can be longer than one batch
            pending, carry = pending[:want], pending[want:]
        take = len(pending) // RECORD.itemsize * RECORD.itemsize
        if take == 0:
            if pending:
                raise ValueError("truncated binary tensor record (%d trailing bytes)" % len(pending))
            break
        if take < len(pending):
            if eof and not carry:
                raise ValueError("truncated binary tensor record (%d trailing bytes)" % (len(pending) - take))
            carry = pending[take:] + carry
        rec = np.frombuffer(pending, dtype=RECORD, count=take // RECORD.itemsize)
        seq_bytes = np.frombuffer(np.ascontiguousarray(rec["seq"]).tobytes(), dtype=np.uint8).reshape(len(rec), 33)
        # a stored sequence shorter than 17 characters has no centre base (S33 fields are NUL padded: NUL is not an IUPAC code)
        keep = (rec["seq_len"] > 16) & _IUPAC_TABLE[seq_bytes[:, 16]]
        if not keep.all():
            rec = rec[keep]
        n = len(rec)
        processed += n
        print("Processed %d tensors" % processed, file=sys.stderr)
        if n == 0:
            continue
        if with_input:
            counts = np.ascontiguousarray(rec["counts"])
            from clair_amd import _hostapi
            x = _hostapi.counts_to_input(counts)
        else:
            counts, x = rec["counts"], None
        yield x, InfoTable(rec["ctg"].copy(), rec["ctg_len"].copy(), rec["pos"].copy(), rec["seq"].copy(), rec["seq_len"].copy()), counts
