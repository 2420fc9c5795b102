"""Binary form of the tensor records: fixed 2 192-byte records instead of ~2.3 KB of decimal text.

The reference's producers and consumers exchange `ctg pos refseq v0 ... v1055` text lines (dataPrepScripts/CreateTensor.py:60-65
-> clair/utils.py:72-109).  This is the same content as little-endian structs, for the cases where the records still cross a
process or file boundary (`create_tensor --binary | call_var`, tensor files kept on disk); `call_var` recognises it by its
first eight bytes, through `gzip -fdc` like the text.  Counts are the RAW pileup counts (before utils.py:96-98 subtracts channel
0): the reader performs the subtraction, and hands the int16 counts on so that the GPU boundary can take them as they are.

    file   := MAGIC record*
    record := pos i64 | seq_len u8 | seq 33 bytes | ctg_len u8 | ctg 37 bytes | counts int16[33][8][4]
"""
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


def read_batches(stream, batch_size, first=b""):
    """Yield (X float32 [n,33,8,4], infos, counts int16 [n,33,8,4]) from a binary record stream positioned after MAGIC.
    Batching follows clair/utils.py:72-109: batch_size records are TAKEN per batch, those whose centre base is not an IUPAC
    code are dropped from it, empty batches are skipped, progress goes to stderr."""
    processed = 0
    pending = first
    want = batch_size * RECORD.itemsize
    eof = False
    while not eof or pending:
        while not eof and len(pending) < want:
            more = stream.read(want - len(pending))
            if more:
                pending += more
            else:
                eof = True
        take = min(len(pending), want) // RECORD.itemsize * RECORD.itemsize
        if take == 0:
            if pending:
                raise ValueError("truncated binary tensor record (%d trailing bytes)" % len(pending))
            break
        rec = np.frombuffer(pending[:take], dtype=RECORD)
        pending = pending[take:]
        seq = rec["seq"]
        keep = np.fromiter((rec["seq_len"][i] > 16 and len(s) > 16 and s[16] in IUPAC for i, s in enumerate(seq)), dtype=bool,
                           count=len(rec))
        rec = rec[keep] if not keep.all() else rec
        n = len(rec)
        processed += n
        print("Processed %d tensors" % processed, file=sys.stderr)
        if n == 0:
            continue
        counts = np.ascontiguousarray(rec["counts"])
        x = counts.astype(np.float32)
        x[:, :, :, 1:] -= x[:, :, :, 0:1]
        infos = [[c[:cl].decode(), str(p), s[:sl].decode()] for c, cl, p, s, sl in
                 zip(rec["ctg"].tolist(), rec["ctg_len"].tolist(), rec["pos"].tolist(), rec["seq"].tolist(), rec["seq_len"].tolist())]
        yield x, infos, counts
