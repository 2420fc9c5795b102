"""Tensor ingest: text records -> float32 batches [n,33,8,4] for Clair.predict.

Counterpart of /root/reference/clair/utils.py:55-109 (``batches_from``,
``tensor_generator_from``).  Record format (dataPrepScripts/CreateTensor.py:60-65):

    ctg pos refseq33 v0 v1 ... v1055        (whitespace separated, %d)

Behaviour kept identical to the reference (pinned by tests/golden/ingest_* minted from it):
  * source is ``gzip -fdc <file>`` or stdin when the path is "PIPE"               utils.py:72-77
  * the last 1056 columns become float32; the columns before must be exactly 3   utils.py:81-88
  * rows whose centre base refseq[16] is not an IUPAC code are dropped           utils.py:90-91
  * channels 1..3 get channel 0 subtracted                                       utils.py:96-98
  * "Processed %d tensors" goes to stderr once per chunk, including the final
    (possibly empty) chunk; empty chunks are not yielded                         utils.py:100-104
  * a partial last batch is yielded as X[:n]                                     utils.py:105
"""
import os
import shlex
import sys
from subprocess import PIPE, Popen

import numpy as np

from clair_amd import param
from clair_amd.task import IUPAC_TO_NUM

N_POS = param.no_of_positions
N_ROW = param.matrixRow
N_CH = param.matrixNum
N_VALUES = param.input_tensor_size
CENTER = param.flankingBaseNum


def setup_environment():
    """Reference: clair/utils.py:39-44 (TF log level, blosc threads).  Nothing to set up here."""
    return None


class _InflateReader(object):
    """read(n) over a gzip file, inflated in-process (zlib, which releases the GIL): 340 MB/s of text where the `gzip -dc` child
    delivers 180.  Members are concatenated and bytes behind the last member passed through as `gzip -fdc` does; a truncated file ends the stream
    with gzip's complaint on stderr."""

    def __init__(self, path):
        import zlib
        self._zlib = zlib
        self._f = open(path, "rb", buffering=0)
        self._d = zlib.decompressobj(16 + zlib.MAX_WBITS)
        self._path = path
        self._fresh = True          # no byte of the current member seen yet
        self._done = False
        self._raw = False           # past the last member: the rest of the file is passed through

    def read(self, n=-1):
        want = n if n is not None and n > 0 else 1 << 23
        out = []
        have = 0
        while have < want and not self._done:
            if self._raw:                                     # bytes behind the last member that are not a member: `gzip -f` copies them through
                more = self._f.read(1 << 20)
                if not more:
                    self._done = True
                    break
                out.append(more)
                have += len(more)
                continue
            if self._d.eof:                                   # the next member, if the bytes behind this one are one
                rest = self._d.unused_data
                if len(rest) < 2:
                    rest += self._f.read(2 - len(rest))
                self._d = self._zlib.decompressobj(16 + self._zlib.MAX_WBITS)
                self._fresh = True
                if rest:
                    if rest[:2] != b"\x1f\x8b":
                        self._raw = True
                        out.append(rest)
                        have += len(rest)
                        continue
                    chunk = self._d.decompress(rest, want - have)
                    self._fresh = False
                    if chunk:
                        out.append(chunk)
                        have += len(chunk)
                    continue
            pending = self._d.unconsumed_tail
            raw = pending if pending else self._f.read(1 << 20)
            if not raw:
                if not self._fresh and not self._d.eof:
                    sys.stderr.write("gzip: %s: unexpected end of file\n" % self._path)
                self._done = True
                break
            self._fresh = False
            chunk = self._d.decompress(raw, want - have)
            if chunk:
                out.append(chunk)
                have += len(chunk)
        return b"".join(out)

    def close(self):
        self._f.close()


def _open_source(tensor_file_path, binary=False):
    if tensor_file_path == "PIPE":
        return None, (sys.stdin.buffer if binary else sys.stdin)
    if binary and os.path.isfile(tensor_file_path):
        # `gzip -fdc` hands a file that is in none of the formats it knows through unchanged -- at ~400 MB/s of pipe, which bounds
        # an uncompressed tensor file at 160 k candidates/s.  Such a file (not starting with gzip's 0x1f lead byte of the gzip /
        # compress / pack / lzh formats, nor with pkzip's "PK") is opened directly: the same bytes.
        with open(tensor_file_path, "rb") as f:
            lead = f.read(2)
        if lead and lead[:1] != b"\x1f" and lead != b"PK":
            return None, open(tensor_file_path, "rb", buffering=8388608)
        if lead == b"\x1f\x8b":
            return None, _InflateReader(tensor_file_path)
    proc = Popen(shlex.split("gzip -fdc %s" % tensor_file_path), stdout=PIPE,
                 bufsize=8388608, universal_newlines=not binary)
    return proc, proc.stdout


def tensor_generator_from(tensor_file_path, batch_size, with_input=True, record_buffers=None):
    """Yield (X float32 [n,33,8,4], infos [[ctg, pos, seq], ...]) with n <= batch_size.  (with_input=False: binary records may
    leave X as None and hand their raw counts on as a third element instead, tensor_binary.read_batches; record_buffers: a
    tensor_binary.BufferPool the binary records are read INTO, a fourth element names the buffer to give back.)

    The text is parsed by the native helper (include/clair_host.h: clair_host_parse_tensors, ~20x the NumPy path below);
    `tensor_generator_from_py` is the line-by-line restatement of the reference it is tested against."""
    from clair_amd import _hostapi, tensor_binary
    import queue
    import threading
    proc, stream = _open_source(tensor_file_path, binary=True)
    head = stream.read(len(tensor_binary.MAGIC))
    if head == tensor_binary.MAGIC:              # fixed-size binary records (clair_amd/tensor_binary.py) instead of text
        if record_buffers is not None and not with_input:      # straight into the consumer's (page-locked) buffers
            batches = tensor_binary.read_batches_into(stream, batch_size, record_buffers)
        else:
            batches = tensor_binary.read_batches(stream, batch_size, with_input=with_input)
        for batch in batches:
            yield batch
        if proc is not None:
            stream.close()
            proc.wait()
        elif tensor_file_path != "PIPE":
            stream.close()
        return
    chunks = queue.Queue(maxsize=4)
    if head:
        chunks.put(head)

    def reader():                                # decompressed text arrives while the previous chunk is being parsed
        while True:
            more = stream.read(1 << 23)
            chunks.put(more)
            if not more:
                return

    threading.Thread(target=reader, daemon=True).start()
    processed = 0
    buf, off, lines_ahead = b"", 0, 0            # unparsed text = buf[off:], holding `lines_ahead` newline characters
    eof = exhausted = False
    while not exhausted:
        flat = np.empty((batch_size, N_VALUES), dtype=np.float32)
        pieces, kept = [], 0                     # MetaInfoTable pieces of this batch: the fields stay bytes until someone asks for strings
        taken = 0
        while taken < batch_size:
            need = batch_size - taken
            while not eof and lines_ahead < need:
                more = chunks.get()
                if more:
                    buf, off = buf[off:] + more, 0
                    lines_ahead += more.count(b"\n")
                else:
                    eof = True
            if off >= len(buf):                  # the reference's readline() returned '' (utils.py:75-77)
                exhausted = True
                break
            t, inf, used = _hostapi.parse_tensors(buf, eof, need, flat, kept, off)
            taken += t
            pieces.append(inf)
            kept += len(inf)
            off += used
            lines_ahead -= t
        n = kept
        processed += n
        print("Processed %d tensors" % processed, file=sys.stderr)
        if n > 0:
            yield flat.reshape(batch_size, N_POS, N_ROW, N_CH)[:n], _hostapi.MetaInfoTable.concat(pieces)
    if proc is not None:
        stream.close()
        proc.wait()
    elif tensor_file_path != "PIPE":
        stream.close()


def tensor_generator_from_py(tensor_file_path, batch_size):
    """The reference's line-by-line NumPy ingest (clair/utils.py:72-109), kept as the checker of the native parser."""
    proc, lines = _open_source(tensor_file_path)
    processed = 0
    exhausted = False
    while not exhausted:
        flat = np.empty((batch_size, N_VALUES), dtype=np.float32)
        infos = []
        taken = 0
        while taken < batch_size:
            row = lines.readline()
            if not row:
                exhausted = True
                break
            taken += 1
            cols = row.split()
            head = cols[:-N_VALUES]
            values = np.array(cols[-N_VALUES:], dtype=np.float32)
            _, _, seq = head                      # exactly three leading columns, as the reference requires
            if seq[CENTER] not in IUPAC_TO_NUM:
                continue
            flat[len(infos)] = values
            infos.append(head)
        n = len(infos)
        X = flat.reshape(batch_size, N_POS, N_ROW, N_CH)
        X[:n, :, :, 1:] -= X[:n, :, :, 0:1]
        processed += n
        print("Processed %d tensors" % processed, file=sys.stderr)
        if n > 0:
            yield X[:n], infos
    if proc is not None:
        lines.close()
        proc.wait()
