"""BAM -> VCF for one contig / region: candidate extraction, pileup tensors, network, decode, in ONE process.

Host-side mirror of the reference's clair/callVarBam.py (same flags).  The reference wires three interpreters together with
text pipes (callVarBam.py:124-199): `pypy ExtractVariantCandidates | pypy CreateTensor | python call_var`, 2.3 KB of decimal
text per candidate on the second pipe.  Here the same three stages are native code in one process and hand each other
arrays:

    samtools view  -> clair_host_evc_*     -> candidate positions (int64)
    samtools view  -> clair_host_pileup_*  -> count windows int32 [n,33,8,4] + (position, refseq)
                   -> float32, channels 1..3 minus channel 0 (clair/utils.py:96-98), centre-base filter (:90-91)
                   -> HIP forward pass (libclair_amd.so) -> native decode -> VCF rows

The VCF is the one `extract_variant_candidates | create_tensor | call_var` produce through their text interfaces
(tests/test_pileup.py pins that), which in turn are pinned against the reference scripts.  Like the reference, alignments
are read twice (`samtools view` once per stage): the pileup needs the candidates ahead of the reads.

--front_end device (the default where it applies) moves both stages to the GPU (include/clair_amd.h: clair_frontend_*):

    samtools view  -> clair_host_sampack_* (ONE pass over the text) -> packed alignments, resident in HBM
                   -> per-position tables -> candidate filter -> windows int16 [n,33,8,4] that never leave the device
                   -> HIP forward pass + decode kernel -> call records -> VCF rows

Same candidates, same windows, same VCF (tests/test_frontend_gpu.py, tests/test_e2e_gpu.py); where the input leaves the regime
that formulation reproduces exactly (CLAIR_FE_* in include/clair_reads.h, a tuple budget that binds) the run falls back to the
host stages above and says so.
"""
import logging
import os
import shlex
import sys
from argparse import ArgumentParser

import numpy as np

from . import call_var as cv
from . import create_tensor as ct
from . import extract_variant_candidates as evc
from . import param

IUPAC = frozenset("ACGTURYSWKMBDHVN")


def view_command(args, region):
    """`samtools view -F 2316 <bam> <region>` as the reference spawns it (CreateTensor.py:163-170), plus -@ N with --samtools_threads N
    (BGZF blocks inflated on N extra threads: the text that comes out is the same)."""
    threads = getattr(args, "samtools_threads", 0) or 0
    extra = getattr(args, "samtools_view_args", None) or ""
    return shlex.split("%s view %s%s-F %d %s %s" % (args.samtools, "-@ %d " % threads if threads > 0 else "", extra + " " if extra else "",
                                                  ct.SAMTOOLS_VIEW_FILTER_FLAG, args.bam_fn, region))


def candidate_positions(args, quiet=False):
    """Stage 1.  -> int64 positions (1-based, ascending for sorted alignments)."""
    from . import _hostapi
    if args.vcf_fn is not None:
        return positions_from_vcf(args.vcf_fn, args.ctgName, args.ctgStart, args.ctgEnd)
    if not os.path.isfile("%s.fai" % args.ref_fn):
        sys.exit("Fasta index %s.fai doesn't exist." % args.ref_fn)
    have_range = args.ctgStart is not None and args.ctgEnd is not None
    region, ref_start = evc.reference_region(args.ctgName, args.ctgStart if have_range else None, args.ctgEnd if have_range else None)
    seq = evc.load_reference(args.samtools, args.ref_fn, region)
    if not seq:
        sys.exit("[ERROR] Failed to load reference seqeunce from file (%s)." % args.ref_fn)
    tree = evc.bed_regions_from(args.bed_fn)
    if tree is not None and args.ctgName not in tree:
        sys.exit("[ERROR] ctg_name(%s) not exists in bed file(%s)." % (args.ctgName, args.bed_fn))
    finder = _hostapi.CandidateFinder(args.ctgName, seq, 0 if ref_start is None else ref_start - 1,
                                      ctg_start=args.ctgStart if have_range else None, ctg_end=args.ctgEnd if have_range else None,
                                      bed=None if tree is None else tree[args.ctgName],
                                      min_coverage=int(args.minCoverage), threshold=args.threshold, min_mq=0)   # callVarBam.py:75: int() before it reaches the extractor
    view = ct.subprocess_popen(view_command(args, region),
                               text=False)
    chunks, tail = [], None
    while True:
        chunk = view.stdout.read(1 << 22)
        if not chunk:
            break
        tail = finder.feed(chunk if tail is None else tail + chunk)
        if finder.pending():
            chunks.append(finder.take_positions())
    if tail:
        finder.feed(tail, final=True)
    finder.finish()
    chunks.append(finder.take_positions())
    view.stdout.close()
    view.wait()
    if view.returncode != 0:
        sys.exit("[ERROR] `samtools view` failed on %s" % args.bam_fn)
    if finder.reads == 0 and not quiet:
        print("No read has been process, either the genome region you specified has no read cover, or please check the correctness of your BAM input (%s)."
              % args.bam_fn, file=sys.stderr)
    return np.concatenate(chunks) if chunks else np.zeros(0, np.int64)


def positions_from_vcf(vcf_fn, ctg_name, ctg_start, ctg_end):
    """--vcf_fn: call only at the sites of a VCF.  The reference pipes dataPrepScripts/GetTruth.py into CreateTensor, which
    reads column 2 of its rows; this is that column, in GetTruth's order (GetTruth.py:75-134): every record of the contig inside
    the range, one row per run of equal positions (:127-133); a '*' alternate adds the base before the record -- first when
    '*' is one of the first two alternates, after it otherwise (:30-48), so the stream need not be sorted (the pileup handles that
    as the reference does)."""
    have_range = ctg_start is not None and ctg_end is not None
    p = ct.subprocess_popen(shlex.split("gzip -fdc %s" % vcf_fn))
    out = []
    for row in p.stdout:
        col = row.strip().split()
        if not col or col[0][0] == "#" or col[0] != ctg_name:
            continue
        pos = int(col[1])
        if have_range and not ctg_start <= pos <= ctg_end:
            continue
        alts = col[4].split(",") if "*" in col[4] else [col[4]]
        if "*" in col[4]:
            if len(alts) < 2:
                sys.exit("[ERROR] %s: a lone '*' alternate at %s:%d (the reference's GetTruth cannot read it either)" % (vcf_fn, ctg_name, pos))
            if alts[1] == "*":
                alts = ["*", col[4][0]]
        for alt in alts:
            site = pos - 1 if alt == "*" else pos
            if not out or out[-1] != site:
                out.append(site)
    p.stdout.close()
    p.wait()
    return np.array(out, dtype=np.int64)


def tensor_batches(args, positions, batch_size, read_flank=(0, 0), progress=True):
    """Stage 2 as a generator of (X float32 [n,33,8,4], [[ctg, pos, refseq], ...]) -- what clair_amd.utils.tensor_generator_from
    yields for the text records of the same windows.  read_flank = (left, right) widens the region the ALIGNMENTS are taken from
    (not the candidates): a sub-range of a larger run needs the reads that touch only the flanks of its outermost windows."""
    from . import _hostapi
    seq, ref_start = ct.reference_sequence_from(args.samtools, args.ref_fn, args.ctgName, args.ctgStart, args.ctgEnd)
    if not seq:
        sys.exit("Failed to load reference seqeunce. Please check if the provided reference fasta %s and the ctgName %s are correct."
                 % (args.ref_fn, args.ctgName))
    have_range = args.ctgStart is not None and args.ctgEnd is not None
    if have_range:
        positions = positions[(positions >= args.ctgStart) & (positions <= args.ctgEnd)]
    builder = _hostapi.PileupBuilder(args.ctgName, seq, 0 if ref_start is None else ref_start - 1, positions,
                                     consider_left_edge=not args.stop_consider_left_edge, dcov=args.dcov, set_order=ct.set_order_of(getattr(args, "pypy", None)))
    region = "%s:%d-%d" % (args.ctgName, max(1, args.ctgStart - read_flank[0]), args.ctgEnd + read_flank[1]) if have_range else args.ctgName
    view = ct.subprocess_popen(view_command(args, region),
                               text=False)
    from .tensor_binary import MAX_CTG, InfoTable, _IUPAC_TABLE
    total = 0
    ctg_bytes = args.ctgName.encode()
    use_table = len(ctg_bytes) <= MAX_CTG           # the column form of [[ctg, pos, refseq], ...] (no Python string per window)
    held = [[], [], []]                             # centres, refseq bytes [n,34], counts of windows not yet handed out
    held_n = 0

    def emit(centres, seqs, counts):
        nonlocal total
        n = len(centres)
        total += n
        if progress:
            print("Processed %d tensors" % total, file=sys.stderr)
        if use_table:
            seq_col = np.ascontiguousarray(seqs[:, :33]).view("S33").ravel()
            infos = InfoTable(np.full(n, ctg_bytes, dtype="S%d" % MAX_CTG), np.full(n, len(ctg_bytes), dtype=np.uint8), centres, seq_col,
                              (seqs[:, :33] != 0).sum(axis=1).astype(np.uint8))
        else:
            raw = seqs.tobytes()
            infos = [[args.ctgName, str(c), raw[i * 34:i * 34 + 34].split(b"\0", 1)[0].decode("latin-1")] for i, c in enumerate(centres.tolist())]
        x = _hostapi.counts_to_input(counts)              # the decode reads depth and allele support from the tensor
        # the GPU takes the raw counts (half the bytes on the host link) when they fit int16
        small = counts.astype(np.int16) if int(counts.max()) <= 32767 and int(counts.min()) >= -32768 else None
        return x, infos, small

    def drain(final):
        nonlocal held, held_n
        while builder.pending():
            centres, seqs, counts = builder.take_columns(batch_size)    # at most one batch per take: every copy below is O(batch)
            keep = _IUPAC_TABLE[seqs[:, 16]]                             # a refseq shorter than 17 has NUL there: not an IUPAC code
            if not keep.all():
                centres, seqs, counts = centres[keep], seqs[keep], counts[keep]
            for col, piece in zip(held, (centres, seqs, counts)):
                col.append(piece)
            held_n += len(centres)
            if held_n >= batch_size:
                cols = [np.concatenate(c) if len(c) > 1 else c[0] for c in held]
                held = [[c[batch_size:]] if held_n > batch_size else [] for c in cols]
                held_n -= batch_size
                yield emit(*(c[:batch_size] for c in cols))
        if final and held_n > 0:
            cols = [np.concatenate(c) if len(c) > 1 else c[0] for c in held]
            held, held_n = [[], [], []], 0
            yield emit(*cols)

    tail = None
    while True:
        chunk = view.stdout.read(1 << 22)
        if not chunk:
            break
        tail = builder.feed(chunk if tail is None else tail + chunk)
        for batch in drain(False):
            yield batch
    if tail:
        builder.feed(tail, final=True)
    builder.finish()
    for batch in drain(True):
        yield batch
    view.stdout.close()
    view.wait()
    if view.returncode != 0:
        sys.exit("[ERROR] `samtools view` failed on %s" % args.bam_fn)


SLAB_BYTES = 64 << 20      # SEQ bytes per slab of packed alignments sent to the device (host packer)
TEXT_CHUNK = 64 << 20      # bytes of `samtools view` text handed to the device at a time (device parser; the page-locked buffer it is read into)
TABLE_MARGIN = 64          # positions the device tables extend beyond the region (a window reaches 17 beyond its centre)
FE_REASONS = ((1, "alignments not sorted by position"), (2, "a zero-length insertion/deletion"), (4, "an alignment spanning > 100 kb more than its bases"),
              (8, "a CIGAR longer than its SEQ"), (16, "a read base outside the IUPAC alphabet"), (32, "a reference base outside the IUPAC alphabet"),
              (64, "a count beyond int16"), (128, "the reference's budget of 5 M outstanding tuples would run out"), (256, "candidate sites not strictly ascending"),
              (512, "an alignment that begins with an insertion/deletion after another one at the same start position"))


class AlignmentStream(object):
    """The text of `samtools view <bam> ctg:a-b`, produced by K `samtools view` processes over K consecutive pieces of [a, b] at once
    (--view_readers K): BAM decoding is what the device front end waits for, and one samtools formats a few hundred MB of text per second.
    The pieces' outputs, taken in order, are the single stream's lines exactly once each, in its order, when the alignments every piece
    but the first prints that START before the piece are dropped: they overlap an earlier piece too and were printed there (sorted BAM:
    those lines are a prefix of the piece's output).  readinto() / close() like the single pipe's."""

    def __init__(self, args, ctg, first, last, readers):
        import queue
        import threading
        edges = [first + (last - first + 1) * k // readers for k in range(readers)] + [last + 1]
        self.starts = edges[:-1]
        self.procs = [ct.subprocess_popen(view_command(args, "%s:%d-%d" % (ctg, edges[k], edges[k + 1] - 1)), text=False) for k in range(readers)]
        self.queues = [queue.Queue(maxsize=16) for _ in range(readers)]          # 16 x 8 MB read ahead per piece
        self.stop = threading.Event()
        self.threads = [threading.Thread(target=self._drain, args=(k,), daemon=True) for k in range(readers)]
        for t in self.threads:
            t.start()
        self.piece, self.pending, self.dropping, self.carry = 0, b"", False, b""

    def _drain(self, k):
        import queue
        out = self.procs[k].stdout
        while not self.stop.is_set():
            chunk = out.read(1 << 23)
            while not self.stop.is_set():
                try:
                    self.queues[k].put(chunk, timeout=0.2)      # b"" = end of this piece
                    break
                except queue.Full:
                    continue
            if not chunk:
                return

    def _next_chunk(self):
        """-> bytes of the merged stream, b"" at its end"""
        while self.piece < len(self.procs):
            chunk = self.queues[self.piece].get()
            if not chunk:
                last, self.carry = self.carry, b""                   # a last line of the piece without its line end, still undecided
                keep = b""
                if last:
                    col = last.split(None, 4)
                    if not (self.piece > 0 and len(col) >= 4 and col[3].isdigit() and int(col[3]) < self.starts[self.piece]):
                        keep = last + b"\n"
                self.piece += 1
                self.dropping = True
                if keep:
                    return keep
                continue
            if self.piece > 0 and self.dropping:
                # the lines at the head of this piece that start before it: POS is the fourth column
                data, at, start = self.carry + chunk, 0, self.starts[self.piece]
                self.carry = b""
                while True:
                    nl = data.find(b"\n", at)
                    if nl < 0:
                        self.carry = data[at:]                        # no complete line yet
                        data = b""
                        break
                    col = data[at:nl].split(None, 4)
                    if len(col) >= 4 and col[0][:1] != b"@" and col[3].isdigit() and int(col[3]) < start:
                        at = nl + 1
                        continue
                    self.dropping = False
                    data = data[at:]
                    break
                if not data:
                    continue
                return data
            return chunk
        return b""

    def readinto(self, mv):
        if not self.pending:
            self.pending = self._next_chunk()
            if not self.pending:
                return 0
        n = min(len(mv), len(self.pending))
        mv[:n] = self.pending[:n]
        self.pending = self.pending[n:]
        return n

    def read(self, n):
        if not self.pending:
            self.pending = self._next_chunk()
        out, self.pending = self.pending[:n], self.pending[n:]
        return out

    def abort(self):
        """The consumer gave up: do not leave K samtools writing into full pipes."""
        self.stop.set()
        for p in self.procs:
            try:
                p.kill()
                p.stdout.close()
                p.wait()
            except OSError:
                pass

    def finish(self):
        """-> 0 when every samtools ended well"""
        self.stop.set()
        code = 0
        for p in self.procs:
            p.stdout.close()
            p.wait()
            code = code or p.returncode
        return code


class _OnePipe(object):
    def __init__(self, args, region):
        self.proc = ct.subprocess_popen(view_command(args, region), text=False)
        self.readinto, self.read = self.proc.stdout.readinto, self.proc.stdout.read

    def abort(self):
        try:
            self.proc.kill()
            self.proc.stdout.close()
            self.proc.wait()
        except OSError:
            pass

    def finish(self):
        self.proc.stdout.close()
        self.proc.wait()
        return self.proc.returncode


class DeviceFrontEnd(object):
    """Both pileup stages on the GPU for one contig / region.  run() -> number of windows, or None when the run has to take the host
    path (reason logged); batches() then yields what tensor_batches yields, with the counts as clair_amd._capi.DeviceWindows."""

    def __init__(self, args, device, pinned=None):
        """pinned: nbytes -> page-locked uint8 array (Clair.pinned_buffer): the text is then read straight into it and parsed on the
        device; without it (or with CLAIR_AMD_FE_PACK=host) the host packer (clair_host_sampack_*) makes the slabs."""
        self.args, self.device, self.pinned = args, device, pinned
        self.frontend = None
        self.n_windows = 0

    def close(self):
        if self.frontend is not None:
            self.frontend.close()
            self.frontend = None

    def _fallback(self, why):
        """The input is outside the regime the device formulation reproduces bit for bit (never: the device or the library is missing --
        that is an error).  --front_end device makes it an error too."""
        self.close()
        if self.args.front_end == "device":
            sys.exit("[ERROR] --front_end device: %s; the sequential host stages (--front_end host) reproduce the reference there" % why)
        logging.info("device front end not used (%s): running the candidate search and the pileup on the host" % why)
        return None

    def run(self):
        """-> number of windows, or None after a logged hand-back to the host stages.  Errors of the device library (out of memory, a lost
        device) end the run like the engine's do: a message and a non-zero exit."""
        from . import _capi
        try:
            return self._run()
        except _capi.MalformedText:
            raise
        except _capi.EngineError as exc:
            if "out of memory" in str(exc).lower():       # tables for the whole region + the resident text: too much for what is free on this GPU
                return self._fallback("not enough device memory for the region: %s" % exc)
            sys.exit("[ERROR] %s" % exc)

    def _run(self):
        from . import _capi, _hostapi
        args = self.args
        have_range = args.ctgStart is not None and args.ctgEnd is not None
        given = None
        if args.vcf_fn is not None:
            given = positions_from_vcf(args.vcf_fn, args.ctgName, args.ctgStart, args.ctgEnd)
            if have_range:
                given = given[(given >= args.ctgStart) & (given <= args.ctgEnd)]
            if len(given) > 1 and not (np.diff(given) > 0).all():
                return self._fallback(FE_REASONS[8][1])
        elif not os.path.isfile("%s.fai" % args.ref_fn):
            sys.exit("Fasta index %s.fai doesn't exist." % args.ref_fn)
        # one reference slice serves both stages: they load the same region (ExtractVariantCandidates.py:228-236, CreateTensor.py:113-156)
        seq, ref_start = ct.reference_sequence_from(args.samtools, args.ref_fn, args.ctgName, args.ctgStart, args.ctgEnd)
        if not seq:
            sys.exit("Failed to load reference seqeunce. Please check if the provided reference fasta %s and the ctgName %s are correct."
                     % (args.ref_fn, args.ctgName))
        ref0 = 0 if ref_start is None else ref_start - 1
        bed = None
        if given is None:
            tree = evc.bed_regions_from(args.bed_fn)
            if tree is not None and args.ctgName not in tree:
                sys.exit("[ERROR] ctg_name(%s) not exists in bed file(%s)." % (args.ctgName, args.bed_fn))
            bed = None if tree is None else tree[args.ctgName]
        if have_range:
            lo, hi = args.ctgStart - 1 - TABLE_MARGIN, args.ctgEnd + TABLE_MARGIN
            # the candidate search reads the widened region, the pileup the plain one (callVarBam.py:124-199); an alignment can only
            # matter to a position of [ctgStart, ctgEnd] if it reaches within one base of it -- the packer marks which of the two
            # stages would have been given it
            region = "%s:%d-%d" % (args.ctgName, max(1, args.ctgStart - 2), args.ctgEnd + 2)
        else:
            lo, hi = ref0 - TABLE_MARGIN, ref0 + len(seq) + TABLE_MARGIN
            region = args.ctgName
        f = self.frontend = _capi.Frontend(self.device, seq, ref0, lo, hi)
        pack_kw = dict(dcov=args.dcov, evc_min_mq=0, pile_min_mq=0, pile_region=(args.ctgStart, args.ctgEnd) if have_range else None)
        readers = max(1, int(getattr(args, "view_readers", 1) or 1))
        span = (max(1, args.ctgStart - 2), args.ctgEnd + 2) if have_range else (1, contig_length(args.ref_fn, args.ctgName) or 0)
        if readers > 1 and span[1] - span[0] + 1 >= 2 * readers:
            view = AlignmentStream(args, args.ctgName, span[0], span[1], readers)
        else:
            view = _OnePipe(args, region)
        from time import time
        t_start, t_pack, t_dev = time(), 0.0, 0.0
        try:
            if self.pinned is not None and os.environ.get("CLAIR_AMD_FE_PACK", "device") != "host":
                # the text goes to the GPU as it comes out of the pipe (read into a page-locked buffer, whole lines at a time) and is
                # parsed there: the host touches no byte of it but the last megabyte of each chunk, looking for the line end
                f.text_options(args.ctgName, **pack_kw)
                buf = self.pinned(TEXT_CHUNK + 16)
                mv = memoryview(buf)
                fill, eof = 0, False
                while not eof:
                    while fill < TEXT_CHUNK:
                        n = view.readinto(mv[fill:TEXT_CHUNK])
                        if not n:
                            eof = True
                            break
                        fill += n
                    cut = fill
                    if not eof:
                        cut, span = -1, 1 << 20
                        while cut < 0:
                            lo_ = max(0, fill - span)
                            k = bytes(mv[lo_:fill]).rfind(b"\n")
                            cut = lo_ + k + 1 if k >= 0 else -1
                            if lo_ == 0:
                                break
                            span *= 4
                        if cut <= 0:
                            sys.exit("[ERROR] an alignment line longer than %d bytes" % TEXT_CHUNK)
                    elif fill and buf[fill - 1] != 10:             # the stream ended without a line end
                        buf[fill] = 10
                        fill = cut = fill + 1
                    if cut:
                        t0 = time()
                        try:
                            f.add_text(buf.ctypes.data, cut)
                        except _capi.MalformedText:
                            _hostapi.SamPacker(args.ctgName, **pack_kw).feed(bytes(mv[:cut]), final=True)     # raises with the line and the column
                            raise
                        t_dev += time() - t0
                    buf[:fill - cut] = buf[cut:fill]
                    fill -= cut
                pst = f.text_stats()
            else:
                packer = _hostapi.SamPacker(args.ctgName, **pack_kw)
                tail = None
                while True:
                    chunk = view.read(1 << 23)
                    if not chunk:
                        break
                    t0 = time()
                    tail = packer.feed(chunk if tail is None else tail + chunk)
                    t_pack += time() - t0
                    if packer.stats()["seq_bytes"] >= SLAB_BYTES:
                        t0 = time()
                        f.add_slab(packer)
                        t_dev += time() - t0
                t0 = time()
                if tail:
                    packer.feed(tail, final=True)
                t_pack += time() - t0
                t0 = time()
                f.add_slab(packer)
                t_dev += time() - t0
                pst = packer.stats()
        except BaseException:
            view.abort()
            raise
        t0 = time()
        if view.finish() != 0:
            sys.exit("[ERROR] `samtools view` failed on %s" % args.bam_fn)
        if given is None:
            if pst["evc_reads"] == 0:
                print("No read has been process, either the genome region you specified has no read cover, or please check the correctness of your BAM input (%s)."
                      % args.bam_fn, file=sys.stderr)
            n_cand = f.find_candidates(min_coverage=int(args.minCoverage), threshold=args.threshold,     # callVarBam.py:75: int() before it reaches the extractor
                                       ctg_start=args.ctgStart if have_range else None, ctg_end=args.ctgEnd if have_range else None, bed=bed)
        else:
            f.set_candidates(given)
            n_cand = len(given)
        n = f.build_windows(min_coverage=0, drop_non_iupac_centre=True, consider_left_edge=not args.stop_consider_left_edge)
        bits = pst["anomalies"] | f.stats()["anomalies"]
        if not bits and f.budget_binds():
            bits = 128
        if bits:
            return self._fallback("; ".join(why for bit, why in FE_REASONS if bits & bit))
        t_dev += time() - t0
        logging.info("%d candidate sites" % n_cand)
        logging.info("device front end: %d alignments, %d windows in %.2f s (%s, device %.2f s, the rest waiting for `samtools view`)"
                     % (f.stats()["reads"], n, time() - t_start, "packing the text on the host %.2f s" % t_pack if t_pack else "text parsed on the device", t_dev))
        self.n_windows = n
        return n

    def batches(self, batch_size, lean=True, progress=True):
        """(X float32 or None, infos, counts) per batch.  lean: nobody on the host reads the tensor (decode on the device, no BAM open):
        the counts stay in HBM (DeviceWindows) and X is None; otherwise both come back to the host, as from tensor_batches."""
        from . import _capi, _hostapi
        from .tensor_binary import MAX_CTG, InfoTable
        f, ctg = self.frontend, self.args.ctgName.encode()
        use_table = len(ctg) <= MAX_CTG
        total = 0
        for first in range(0, self.n_windows, batch_size):
            n = min(batch_size, self.n_windows - first)
            centres, seqs = f.window_info(first, n)
            total += n
            if progress:
                print("Processed %d tensors" % total, file=sys.stderr)
            if use_table:
                infos = InfoTable(np.full(n, ctg, dtype="S%d" % MAX_CTG), np.full(n, len(ctg), dtype=np.uint8), centres,
                                  np.ascontiguousarray(seqs[:, :33]).view("S33").ravel(), (seqs[:, :33] != 0).sum(axis=1).astype(np.uint8))
            else:
                raw = seqs.tobytes()
                infos = [[self.args.ctgName, str(c), raw[i * 34:i * 34 + 34].split(b"\0", 1)[0].decode("latin-1")] for i, c in enumerate(centres.tolist())]
            if lean:
                yield None, infos, _capi.DeviceWindows(f, first, n)
            else:
                counts = f.window_counts(first, n)
                yield _hostapi.counts_to_input(counts.astype(np.int32)), infos, counts


WINDOW_FLANK = 33          # reads that touch only the flank of a sub-range's outermost windows (16 positions) must still be seen
MIN_SPAN_PER_WORKER = 50000


def contig_length(ref_fn, ctg_name):
    with open(ref_fn + ".fai") as f:
        for row in f:
            col = row.split("\t")
            if col[0] == ctg_name:
                return int(col[1])
    return None


def front_end_workers(args):
    """How many sub-ranges the two host stages are split into (--front_end_workers; default: by the CPUs this process may use)."""
    if args.vcf_fn is not None:
        return 1, None, None
    have_range = args.ctgStart is not None and args.ctgEnd is not None
    lo, hi = (args.ctgStart, args.ctgEnd) if have_range else (1, contig_length(args.ref_fn, args.ctgName) if os.path.isfile(args.ref_fn + ".fai") else None)
    if hi is None:
        return 1, None, None
    want = args.front_end_workers
    if want is None or want == 1:
        return 1, None, None        # the default is the single pass: see --front_end_workers for when the split run can differ from it
    if want <= 0:                   # 0 = by the CPUs this process may use
        want = max(1, min(8, ingest_cpus() // 2))
    want = max(1, min(want, (hi - lo + 1) // MIN_SPAN_PER_WORKER))
    return want, lo, hi


def ingest_cpus():
    """CPUs this process may use: affinity mask, capped by a cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return n


def parallel_front_end(args, batch_size, workers, lo, hi):
    """Both host stages (candidate search, pileup windows) over `workers` consecutive sub-ranges of [lo, hi] at once, each on its
    own thread with its own `samtools view` streams (the native cores release the GIL) -- the reference's process-per-chunk
    fan-out (clair/callVarBamParallel.py:90-119) inside one process feeding one GPU.  Batches come out in position order: a
    sub-range's batches are consumed when every earlier sub-range is done (bounded queues hold the others back).  Candidates
    and windows are those of the unsplit run (a sub-range takes its alignments from WINDOW_FLANK positions beyond its INNER ends) as
    long as CreateTensor's tuple budget does not run out: each sub-range has its own, like each chunk of callVarBamParallel."""
    import copy
    import queue
    import threading
    edges = [lo + (hi - lo + 1) * k // workers for k in range(workers)] + [hi + 1]
    queues = [queue.Queue(maxsize=6) for _ in range(workers)]
    counts = [0] * workers
    stop = threading.Event()

    def put(q, item):
        while not stop.is_set():
            try:
                q.put(item, timeout=0.2)
                return True
            except queue.Full:
                continue
        return False

    def work(k):
        sub = copy.copy(args)
        sub.ctgStart, sub.ctgEnd = edges[k], edges[k + 1] - 1
        try:
            positions = candidate_positions(sub, quiet=True)
            counts[k] = len(positions)
            # the run's own outer ends keep the reference's region semantics (alignments of exactly ctgStart-ctgEnd, CreateTensor.py);
            # on a whole contig they are the contig's ends
            flank = (WINDOW_FLANK if k > 0 else 0, WINDOW_FLANK if k + 1 < workers else 0)
            for batch in tensor_batches(sub, positions, batch_size, read_flank=flank, progress=False):
                if not put(queues[k], batch):
                    return
            put(queues[k], None)
        except BaseException:                       # sys.exit of a failing samtools included: handed to the consumer
            put(queues[k], sys.exc_info())

    threads = [threading.Thread(target=work, args=(k,), daemon=True) for k in range(workers)]
    for t in threads:
        t.start()
    total = 0
    try:
        for k in range(workers):
            while True:
                item = queues[k].get()
                if item is None:
                    break
                if isinstance(item, tuple) and len(item) == 3 and isinstance(item[1], BaseException):
                    raise item[1].with_traceback(item[2])
                total += len(item[1])
                print("Processed %d tensors" % total, file=sys.stderr)
                yield item
        logging.info("%d candidate sites" % sum(counts))
    finally:
        stop.set()


def normalise(args):
    """The argument checks of callVarBam.py:62-101."""
    if args.ctgName is None:
        sys.exit("--ctgName must be specified. You can call variants on multiple chromosomes simultaneously.")
    for path, suffix in ((args.bam_fn, ""), (args.ref_fn, "")):
        if not os.path.isfile(path + suffix):
            sys.exit("[ERROR] file %s not found" % (path + suffix))
    if args.ctgStart is not None and args.ctgEnd is not None and int(args.ctgStart) > int(args.ctgEnd):
        args.ctgStart = args.ctgEnd = None          # callVarBam.py:97-101
    if (args.ctgStart is None) != (args.ctgEnd is None):
        args.ctgStart = args.ctgEnd = None
    return args


def load_model(args):
    from .model import Clair
    batch = args.batch_size or param.engineBatchSize
    try:
        m = Clair(device=args.device, max_batch=batch, n_slots=param.pipeline_slots())
        m.init()
        m.restore_parameters(os.path.abspath(args.chkpnt_fn))
    except Exception as exc:
        sys.exit("[ERROR] %s" % exc)
    return m


def wants_device_front_end(args):
    return args.front_end != "host" and front_end_workers(args)[0] == 1


def call_region(args, m, prepared=None):
    """One contig / region through an engine that is already up: front end, network, decode, VCF.  prepared: a DeviceFrontEnd whose run()
    has been called (clair_amd.callVarBamParallel --run reads the next regions' alignments while this one is called); it is closed here."""
    config = cv.OutputConfig(
        is_show_reference=False, is_debug=args.debug,
        is_haploid_precision_mode_enabled=args.haploid_precision,
        is_haploid_sensitive_mode_enabled=args.haploid_sensitive,
        is_output_for_ensemble=args.output_for_ensemble, quality_score_for_pass=args.qual)
    lookup = cv.AlignmentLookup(args.bam_fn, args.ref_fn)
    decoder = cv.VariantDecoder(config, lookup, always_use_bam=args.pysam_for_all_indel_bases, arith=args.arith)
    writer = cv.VcfWriter(args.call_fn, args.sampleName, args.ref_fn, args.output_for_ensemble)
    device_fe = prepared
    try:
        batch = args.batch_size or param.engineBatchSize
        source = None
        workers, lo, hi = front_end_workers(args)
        if wants_device_front_end(args):
            # nobody on the host reads the tensors when the decode runs on the device and no BAM is consulted (cv.call_variants)
            lean = decoder.native_applies() and lookup.sam is None and os.environ.get("CLAIR_AMD_DEVICE_DECODE", "1") != "0"
            if device_fe is None:
                device_fe = DeviceFrontEnd(args, args.device, pinned=getattr(m, "pinned_buffer", None))
                device_fe.run()
            if device_fe.frontend is not None:
                def source(batch):
                    return device_fe.batches(batch, lean=lean)
        if source is None and workers > 1:
            def source(batch):
                return parallel_front_end(args, batch, workers, lo, hi)
        elif source is None:
            positions = candidate_positions(args)
            logging.info("%d candidate sites" % len(positions))

            def source(batch):
                return tensor_batches(args, positions, batch)
        cv.call_variants(args, m, decoder, writer, batch, generator=source(batch))
    finally:
        if device_fe is not None:
            device_fe.close()
        writer.close()
        lookup.close()


def Run(args):
    normalise(args)
    logging.basicConfig(format="%(message)s", level=logging.INFO)
    cv.ingest.setup_environment()
    if args.activation_only:
        cv.VcfWriter(args.call_fn, args.sampleName, args.ref_fn, args.output_for_ensemble).close()
        return
    m = load_model(args)
    try:
        call_region(args, m)
    finally:
        m.close()


def build_parser():
    """Flag names and defaults of clair/callVarBam.py:237-322 (help texts are this build's), plus --batch_size / --device / --arith."""
    parser = ArgumentParser(description="BAM to VCF for one contig or region, in one process on one GPU")
    add = parser.add_argument
    add('--chkpnt_fn', type=str, default=None, help="model checkpoint prefix (.npz container or TensorFlow bundle)")
    add('--ref_fn', type=str, default="ref.fa", help="reference FASTA with its .fai")
    add('--bed_fn', type=str, default=None, help="restrict candidates to these intervals (intersected with the region)")
    add('--bam_fn', type=str, default="bam.bam", help="sorted alignments")
    add('--call_fn', type=str, default=None, help="output VCF")
    add('--vcf_fn', type=str, default=None, help="call only at the sites of this VCF instead of extracting candidates")
    add('--threshold', type=float, default=0.125, help="minimum allele frequency of a candidate site, default: %(default)s")
    add('--minCoverage', type=float, default=4, help="minimum depth of a candidate site, default: %(default)s")
    add('--qual', type=int, default=None, help="PASS / LowQual cut-off, optional")
    add('--sampleName', type=str, default="SAMPLE", help="sample column of the VCF")
    add('--ctgName', type=str, default=None, help="contig to process (required)")
    add('--ctgStart', type=int, default=None, help="1-based first position of the region")
    add('--ctgEnd', type=int, default=None, help="1-based last position of the region (inclusive)")
    add('--stop_consider_left_edge', action='store_true', help="open a window only for reads that cover its left edge")
    add('--dcov', type=int, default=250, help="at most this many reads per start position, default: %(default)s")
    add('--samtools', type=str, default="samtools", help="samtools executable")
    add('--pypy', type=str, default="pypy3", help="no stage of this pipeline runs under pypy; the name only selects whose set order the pileup follows where the reference's tuple budget binds (pypy*: insertion order, anything else: CPython's)")
    add('--threads', type=int, default=None, help="host threads, optional")
    add('--delay', type=int, default=10, help="ignored: there is no TensorFlow start-up thread storm to stagger")
    add('--debug', action='store_true', help="debug lines in the VCF body")
    add('--pysam_for_all_indel_bases', action='store_true', help="look every indel up in the BAM (needs pysam)")
    add('--haploid_precision', action='store_true', help="haploid calling: homozygous variants only")
    add('--haploid_sensitive', action='store_true', help="haploid calling: everything but multi-allelic variants")
    add('--activation_only', action='store_true', help="kept for flag compatibility (plotting is a dead path)")
    add('--max_plot', type=int, default=10, help="kept for flag compatibility")
    add('--log_path', type=str, nargs='?', default=None, help="kept for flag compatibility")
    add('-p', '--parallel_level', type=int, default=2, help="kept for flag compatibility")
    add('--fast_plotting', action='store_true', help="kept for flag compatibility")
    add('-w', '--workers', type=int, default=8, help="kept for flag compatibility")
    add('--output_for_ensemble', action='store_true', help="write probabilities for ensembling instead of a VCF")
    # additions of this implementation
    add('--batch_size', type=int, default=None, help="candidates per forward pass, default: %d" % param.engineBatchSize)
    add('--front_end_workers', type=int, default=None,
        help="run the candidate search and the pileup over this many consecutive sub-ranges at once (threads, one pair of samtools streams each; "
             "0 = half the usable CPUs, at most 8; at least 50 kb per sub-range).  Default 1: the single pass.  The split run yields the single "
             "pass's candidates and windows EXCEPT where CreateTensor's 5 M-tuple budget runs out (candidates every few bases at high depth): every "
             "sub-range has its own budget, as every chunk of callVarBamParallel has")
    add('--front_end', type=str, default="auto", choices=("auto", "device", "host"),
        help="where the candidate search and the pileup run: on the GPU (one pass over the alignments; auto = device, falling back to the host "
             "stages, with a message, where the device formulation does not reproduce the reference exactly) or on the host (two passes, the "
             "sequential code).  --front_end_workers > 1 implies host")
    add('--view_readers', type=int, default=1,
        help="with the front end on the device: read the alignments with this many `samtools view` processes at once, each over a consecutive piece "
             "of the region (same lines, same order as one)")
    add('--samtools_view_args', type=str, default=None,
        help="extra options for `samtools view`, e.g. \"--keep-tag NM\": nothing beyond SEQ is read from a line, and the tags of an ONT BAM (move "
             "tables) can be several times the size of the rest.  Write a value that is itself one option with `=`: --samtools_view_args=-x")
    add('--samtools_threads', type=int, default=0,
        help="extra decompression threads for `samtools view` (its -@): with the front end on the device the BAM decoder is what the run waits for")
    add('--device', type=int, default=0, help="HIP device ordinal, default: %(default)s")
    add('--arith', type=str, default="legacy", choices=("legacy", "numpy2"),
        help="QUAL/AF arithmetic: float64 as under the reference's NumPy 1.x (legacy) or float32 (numpy2)")
    return parser


def main(argv=None):
    parser = build_parser()
    argv = sys.argv[1:] if argv is None else argv
    if len(argv) == 0:
        parser.print_help()
        sys.exit(1)
    Run(parser.parse_args(argv))


if __name__ == "__main__":
    main()
