"""Pileup tensor generation: sorted alignments + candidate positions -> the [33, 8, 4] count tensors `call_var` consumes.

Host-side mirror of the reference's dataPrepScripts/CreateTensor.py (same command-line flags, same text records on
stdout / in the gzip file, byte for byte -- tests/golden/pileup_ct_*.json.gz are minted from the real script).  Like the
reference it never links htslib: `samtools view` / `samtools faidx` are spawned and their TEXT is read
(CreateTensor.py:113-170).  The work itself -- the CIGAR walk and the scatter of base counts -- runs in
libclair_host.so (include/clair_host.h, clair_host_pileup_*); `PileupBuilderPy` below is the same algorithm in plain
Python, kept as the checker the native code is pinned against (tests/test_pileup.py).

What is computed (CreateTensor.py:29-65, 251-388).  For every candidate centre c (1-based) a window of 33 reference
positions c-16 .. c+16; per window position, 8 rows (A, C, G, T on the forward strand, then the reverse strand) and 4
channels:
    0: reference base of every read base aligned there (M/=/X)          2: channel 0 + reference base of deleted positions (D)
    1: read base of every aligned base + every inserted base (inserted  3: read base of every aligned base
       base k of an insertion lands on window position +k, clamped)
A read contributes to a window from the first walked reference position inside [c-17, c+17) ("left edge" mode, the
default; with --stop_consider_left_edge only a read that walks over c-17 itself) until it walks over c+17.  Reads starting
at one position are capped at --dcov; mapping quality below --minMQ drops the read.  A window is written when a read with a
new start position begins beyond it (and at the end of input), in the order the windows were first touched; it is dropped
when its centre depth is below --minCoverage or it would start before the loaded reference.

Instead of the reference's per-read lists of (position, base) tuples the counts are accumulated as the read is walked
(the sum is order-independent); the reference's budget of 5 000 000 outstanding tuples ("available_slots",
CreateTensor.py:181, 303-309) is accounted for tuple by tuple, so the same bases are dropped once it runs out -- with one
documented liberty: when the budget runs out in the MIDDLE of one reference position, the reference serves the windows in
CPython set-iteration order; here they are served in ascending centre order.
"""
import os
import shlex
import subprocess
import sys
from argparse import ArgumentParser

import numpy as np

FLANK = 16                      # shared/param.py:9 flankingBaseNum
N_POS = 2 * FLANK + 1           # 33
EXPAND_REFERENCE_REGION = 1000000   # shared/param.py:5
SAMTOOLS_VIEW_FILTER_FLAG = 2316    # shared/param.py:6
AVAILABLE_SLOTS = 5000000       # CreateTensor.py:181
CANDIDATE_LOOKAHEAD = 100000    # CreateTensor.py:275
# shared/utils.py:24-27: IUPAC code -> row (first listed base of the ambiguity set)
BASE2NUM = dict(zip("ACGTURYSWKMBDHVN", (0, 1, 2, 3, 3, 0, 1, 1, 0, 2, 0, 1, 0, 0, 0, 0)))


class PileupError(RuntimeError):
    pass


class PileupBuilderPy(object):
    """Streaming restatement of OutputAlnTensor's main loop (CreateTensor.py:251-388) -- the checker for the native code."""

    def __init__(self, ctg_name, reference_sequence, reference_start_0_based, candidates, consider_left_edge=True,
                 dcov=250, min_coverage=0, min_mq=0, available_slots=AVAILABLE_SLOTS, set_order="ascending"):
        """set_order: the order in which a read base is offered to the windows open at its position, which only shows when the tuple budget
        runs out in the middle of one base (CreateTensor.py:181, 296-310): "ascending" = the order the windows were opened in, what an
        insertion-ordered set gives (PyPy, the interpreter clair/callVarBam.py starts the script with by default); "cpython" = the iteration order
        of CPython's hash set -- here simply a real `set` driven by the reference's own sequence of add / remove calls."""
        if set_order not in ("ascending", "cpython"):
            raise ValueError("set_order: 'ascending' or 'cpython'")
        self.set_order = set_order
        self.ctg = ctg_name
        self.ref = reference_sequence
        self.ref0 = reference_start_0_based
        self.cands = list(candidates)          # region-filtered, in stream order (CreateTensor.py:88-92)
        self.left_edge = consider_left_edge
        self.dcov, self.min_cov, self.min_mq = dcov, min_coverage, min_mq
        self.slots = available_slots
        self.next_cand = 0
        self.cand_pos = 0                       # the generator's last answer (CreateTensor.py:219), -1 once exhausted
        self.begin = {}                         # 0-based reference position -> [(end, centre), ...]
        self.windows = {}                       # centre -> [counts int32[33,8,4], used tuples]; insertion-ordered
        self.prev_pos, self.depth_cap = 0, 0
        self.out = []                           # finished records (centre, refseq, counts)

    # -- the candidate generator, advanced lazily ahead of the reads (CreateTensor.py:68-109, 274-275)
    def _load_candidates(self, limit):
        while self.cand_pos != -1 and self.cand_pos < limit:
            if self.next_cand >= len(self.cands):
                self.cand_pos = -1
                break
            p = self.cands[self.next_cand]
            self.next_cand += 1
            if self.left_edge:
                for i in range(p - (FLANK + 1), p + (FLANK + 1)):
                    self.begin.setdefault(i, []).append((p + FLANK + 1, p))
            else:
                self.begin[p - (FLANK + 1)] = [(p + FLANK + 1, p)]
            self.cand_pos = p

    def _ref_base(self, reference_position):
        i = reference_position - self.ref0
        if i < 0:
            i += len(self.ref)                  # the reference indexes a Python str: negative offsets wrap
        if not 0 <= i < len(self.ref):
            raise PileupError("reference position %d is outside the loaded reference sequence" % (reference_position + 1))
        return self.ref[i]

    def _count(self, centre, reference_position, query_adv, ref_base, query_base, strand):
        """One (position, base) tuple of the reference's alignment lists, applied at once (generate_tensor, :29-56)."""
        w = self.windows[centre]
        w[1] += 1
        self.slots -= 1
        if (ref_base != "-" and ref_base not in BASE2NUM) or (query_base != "-" and query_base not in BASE2NUM):
            return
        idx = reference_position - centre + (FLANK + 1)
        if not 0 <= idx < N_POS:
            return
        so = 4 if strand else 0
        t = w[0]
        if query_base != "-" and ref_base != "-":
            r, q = BASE2NUM[ref_base] + so, BASE2NUM[query_base] + so
            t[idx, r, 0] += 1
            t[idx, q, 1] += 1
            t[idx, r, 2] += 1
            t[idx, q, 3] += 1
            w[2][idx] += 1
        elif query_base != "-":
            t[min(idx + query_adv, N_POS - 1), BASE2NUM[query_base] + so, 1] += 1
        else:
            t[idx, BASE2NUM[ref_base] + so, 2] += 1

    def _finish(self, centre):
        counts, used, depth = self.windows[centre]
        nrp = centre - self.ref0
        if nrp - (FLANK + 1) < 0 or depth[FLANK] < self.min_cov:
            return
        self.out.append((centre, self.ref[nrp - (FLANK + 1):nrp + FLANK], counts))

    def add_read(self, flag, pos_1_based, mapq, cigar, seq):
        pos = pos_1_based - 1
        seq = seq.upper()
        strand = (flag & 16) == 16
        if mapq < self.min_mq:
            return
        self._load_candidates(pos + len(seq) + CANDIDATE_LOOKAHEAD)
        if self.prev_pos != pos:
            self.prev_pos, self.depth_cap = pos, 0
        else:
            self.depth_cap += 1
            if self.depth_cap >= self.dcov:
                return
        active, end_to_centre = [], {}          # this read's open windows (ascending insertion), their closing positions
        hashed = set() if self.set_order == "cpython" else None      # the reference's active_set itself: same adds, same removes, same order
        rp, qp, adv = pos, 0, 0

        def offered():
            return list(hashed) if hashed is not None else sorted(active)

        def close_window(centre):
            active.remove(centre)
            if hashed is not None:
                hashed.remove(centre)

        def open_windows():
            for end, centre in self.begin.get(rp, ()):
                if centre in active:
                    continue
                end_to_centre[end] = centre
                active.append(centre)
                if hashed is not None:
                    hashed.add(centre)
                if centre not in self.windows:
                    self.windows[centre] = [np.zeros((N_POS, 8, 4), np.int32), 0, np.zeros(N_POS, np.int64)]

        for ch in cigar:
            if self.slots <= 0:
                break
            if ch.isdigit():
                adv = adv * 10 + int(ch)
                continue
            if ch == "S":
                qp += adv
            if ch in "M=X":
                for _ in range(adv):
                    open_windows()
                    if active:
                        if qp >= len(seq):
                            raise PileupError("CIGAR %s walks past the end of SEQ (%d bases)" % (cigar, len(seq)))
                        rb, qb = self._ref_base(rp), seq[qp]
                        for centre in offered():
                            if self.slots <= 0:
                                break
                            self._count(centre, rp, 0, rb, qb, strand)
                    if rp in end_to_centre:
                        close_window(end_to_centre[rp])
                    rp += 1
                    qp += 1
            if ch == "I":
                for k in range(adv):
                    if active:
                        if qp >= len(seq):
                            raise PileupError("CIGAR %s walks past the end of SEQ (%d bases)" % (cigar, len(seq)))
                        for centre in offered():
                            if self.slots <= 0:
                                break
                            self._count(centre, rp, k, "-", seq[qp], strand)
                    qp += 1
            if ch == "D":
                for _ in range(adv):
                    if active:
                        rb = self._ref_base(rp)
                        for centre in offered():
                            if self.slots <= 0:
                                break
                            self._count(centre, rp, 0, rb, "-", strand)
                    open_windows()
                    if rp in end_to_centre:
                        close_window(end_to_centre[rp])
                    rp += 1
            adv = 0

        if self.depth_cap == 0:                 # a new start position: windows that end before it are complete
            for centre in [c for c in self.windows if c + (FLANK + 1) < pos]:
                self._finish(centre)
                self.slots += self.windows[centre][1]
                del self.windows[centre]

    def add_sam_line(self, line):
        col = line.split()
        if not col:
            raise PileupError("empty alignment line")
        if col[0][0] == "@":
            return
        if len(col) < 10:
            raise PileupError("alignment line with %d columns" % len(col))
        self.add_read(int(col[1]), int(col[3]), int(col[4]), col[5], col[9])

    def finish(self):
        for centre in list(self.windows):
            self._finish(centre)
        self.windows.clear()

    def take(self):
        out, self.out = self.out, []
        return out


def format_record(ctg_name, centre, refseq, counts):
    """One text record (CreateTensor.py:60-65): 'ctg pos refseq v0 ... v1055'."""
    return "%s %d %s %s" % (ctg_name, centre, refseq, " ".join("%d" % v for v in np.asarray(counts).reshape(-1)))


# ---------------------------------------------------------------------------------------------------------------------
# inputs: reference slice, candidate stream, alignments

def subprocess_popen(args, stdin=None, stdout=subprocess.PIPE, text=True):
    return subprocess.Popen(args, stdin=stdin, stdout=stdout, stderr=sys.stderr, bufsize=8388608, universal_newlines=text)


def reference_sequence_from(samtools, ref_fn, ctg_name, ctg_start, ctg_end):
    """`samtools faidx` for the region widened by 1 Mbp (CreateTensor.py:113-156) -> (sequence upper-cased, 1-based start or None)."""
    start = end = None
    if ctg_start is not None and ctg_end is not None:
        start, end = max(1, ctg_start - EXPAND_REFERENCE_REGION), ctg_end + EXPAND_REFERENCE_REGION
        region = "%s:%d-%d" % (ctg_name, start, end)
    else:
        region = ctg_name
    try:
        p = subprocess_popen(shlex.split("%s faidx %s %s" % (samtools, ref_fn, region)))
    except OSError:
        return None, start
    rows = p.stdout.read().split("\n")
    p.stdout.close()
    p.wait()
    if p.returncode != 0:
        return None, start
    return "".join(r.rstrip() for r in rows[1:]).upper(), start


def candidate_positions_from(handle, ctg_start, ctg_end):
    """1-based positions, column 2 of every row, those outside [ctgStart, ctgEnd] dropped (CreateTensor.py:86-92)."""
    region = ctg_start is not None and ctg_end is not None
    out = []
    for row in handle:
        col = row.split(None, 2)
        if len(col) < 2:
            raise PileupError("candidate row with %d columns" % len(col))
        p = int(col[1])
        if region and not ctg_start <= p <= ctg_end:
            continue
        out.append(p)
    return out


def set_order_of(interpreter=None):
    """Which order the reference would have offered a base to its open windows in, given the interpreter its script runs under: PyPy's sets
    keep insertion order ("ascending" here), CPython's iterate in hash order ("cpython").  CLAIR_AMD_SET_ORDER overrides; the default is the
    reference pipeline's own default interpreter, pypy3 (clair/callVarBam.py: --pypy).  It only matters where the tuple budget binds."""
    forced = os.environ.get("CLAIR_AMD_SET_ORDER")
    if forced in ("ascending", "cpython"):
        return forced
    name = os.path.basename(interpreter or "pypy3")
    return "ascending" if name.startswith("pypy") else "cpython"


def make_builder(native, *args, **kwargs):
    if native:
        from . import _hostapi
        return _hostapi.PileupBuilder(*args, **kwargs)
    return PileupBuilderPy(*args, **kwargs)


def records_from_sam(builder, sam_handle, chunk_bytes=1 << 22):
    """Feed SAM text to a builder, yielding finished (centre, refseq, counts) records as they complete."""
    if isinstance(builder, PileupBuilderPy):
        for line in sam_handle:
            builder.add_sam_line(line)
            if builder.out:
                for rec in builder.take():
                    yield rec
    else:
        tail = None
        while True:
            chunk = sam_handle.read(chunk_bytes)
            if not chunk:
                break
            tail = builder.feed(chunk if tail is None else tail + chunk)
            for rec in builder.take():
                yield rec
        if tail:
            builder.feed(tail, final=True)
    builder.finish()
    for rec in builder.take():
        yield rec


def output_aln_tensor(args, native=True):
    """OutputAlnTensor (CreateTensor.py:179-394)."""
    seq, ref_start = reference_sequence_from(args.samtools, args.ref_fn, args.ctgName, args.ctgStart, args.ctgEnd)
    if not seq:
        print("Failed to load reference seqeunce. Please check if the provided reference fasta %s and the ctgName %s are correct."
              % (args.ref_fn, args.ctgName), file=sys.stderr)
        sys.exit(1)
    ref0 = 0 if ref_start is None else ref_start - 1

    can_proc = None
    if args.can_fn == "PIPE":
        can_handle = sys.stdin
    else:
        can_proc = subprocess_popen(shlex.split("gzip -fdc %s" % args.can_fn))
        can_handle = can_proc.stdout
    # The reference pulls candidates lazily while it reads alignments; which candidates are known when a read is walked
    # only depends on their order, so the list is read up front and the builder replays the same look-ahead rule.
    cands = candidate_positions_from(can_handle, args.ctgStart, args.ctgEnd)
    if can_proc is not None:
        can_handle.close()
        can_proc.wait()

    builder = make_builder(native, args.ctgName, seq, ref0, cands, consider_left_edge=not args.stop_consider_left_edge,
                           dcov=args.dcov, min_coverage=args.minCoverage, min_mq=args.minMQ, set_order=set_order_of())
    have_region = args.ctgStart is not None and args.ctgEnd is not None
    region = "%s:%d-%d" % (args.ctgName, args.ctgStart, args.ctgEnd) if have_region else args.ctgName
    is_native = not isinstance(builder, PileupBuilderPy)      # the native builder takes the bytes as they come
    if getattr(args, "sam_fn", None):
        view, sam_handle = None, open(args.sam_fn, "rb" if is_native else "r")
    else:
        view = subprocess_popen(shlex.split("%s view -F %d %s %s" % (args.samtools, SAMTOOLS_VIEW_FILTER_FLAG, args.bam_fn, region)),
                                text=not is_native)
        sam_handle = view.stdout

    gz = None
    if args.tensor_fn != "PIPE":
        raw = open(args.tensor_fn, "wb")
        gz = subprocess_popen(shlex.split("gzip -c"), stdin=subprocess.PIPE, stdout=raw, text=not getattr(args, "binary", False))
        sink = gz.stdin
    else:
        sink = sys.stdout
    try:
        if getattr(args, "binary", False):
            from . import tensor_binary
            out = sink.buffer if hasattr(sink, "buffer") else sink
            out.write(tensor_binary.MAGIC)
            group = []
            for rec in records_from_sam(builder, sam_handle):
                group.append(rec)
                if len(group) >= 512:
                    out.write(tensor_binary.pack_records(args.ctgName, [r[0] for r in group], [r[1] for r in group], np.stack([r[2] for r in group])))
                    group = []
            if group:
                out.write(tensor_binary.pack_records(args.ctgName, [r[0] for r in group], [r[1] for r in group], np.stack([r[2] for r in group])))
            out.flush()
        elif isinstance(builder, PileupBuilderPy):
            for centre, refseq, counts in records_from_sam(builder, sam_handle):
                sink.write(format_record(args.ctgName, centre, refseq, counts))
                sink.write("\n")
        else:
            for text in builder.text_from_sam(sam_handle):
                sink.write(text)
    finally:
        sam_handle.close()
        if view is not None:
            view.wait()
        if gz is not None:
            gz.stdin.close()
            gz.wait()
            raw.close()
        else:
            sink.flush()


def build_parser():
    """Flag names and defaults of dataPrepScripts/CreateTensor.py:397-437 (help texts are this build's), plus three additions."""
    parser = ArgumentParser(description="Pileup count tensors for a list of candidate positions")
    add = parser.add_argument
    add('--bam_fn', type=str, default="input.bam", help="sorted alignments")
    add('--ref_fn', type=str, default="ref.fa", help="reference FASTA")
    add('--can_fn', type=str, default="PIPE", help="candidate rows (column 2 = 1-based position), gzip or plain; PIPE = standard input")
    add('--tensor_fn', type=str, default="PIPE", help="output records, gzip file; PIPE = standard output")
    add('--minMQ', type=int, default=0, help="drop alignments below this mapping quality, default: %(default)d")
    add('--ctgName', type=str, default="chr17", help="contig to process, default: %(default)s")
    add('--ctgStart', type=int, default=None, help="1-based first position of the region")
    add('--ctgEnd', type=int, default=None, help="1-based last position of the region (inclusive)")
    add('--samtools', type=str, default="samtools", help="samtools executable")
    add('--stop_consider_left_edge', action='store_true', help="open a window only for reads that cover its left edge")
    add('--dcov', type=int, default=250, help="at most this many reads per start position, default: %(default)d")
    add('--minCoverage', type=int, default=0, help="drop windows whose centre depth is below this, default: %(default)d")
    # additions (not in the reference)
    add('--sam_fn', type=str, default=None, help="read alignments as SAM text from this file instead of spawning `samtools view`")
    add('--python_pileup', action='store_true', help="use the pure-Python pileup instead of libclair_host.so (slow)")
    add('--binary', action='store_true', help="write fixed-size binary records (clair_amd/tensor_binary.py) instead of text")
    return parser


def main(argv=None):
    parser = build_parser()
    argv = sys.argv[1:] if argv is None else argv
    if len(argv) == 0:
        parser.print_help()
        sys.exit(1)
    args = parser.parse_args(argv)
    output_aln_tensor(args, native=not args.python_pileup)


if __name__ == "__main__":
    main()
