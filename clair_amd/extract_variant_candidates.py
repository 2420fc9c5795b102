"""Candidate extraction: sorted alignments -> the sites worth showing to the network.

Host-side mirror of the reference's dataPrepScripts/ExtractVariantCandidates.py in inference mode (same flags, same rows on
stdout / in the gzip file, byte for byte -- tests/golden/pileup_evc_*.json.gz are minted from the real script).  A site is a
candidate when enough reads cover it and the reference base is not the most frequent observation, or the runner-up
observation (another base, "an insertion starts after here", "a deletion starts after here") reaches --threshold of the depth
(ExtractVariantCandidates.py:357-371).  The tallying runs in libclair_host.so (clair_host_evc_*); `CandidateFinderPy` is the
same algorithm in plain Python, the checker the native code is pinned against (tests/test_pileup.py).

Not restated: the training-set switches --gen4Training / --var_fn / --outputProb (they thin the candidates with Python's
`random` module; the build has no training path) -- they are accepted and rejected with a message.
"""
import bisect
import shlex
import subprocess
import sys
from argparse import ArgumentParser
from os.path import isfile


from .create_tensor import EXPAND_REFERENCE_REGION, SAMTOOLS_VIEW_FILTER_FLAG, PileupError, subprocess_popen

RATIO_OF_NON_VARIANT_TO_VARIANT = 2.0
# shared/utils.py:19-22
BASE2ACGT = dict(zip("ACGTURYSWKMBDHVN", "ACGTTACCAGACAAAA"))
TALLY_KEYS = ("A", "C", "G", "T", "I", "D", "N")     # the reference's dict order (:265): ties in the frequency sort keep it


def evc_base_from(base):
    return base if base == "N" else BASE2ACGT[base]


class BedRegions(object):
    """Membership of a 0-based position in the bed intervals of one contig (shared/interval_tree.py: IntervalTree.at)."""

    def __init__(self, intervals):
        iv = sorted((s, e + 1 if s == e else e) for s, e in intervals)
        self.start, self.end = [], []
        for s, e in iv:
            if e <= s:
                continue
            if self.start and s <= self.end[-1]:
                self.end[-1] = max(self.end[-1], e)
            else:
                self.start.append(s)
                self.end.append(e)

    def __contains__(self, p):
        i = bisect.bisect_right(self.start, p)
        return i > 0 and p < self.end[i - 1]


def bed_regions_from(bed_fn):
    """-> {ctg: [(start, end), ...]} (shared/interval_tree.py:7-39); None without a bed file."""
    if bed_fn is None:
        return None
    p = subprocess_popen(shlex.split("gzip -fdc %s" % bed_fn))
    tree = {}
    for row in p.stdout:
        col = row.strip().split()
        if not col:
            continue
        tree.setdefault(col[0], []).append((int(col[1]), int(col[2])))
    p.stdout.close()
    p.wait()
    return tree


class CandidateFinderPy(object):
    """Streaming restatement of make_candidates' loop (ExtractVariantCandidates.py:263-393) -- the checker for the native code."""

    def __init__(self, ctg_name, reference_sequence, reference_start_0_based, ctg_start=None, ctg_end=None, bed=None,
                 min_coverage=4, threshold=0.125, min_mq=0):
        self.ctg, self.ref, self.ref0 = ctg_name, reference_sequence, reference_start_0_based
        self.range = (ctg_start, ctg_end) if ctg_start is not None and ctg_end is not None else None
        self.bed = None if bed is None else BedRegions(bed)
        self.min_cov, self.min_af, self.min_mq = min_coverage, threshold, min_mq
        self.pileup = {}
        self.reads = 0
        self.out = []

    def _emit(self, p0, tally):
        if self.range is not None and not self.range[0] <= p0 + 1 <= self.range[1]:
            return
        if self.bed is not None and p0 not in self.bed:
            return
        i = p0 - self.ref0
        if i < 0:
            i += len(self.ref)
        if not 0 <= i < len(self.ref) or (self.ref[i] != "N" and self.ref[i] not in BASE2ACGT):
            return
        ref_base = evc_base_from(self.ref[i])
        depth = sum(tally) - tally[4] - tally[5]
        if depth < self.min_cov:
            return
        order = sorted(range(7), key=lambda k: -tally[k])
        if TALLY_KEYS[order[0]] == ref_base and float(tally[order[1]]) / (depth if depth > 0 else 1) < self.min_af:
            return
        self.out.append((p0 + 1, "%s %d %s %d %s" % (self.ctg, p0 + 1, ref_base, depth,
                                                     " ".join("%s %d" % (TALLY_KEYS[k], tally[k]) for k in order))))

    def _flush(self, before=None):
        for p0 in sorted(p for p in self.pileup if before is None or p < before):
            self._emit(p0, self.pileup.pop(p0))

    def add_sam_line(self, line):
        col = line.strip().split()
        if not col:
            raise PileupError("empty alignment line")
        if col[0][0] == "@":
            return
        if len(col) < 10:
            raise PileupError("alignment line with %d columns" % len(col))
        if col[2] != self.ctg:
            return
        pos, mapq, cigar, seq = int(col[3]) - 1, int(col[4]), col[5], col[9].upper()
        if mapq < self.min_mq or cigar == "*":
            return
        soft = total = adv = 0
        for ch in cigar:
            if ch.isdigit():
                adv = adv * 10 + int(ch)
                continue
            if ch == "S":
                soft += adv
            total += adv
            adv = 0
        if 1.0 - float(soft) / (total + 1) < 0.55:
            return
        self.reads += 1
        rp, qp, adv = pos, 0, 0
        for ch in cigar:
            if ch.isdigit():
                adv = adv * 10 + int(ch)
                continue
            if ch == "S":
                qp += adv
            elif ch in "M=X":
                for _ in range(adv):
                    if qp >= len(seq):
                        raise PileupError("CIGAR %s walks past the end of SEQ (%d bases)" % (cigar, len(seq)))
                    if seq[qp] != "N" and seq[qp] not in BASE2ACGT:
                        raise PileupError("SEQ holds '%s', not an IUPAC base code" % seq[qp])
                    self.pileup.setdefault(rp, [0] * 7)[TALLY_KEYS.index(evc_base_from(seq[qp]))] += 1
                    rp += 1
                    qp += 1
            elif ch == "I":
                self.pileup.setdefault(rp - 1, [0] * 7)[4] += 1
                qp += adv
            elif ch == "D":
                self.pileup.setdefault(rp - 1, [0] * 7)[5] += 1
                rp += adv
            adv = 0
        self._flush(pos)

    def finish(self):
        self._flush()

    def take(self):
        out, self.out = self.out, []
        return out


def make_finder(native, *args, **kwargs):
    if native:
        from . import _hostapi
        return _hostapi.CandidateFinder(*args, **kwargs)
    return CandidateFinderPy(*args, **kwargs)


def rows_from_sam(finder, handle, chunk_bytes=1 << 22):
    """Feed SAM text to a finder, yielding candidate rows (text, '\\n'-terminated chunks) as positions complete."""
    if isinstance(finder, CandidateFinderPy):
        for line in handle:
            finder.add_sam_line(line)
            if finder.out:
                yield "".join(row + "\n" for _, row in finder.take())
        finder.finish()
        if finder.out:
            yield "".join(row + "\n" for _, row in finder.take())
    else:
        for text in finder.text_from_sam(handle, chunk_bytes):
            yield text


def reference_region(ctg_name, ctg_start, ctg_end):
    """-> (region string for samtools, 1-based start of the loaded slice or None) (ExtractVariantCandidates.py:228-236)."""
    if ctg_name is not None and ctg_start is not None and ctg_end is not None:
        start = max(1, ctg_start - EXPAND_REFERENCE_REGION)
        return "%s:%d-%d" % (ctg_name, start, ctg_end + EXPAND_REFERENCE_REGION), start
    return ("%s" % ctg_name if ctg_name is not None else ""), None


def load_reference(samtools, ref_fn, region):
    try:
        p = subprocess_popen(shlex.split("%s faidx %s %s" % (samtools, ref_fn, region)))
    except OSError:
        return None
    rows = p.stdout.read().split("\n")
    p.stdout.close()
    p.wait()
    if p.returncode != 0:
        return None
    return "".join(r.rstrip() for r in rows[1:]).upper()


def make_candidates(args, native=True):
    """make_candidates (ExtractVariantCandidates.py:160-405), inference mode."""
    if args.gen4Training or args.var_fn is not None:
        sys.exit("[ERROR] --gen4Training / --var_fn build training sets by random sampling; this build covers variant calling only.")
    if not isfile("%s.fai" % args.ref_fn):
        print("Fasta index %s.fai doesn't exist." % args.ref_fn, file=sys.stderr)
        sys.exit(1)
    region, ref_start = reference_region(args.ctgName, args.ctgStart, args.ctgEnd)
    seq = load_reference(args.samtools, args.ref_fn, region)
    if not seq:
        print("[ERROR] Failed to load reference seqeunce from file (%s)." % args.ref_fn, file=sys.stderr)
        sys.exit(1)
    tree = bed_regions_from(args.bed_fn)
    if tree is not None and args.ctgName not in tree:
        print("[ERROR] ctg_name(%s) not exists in bed file(%s)." % (args.ctgName, args.bed_fn), file=sys.stderr)
        sys.exit(1)

    finder = make_finder(native, args.ctgName, seq, 0 if ref_start is None else ref_start - 1,
                         ctg_start=args.ctgStart if ref_start is not None else None,
                         ctg_end=args.ctgEnd if ref_start is not None else None,
                         bed=None if tree is None else tree[args.ctgName],
                         min_coverage=args.minCoverage, threshold=args.threshold, min_mq=args.minMQ)
    if getattr(args, "sam_fn", None):
        view, handle = None, open(args.sam_fn, "rb" if native else "r")
    else:
        view = subprocess_popen(shlex.split("%s view -F %d %s %s" % (args.samtools, SAMTOOLS_VIEW_FILTER_FLAG, args.bam_fn, region)),
                                text=not native)
        handle = view.stdout
    gz = raw = None
    if args.can_fn != "PIPE":
        raw = open(args.can_fn, "wb")
        gz = subprocess_popen(shlex.split("gzip -c"), stdin=subprocess.PIPE, stdout=raw)
        sink = gz.stdin
    else:
        sink = sys.stdout
    try:
        for text in rows_from_sam(finder, handle):
            sink.write(text)
    finally:
        handle.close()
        if view is not None:
            view.wait()
        if gz is not None:
            gz.stdin.close()
            gz.wait()
            raw.close()
        else:
            sink.flush()
    if finder.reads == 0:
        print("No read has been process, either the genome region you specified has no read cover, or please check the correctness of your BAM input (%s)."
              % args.bam_fn, file=sys.stderr)
        sys.exit(0)


def build_parser():
    """Flag names and defaults of dataPrepScripts/ExtractVariantCandidates.py:408-457 (help texts are this build's), plus two additions."""
    parser = ArgumentParser(description="Candidate sites (1-based) from sorted alignments")
    add = parser.add_argument
    add('--bam_fn', type=str, default="input.bam", help="sorted alignments")
    add('--ref_fn', type=str, default="ref.fa", help="reference FASTA with its .fai")
    add('--bed_fn', type=str, default=None, help="restrict candidates to these intervals (intersected with the region)")
    add('--can_fn', type=str, default="PIPE", help="output rows, gzip file; PIPE = standard output")
    add('--var_fn', type=str, default=None, help="training-set switch, not supported")
    add('--threshold', type=float, default=0.125, help="minimum frequency of the runner-up observation, default: %(default)f")
    add('--minCoverage', type=float, default=4, help="minimum depth, default: %(default)f")
    add('--minMQ', type=int, default=0, help="drop alignments below this mapping quality, default: %(default)d")
    add('--gen4Training', action='store_true', help="training-set switch, not supported")
    add('--outputProb', type=float, default=(7000000.0 * RATIO_OF_NON_VARIANT_TO_VARIANT / 3000000000), help="training-set switch, ignored")
    add('--ctgName', type=str, default="chr17", help="contig to process, default: %(default)s")
    add('--ctgStart', type=int, default=None, help="1-based first position of the region")
    add('--ctgEnd', type=int, default=None, help="1-based last position of the region (inclusive)")
    add('--samtools', type=str, default="samtools", help="samtools executable")
    # additions (not in the reference)
    add('--sam_fn', type=str, default=None, help="read alignments as SAM text from this file instead of spawning `samtools view`")
    add('--python_pileup', action='store_true', help="use the pure-Python tally instead of libclair_host.so (slow)")
    return parser


def main(argv=None):
    parser = build_parser()
    argv = sys.argv[1:] if argv is None else argv
    if len(argv) == 0:
        parser.print_help()
        sys.exit(1)
    args = parser.parse_args(argv)
    make_candidates(args, native=not args.python_pileup)


if __name__ == "__main__":
    main()
