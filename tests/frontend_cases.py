"""Inputs and expected outputs for the device front end's tests (test support): the reference-minted golden records of the two
pileup stages (tests/golden/pileup_ct_*, pileup_evc_*), fresh synthetic alignments (tests/pileup_synth.py), and the sequential host
code (libclair_host.so, itself pinned against those golden records) as the source of expected values for the synthetic ones."""
import glob
import gzip
import io
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import pileup_synth  # noqa: E402

from clair_amd import _hostapi, create_tensor as ct, extract_variant_candidates as evc  # noqa: E402

BUDGET_GOLDEN = os.path.join(HERE, "golden", "pileup_ct_budget_binds.json.gz")     # the reference's tuple budget binds: tests of their own
CT_GOLDEN = sorted(p for p in glob.glob(os.path.join(HERE, "golden", "pileup_ct_*.json.gz")) if p != BUDGET_GOLDEN)
EVC_GOLDEN = sorted(glob.glob(os.path.join(HERE, "golden", "pileup_evc_*.json.gz")))
VIEW_FILTER = ct.SAMTOOLS_VIEW_FILTER_FLAG


def load(path):
    with gzip.open(path, "rt") as f:
        return json.load(f)


def parse_fasta(text, ctg):
    seq, on = [], False
    for line in text.splitlines():
        if line.startswith(">"):
            on = line[1:].split()[0] == ctg
        elif on:
            seq.append(line)
    return "".join(seq)


def viewed(sam_text, ctg=None):
    """what `samtools view -F 2316 <bam> [ctg]` prints of a SAM file: no header, no filtered flags, one contig"""
    keep = []
    for line in sam_text.splitlines():
        if line.startswith("@"):
            continue
        col = line.split("\t")
        if int(col[1]) & VIEW_FILTER or (ctg is not None and col[2] != ctg):
            continue
        keep.append(line + "\n")
    return "".join(keep).encode()


def reference_of(fasta, ctg, ctg_start=None, ctg_end=None):
    """(sequence upper-cased, 0-based start): what both stages load with `samtools faidx` (region widened by 1 Mbp)"""
    full = parse_fasta(fasta, ctg)
    if ctg_start is not None and ctg_end is not None:
        start = max(1, ctg_start - ct.EXPAND_REFERENCE_REGION)
        return full[start - 1:ctg_end + ct.EXPAND_REFERENCE_REGION].upper(), start - 1
    return full.upper(), 0


def ct_golden_case(path):
    """-> dict(ctg, ref, ref0, candidates int64, sam bytes (the contig's alignments), pile_region, kw of the builder, expected text)"""
    doc = load(path)
    args = ct.build_parser().parse_args(["--ctgName", doc["ctg"]] + doc["args"])
    ref, ref0 = reference_of(doc["fasta"], doc["ctg"], args.ctgStart, args.ctgEnd)
    cands = np.array(ct.candidate_positions_from(io.StringIO(doc["candidates"]), args.ctgStart, args.ctgEnd), np.int64)
    region = (args.ctgStart, args.ctgEnd) if args.ctgStart is not None and args.ctgEnd is not None else None
    return dict(ctg=doc["ctg"], ref=ref, ref0=ref0, candidates=cands, sam=viewed(doc["sam"], doc["ctg"]), pile_region=region,
                left_edge=not args.stop_consider_left_edge, dcov=args.dcov, min_coverage=args.minCoverage, min_mq=args.minMQ,
                expected=doc["expected"])


def evc_golden_case(path):
    doc = load(path)
    args = evc.build_parser().parse_args(["--ctgName", doc["ctg"]] + doc["args"])
    have = args.ctgStart is not None and args.ctgEnd is not None
    ref, ref0 = reference_of(doc["fasta"], doc["ctg"], args.ctgStart if have else None, args.ctgEnd if have else None)
    bed = None
    if doc["bed"] is not None:
        bed = [(int(r.split()[1]), int(r.split()[2])) for r in doc["bed"].splitlines() if r.split() and r.split()[0] == doc["ctg"]]
    return dict(ctg=doc["ctg"], ref=ref, ref0=ref0, sam=viewed(doc["sam"], doc["ctg"]), ctg_range=(args.ctgStart, args.ctgEnd) if have else None,
                bed=bed, min_coverage=args.minCoverage, threshold=args.threshold, min_mq=args.minMQ,
                expected_positions=np.array([int(r.split()[1]) for r in doc["expected"].splitlines()], np.int64))


def synth(seed, **kw):
    case = pileup_synth.synth_case(seed=seed, **kw)
    ref, ref0 = reference_of(case["fasta"], case["ctg"])
    cands = np.array(ct.candidate_positions_from(io.StringIO(case["candidates"]), None, None), np.int64)
    return dict(ctg=case["ctg"], ref=ref, ref0=ref0, candidates=cands, sam=viewed(case["sam"], case["ctg"]))


def host_windows(case, candidates=None, pile_region=None, **kw):
    """The sequential builder's windows -> (centres, refseq uint8 [n,34], counts int32 [n,33,8,4]).  pile_region: only the alignments
    `samtools view ctg:start-end` prints (tests/fake_samtools.py's rule = htslib's)."""
    sam = case["sam"]
    if pile_region is not None:
        import re
        keep = []
        for line in sam.decode().splitlines():
            col = line.split("\t")
            span = sum(int(n) for n, op in re.findall(r"(\d+)([MIDNSHP=X])", col[5]) if op in "MDN=X")
            pos = int(col[3])
            if pos <= pile_region[1] and pos + max(span, 1) - 1 >= pile_region[0]:
                keep.append(line + "\n")
        sam = "".join(keep).encode()
    b = _hostapi.PileupBuilder(case["ctg"], case["ref"], case["ref0"], case["candidates"] if candidates is None else candidates, **kw)
    assert b.feed(sam) == b""
    b.finish()
    return b.take_columns()


def host_candidates(case, **kw):
    f = _hostapi.CandidateFinder(case["ctg"], case["ref"], case["ref0"], **kw)
    assert f.feed(case["sam"]) == b""
    f.finish()
    return f.take_positions()


def text_of(ctg, centres, seqs, counts):
    """windows as CreateTensor's records (CreateTensor.py:60-65)"""
    raw = np.ascontiguousarray(seqs).tobytes()
    return "".join(ct.format_record(ctg, int(c), raw[i * 34:i * 34 + 34].split(b"\0", 1)[0].decode("latin-1"), counts[i]) + "\n"
                   for i, c in enumerate(centres.tolist()))


def fuzz_case(seed, **synth_kw):
    """A random small case with random options of both stages -> (case, pileup kw, candidate-search kw, region or None).
    synth_kw: further options of pileup_synth.synth_case (lead_indel: drawn from a generator of its own, the rest of the case is unchanged)."""
    rng = np.random.default_rng(1000 + seed)
    ref_len = int(rng.integers(600, 2500))
    case = synth(seed, n_reads=int(rng.integers(20, 260)), ref_len=ref_len, read_len=(30, int(rng.integers(60, 400))),
                 cand_step=(1, int(rng.integers(3, 50))), sub_rate=float(rng.choice([0.01, 0.04, 0.15])), ins_rate=float(rng.choice([0.0, 0.02, 0.1])),
                 del_rate=float(rng.choice([0.0, 0.02, 0.1])), dup_burst=int(rng.choice([0, 0, 5, 12])), skip_ops=bool(rng.integers(0, 2)), **synth_kw)
    pile_kw = dict(dcov=int(rng.choice([250, 250, 1, 3])), min_mq=int(rng.choice([0, 0, 10, 40])), min_coverage=int(rng.choice([0, 0, 2, 6])))
    evc_kw = dict(min_coverage=float(rng.choice([4, 1, 8, 2.5])), threshold=float(rng.choice([0.125, 0.05, 0.3])), min_mq=int(rng.choice([0, 0, 15])))
    region = None
    if rng.random() < 0.4:
        a = int(rng.integers(1, ref_len - 100))
        region = (a, int(rng.integers(a, ref_len + 50)))
    bed = None
    if rng.random() < 0.3:
        bed = sorted((int(x), int(x + rng.integers(0, 300))) for x in rng.integers(0, ref_len, int(rng.integers(1, 5))))
    evc_kw["bed"] = bed
    return case, pile_kw, evc_kw, region
