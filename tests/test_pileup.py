"""Pileup front end: clair_amd.create_tensor / libclair_host.so vs records minted from the real
dataPrepScripts/CreateTensor.py (tests/golden/pileup_ct_*.json.gz, tools/make_pileup_goldens.py), and the native code vs its
Python twin on fresh synthetic alignments."""
import glob
import gzip
import io
import json
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
import pileup_synth  # noqa: E402

from clair_amd import _hostapi, create_tensor as ct  # noqa: E402

FAKE_SAMTOOLS = "%s %s" % (sys.executable, os.path.join(HERE, "fake_samtools.py"))
GOLDEN = sorted(glob.glob(os.path.join(HERE, "golden", "pileup_ct_*.json.gz")))


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


def builder_inputs(doc):
    """What output_aln_tensor derives from the CLI arguments of a golden case."""
    args = ct.build_parser().parse_args(["--ctgName", doc["ctg"]] + doc["args"])
    full = parse_fasta(doc["fasta"], doc["ctg"])
    if args.ctgStart is not None and args.ctgEnd is not None:
        start = max(1, args.ctgStart - ct.EXPAND_REFERENCE_REGION)
        seq, ref0 = full[start - 1:args.ctgEnd + ct.EXPAND_REFERENCE_REGION].upper(), start - 1
    else:
        seq, ref0 = full.upper(), 0
    cands = ct.candidate_positions_from(io.StringIO(doc["candidates"]), args.ctgStart, args.ctgEnd)
    kw = dict(consider_left_edge=not args.stop_consider_left_edge, dcov=args.dcov, min_coverage=args.minCoverage, min_mq=args.minMQ)
    # what `samtools view -F 2316 <bam> <region>` would print
    sam = "".join(line + "\n" for line in doc["sam"].splitlines()
                  if not line.startswith("@") and not int(line.split("\t")[1]) & ct.SAMTOOLS_VIEW_FILTER_FLAG
                  and line.split("\t")[2] == doc["ctg"])
    return (doc["ctg"], seq, ref0, cands), kw, sam


def text_of(builder, ctg):
    return "".join(ct.format_record(ctg, c, s, t) + "\n" for c, s, t in builder.take())


def test_golden_files_present():
    assert len(GOLDEN) >= 6


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[10:-8] for p in GOLDEN])
def test_native_matches_reference_records(path):
    doc = load(path)
    if doc["args"] and "--ctgStart" in doc["args"]:
        pytest.skip("region cases go through the CLI test (samtools view selects the reads)")
    a, kw, sam = builder_inputs(doc)
    b = _hostapi.PileupBuilder(*a, **kw)
    got = "".join(b.text_from_sam(io.BytesIO(sam.encode())))
    assert got == doc["expected"]
    # the array hand-off holds the same numbers as the text
    b2 = _hostapi.PileupBuilder(*a, **kw)
    assert b2.feed(sam.encode()) == b""
    b2.finish()
    assert text_of(b2, doc["ctg"]) == doc["expected"]


@pytest.mark.parametrize("path", GOLDEN[:3] + GOLDEN[-1:], ids=[os.path.basename(p)[10:-8] for p in GOLDEN[:3] + GOLDEN[-1:]])
def test_python_twin_matches_reference_records(path):
    doc = load(path)
    if doc["args"] and "--ctgStart" in doc["args"]:
        pytest.skip("region cases go through the CLI test")
    a, kw, sam = builder_inputs(doc)
    b = ct.PileupBuilderPy(*a, **kw)
    got = "".join(ct.format_record(doc["ctg"], c, s, t) + "\n" for c, s, t in ct.records_from_sam(b, io.StringIO(sam)))
    assert got == doc["expected"]


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[10:-8] for p in GOLDEN])
def test_cli_matches_reference_stdout(path):
    """The whole command line, sub-processes included (`samtools` = tests/fake_samtools.py over text files)."""
    doc = load(path)
    with tempfile.TemporaryDirectory() as tmp:
        fa, sam, can, out = (os.path.join(tmp, n) for n in ("ref.fa", "reads.sam", "cands.txt.gz", "tensors.gz"))
        open(fa, "w").write(doc["fasta"])
        open(sam, "w").write(doc["sam"])
        args = ["--bam_fn", sam, "--ref_fn", fa, "--ctgName", doc["ctg"], "--samtools", FAKE_SAMTOOLS] + doc["args"]
        stdin = doc["candidates"]
        if doc["candidates_via_file"]:
            with gzip.open(can, "wt") as f:
                f.write(doc["candidates"])
            args += ["--can_fn", can, "--tensor_fn", out]      # and the gzip sink
            stdin = None
        r = subprocess.run([sys.executable, "-m", "clair_amd.create_tensor"] + args, input=stdin, capture_output=True, text=True,
                           cwd=ROOT)
        assert r.returncode == 0, r.stderr
        got = gzip.open(out, "rt").read() if doc["candidates_via_file"] else r.stdout
    assert got == doc["expected"]


OPTION_SETS = [
    dict(),
    dict(consider_left_edge=False, min_mq=10),
    dict(dcov=2, min_coverage=3),
    dict(available_slots=3000),          # the reference's tuple budget runs out: bases are dropped, windows still written
    dict(available_slots=40, consider_left_edge=False),
]


@pytest.mark.parametrize("seed", [101, 102, 103])
@pytest.mark.parametrize("opt", range(len(OPTION_SETS)))
def test_native_paths_match_python_twin(seed, opt):
    kw = OPTION_SETS[opt]
    case = pileup_synth.synth_case(seed=seed, n_reads=160, ref_len=1800, dup_burst=6 if opt == 2 else 0,
                                   cand_step=(1, 12) if seed == 103 else (3, 40))
    doc = {"ctg": case["ctg"], "args": [], "fasta": case["fasta"], "candidates": case["candidates"], "sam": case["sam"]}
    a, _, sam = builder_inputs(doc)
    py = ct.PileupBuilderPy(*a, **kw)
    want = "".join(ct.format_record(case["ctg"], c, s, t) + "\n" for c, s, t in ct.records_from_sam(py, io.StringIO(sam)))
    assert want.count("\n") > 20
    for general in (False, True):
        b = _hostapi.PileupBuilder(*a, force_general_path=general, **kw)
        assert b.stats()["sorted_path"] == (not general)
        got = "".join(b.text_from_sam(io.BytesIO(sam.encode()), chunk_bytes=257))     # lines split across feeds
        assert got == want, "general=%s" % general


def test_windows_are_released_as_reads_move_on():
    case = pileup_synth.synth_case(seed=7, n_reads=200, ref_len=4000, read_len=(40, 120))
    doc = {"ctg": case["ctg"], "args": [], "fasta": case["fasta"], "candidates": case["candidates"], "sam": case["sam"]}
    a, kw, sam = builder_inputs(doc)
    b = _hostapi.PileupBuilder(*a, **kw)
    lines = sam.encode().splitlines(keepends=True)
    half = b"".join(lines[:len(lines) // 2])
    assert b.feed(half) == b""
    st = b.stats()
    assert b.pending() > 0 and st["open_windows"] < 60 and st["slots_left"] > 5000000 - 200000
    centres, seqs, counts = b.take_arrays(5)
    assert len(centres) == 5 and counts.shape == (5, 33, 8, 4) and all(len(s) == 33 for s in seqs)
    assert (counts[:, :, :, 0] >= 0).all() and counts.sum() > 0


def test_malformed_alignments_are_reported():
    ref = "ACGT" * 50
    mk = lambda: _hostapi.PileupBuilder("c", ref, 0, [20, 198])  # noqa: E731
    with pytest.raises(ct.PileupError, match="columns"):
        mk().feed(b"r1\t0\tc\t1\t60\t10M\n")
    with pytest.raises(ct.PileupError, match="not an integer"):
        mk().feed(b"r1\t0\tc\tx\t60\t10M\t*\t0\t0\tACGTACGTAC\tIIIIIIIIII\n")
    with pytest.raises(ct.PileupError, match="past the end of SEQ"):
        mk().feed(b"r1\t0\tc\t1\t60\t30M\t*\t0\t0\tACGTACGTAC\tIIIIIIIIII\n")
    with pytest.raises(ct.PileupError, match="outside the loaded reference"):
        mk().feed(b"r1\t0\tc\t195\t60\t10M\t*\t0\t0\tACGTACGTAC\tIIIIIIIIII\n")
    with pytest.raises(ct.PileupError):
        ct.PileupBuilderPy("c", ref, 0, [20, 40]).add_sam_line("r1\t0\tc\t1\t60\t30M\t*\t0\t0\tACGTACGTAC\tIIIIIIIIII\n")
    b = mk()
    assert b.feed(b"@HD\tVN:1.6\nr1\t0\tc\t1\t60\t10M\t*\t0\t0\tACGTACGTAC\tIIIIIIIIII\npartial") == b"partial"


def test_cli_without_arguments_prints_help_and_exits_1():
    r = subprocess.run([sys.executable, "-m", "clair_amd.create_tensor"], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 1 and "--can_fn" in r.stdout


def test_cli_reports_missing_reference():
    with tempfile.TemporaryDirectory() as tmp:
        fa = os.path.join(tmp, "ref.fa")
        open(fa, "w").write(">chrA\nACGT\n")
        r = subprocess.run([sys.executable, "-m", "clair_amd.create_tensor", "--ref_fn", fa, "--ctgName", "nope", "--samtools",
                            FAKE_SAMTOOLS, "--bam_fn", fa], input="", capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 1 and "Failed to load reference" in r.stderr
