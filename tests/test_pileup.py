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
BUDGET_GOLDEN = os.path.join(HERE, "golden", "pileup_ct_budget_binds.json.gz")     # the reference's tuple budget binds: tests of their own below
GOLDEN = sorted(p for p in glob.glob(os.path.join(HERE, "golden", "pileup_ct_*.json.gz")) if p != BUDGET_GOLDEN)


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


@pytest.mark.skipif(sys.implementation.name != "cpython", reason="the restatement is of CPython's set")
def test_cpython_set_order_is_restated():
    """host_pileup.cpp: PySetOrder against the interpreter's own set -- random histories of add / remove over clustered and scattered ints,
    through dummies, probe-path reuse and table rebuilds; the iteration order must be the interpreter's after every history."""
    import random
    for trial in range(400):
        rng = random.Random(trial)
        live, s, ops = [], set(), []
        base = rng.randrange(0, 10 ** 8)
        for _ in range(rng.randrange(1, 500)):
            if live and rng.random() < 0.45:
                k = live.pop(rng.randrange(len(live)) if rng.random() < 0.3 else 0)      # mostly the oldest: windows close in the order they opened
                s.remove(k)
                ops.append(-(k + 1))
            else:
                k = base + rng.randrange(0, 80) if rng.random() < 0.7 else rng.randrange(0, 10 ** 9)
                if k not in s:
                    live.append(k)
                s.add(k)
                ops.append(k)
        assert _hostapi.pyset_order(ops) == list(s), trial


def test_budget_that_binds_reproduces_the_reference_records():
    """The reference's budget of 5 000 000 outstanding tuples BINDS in this golden case (64 reads of 3-3.9 kb over a candidate at every
    position).  Where it runs out in the middle of one read base, the windows the base still reaches are the first ones of the interpreter's
    set (CreateTensor.py:296-310): with CPython's order restated (set_order="cpython") the native builder -- sorted-candidates path and
    dict path -- and the Python twin (a real set) give the reference's records byte for byte; with insertion order (PyPy's sets, the
    default) only the records of those few bases differ."""
    doc = load(BUDGET_GOLDEN)
    assert doc["python"].startswith("3.")
    a, kw, sam = builder_inputs(doc)
    free = _hostapi.PileupBuilder(*a, available_slots=10 ** 9, **kw)
    unbounded = "".join(free.text_from_sam(io.BytesIO(sam.encode())))
    assert unbounded != doc["expected"]                                    # the budget does bind
    for general in (False, True):
        b = _hostapi.PileupBuilder(*a, set_order="cpython", force_general_path=general, **kw)
        assert "".join(b.text_from_sam(io.BytesIO(sam.encode()))) == doc["expected"]
    b = _hostapi.PileupBuilder(*a, **kw)                                   # insertion order
    got = "".join(b.text_from_sam(io.BytesIO(sam.encode()))).splitlines()
    want = doc["expected"].splitlines()
    differ = [i for i, (x, y) in enumerate(zip(got, want)) if x != y]
    assert len(got) == len(want) and 0 < len(differ) <= 40
    for i in differ:                                                       # same windows, same reference bases; a handful of counts moved
        gx, wx = got[i].split(), want[i].split()
        assert gx[:3] == wx[:3] and sum(x != y for x, y in zip(gx[3:], wx[3:])) <= 8
    if sys.implementation.name == "cpython":
        py = ct.PileupBuilderPy(*a, set_order="cpython", **kw)
        for line in sam.splitlines():
            py.add_sam_line(line)
        py.finish()
        assert text_of(py, doc["ctg"]) == doc["expected"]


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


# ---------------------------------------------------------------------------------------------------------------------
# candidate extraction (dataPrepScripts/ExtractVariantCandidates.py)
from clair_amd import extract_variant_candidates as evc  # noqa: E402

EVC_GOLDEN = sorted(glob.glob(os.path.join(HERE, "golden", "pileup_evc_*.json.gz")))


def run_evc_cli(doc, extra=()):
    with tempfile.TemporaryDirectory() as tmp:
        fa, sam, bedf, can = (os.path.join(tmp, n) for n in ("ref.fa", "reads.sam", "regions.bed", "cands.gz"))
        open(fa, "w").write(doc["fasta"])
        open(fa + ".fai", "w").write("%s\t%d\t6\t60\t61\n" % (doc["ctg"], doc["ref_len"]))
        open(sam, "w").write(doc["sam"])
        args = ["--bam_fn", sam, "--ref_fn", fa, "--ctgName", doc["ctg"], "--samtools", FAKE_SAMTOOLS] + doc["args"] + list(extra)
        if doc["bed"] is not None:
            open(bedf, "w").write(doc["bed"])
            args += ["--bed_fn", bedf]
        if doc["via_file"]:
            args += ["--can_fn", can]
        r = subprocess.run([sys.executable, "-m", "clair_amd.extract_variant_candidates"] + args, capture_output=True, text=True, cwd=ROOT)
        assert r.returncode == 0, r.stderr
        return gzip.open(can, "rt").read() if doc["via_file"] else r.stdout


def test_evc_golden_files_present():
    assert len(EVC_GOLDEN) >= 5


@pytest.mark.parametrize("path", EVC_GOLDEN, ids=[os.path.basename(p)[11:-8] for p in EVC_GOLDEN])
def test_evc_cli_matches_reference_stdout(path):
    doc = load(path)
    assert run_evc_cli(doc) == doc["expected"]


@pytest.mark.parametrize("path", EVC_GOLDEN[:2] + EVC_GOLDEN[-1:], ids=[os.path.basename(p)[11:-8] for p in EVC_GOLDEN[:2] + EVC_GOLDEN[-1:]])
def test_evc_python_twin_matches_reference_stdout(path):
    doc = load(path)
    assert run_evc_cli(doc, ["--python_pileup"]) == doc["expected"]


@pytest.mark.parametrize("seed", [201, 202])
def test_evc_native_matches_python_twin_and_chains_into_pileup(seed):
    case = pileup_synth.synth_case(seed=seed, n_reads=300, ref_len=2500)
    ref = parse_fasta(case["fasta"], case["ctg"]).upper()
    sam = "".join(l + "\n" for l in case["sam"].splitlines() if not l.startswith("@") and not int(l.split("\t")[1]) & 2316)
    kw = dict(ctg_start=200, ctg_end=2300, bed=[(0, 1000), (900, 1200), (1800, 1800), (2000, 2600)], min_coverage=3, threshold=0.1, min_mq=5)
    py = evc.CandidateFinderPy(case["ctg"], ref, 0, **kw)
    want = "".join(evc.rows_from_sam(py, io.StringIO(sam)))
    nat = _hostapi.CandidateFinder(case["ctg"], ref, 0, **kw)
    got = "".join(nat.text_from_sam(io.BytesIO(sam.encode()), chunk_bytes=311))
    assert got == want and want.count("\n") > 30 and nat.reads == py.reads
    # positions-only hand-off = column 2 of the rows; feeding them to the pileup builder works end to end
    nat2 = _hostapi.CandidateFinder(case["ctg"], ref, 0, **kw)
    assert nat2.feed(sam.encode()) == b""
    nat2.finish()
    pos = nat2.take_positions()
    assert pos.tolist() == [int(r.split()[1]) for r in want.splitlines()]
    b = _hostapi.PileupBuilder(case["ctg"], ref, 0, pos)
    b.feed(sam.encode())
    b.finish()
    centres, seqs, counts = b.take_arrays()
    assert centres.tolist() == [p for p in pos.tolist() if p - 17 >= 0]
    # centre column of a window: channel 0 summed over the 8 rows = matched depth = A+C+G+T(+N) tallies of the candidate row
    rows = {int(r.split()[1]): r.split() for r in want.splitlines()}
    for c, t in zip(centres.tolist()[:50], counts[:50]):
        tally = dict(zip(rows[c][4::2], map(int, rows[c][5::2])))
        # the finder skips mostly-clipped reads, the pileup does not: equal unless such a read covers the site
        assert t[16, :, 0].sum() >= tally["A"] + tally["C"] + tally["G"] + tally["T"] + tally["N"]


def test_evc_rejects_training_switches_and_missing_index():
    r = subprocess.run([sys.executable, "-m", "clair_amd.extract_variant_candidates", "--gen4Training"], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode != 0 and "training" in r.stderr
    with tempfile.TemporaryDirectory() as tmp:
        fa = os.path.join(tmp, "ref.fa")
        open(fa, "w").write(">chrA\nACGT\n")
        r = subprocess.run([sys.executable, "-m", "clair_amd.extract_variant_candidates", "--ref_fn", fa, "--ctgName", "chrA"],
                           capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 1 and ".fai doesn't exist" in r.stderr


# ---------------------------------------------------------------------------------------------------------------------
# the in-process BAM -> tensors path of clair_amd.callVarBam vs the text pipeline
def _bam_case(tmp, seed=301):
    case = pileup_synth.synth_case(seed=seed, n_reads=500, ref_len=3000)
    fa, sam = os.path.join(tmp, "ref.fa"), os.path.join(tmp, "reads.sam")
    open(fa, "w").write(case["fasta"])
    open(fa + ".fai", "w").write("%s\t%d\t6\t60\t61\nchrOther\t120\t3100\t120\t121\n" % (case["ctg"], case["ref_len"]))
    open(sam, "w").write(case["sam"])
    return case, fa, sam


@pytest.mark.parametrize("region", [[], ["--ctgStart", "300", "--ctgEnd", "2500"]], ids=["contig", "region"])
def test_in_process_tensor_batches_equal_the_text_pipeline(region):
    from clair_amd import callVarBam, utils
    with tempfile.TemporaryDirectory() as tmp:
        case, fa, sam = _bam_case(tmp)
        common = ["--bam_fn", sam, "--ref_fn", fa, "--ctgName", case["ctg"], "--samtools", FAKE_SAMTOOLS] + region
        r1 = subprocess.run([sys.executable, "-m", "clair_amd.extract_variant_candidates", "--threshold", "0.15", "--minCoverage", "5"] + common,
                            capture_output=True, text=True, cwd=ROOT)
        assert r1.returncode == 0, r1.stderr
        tensors = os.path.join(tmp, "t.gz")
        r2 = subprocess.run([sys.executable, "-m", "clair_amd.create_tensor", "--tensor_fn", tensors] + common, input=r1.stdout,
                            capture_output=True, text=True, cwd=ROOT)
        assert r2.returncode == 0, r2.stderr
        want = list(utils.tensor_generator_from(tensors, 64))
        args = callVarBam.build_parser().parse_args(common + ["--threshold", "0.15", "--minCoverage", "5", "--chkpnt_fn", "x", "--call_fn", "y"])
        pos = callVarBam.candidate_positions(args)
        assert pos.tolist() == [int(r.split()[1]) for r in r1.stdout.splitlines()]
        got = list(callVarBam.tensor_batches(args, pos, 64))
    assert len(want) > 2 and len(got) == len(want)
    for (xg, ig, cg), (xw, iw) in zip(got, want):
        assert xg.dtype == np.float32 and xg.shape == xw.shape and np.array_equal(xg, xw)
        assert cg.dtype == np.int16 and np.array_equal(cg[..., 0], xw[..., 0]) and np.array_equal(cg[..., 1:] - cg[..., 0:1], xw[..., 1:])
        assert [list(map(str, i)) for i in ig] == [list(map(str, i)) for i in iw]


@pytest.mark.parametrize("region", [[], ["--ctgStart", "300", "--ctgEnd", "2500"]], ids=["contig", "region"])
@pytest.mark.parametrize("workers", [2, 5])
def test_parallel_front_end_gives_the_windows_of_the_unsplit_run(region, workers, capfd):
    """callVarBam's host stages over consecutive sub-ranges on threads (each with its own samtools streams, alignments taken from
    33 positions beyond the sub-range) yield the candidates and windows of the single pass, in the same order -- also when a
    worker's samtools dies (the failure reaches the consumer instead of reading as end of input)."""
    from clair_amd import callVarBam
    with tempfile.TemporaryDirectory() as tmp:
        case, fa, sam = _bam_case(tmp, seed=307)
        common = ["--bam_fn", sam, "--ref_fn", fa, "--ctgName", case["ctg"], "--samtools", FAKE_SAMTOOLS] + region
        args = callVarBam.build_parser().parse_args(common + ["--threshold", "0.15", "--minCoverage", "5", "--chkpnt_fn", "x", "--call_fn", "y"])
        pos = callVarBam.candidate_positions(args)
        want = list(callVarBam.tensor_batches(args, pos, 64))
        lo, hi = (args.ctgStart, args.ctgEnd) if region else (1, callVarBam.contig_length(fa, case["ctg"]))
        assert hi == (2500 if region else case["ref_len"])
        capfd.readouterr()
        got = list(callVarBam.parallel_front_end(args, 64, workers, lo, hi))
        progress = [l for l in capfd.readouterr().err.splitlines() if l.startswith("Processed")]
        bad = callVarBam.build_parser().parse_args(common + ["--chkpnt_fn", "x", "--call_fn", "y"])
        bad.samtools = "%s %s" % (sys.executable, os.path.join(HERE, "no_such_tool.py"))
        with pytest.raises(SystemExit):
            list(callVarBam.parallel_front_end(bad, 64, workers, lo, hi))
    cat = lambda batches, k: np.concatenate([b[k] for b in batches])  # noqa: E731
    assert len(pos) > 100 and sum(len(b[1]) for b in got) == sum(len(b[1]) for b in want)
    assert np.array_equal(cat(got, 0), cat(want, 0)) and np.array_equal(cat(got, 2), cat(want, 2))
    assert [list(map(str, i)) for b in got for i in b[1]] == [list(map(str, i)) for b in want for i in b[1]]
    assert progress[-1] == "Processed %d tensors" % sum(len(b[1]) for b in want)


def test_vcf_sites_as_candidates():
    """--vcf_fn: the candidate stream = column 2 of what the reference's GetTruth prints (tests/golden/get_truth.json, minted by
    running dataPrepScripts/GetTruth.py here): '*' alternates add the base before, equal positions collapse, order as printed."""
    from clair_amd import callVarBam
    doc = json.load(open(os.path.join(HERE, "golden", "get_truth.json")))
    with tempfile.TemporaryDirectory() as tmp:
        vcf = os.path.join(tmp, "sites.vcf")
        open(vcf, "w").write(doc["vcf"])
        for name, case in doc["cases"].items():
            a = callVarBam.build_parser().parse_args(["--ctgName", "chrS"] + case["extra"])
            want = [int(r.split()[1]) for r in case["stdout"].splitlines()]
            assert callVarBam.positions_from_vcf(vcf, "chrS", a.ctgStart, a.ctgEnd).tolist() == want, name
        assert 499 in want or name != "all"


def test_callVarBamParallel_commands_match_reference():
    """Chunking, contig selection, bed filtering and option spelling vs the reference's own output
    (tests/golden/parallel_cmds.json, tools/make_pileup_goldens.py); only the program name differs."""
    from clair_amd import callVarBamParallel as par
    doc = json.load(open(os.path.join(HERE, "golden", "parallel_cmds.json")))
    for name, case in doc["cases"].items():
        with tempfile.TemporaryDirectory() as tmp:
            for fn, text in (("ref.fa", ">x\n"), ("ref.fa.fai", doc["fai"]), ("a.bam", ""), ("model.meta", ""), ("r.bed", doc["bed"])):
                open(os.path.join(tmp, fn), "w").write(text)
            argv = ["--chkpnt_fn", os.path.join(tmp, "model"), "--ref_fn", os.path.join(tmp, "ref.fa"), "--bam_fn", os.path.join(tmp, "a.bam"),
                    "--output_prefix", os.path.join(tmp, "out", "var"), "--pypy", "python3", "--samtools", "gzip", "--python", "PY"] + case["extra"]
            if case["use_bed"]:
                argv += ["--bed_fn", os.path.join(tmp, "r.bed")]
            got = par.commands(par.build_parser().parse_args(argv))
            want = [l.replace("@TMP@", tmp).replace("python /root/reference/clair/../clair.py callVarBam", "PY -m clair_amd.callVarBam")
                    for l in case["expected"].splitlines()[2:]]
            assert got == want, name
            dealt = par.commands(par.build_parser().parse_args(argv + ["--devices", "8"]))
            assert [l.rsplit(" ", 2)[1:] for l in dealt] == [["--device", '"%d"' % (i % 8)] for i in range(len(dealt))]
            passed = par.commands(par.build_parser().parse_args(argv + ["--front_end", "host", "--batch_size", "4096"]))
            assert all(' --front_end "host" --batch_size "4096" ' in l for l in passed) and len(passed) == len(got)


def test_binary_tensor_records_equal_the_text_records(tmp_path):
    """create_tensor --binary -> utils.tensor_generator_from: the same batches as the text records of the same windows, plus the
    raw int16 counts for the GPU boundary (clair_amd/tensor_binary.py)."""
    from clair_amd import tensor_binary, utils
    case, fa, sam = _bam_case(str(tmp_path), seed=303)
    common = ["--bam_fn", sam, "--ref_fn", fa, "--ctgName", case["ctg"], "--samtools", FAKE_SAMTOOLS]
    text, binary = str(tmp_path / "t.txt.gz"), str(tmp_path / "t.bin.gz")
    for extra in (["--tensor_fn", text], ["--tensor_fn", binary, "--binary"]):
        r = subprocess.run([sys.executable, "-m", "clair_amd.create_tensor"] + common + extra, input=case["candidates"], capture_output=True,
                           text=True, cwd=ROOT)
        assert r.returncode == 0, r.stderr
    assert gzip.open(binary, "rb").read(8) == tensor_binary.MAGIC
    want = list(utils.tensor_generator_from(text, 50))
    got = list(utils.tensor_generator_from(binary, 50))
    assert len(want) > 2 and len(got) == len(want)
    for (xw, iw), (xg, ig, cg) in zip(want, got):
        assert np.array_equal(xw, xg) and [list(map(str, i)) for i in iw] == ig
        assert cg.dtype == np.int16 and np.array_equal(cg[..., 0], xg[..., 0])
    # stdout sink and the Python twin write the same bytes
    r1 = subprocess.run([sys.executable, "-m", "clair_amd.create_tensor", "--binary"] + common, input=case["candidates"].encode(),
                        capture_output=True, cwd=ROOT)
    r2 = subprocess.run([sys.executable, "-m", "clair_amd.create_tensor", "--binary", "--python_pileup"] + common,
                        input=case["candidates"].encode(), capture_output=True, cwd=ROOT)
    assert r1.returncode == 0 and r1.stdout == gzip.open(binary, "rb").read() and r2.stdout == r1.stdout
    with pytest.raises(ValueError):
        tensor_binary.pack_records("c" * 40, [1], ["A" * 33], np.zeros((1, 33, 8, 4), np.int32))
    with pytest.raises(ValueError):
        tensor_binary.pack_records("c", [1], ["A" * 33], np.full((1, 33, 8, 4), 40000, np.int32))
    with pytest.raises(ValueError):
        list(tensor_binary.read_batches(io.BytesIO(b"x" * 100), 4))


def test_cli_mirrors_keep_every_reference_flag_and_default():
    """Option strings, defaults, types, nargs and action kinds of the four reference command lines
    (tests/golden/cli_flags.json, tools/make_cli_flag_golden.py); this build may add flags, never drop or change one."""
    from clair_amd import callVarBam, callVarBamParallel
    ref = json.load(open(os.path.join(HERE, "golden", "cli_flags.json")))
    mine = {"CreateTensor": ct.build_parser(), "ExtractVariantCandidates": evc.build_parser(), "callVarBam": callVarBam.build_parser(),
            "callVarBamParallel": callVarBamParallel.build_parser()}
    for tool, flags in ref.items():
        have = {tuple(a.option_strings): a for a in mine[tool]._actions if a.option_strings}
        for f in flags:
            a = have.get(tuple(f["flags"]))
            assert a is not None, (tool, f["flags"])
            assert a.default == f["default"] and getattr(a.type, "__name__", None) == f["type"] and a.nargs == f["nargs"] \
                and type(a).__name__ == f["action"], (tool, f["flags"])


def test_submodule_dispatcher():
    """python -m clair_amd <submodule> ... = the reference's `python clair.py <submodule> ...` (clair.py:60-86)."""
    r = subprocess.run([sys.executable, "-m", "clair_amd"], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0 and "callVarBam" in r.stdout and "CreateTensor" in r.stdout
    r = subprocess.run([sys.executable, "-m", "clair_amd", "CreateTensor"], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 1 and "--can_fn" in r.stdout            # the submodule's own help, exit 1 without options
    r = subprocess.run([sys.executable, "-m", "clair_amd", "train"], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode != 0 and "outside this build" in r.stderr
    r = subprocess.run([sys.executable, "-m", "clair_amd", "nope"], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode != 0 and "not found" in r.stderr


def test_set_order_follows_the_interpreter_the_reference_would_run_under(monkeypatch):
    """clair/callVarBam.py starts the pileup script with --pypy (default pypy3): insertion-ordered sets; any other interpreter name means CPython."""
    monkeypatch.delenv("CLAIR_AMD_SET_ORDER", raising=False)
    assert ct.set_order_of(None) == ct.set_order_of("pypy3") == ct.set_order_of("/opt/pypy3.6/bin/pypy3") == "ascending"
    assert ct.set_order_of("python3") == ct.set_order_of("/usr/bin/python") == "cpython"
    monkeypatch.setenv("CLAIR_AMD_SET_ORDER", "cpython")
    assert ct.set_order_of("pypy3") == "cpython"
    with pytest.raises(ValueError):
        ct.PileupBuilderPy("c", "ACGT", 0, [], set_order="sorted")
    with pytest.raises(ValueError):
        _hostapi.PileupBuilder("c", "ACGT", 0, [], set_order="sorted")
