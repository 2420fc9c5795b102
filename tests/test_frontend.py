"""Device front end, the parts that run without a GPU: the packer (libclair_host.so: clair_host_sampack_*) against
its Python twin, and the column formulation the kernels implement (oracle/frontend_np.py) against the reference-minted golden records
of both pileup stages and against the sequential host code on fresh synthetic alignments."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import frontend_cases as fc  # noqa: E402

from clair_amd import _hostapi  # noqa: E402
from oracle import frontend_np as fe  # noqa: E402


def columns_of(case, slabs=1, **pack_kw):
    packed = fe.pack_sam(case["sam"], case["ctg"], **pack_kw)
    col = fe.Columns(case["ref"], case["ref0"], case["ref0"] - 64, case["ref0"] + len(case["ref"]) + 64)
    col.add_reads(packed)
    return col, packed


# ---- the packer ---------------------------------------------------------------------------------------------------------------------
PACK_OPTIONS = [dict(), dict(dcov=3), dict(evc_min_mq=20, pile_min_mq=30), dict(pile_region=(500, 1500)), dict(pile_region=(1, 40), dcov=2)]


@pytest.mark.parametrize("opt", range(len(PACK_OPTIONS)))
def test_packer_matches_python_twin(opt):
    kw = PACK_OPTIONS[opt]
    case = fc.synth(seed=40 + opt, n_reads=400, ref_len=3000, dup_burst=8)
    want = fe.pack_sam(case["sam"], case["ctg"], **kw)
    p = _hostapi.SamPacker(case["ctg"], **kw)
    tail = b""
    for i in range(0, len(case["sam"]), 997):            # lines split across feeds
        tail = p.feed(tail + case["sam"][i:i + 997])
    assert p.feed(tail, final=True) == b""
    r, o, e, s = p.slab_arrays()
    st = p.stats()
    assert st["reads"] == len(want["pos0"]) > 100 and st["anomalies"] == want["anomalies"] == 0
    for name, got in (("pos0", r["pos0"]), ("flags", r["flags"]), ("seq0", r["seq0"]), ("seq_len", r["seq_len"]), ("op0", r["op0"]), ("n_ops", r["n_ops"]),
                      ("op_read", o["read"]), ("op_code", o["code_len"] & 3), ("op_len", o["code_len"] >> 2), ("op_ref", o["ref_off"]), ("op_q", o["q_off"]),
                      ("op_elem", e), ("seq", s)):
        assert np.array_equal(got, want[name]), name
    if "pile_region" in kw:
        assert 0 < st["pile_reads"] < st["evc_reads"]
    # slabs: taking the slab in the middle of the stream changes nothing but the offsets
    p2 = _hostapi.SamPacker(case["ctg"], **kw)
    half = case["sam"].index(b"\n", len(case["sam"]) // 2) + 1
    p2.feed(case["sam"][:half])
    r1 = p2.slab_arrays()[0]
    p2.reset()
    p2.feed(case["sam"][half:], final=True)
    r2 = p2.slab_arrays()[0]
    assert np.array_equal(np.concatenate([r1["flags"], r2["flags"]]), r["flags"]) and np.array_equal(np.concatenate([r1["pos0"], r2["pos0"]]), r["pos0"])
    assert r2["seq0"][0] == 0 and r2["op0"][0] == 0


def test_packer_reports_what_leaves_the_regime():
    ok = b"r1\t0\tchrS\t10\t60\t5M\t*\t0\t0\tACGTA\tIIIII\n"
    p = _hostapi.SamPacker("chrS")
    p.feed(ok + b"r2\t0\tchrS\t5\t60\t5M\t*\t0\t0\tACGTA\tIIIII\n", final=True)
    assert p.stats()["anomalies"] == fe.A_UNSORTED
    p = _hostapi.SamPacker("chrS")
    p.feed(b"r1\t0\tchrS\t10\t60\t3M0I2M\t*\t0\t0\tACGTA\tIIIII\n", final=True)
    assert p.stats()["anomalies"] == fe.A_ZERO_INDEL
    for cigar in (b"2M200000D3M", b"3000000000S5M"):
        p = _hostapi.SamPacker("chrS")
        p.feed(b"r1\t0\tchrS\t10\t60\t" + cigar + b"\t*\t0\t0\tACGTA\tIIIII\n", final=True)
        assert p.stats()["anomalies"] == fe.A_LONG_SPAN == fe.pack_sam(b"r1\t0\tchrS\t10\t60\t" + cigar + b"\t*\t0\t0\tACGTA\tIIIII\n", "chrS")["anomalies"]
    from clair_amd.create_tensor import PileupError
    for bad in (b"r1\t0\tchrS\n", b"r1\tx\tchrS\t10\t60\t5M\t*\t0\t0\tACGTA\tIIIII\n", b"\n"):
        with pytest.raises(PileupError):
            _hostapi.SamPacker("chrS").feed(bad, final=True)
    # an alignment neither stage walks leaves nothing behind
    p = _hostapi.SamPacker("chrS", evc_min_mq=10, pile_min_mq=10)
    p.feed(b"r1\t0\tchrS\t10\t5\t5M\t*\t0\t0\tACGTA\tIIIII\n" + ok, final=True)
    assert p.stats()["reads"] == 1 and p.stats()["ops"] == 1 and p.stats()["lines"] == 2


def _lead_indel_sam(first_cigar, later_cigar, later_mq=60, second_pos=102):
    """ten reads of 50M at POS 81, then two reads at POS `second_pos` -- the advisor's case: the sequential search
    evaluates position POS - 1 after EVERY accepted alignment, so a leading I / D of the second read at that POS stands alone"""
    rows = ["r%d\t0\tchrS\t81\t60\t50M\t*\t0\t0\t%s\t%s\n" % (i, "A" * 50, "I" * 50) for i in range(10)]
    for k, (cigar, mq) in enumerate(((first_cigar, 60), (later_cigar, later_mq))):
        n = sum(int(a) for a, op in __import__("re").findall(r"(\d+)([MIS=X])", cigar))
        rows.append("q%d\t0\tchrS\t%d\t%d\t%s\t*\t0\t0\t%s\t%s\n" % (k, second_pos, mq, cigar, "A" * n, "I" * n))
    return "".join(rows).encode()


@pytest.mark.parametrize("first,later,flagged", [("2I30M", "2I30M", True), ("30M", "2I30M", True), ("30M", "3D30M", True), ("30M", "2I3D30M", True),
                                                  ("2I30M", "30M", False), ("3D30M", "30M", False), ("30M", "4S2I30M", True), ("30M", "2M2I28M", False)])
def test_a_leading_indel_after_another_alignment_of_the_same_start_is_reported(first, later, flagged):
    sam = _lead_indel_sam(first, later)
    ref = "A" * 400
    case = dict(ctg="chrS", ref=ref, ref0=0, sam=sam)
    want = fc.host_candidates(case, min_coverage=4, threshold=0.125)                 # the sequential search, pinned to the reference's records
    packed = fe.pack_sam(sam, "chrS")
    p = _hostapi.SamPacker("chrS")
    assert p.feed(sam, final=True) == b""
    assert p.stats()["anomalies"] == packed["anomalies"] == (fe.A_LEAD_INDEL if flagged else 0)
    col = fe.Columns(ref, 0, 0, 400)
    col.add_reads(packed)
    got = col.candidates(min_depth=4, min_af=0.125)
    if not flagged:
        assert np.array_equal(got, want)                                              # one sum per position is what the reference computes
    elif first == later == "2I30M":
        assert len(want) == 0 and got.tolist() == [101]                               # ... and here it is not: 2/10 against 1/10 twice


def test_a_leading_indel_is_not_reported_when_the_earlier_alignment_is_not_the_searchs():
    sam = _lead_indel_sam("30M", "2I30M").replace(b"q0\t0\tchrS\t102\t60", b"q0\t0\tchrS\t102\t5")     # the first read at POS 102 fails --minMQ
    for kw in (dict(evc_min_mq=10), dict(evc_min_mq=10, pile_min_mq=0)):
        p = _hostapi.SamPacker("chrS", **kw)
        assert p.feed(sam, final=True) == b""
        assert p.stats()["anomalies"] == fe.pack_sam(sam, "chrS", **kw)["anomalies"] == 0
    assert fe.pack_sam(sam, "chrS")["anomalies"] == fe.A_LEAD_INDEL


# ---- the column formulation against the reference's records --------------------------------------------------------------------------
@pytest.mark.parametrize("path", fc.CT_GOLDEN, ids=[os.path.basename(p)[10:-8] for p in fc.CT_GOLDEN])
def test_columns_reproduce_reference_tensor_records(path):
    case = fc.ct_golden_case(path)
    col, _ = columns_of(case, dcov=case["dcov"], pile_min_mq=case["min_mq"], pile_region=case["pile_region"])
    w = col.windows(case["candidates"], min_cov=case["min_coverage"])
    if "unsorted" in path:
        assert w is None and col.anomalies == fe.A_CANDIDATES          # list order matters to the reference: the host path's business
        return
    if not case["left_edge"]:
        w = col.windows(case["candidates"], min_cov=case["min_coverage"], left_edge=False)
    # (a late leading I / D is the candidate search's concern: with a GIVEN candidate list the pileup is indifferent to it)
    assert col.anomalies == (fe.A_LEAD_INDEL if "lead_indel" in path else 0)
    assert fc.text_of(case["ctg"], w["centres"], w["refseq"], w["counts"]) == case["expected"]
    assert sum(int(t.sum()) for t in w["tuples"]) == int(w["totals"][w["opened"]].sum()) > 0


@pytest.mark.parametrize("path", fc.EVC_GOLDEN, ids=[os.path.basename(p)[11:-8] for p in fc.EVC_GOLDEN])
def test_columns_reproduce_reference_candidates(path):
    case = fc.evc_golden_case(path)
    col, _ = columns_of(case, evc_min_mq=case["min_mq"])
    bed = None
    if case["bed"] is not None:
        iv, st, en = sorted((s, e + 1 if e == s else e) for s, e in case["bed"]), [], []
        for s, e in iv:
            if e <= s:
                continue
            if st and s <= en[-1]:
                en[-1] = max(en[-1], e)
            else:
                st.append(s)
                en.append(e)
        bed = (np.array(st, np.int64), np.array(en, np.int64))
    got = col.candidates(min_depth=case["min_coverage"], min_af=case["threshold"], ctg_range=case["ctg_range"], bed=bed)
    if "lead_indel_late" in path:     # the reference evaluated some POS - 1 twice: reported (what that can change:
        assert col.anomalies == fe.A_LEAD_INDEL     # test_a_leading_indel_after_another_alignment_of_the_same_start_is_reported)
        return
    assert col.anomalies == 0 and len(got) > 10
    assert np.array_equal(got, case["expected_positions"])


# ---- ... and against the sequential host code on fresh alignments -------------------------------------------------------------------
SYNTH = [
    (101, dict(n_reads=160, ref_len=1800), dict()),
    (103, dict(n_reads=160, ref_len=1800, cand_step=(1, 12)), dict()),
    (5, dict(n_reads=300, ref_len=3000, dup_burst=6), dict(dcov=2, min_coverage=3)),
    (6, dict(n_reads=300, ref_len=3000), dict(min_mq=10)),
    (8, dict(n_reads=250, ref_len=2500, ins_rate=0.12, del_rate=0.1, cand_step=(1, 6)), dict()),
]


@pytest.mark.parametrize("left_edge", [True, False], ids=["left_edge", "no_left_edge"])
@pytest.mark.parametrize("k", range(len(SYNTH)))
def test_columns_equal_the_sequential_pileup(k, left_edge):
    seed, synth_kw, kw = SYNTH[k]
    case = fc.synth(seed, **synth_kw)
    hc, hs, hcounts = fc.host_windows(case, consider_left_edge=left_edge, **kw)
    col, _ = columns_of(case, dcov=kw.get("dcov", 250), pile_min_mq=kw.get("min_mq", 0))
    w = col.windows(case["candidates"], min_cov=kw.get("min_coverage", 0), left_edge=left_edge)
    assert col.anomalies == 0 and len(hc) > 50
    assert np.array_equal(hc, w["centres"]) and np.array_equal(hs, w["refseq"]) and np.array_equal(hcounts, w["counts"])
    assert sum(int(t.sum()) for t in w["tuples"]) == int(w["totals"][w["opened"]].sum())


def test_columns_with_a_region_take_the_alignments_samtools_would_print():
    case = fc.synth(21, n_reads=300, ref_len=3000)
    region = (700, 1900)
    cands = case["candidates"][(case["candidates"] >= region[0]) & (case["candidates"] <= region[1])]
    hc, hs, hcounts = fc.host_windows(case, candidates=cands, pile_region=region)
    col, packed = columns_of(case, pile_region=region)
    w = col.windows(cands)
    assert np.array_equal(hc, w["centres"]) and np.array_equal(hcounts, w["counts"]) and len(hc) > 20
    assert 0 < ((packed["flags"] & fe.F_PILE) != 0).sum() < len(packed["flags"])


@pytest.mark.parametrize("seed", [201, 207])
def test_columns_equal_the_sequential_candidate_search(seed):
    case = fc.synth(seed, n_reads=400, ref_len=2500, ins_rate=0.06 if seed == 207 else 0.02)
    kw = dict(ctg_start=200, ctg_end=2300, bed=[(0, 1000), (900, 1200), (1800, 1800), (2000, 2600)], min_coverage=3, threshold=0.1, min_mq=5)
    want = fc.host_candidates(case, **kw)
    col, _ = columns_of(case, evc_min_mq=5)
    got = col.candidates(min_depth=3, min_af=0.1, ctg_range=(200, 2300), bed=(np.array([0, 1800, 2000], np.int64), np.array([1200, 1801, 2600], np.int64)))
    assert np.array_equal(want, got) and len(want) > 30


def test_reference_case_whose_budget_binds_is_reported_with_the_real_budget():
    """tests/golden/pileup_ct_budget_binds: minted from the real script, whose 5 000 000-tuple budget runs out.  The column formulation cannot
    reproduce that (its result depends on the order bases are offered in, down to the interpreter's set order): the replay of the budget
    says so, which is what sends callVarBam to the sequential stage (tests/test_pileup.py pins THAT against the same records)."""
    case = fc.ct_golden_case(fc.BUDGET_GOLDEN)
    col, _ = columns_of(case, dcov=case["dcov"], pile_min_mq=case["min_mq"], pile_region=case["pile_region"])
    w = col.windows(case["candidates"], min_cov=case["min_coverage"])
    assert col.anomalies == fe.A_BUDGET                       # said by windows() itself, with the reference's 5 000 000
    totals = np.where(w["opened"], w["totals"], 0)
    assert fe.budget_binds(col.slabs, w["tuples"], case["candidates"], totals, 5000000)
    assert not fe.budget_binds(col.slabs, w["tuples"], case["candidates"], totals, 10 ** 9)
    assert fc.text_of(case["ctg"], w["centres"], w["refseq"], w["counts"]) != case["expected"]     # unbounded counts are not the reference's here


def test_budget_replay_is_safe_without_left_edge_windows():
    case = fc.synth(33, n_reads=200, ref_len=1500, cand_step=(1, 4))
    col, _ = columns_of(case)
    w = col.windows(case["candidates"], left_edge=False)
    free = fc.host_windows(case, consider_left_edge=False)
    assert np.array_equal(free[2], w["counts"]) and sum(int(t.sum()) for t in w["tuples"]) == int(w["totals"].sum())
    said_no = 0
    for slots in (200, 1000, 3000, 6000, 10000, 20000, 40000, 80000):
        binds = fe.budget_binds(col.slabs, w["tuples"], case["candidates"], np.where(w["opened"], w["totals"], 0), slots)
        same = all(np.array_equal(a, b) for a, b in zip(free, fc.host_windows(case, consider_left_edge=False, available_slots=slots)))
        assert binds or same, slots
        said_no += not binds
    assert 0 < said_no < 8


def test_budget_replay_is_safe():
    """Whenever the replay says the budget does not bind, the sequential code run WITH that budget gives the unbounded result; and
    it does say "binds" for budgets that change the result."""
    case = fc.synth(31, n_reads=200, ref_len=1500, cand_step=(1, 4))
    col, _ = columns_of(case)
    w = col.windows(case["candidates"])
    free = fc.host_windows(case)
    need = None
    for slots in (200, 1000, 3000, 6000, 10000, 20000, 40000, 80000):
        binds = fe.budget_binds(col.slabs, w["tuples"], case["candidates"], np.where(w["opened"], w["totals"], 0), slots)
        got = fc.host_windows(case, available_slots=slots)
        same = all(np.array_equal(a, b) for a, b in zip(free, got))
        assert binds or same, slots
        if not binds and need is None:
            need = slots
    assert need is not None and need > 1000                # the small budgets were reported, the large ones passed
    assert not all(np.array_equal(a, b) for a, b in zip(free, fc.host_windows(case, available_slots=200)))
    # the C replay (libclair_host.so) says the same as the Python one, slab by slab
    for slots in (3000, need, 80000):
        reads = np.zeros(len(col.slabs[0]["pos0"]), dtype=_hostapi.READ_DTYPE)
        reads["pos0"], reads["flags"] = col.slabs[0]["pos0"], col.slabs[0]["flags"]
        state = np.array([slots, 0], np.int64)
        got = _hostapi.tuple_budget_binds(reads, w["tuples"][0].astype(np.uint64), case["candidates"], np.where(w["opened"], w["totals"], 0).astype(np.uint64), state)
        assert got == fe.budget_binds(col.slabs, w["tuples"], case["candidates"], np.where(w["opened"], w["totals"], 0), slots)


def test_columns_flag_bases_the_reference_would_trip_over():
    case = fc.synth(3, n_reads=60, ref_len=900)
    sam = case["sam"].decode().splitlines()
    col7 = sam[5].split("\t")
    col7[9] = col7[9][:10] + "*" + col7[9][11:]
    bad = dict(case, sam=("\n".join(sam[:5] + ["\t".join(col7)] + sam[6:]) + "\n").encode())
    col, _ = columns_of(bad)
    assert col.anomalies & fe.A_BAD_BASE
    short = sam[7].split("\t")
    short[9] = short[9][:5]
    col, _ = columns_of(dict(case, sam=("\n".join(sam[:7] + ["\t".join(short)] + sam[8:]) + "\n").encode()))
    assert col.anomalies & fe.A_SEQ_OVERRUN
    col, _ = columns_of(dict(case, ref=case["ref"][:300] + "-" + case["ref"][301:]))
    assert col.anomalies & fe.A_BAD_REF


@pytest.mark.parametrize("block", range(4))
def test_differential_fuzz_of_the_column_formulation(block):
    """Random alignments x random options of both stages: the restatement = the sequential host code, candidates then windows."""
    for seed in range(block * 8, block * 8 + 8):
        case, pile_kw, evc_kw, region = fc.fuzz_case(seed)
        rng = dict(ctg_start=region[0], ctg_end=region[1]) if region else {}
        want_pos = fc.host_candidates(case, **rng, **evc_kw)
        col, _ = columns_of(case, dcov=pile_kw["dcov"], pile_min_mq=pile_kw["min_mq"], evc_min_mq=evc_kw["min_mq"], pile_region=region)
        bed = None
        if evc_kw["bed"] is not None:
            st, en = [], []
            for a, b in sorted((a, b + 1 if b == a else b) for a, b in evc_kw["bed"]):
                if st and a <= en[-1]:
                    en[-1] = max(en[-1], b)
                else:
                    st.append(a)
                    en.append(b)
            bed = (np.array(st, np.int64), np.array(en, np.int64))
        got_pos = col.candidates(min_depth=evc_kw["min_coverage"], min_af=evc_kw["threshold"], ctg_range=region, bed=bed)
        assert np.array_equal(want_pos, got_pos), seed
        hc, hs, hcounts = fc.host_windows(case, candidates=want_pos, pile_region=region, dcov=pile_kw["dcov"], min_mq=pile_kw["min_mq"],
                                          min_coverage=pile_kw["min_coverage"], consider_left_edge=seed % 3 != 0)
        w = col.windows(want_pos, min_cov=pile_kw["min_coverage"], left_edge=seed % 3 != 0)
        assert col.anomalies == 0, seed
        assert np.array_equal(hc, w["centres"]) and np.array_equal(hs, w["refseq"]) and np.array_equal(hcounts, w["counts"]), seed


def test_differential_fuzz_with_leading_indels():
    """Alignments that begin with an I / D: whenever the packer stays silent the one-sum-per-position candidates ARE the
    sequential search's; the windows over given candidates are the sequential pileup's either way."""
    silent = reported = 0
    for seed in range(300, 330):
        case, pile_kw, evc_kw, region = fc.fuzz_case(seed, lead_indel=0.25, lead_indel_late=bool(seed & 1))
        rng = dict(ctg_start=region[0], ctg_end=region[1]) if region else {}
        want_pos = fc.host_candidates(case, **rng, **{k: v for k, v in evc_kw.items() if k != "bed"})
        col, packed = columns_of(case, dcov=pile_kw["dcov"], pile_min_mq=pile_kw["min_mq"], evc_min_mq=evc_kw["min_mq"], pile_region=region)
        p = _hostapi.SamPacker(case["ctg"], dcov=pile_kw["dcov"], pile_min_mq=pile_kw["min_mq"], evc_min_mq=evc_kw["min_mq"], pile_region=region)
        assert p.feed(case["sam"], final=True) == b"" and p.stats()["anomalies"] == packed["anomalies"], seed
        assert packed["anomalies"] & ~fe.A_LEAD_INDEL == 0, seed
        got_pos = col.candidates(min_depth=evc_kw["min_coverage"], min_af=evc_kw["threshold"], ctg_range=region)
        if packed["anomalies"] & fe.A_LEAD_INDEL:
            reported += 1
        else:
            silent += 1
            assert np.array_equal(want_pos, got_pos), seed
        hc, hs, hcounts = fc.host_windows(case, candidates=want_pos, pile_region=region, dcov=pile_kw["dcov"], min_mq=pile_kw["min_mq"],
                                          min_coverage=pile_kw["min_coverage"])
        w = col.windows(want_pos, min_cov=pile_kw["min_coverage"])
        assert np.array_equal(hc, w["centres"]) and np.array_equal(hs, w["refseq"]) and np.array_equal(hcounts, w["counts"]), seed
    assert silent >= 8 and reported >= 4, (silent, reported)


# ---- callVarBam's device front end driver with a stand-in for the device (the restatement behind the Frontend interface) ---------------
class _StandInFrontend(object):
    """What clair_amd.callVarBam.DeviceFrontEnd uses of clair_amd._capi.Frontend, computed by oracle/frontend_np.py."""

    def __init__(self, device, ref, ref0, lo, hi):
        self.ref, self.ref0, self.lo, self.hi = ref, ref0, lo, hi
        self.text, self.chunks, self.col, self.w = [], [], None, None
        self.slab_reads = []

    def text_options(self, ctg, **kw):
        self.ctg, self.kw = ctg, kw

    def add_text(self, address, length):
        import ctypes
        piece = ctypes.string_at(address, length)
        assert piece.endswith(b"\n") and length > 0
        self.chunks.append(length)
        self.text.append(piece)

    def _columns(self):
        if self.col is None:
            self.packed = fe.pack_sam(b"".join(self.text), self.ctg, **self.kw)
            self.col = fe.Columns(self.ref, self.ref0, self.lo, self.hi)
            self.col.add_reads(self.packed)
        return self.col

    def text_stats(self):
        self._columns()
        fl = self.packed["flags"]
        return dict(lines=self.packed["lines"], evc_reads=int(((fl & fe.F_EVC) != 0).sum()), pile_reads=int(((fl & fe.F_PILE) != 0).sum()), anomalies=self.packed["anomalies"])

    def find_candidates(self, min_coverage=4, threshold=0.125, ctg_start=None, ctg_end=None, bed=None):
        assert bed is None
        self.pos = self._columns().candidates(min_depth=min_coverage, min_af=threshold, ctg_range=(ctg_start, ctg_end) if ctg_start is not None else None)
        return len(self.pos)

    def set_candidates(self, positions):
        self.pos = np.asarray(positions, np.int64)
        return len(self.pos)

    def build_windows(self, min_coverage=0, drop_non_iupac_centre=True, consider_left_edge=True):
        w = self._columns().windows(self.pos, min_cov=min_coverage, left_edge=consider_left_edge)
        keep = w["centre_ok"] if drop_non_iupac_centre else np.ones(len(w["centres"]), bool)
        self.w = {k: w[k][keep] for k in ("centres", "refseq", "counts")}
        return len(self.w["centres"])

    def stats(self):
        return dict(anomalies=self._columns().anomalies & ~fe.A_BUDGET, slabs=1, reads=len(self.packed["pos0"]), elements=0,
                    candidates=len(self.pos), windows=-1 if self.w is None else len(self.w["centres"]))

    def budget_binds(self, available_slots=5000000):
        return bool(self._columns().anomalies & fe.A_BUDGET)

    def window_info(self, first, n):
        return self.w["centres"][first:first + n], self.w["refseq"][first:first + n]

    def window_counts(self, first, n):
        return self.w["counts"][first:first + n].astype(np.int16)

    def counts_address(self, first):
        return 1

    def close(self):
        pass


@pytest.mark.parametrize("readers", [1, 3])
@pytest.mark.parametrize("region", [[], ["--ctgStart", "300", "--ctgEnd", "2500"]], ids=["contig", "region"])
@pytest.mark.parametrize("chunk", [3000, 1 << 20])
def test_callVarBam_device_driver_with_a_stand_in_device(tmp_path, monkeypatch, region, chunk, readers):
    """clair_amd.callVarBam.DeviceFrontEnd: the pipe read into a "page-locked" buffer in chunks, whole lines handed on, the unfinished
    line carried over, a stream that ends without a line end -- the batches it yields = those of the host stages (tensor_batches)."""
    import pileup_synth
    from clair_amd import _capi, callVarBam
    tmp = str(tmp_path)
    case = pileup_synth.synth_case(seed=301)
    fa, sam = os.path.join(tmp, "ref.fa"), os.path.join(tmp, "reads.sam")
    open(fa, "w").write(case["fasta"])
    open(fa + ".fai", "w").write("%s\t%d\t6\t60\t61\nchrOther\t120\t3100\t120\t121\n" % (case["ctg"], case["ref_len"]))
    open(sam, "w").write(case["sam"].rstrip("\n"))                       # no line end after the last alignment
    fake = "%s %s" % (sys.executable, os.path.join(HERE, "fake_samtools.py"))
    args = callVarBam.build_parser().parse_args(["--bam_fn", sam, "--ref_fn", fa, "--ctgName", case["ctg"], "--samtools", fake, "--threshold", "0.15",
                                                 "--minCoverage", "5", "--samtools_threads", "2", "--view_readers", str(readers)] + region)
    assert callVarBam.view_command(args, "x:1-2")[1:5] == [os.path.join(HERE, "fake_samtools.py"), "view", "-@", "2"]
    positions = callVarBam.candidate_positions(args, quiet=True)
    want = list(callVarBam.tensor_batches(args, positions, 64, progress=False))
    monkeypatch.setattr(_capi, "Frontend", _StandInFrontend)
    monkeypatch.setattr(callVarBam, "TEXT_CHUNK", chunk)
    d = callVarBam.DeviceFrontEnd(args, 0, pinned=lambda n: np.zeros(n, np.uint8))
    assert d.run() == sum(len(b[1]) for b in want) > 100
    assert len(d.frontend.chunks) == (1 if chunk > 100000 else len(d.frontend.chunks)) and (chunk > 100000 or len(d.frontend.chunks) > 10)
    got = list(d.batches(64, lean=False, progress=False))
    assert len(got) == len(want)
    for (gx, ginfo, gc), (wx, winfo, wc) in zip(got, want):
        assert np.array_equal(gx, wx) and np.array_equal(gc, wc)
        assert [list(map(str, r)) for r in ginfo] == [list(map(str, r)) for r in winfo]
    lean = list(d.batches(64, lean=True, progress=False))
    assert all(x is None and isinstance(c, _capi.DeviceWindows) and len(c) == len(i) for x, i, c in lean)


def test_callVarBamParallel_worker_with_stand_ins(tmp_path, monkeypatch):
    """callVarBamParallel's --run worker (one engine for all chunks, front ends read ahead on threads, chunks called in order) with the
    restatement standing in for the device front end and the oracle for the network: the per-chunk VCFs of callVarBam --front_end host."""
    import pileup_synth
    from test_decode import _CallsModel
    from clair_amd import _capi, callVarBam, callVarBamParallel as par, weights
    tmp = str(tmp_path)
    case = pileup_synth.synth_case(seed=77, n_reads=200)
    fa, sam = os.path.join(tmp, "ref.fa"), os.path.join(tmp, "reads.sam")
    open(fa, "w").write(case["fasta"])
    open(fa + ".fai", "w").write("%s\t%d\t6\t60\t61\n" % (case["ctg"], case["ref_len"]))
    open(sam, "w").write(case["sam"])
    open(os.path.join(tmp, "model.npz"), "w").close()
    fake = "%s %s" % (sys.executable, os.path.join(HERE, "fake_samtools.py"))
    w = weights.synthetic_weights(seed=4242, head_gain=6.0, lstm_bias_scale=0.1)

    class Model(_CallsModel):
        def submit_calls(self, slot, batch, centre, counts=False, with_probabilities=False):
            if isinstance(batch, _capi.DeviceWindows):
                batch, counts = batch.host(), True
            _CallsModel.submit_calls(self, slot, batch, centre, counts=counts, with_probabilities=with_probabilities)

        def close(self):
            pass
    models = []
    monkeypatch.setattr(callVarBam, "load_model", lambda args: models.append(Model(w)) or models[-1])
    monkeypatch.setattr(_capi, "Frontend", _StandInFrontend)
    monkeypatch.setattr(callVarBam, "TEXT_CHUNK", 50000)
    common = ["--chkpnt_fn", os.path.join(tmp, "model"), "--bam_fn", sam, "--ref_fn", fa, "--samtools", fake, "--includingAllContigs", "--refChunkSize", "800",
              "--threshold", "0.15", "--minCoverage", "5", "--batch_size", "64", "--python", "PY"]
    lines = par.commands(par.build_parser().parse_args(common + ["--output_prefix", os.path.join(tmp, "all", "var")]))
    os.makedirs(os.path.join(tmp, "all"))
    assert len(lines) == 4 and par.run_worker(lines, 0, 2) == 0 and len(models) == 1 and models[0].calls_submits > 4
    os.makedirs(os.path.join(tmp, "one"))
    import shlex
    rows = 0
    for line, (_, out) in zip(lines, par.commands.chunks):
        argv = shlex.split(line.replace(os.path.join(tmp, "all"), os.path.join(tmp, "one")))
        args = callVarBam.normalise(callVarBam.build_parser().parse_args(argv[argv.index("clair_amd.callVarBam") + 1:] + ["--front_end", "host"]))
        callVarBam.call_region(args, Model(w))
        a, b = open(out).read(), open(args.call_fn).read()
        assert a == b, out
        rows += len([x for x in a.splitlines() if not x.startswith("#")])
    assert rows > 20


def test_callVarBamParallel_worker_goes_on_after_a_failing_chunk(tmp_path, monkeypatch):
    """one chunk that fails (an exception or a stage's sys.exit) is reported and costs only its own VCF, as with the printed
    one-process-per-chunk commands; its front end is closed; the worker's exit code says so."""
    import pileup_synth
    from test_decode import _CallsModel
    from clair_amd import _capi, callVarBam, callVarBamParallel as par, weights
    tmp = str(tmp_path)
    case = pileup_synth.synth_case(seed=78, n_reads=200)
    fa, sam = os.path.join(tmp, "ref.fa"), os.path.join(tmp, "reads.sam")
    open(fa, "w").write(case["fasta"])
    open(fa + ".fai", "w").write("%s\t%d\t6\t60\t61\n" % (case["ctg"], case["ref_len"]))
    open(sam, "w").write(case["sam"])
    open(os.path.join(tmp, "model.npz"), "w").close()
    fake = "%s %s" % (sys.executable, os.path.join(HERE, "fake_samtools.py"))
    w = weights.synthetic_weights(seed=4242, head_gain=6.0, lstm_bias_scale=0.1)

    class Model(_CallsModel):
        def submit_calls(self, slot, batch, centre, counts=False, with_probabilities=False):
            if isinstance(batch, _capi.DeviceWindows):
                batch, counts = batch.host(), True
            _CallsModel.submit_calls(self, slot, batch, centre, counts=counts, with_probabilities=with_probabilities)

        def close(self):
            pass
    monkeypatch.setattr(callVarBam, "load_model", lambda args: Model(w))
    monkeypatch.setattr(_capi, "Frontend", _StandInFrontend)
    closed = []
    real_close = callVarBam.DeviceFrontEnd.close
    monkeypatch.setattr(callVarBam.DeviceFrontEnd, "close", lambda self: (closed.append(self.args.ctgStart), real_close(self))[1])
    real_call = callVarBam.call_region

    def flaky(args, m, prepared=None):
        if args.ctgStart == 800:
            sys.exit("chunk 2 fell over")          # without closing its front end
        if args.ctgStart == 1600:
            raise RuntimeError("so did chunk 3")
        return real_call(args, m, prepared=prepared)
    monkeypatch.setattr(callVarBam, "call_region", flaky)
    common = ["--chkpnt_fn", os.path.join(tmp, "model"), "--bam_fn", sam, "--ref_fn", fa, "--samtools", fake, "--includingAllContigs", "--refChunkSize", "800",
              "--threshold", "0.15", "--minCoverage", "5", "--batch_size", "64", "--python", "PY", "--output_prefix", os.path.join(tmp, "all", "var")]
    lines = par.commands(par.build_parser().parse_args(common))
    os.makedirs(os.path.join(tmp, "all"))
    assert len(lines) == 4 and par.run_worker(lines, 0, 2) == 1
    outs = [out for _, out in par.commands.chunks]
    assert [os.path.isfile(o) for o in outs] == [True, False, False, True]
    assert closed.count(800) >= 1 and closed.count(1600) >= 1
    assert sum(1 for row in open(outs[3]) if not row.startswith("#")) > 3


def test_samtools_view_args_reach_samtools_also_when_the_value_is_one_option(tmp_path):
    """`--samtools_view_args -x` is read by argparse as a flag without its value; callVarBamParallel writes the `=` form."""
    import shlex
    from clair_amd import callVarBam, callVarBamParallel as par
    tmp = str(tmp_path)
    for fn, text in (("ref.fa", ">chrS\nACGT\n"), ("ref.fa.fai", "chrS\t4\t6\t60\t61\n"), ("a.sam", ""), ("model.npz", "")):
        open(os.path.join(tmp, fn), "w").write(text)
    common = ["--chkpnt_fn", os.path.join(tmp, "model"), "--bam_fn", os.path.join(tmp, "a.sam"), "--ref_fn", os.path.join(tmp, "ref.fa"), "--includingAllContigs",
              "--output_prefix", os.path.join(tmp, "var"), "--python", "PY"]
    for value in ("--no-PG", "-x MM", "--keep-tag NM,MD"):
        (line,) = par.commands(par.build_parser().parse_args(common + ["--samtools_view_args=" + value]))
        argv = shlex.split(line)
        args = callVarBam.build_parser().parse_args(argv[argv.index("clair_amd.callVarBam") + 1:])
        assert args.samtools_view_args == value
        assert callVarBam.view_command(args, "chrS")[2:2 + len(value.split())] == value.split()
    # ... and samtools acts on them (the stand-in implements these three)
    import subprocess
    sam = os.path.join(tmp, "t.sam")
    open(sam, "w").write("r1\t0\tchrS\t1\t60\t4M\t*\t0\t0\tACGT\tIIII\tNM:i:0\tMM:Z:C+m\n")
    args = callVarBam.build_parser().parse_args(["--bam_fn", sam, "--samtools", "%s %s" % (sys.executable, os.path.join(HERE, "fake_samtools.py")), "--samtools_view_args=-x MM"])
    out = subprocess.run(callVarBam.view_command(args, "chrS"), capture_output=True, text=True, check=True).stdout
    assert out.endswith("IIII\tNM:i:0\n")


@pytest.mark.parametrize("readers", [2, 5])
def test_several_samtools_at_once_print_the_single_streams_lines(tmp_path, readers):
    """clair_amd.callVarBam.AlignmentStream: K `samtools view` over K consecutive pieces of a region = the one stream, line for line."""
    import pileup_synth
    from clair_amd import callVarBam
    tmp = str(tmp_path)
    case = pileup_synth.synth_case(seed=5, n_reads=500, ref_len=4000, read_len=(40, 900), dup_burst=7)
    sam = os.path.join(tmp, "reads.sam")
    open(sam, "w").write(case["sam"])
    fake = "%s %s" % (sys.executable, os.path.join(HERE, "fake_samtools.py"))
    args = callVarBam.build_parser().parse_args(["--bam_fn", sam, "--ctgName", case["ctg"], "--samtools", fake])
    for first, last in ((1, 4000), (700, 2900), (3990, 4000)):
        one = callVarBam._OnePipe(args, "%s:%d-%d" % (case["ctg"], first, last))
        want = one.read(1 << 30)
        assert one.finish() == 0
        many = callVarBam.AlignmentStream(args, case["ctg"], first, last, readers)
        got, buf = b"", bytearray(777)
        while True:
            n = many.readinto(memoryview(buf))
            if not n:
                break
            got += bytes(buf[:n])
        assert many.finish() == 0
        assert got == want and (want.count(b"\n") > 100 or first > 3000)


def test_callVarBamParallel_worker_reports_the_chunk_whose_samtools_fails(tmp_path, monkeypatch, caplog):
    """A `samtools view` that dies on one chunk makes THAT chunk fail -- no short VCF passed off as complete -- and the worker's exit code
    non-zero; the other chunks of the GPU are called (round 4, ADVICE r03: as the printed one-process-per-chunk commands would)."""
    import pileup_synth
    from test_decode import _CallsModel
    from clair_amd import _capi, callVarBam, callVarBamParallel as par, weights
    tmp = str(tmp_path)
    case = pileup_synth.synth_case(seed=78, n_reads=150)
    fa, sam = os.path.join(tmp, "ref.fa"), os.path.join(tmp, "reads.sam")
    open(fa, "w").write(case["fasta"])
    open(fa + ".fai", "w").write("%s\t%d\t6\t60\t61\n" % (case["ctg"], case["ref_len"]))
    open(sam, "w").write(case["sam"])
    open(os.path.join(tmp, "model.npz"), "w").close()
    flaky = os.path.join(tmp, "flaky.py")
    open(flaky, "w").write("import subprocess, sys\n"
                           "if sys.argv[1] == 'view' and any(a.startswith('%s:1598-') for a in sys.argv): sys.exit(3)\n"
                           "sys.exit(subprocess.run([sys.executable, %r] + sys.argv[1:]).returncode)\n" % (case["ctg"], os.path.join(HERE, "fake_samtools.py")))
    w = weights.synthetic_weights(seed=4242, head_gain=6.0, lstm_bias_scale=0.1)

    class Model(_CallsModel):
        def submit_calls(self, slot, batch, centre, counts=False, with_probabilities=False):
            if isinstance(batch, _capi.DeviceWindows):
                batch, counts = batch.host(), True
            _CallsModel.submit_calls(self, slot, batch, centre, counts=counts, with_probabilities=with_probabilities)

        def close(self):
            pass
    monkeypatch.setattr(callVarBam, "load_model", lambda args: Model(w))
    monkeypatch.setattr(_capi, "Frontend", _StandInFrontend)
    common = ["--chkpnt_fn", os.path.join(tmp, "model"), "--bam_fn", sam, "--ref_fn", fa, "--samtools", "%s %s" % (sys.executable, flaky), "--includingAllContigs",
              "--refChunkSize", "800", "--threshold", "0.15", "--minCoverage", "5", "--batch_size", "64", "--python", "PY"]
    lines = par.commands(par.build_parser().parse_args(common + ["--output_prefix", os.path.join(tmp, "all", "var")]))
    os.makedirs(os.path.join(tmp, "all"))
    assert any('--ctgStart "1600"' in l for l in lines)
    import logging
    with caplog.at_level(logging.ERROR):
        assert par.run_worker(lines, 0, 2) == 1
    assert "samtools view" in caplog.text and "var.chrS_1600_2400.vcf" in caplog.text
    done = [os.path.isfile(out) and any(not r.startswith("#") for r in open(out)) for _, out in par.commands.chunks]
    assert done.count(True) == len(done) - 1 and not done[2]


def test_device_driver_hands_back_when_the_region_does_not_fit(tmp_path, monkeypatch, caplog):
    """Out of device memory for the tables: `auto` runs the host stages and says so, `device` stops."""
    import logging
    import pileup_synth
    from clair_amd import _capi, callVarBam
    tmp = str(tmp_path)
    case = pileup_synth.synth_case(seed=1, n_reads=20)
    fa, sam = os.path.join(tmp, "r.fa"), os.path.join(tmp, "r.sam")
    open(fa, "w").write(case["fasta"])
    open(fa + ".fai", "w").write("%s\t%d\t6\t60\t61\n" % (case["ctg"], case["ref_len"]))
    open(sam, "w").write(case["sam"])

    class NoRoom(object):
        def __init__(self, *a):
            raise _capi.EngineError("clair_frontend_create failed: hipMalloc(read-base counters) failed: out of memory")
    monkeypatch.setattr(_capi, "Frontend", NoRoom)
    fake = "%s %s" % (sys.executable, os.path.join(HERE, "fake_samtools.py"))
    args = callVarBam.build_parser().parse_args(["--bam_fn", sam, "--ref_fn", fa, "--ctgName", case["ctg"], "--samtools", fake])
    with caplog.at_level(logging.INFO):
        assert callVarBam.DeviceFrontEnd(args, 0).run() is None
    assert "not enough device memory" in caplog.text
    args.front_end = "device"
    with pytest.raises(SystemExit, match="not enough device memory"):
        callVarBam.DeviceFrontEnd(args, 0).run()
