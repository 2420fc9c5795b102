"""Device front end on the GPU (clair_amd/csrc/frontend.hip through the C ABI): candidates and pileup windows bit for bit against the
reference-minted golden records, the sequential host code and the NumPy restatement (oracle/frontend_np.py), the tuple counts the
budget replay needs, the CLAIR_FE_* reports, and the hand-off of the windows to the engine without leaving HBM."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import frontend_cases as fc  # noqa: E402

from clair_amd import _capi, _hostapi  # noqa: E402
from oracle import frontend_np as fe  # noqa: E402

pytestmark = pytest.mark.gpu


def device_frontend(case, slabs=1, margin=64, **pack_kw):
    """Pack the case's alignments (in `slabs` pieces) and put them on the device."""
    f = _capi.Frontend(0, case["ref"], case["ref0"], case["ref0"] - margin, case["ref0"] + len(case["ref"]) + margin)
    p = _hostapi.SamPacker(case["ctg"], **pack_kw)
    sam, at = case["sam"], 0
    for k in range(slabs):
        cut = len(sam) if k == slabs - 1 else sam.index(b"\n", len(sam) * (k + 1) // slabs) + 1
        assert p.feed(sam[at:cut], final=(k == slabs - 1)) == b""
        at = cut
        f.add_slab(p)
    f.host_anomalies = p.stats()["anomalies"]
    return f


def windows_of(f):
    n = f.stats()["windows"]
    centres, seqs = f.window_info(0, n)
    return centres, seqs, f.window_counts(0, n).astype(np.int32)


@pytest.mark.parametrize("path", [p for p in fc.CT_GOLDEN if "unsorted" not in p], ids=lambda p: os.path.basename(p)[10:-8])
def test_windows_reproduce_reference_tensor_records(path):
    case = fc.ct_golden_case(path)
    f = device_frontend(case, slabs=3, dcov=case["dcov"], pile_min_mq=case["min_mq"], pile_region=case["pile_region"])
    assert f.set_candidates(case["candidates"]) == len(case["candidates"])
    n = f.build_windows(min_coverage=case["min_coverage"], drop_non_iupac_centre=False, consider_left_edge=case["left_edge"])
    assert f.stats()["anomalies"] == 0 and f.host_anomalies == (fe.A_LEAD_INDEL if "lead_indel" in path else 0) and not f.budget_binds()
    centres, seqs, counts = windows_of(f)
    assert n == len(centres) > 20
    assert fc.text_of(case["ctg"], centres, seqs, counts) == case["expected"]


@pytest.mark.parametrize("path", fc.EVC_GOLDEN, ids=lambda p: os.path.basename(p)[11:-8])
def test_candidates_reproduce_reference_rows(path):
    case = fc.evc_golden_case(path)
    f = device_frontend(case, slabs=2, evc_min_mq=case["min_mq"])
    rng = case["ctg_range"] or (None, None)
    n = f.find_candidates(min_coverage=case["min_coverage"], threshold=case["threshold"], ctg_start=rng[0], ctg_end=rng[1], bed=case["bed"])
    assert f.stats()["anomalies"] == 0
    got = f.candidates()
    if "lead_indel_late" in path:     # reported by the packer (and by the device's own text pass, below): the sequential stage's regime
        assert f.host_anomalies == fe.A_LEAD_INDEL
        g = device_frontend_text(case, chunks=3, evc_min_mq=case["min_mq"])
        assert g.text_stats()["anomalies"] == fe.A_LEAD_INDEL
        return
    assert f.host_anomalies == 0
    assert n == len(got) and np.array_equal(got, case["expected_positions"])


SYNTH = [
    (101, dict(n_reads=160, ref_len=1800), dict(), 1),
    (103, dict(n_reads=160, ref_len=1800, cand_step=(1, 12)), dict(), 2),
    (5, dict(n_reads=300, ref_len=3000, dup_burst=6), dict(dcov=2, min_coverage=3), 3),
    (6, dict(n_reads=300, ref_len=3000), dict(min_mq=10), 1),
    (8, dict(n_reads=250, ref_len=2500, ins_rate=0.12, del_rate=0.1, cand_step=(1, 6)), dict(), 4),
    (9, dict(n_reads=2500, ref_len=30000, read_len=(200, 3000), cand_step=(1, 30)), dict(), 5),
]


@pytest.mark.parametrize("left_edge", [True, False], ids=["left_edge", "no_left_edge"])
@pytest.mark.parametrize("k", range(len(SYNTH)))
def test_windows_equal_the_sequential_pileup_and_the_restatement(k, left_edge):
    seed, synth_kw, kw, slabs = SYNTH[k]
    case = fc.synth(seed, **synth_kw)
    hc, hs, hcounts = fc.host_windows(case, consider_left_edge=left_edge, **kw)
    f = device_frontend(case, slabs=slabs, dcov=kw.get("dcov", 250), pile_min_mq=kw.get("min_mq", 0))
    f.set_candidates(case["candidates"])
    f.build_windows(min_coverage=kw.get("min_coverage", 0), drop_non_iupac_centre=False, consider_left_edge=left_edge)
    assert f.stats()["anomalies"] == 0 and not f.budget_binds()
    centres, seqs, counts = windows_of(f)
    assert np.array_equal(hc, centres) and np.array_equal(hs, seqs) and np.array_equal(hcounts, counts) and len(hc) > 50
    # the tuple counts behind the budget replay, against the NumPy restatement: per alignment and per window
    packed = fe.pack_sam(case["sam"], case["ctg"], dcov=kw.get("dcov", 250), pile_min_mq=kw.get("min_mq", 0))
    col = fe.Columns(case["ref"], case["ref0"], case["ref0"] - 64, case["ref0"] + len(case["ref"]) + 64)
    col.add_reads(packed)
    w = col.windows(case["candidates"], min_cov=kw.get("min_coverage", 0), left_edge=left_edge)
    got = np.concatenate([f.read_tuples(s) for s in range(f.stats()["slabs"])])
    assert np.array_equal(got.astype(np.int64), w["tuples"][0])
    cc, wt = f.window_tuples()
    assert np.array_equal(cc, case["candidates"]) and np.array_equal(wt.astype(np.int64), np.where(w["opened"], w["totals"], 0))
    assert int(got.sum()) == int(wt.sum()) > 0


@pytest.mark.parametrize("margin", [64, 300, 511])
@pytest.mark.parametrize("text", [False, True], ids=["packed", "text"])
def test_tiled_tally_equals_the_per_base_tally(margin, text, monkeypatch):
    """Pass 1 by position tiles in LDS (the default) against pass 1 with one device atomic per base (CLAIR_AMD_FE_TALLY=atomic): every candidate,
    every window count and every anomaly the same, with the tile boundaries (multiples of 512 table positions) moved across the alignments by the
    margin, reads of 200-3000 bases crossing dozens of tiles, dense insertions and deletions (the search tallies an I / D at the position BEFORE
    it: across a tile boundary that word belongs to the neighbour), and the alignments in several slabs."""
    for seed, kw in ((9, dict(n_reads=2500, ref_len=30000, read_len=(200, 3000), cand_step=(1, 30))),
                     (8, dict(n_reads=900, ref_len=4000, ins_rate=0.12, del_rate=0.1, cand_step=(1, 6))),
                     (5, dict(n_reads=600, ref_len=3000, dup_burst=6))):
        case = fc.synth(seed, **kw)
        got = {}
        for mode in ("atomic", "tiles"):
            if mode == "atomic":
                monkeypatch.setenv("CLAIR_AMD_FE_TALLY", "atomic")
            else:
                monkeypatch.delenv("CLAIR_AMD_FE_TALLY", raising=False)
            f = device_frontend_text(case, chunks=3, margin=margin) if text else device_frontend(case, slabs=3, margin=margin)
            n = f.find_candidates(min_coverage=4, threshold=0.125)
            pos = f.candidates()
            f.build_windows(drop_non_iupac_centre=False)
            got[mode] = (n, pos, windows_of(f), f.stats()["anomalies"], f.window_tuples())
            f.close()
        a, b = got["atomic"], got["tiles"]
        assert a[0] == b[0] and a[0] > 50 and np.array_equal(a[1], b[1]) and a[3] == b[3]
        assert all(np.array_equal(x, y) for x, y in zip(a[2], b[2])) and np.array_equal(a[4], b[4])


@pytest.mark.parametrize("inner", [False, True], ids=["whole_span", "tables_start_inside_the_reads"])
def test_pass_two_per_operation_without_left_edge_windows_equals_the_per_base_pass(inner, monkeypatch):
    """--stop_consider_left_edge: tuples per alignment from running sums of the candidate prefix, tuples per window as second differences over the
    centre value (two running sums, a gather), late starters per alignment -- against the per-base kernel (CLAIR_AMD_FE_PASS2=base) and the
    sequential stage.  With the tables starting INSIDE the alignments (a region of a larger run) the trapezoids that begin left of the tables
    go through the fold."""
    for seed, kw in ((8, dict(n_reads=900, ref_len=4000, ins_rate=0.12, del_rate=0.1, cand_step=(1, 6))),
                     (9, dict(n_reads=1500, ref_len=20000, read_len=(200, 3000), cand_step=(1, 30))),
                     (5, dict(n_reads=600, ref_len=3000, dup_burst=6))):
        case = fc.synth(seed, **kw)
        lo, hi = (case["ref0"] + 700, case["ref0"] + len(case["ref"]) - 600) if inner else (case["ref0"] - 64, case["ref0"] + len(case["ref"]) + 64)
        cands = case["candidates"][(case["candidates"] > lo + 40) & (case["candidates"] < hi - 40)]
        hc, hs, hcounts = fc.host_windows(case, candidates=cands, consider_left_edge=False)
        got = {}
        for mode in ("base", "op"):
            if mode == "base":
                monkeypatch.setenv("CLAIR_AMD_FE_PASS2", "base")
            else:
                monkeypatch.delenv("CLAIR_AMD_FE_PASS2", raising=False)
            f = _capi.Frontend(0, case["ref"], case["ref0"], lo, hi)
            p = _hostapi.SamPacker(case["ctg"])
            assert p.feed(case["sam"], final=True) == b""
            f.add_slab(p)
            f.set_candidates(cands)
            f.build_windows(min_coverage=0, drop_non_iupac_centre=False, consider_left_edge=False)
            assert f.stats()["anomalies"] == 0
            got[mode] = (windows_of(f), f.read_tuples(0), f.window_tuples())
            f.close()
        a, b = got["base"], got["op"]
        assert all(np.array_equal(x, y) for x, y in zip(a[0], b[0])) and np.array_equal(a[1], b[1])
        assert np.array_equal(a[2][0], b[2][0]) and np.array_equal(a[2][1], b[2][1]) and int(b[2][1].sum()) > 0
        assert np.array_equal(hc, b[0][0]) and np.array_equal(hs, b[0][1]) and np.array_equal(hcounts, b[0][2]) and len(hc) > 50


def test_candidates_then_windows_equal_the_two_sequential_stages():
    """The whole front end: one packed stream, candidate search on the tallies, windows at those candidates -- against the finder and the
    builder run one after the other on the text, with a region (the two stages then see different alignments) and a bed file."""
    case = fc.synth(77, n_reads=1500, ref_len=12000, read_len=(100, 1500), ins_rate=0.03, del_rate=0.03)
    region = (2000, 9000)
    bed = [(0, 5000), (4800, 5200), (6000, 6000), (7000, 11000)]
    want_pos = fc.host_candidates(case, ctg_start=region[0], ctg_end=region[1], bed=bed, min_coverage=4, threshold=0.125)
    hc, hs, hcounts = fc.host_windows(case, candidates=want_pos, pile_region=region)
    keep = fe.PILE_ROW[hs[:, 16]] != 255
    f = device_frontend(case, slabs=3, pile_region=region)
    n = f.find_candidates(min_coverage=4, threshold=0.125, ctg_start=region[0], ctg_end=region[1], bed=bed)
    assert n == len(want_pos) > 100 and np.array_equal(f.candidates(), want_pos)
    f.build_windows(min_coverage=0, drop_non_iupac_centre=True)
    assert f.stats()["anomalies"] == 0 and not f.budget_binds()
    centres, seqs, counts = windows_of(f)
    assert np.array_equal(hc[keep], centres) and np.array_equal(hs[keep], seqs) and np.array_equal(hcounts[keep], counts)


def test_budget_that_binds_is_reported():
    case = fc.synth(31, n_reads=200, ref_len=1500, cand_step=(1, 4))
    f = device_frontend(case, slabs=2)
    f.set_candidates(case["candidates"])
    f.build_windows(drop_non_iupac_centre=False)
    assert not f.budget_binds() and f.budget_binds(available_slots=3000)
    free = fc.host_windows(case)
    assert not all(np.array_equal(a, b) for a, b in zip(free, fc.host_windows(case, available_slots=3000)))


def test_reference_case_whose_budget_binds_is_reported_by_the_device():
    """The golden case minted from the real script with its budget binding: the device front end builds its (unbounded) windows and reports that
    the reference's budget would have run out -- callVarBam then runs the sequential stage, which tests/test_pileup.py pins against the records."""
    case = fc.ct_golden_case(fc.BUDGET_GOLDEN)
    f = device_frontend(case, slabs=2, dcov=case["dcov"], pile_min_mq=case["min_mq"])
    f.set_candidates(case["candidates"])
    f.build_windows(min_coverage=case["min_coverage"], drop_non_iupac_centre=False)
    assert f.stats()["anomalies"] == 0 and f.budget_binds() and not f.budget_binds(available_slots=10 ** 9)
    f.close()


def test_reports_of_what_leaves_the_regime():
    case = fc.synth(3, n_reads=60, ref_len=900)
    sam = case["sam"].decode().splitlines()
    col = sam[5].split("\t")
    col[9] = col[9][:10] + "*" + col[9][11:]
    f = device_frontend(dict(case, sam=("\n".join(sam[:5] + ["\t".join(col)] + sam[6:]) + "\n").encode()))
    assert f.stats()["anomalies"] & fe.A_BAD_BASE
    short = sam[7].split("\t")
    short[9] = short[9][:5]
    f = device_frontend(dict(case, sam=("\n".join(sam[:7] + ["\t".join(short)] + sam[8:]) + "\n").encode()))
    assert f.stats()["anomalies"] & fe.A_SEQ_OVERRUN
    f = device_frontend(dict(case, ref=case["ref"][:300] + "-" + case["ref"][301:]))
    assert f.stats()["anomalies"] & fe.A_BAD_REF
    f = device_frontend(case)
    f.set_candidates(case["candidates"][::-1].copy())
    assert f.stats()["anomalies"] & fe.A_CANDIDATES
    # depth beyond int16: 40 000 identical alignments (dcov lifted)
    one = b"r\t0\tchrS\t100\t60\t40M\t*\t0\t0\t" + case["ref"][99:139].encode() + b"\t" + b"I" * 40 + b"\n"
    f = device_frontend(dict(case, sam=one * 40000), dcov=100000)
    f.set_candidates(np.array([120], np.int64))
    f.build_windows(drop_non_iupac_centre=False)
    assert f.stats()["anomalies"] & fe.A_OVERFLOW


def test_counters_that_fill_up_are_reported():
    """The per-position counters are 21 bits wide: 2 097 151 alignments over one position is the end of the road, and is said."""
    case = fc.synth(3, n_reads=60, ref_len=900)
    one = b"r\t0\tchrS\t100\t60\t8M\t*\t0\t0\t" + case["ref"][99:107].encode() + b"\t" + b"I" * 8 + b"\n"
    f = _capi.Frontend(0, case["ref"], 0, -64, 964)
    f.text_options("chrS", dcov=10 ** 7)
    f.add_text(one * 1000000)
    assert f.stats()["anomalies"] == 0
    f.add_text(one * 1000000)
    assert f.stats()["anomalies"] == 0            # 2 000 000: still fine
    f.add_text(one * 200000)
    assert f.stats()["anomalies"] & fe.A_OVERFLOW


def test_error_paths():
    with pytest.raises(_capi.EngineError, match="empty span"):
        _capi.Frontend(0, "ACGT", 0, 10, 10)
    f = _capi.Frontend(0, "ACGT" * 100, 0, -64, 464)
    with pytest.raises(_capi.EngineError, match="no candidates yet"):
        f.build_windows()
    with pytest.raises(_capi.EngineError, match="no windows yet"):
        f.window_info(0, 1)
    assert f.set_candidates(np.zeros(0, np.int64)) == 0 and f.build_windows() == 0
    assert f.window_info(0, 0)[0].shape == (0,)
    with pytest.raises(_capi.EngineError, match="out of range"):
        f.window_info(0, 1)


def test_windows_go_to_the_engine_without_leaving_the_device():
    """clair_submit_ex on the device address of the windows = clair_submit_ex on a host copy of them: call records bit for bit."""
    from clair_amd import weights
    case = fc.synth(55, n_reads=1200, ref_len=9000, read_len=(100, 1200))
    f = device_frontend(case, slabs=2)
    f.find_candidates(min_coverage=4, threshold=0.125)
    n = f.build_windows(drop_non_iupac_centre=True)
    assert n > 300
    centres, seqs = f.window_info(0, n)
    centre = np.stack([seqs[:, 16], (seqs[:, :33] != 0).sum(axis=1).astype(np.uint8)], axis=1)
    e = _capi.Engine(0, 256, 2)
    e.load_weights(weights.synthetic_weights(seed=12))
    for first in (0, 256, n - 77):
        m = min(256, n - first)
        e.submit_calls(0, _capi.DeviceWindows(f, first, m), centre[first:first + m])
        on_device = e.wait(0)
        e.submit_calls(1, f.window_counts(first, m), centre[first:first + m], counts=True)
        from_host = e.wait(1)
        assert on_device.tobytes() == from_host.tobytes() and (on_device["status"] != 0).any()
    e.close()


@pytest.mark.parametrize("block", range(4))
def test_differential_fuzz_against_the_sequential_stages(block):
    """Random alignments x random options of both stages (dcov, mapping-quality floors, depth floors, thresholds, regions, bed
    intervals, slab cuts): the device front end = the finder followed by the builder, candidates then windows, bit for bit."""
    done = 0
    for seed in range(block * 12, block * 12 + 12):
        case, pile_kw, evc_kw, region = fc.fuzz_case(seed)
        rng = dict(ctg_start=region[0], ctg_end=region[1]) if region else {}
        want_pos = fc.host_candidates(case, **rng, **evc_kw)
        hc, hs, hcounts = fc.host_windows(case, candidates=want_pos, pile_region=region, dcov=pile_kw["dcov"], min_mq=pile_kw["min_mq"],
                                          min_coverage=pile_kw["min_coverage"], consider_left_edge=seed % 3 != 0)
        f = device_frontend(case, slabs=1 + seed % 4, dcov=pile_kw["dcov"], pile_min_mq=pile_kw["min_mq"], evc_min_mq=evc_kw["min_mq"], pile_region=region)
        n = f.find_candidates(min_coverage=evc_kw["min_coverage"], threshold=evc_kw["threshold"], ctg_start=rng.get("ctg_start"), ctg_end=rng.get("ctg_end"),
                              bed=evc_kw["bed"])
        assert n == len(want_pos) and np.array_equal(f.candidates(), want_pos), seed
        f.build_windows(min_coverage=pile_kw["min_coverage"], drop_non_iupac_centre=False, consider_left_edge=seed % 3 != 0)
        assert f.stats()["anomalies"] == 0 and f.host_anomalies == 0 and not f.budget_binds(), seed
        centres, seqs, counts = windows_of(f)
        assert np.array_equal(hc, centres) and np.array_equal(hs, seqs) and np.array_equal(hcounts, counts), seed
        done += len(hc)
        f.close()
    assert done > 300


def test_span_of_a_whole_chromosome():
    """Tables over 260 M positions (chr1-sized: 25 GB of tables, 64-bit indexing, 63 k scan blocks) with alignments at both ends."""
    rng = np.random.default_rng(1)
    n = 260_000_000
    ref = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, n, dtype=np.uint8)]
    offset = n - 5000
    a = fc.synth(61, n_reads=300, ref_len=3000, iupac=False)
    b = fc.synth(62, n_reads=300, ref_len=3000, iupac=False)
    ref[:3000] = np.frombuffer(a["ref"].encode(), np.uint8)
    ref[offset:offset + 3000] = np.frombuffer(b["ref"].encode(), np.uint8)
    shifted = []
    for line in b["sam"].decode().splitlines():
        col = line.split("\t")
        col[3] = str(int(col[3]) + offset)
        shifted.append("\t".join(col))
    case = dict(ctg=a["ctg"], ref=ref.tobytes().decode(), ref0=0, sam=a["sam"] + ("\n".join(shifted) + "\n").encode())
    del ref
    want_pos = fc.host_candidates(case, min_coverage=4, threshold=0.125)
    hc, hs, hcounts = fc.host_windows(case, candidates=want_pos)
    assert (want_pos > offset).sum() > 50 and (want_pos < 3000).sum() > 50
    f = device_frontend(case, slabs=2)
    assert f.find_candidates(min_coverage=4, threshold=0.125) == len(want_pos) and np.array_equal(f.candidates(), want_pos)
    f.build_windows(drop_non_iupac_centre=False)
    assert f.stats()["anomalies"] == 0 and not f.budget_binds()
    centres, seqs, counts = windows_of(f)
    assert np.array_equal(hc, centres) and np.array_equal(hs, seqs) and np.array_equal(hcounts, counts)
    f.close()


# ---- the same with the text parsed on the device (clair_frontend_add_text) -------------------------------------------------------------
def device_frontend_text(case, chunks=1, margin=64, **pack_kw):
    f = _capi.Frontend(0, case["ref"], case["ref0"], case["ref0"] - margin, case["ref0"] + len(case["ref"]) + margin)
    f.text_options(case["ctg"], **pack_kw)
    sam, at = case["sam"], 0
    for k in range(chunks):
        cut = len(sam) if k == chunks - 1 else sam.index(b"\n", len(sam) * (k + 1) // chunks) + 1
        f.add_text(sam[at:cut])
        at = cut
    return f


def host_packed(case, **pack_kw):
    p = _hostapi.SamPacker(case["ctg"], **pack_kw)
    assert p.feed(case["sam"], final=True) == b""
    return p.slab_arrays(), p.stats()


@pytest.mark.parametrize("chunks", [1, 7])
@pytest.mark.parametrize("opt", range(5))
def test_text_parsed_on_the_device_gives_the_host_packers_alignments(opt, chunks):
    kw = [dict(), dict(dcov=3), dict(evc_min_mq=20, pile_min_mq=30), dict(pile_region=(500, 1500)), dict(pile_region=(1, 40), dcov=2)][opt]
    case = fc.synth(seed=40 + opt, n_reads=400, ref_len=3000, dup_burst=8)
    (r, o, e, q), st = host_packed(case, **kw)
    f = device_frontend_text(case, chunks=chunks, **kw)
    got = np.concatenate(f.slab_reads)
    ts = f.text_stats()
    assert ts == dict(lines=st["lines"], evc_reads=st["evc_reads"], pile_reads=st["pile_reads"], anomalies=st["anomalies"])
    assert len(got) == len(r) and np.array_equal(got["pos0"], r["pos0"]) and np.array_equal(got["flags"], r["flags"])
    assert np.array_equal(got["seq_len"], r["seq_len"]) and np.array_equal(got["n_ops"], r["n_ops"])
    # only the SEQ columns stay on the device, packed a whole number of dwords per alignment (fe_text_seq_kernel): not offsets into the text
    for sr in f.slab_reads:
        assert (sr["seq0"] % 4 == 0).all() and sr["seq0"][0] == 0
        assert np.array_equal(sr["seq0"][1:], np.cumsum((sr["seq_len"][:-1].astype(np.int64) + 3) // 4 * 4))
    # ... and everything downstream of them: tallies -> candidates, windows, tuple counts
    g = _capi.Frontend(0, case["ref"], case["ref0"], case["ref0"] - 64, case["ref0"] + len(case["ref"]) + 64)
    g.add_arrays(r, o, e, q)
    for x in (f, g):
        x.find_candidates(min_coverage=3, threshold=0.1)
        x.build_windows(drop_non_iupac_centre=False)
    assert np.array_equal(f.candidates(), g.candidates()) and f.stats()["windows"] == g.stats()["windows"] > 20
    for a, b in zip(windows_of(f), windows_of(g)):
        assert np.array_equal(a, b)
    assert np.array_equal(np.concatenate([f.read_tuples(k) for k in range(f.stats()["slabs"])]), g.read_tuples(0))
    assert f.stats()["anomalies"] == g.stats()["anomalies"] == 0 and f.budget_binds() == g.budget_binds()


@pytest.mark.parametrize("block", range(3))
def test_differential_fuzz_with_the_text_parsed_on_the_device(block):
    done = 0
    for seed in range(100 + block * 12, 100 + block * 12 + 12):
        case, pile_kw, evc_kw, region = fc.fuzz_case(seed)
        rng = dict(ctg_start=region[0], ctg_end=region[1]) if region else {}
        want_pos = fc.host_candidates(case, **rng, **evc_kw)
        hc, hs, hcounts = fc.host_windows(case, candidates=want_pos, pile_region=region, dcov=pile_kw["dcov"], min_mq=pile_kw["min_mq"],
                                          min_coverage=pile_kw["min_coverage"])
        f = device_frontend_text(case, chunks=1 + seed % 5, dcov=pile_kw["dcov"], pile_min_mq=pile_kw["min_mq"], evc_min_mq=evc_kw["min_mq"], pile_region=region)
        n = f.find_candidates(min_coverage=evc_kw["min_coverage"], threshold=evc_kw["threshold"], ctg_start=rng.get("ctg_start"), ctg_end=rng.get("ctg_end"),
                              bed=evc_kw["bed"])
        assert n == len(want_pos) and np.array_equal(f.candidates(), want_pos), seed
        f.build_windows(min_coverage=pile_kw["min_coverage"], drop_non_iupac_centre=False)
        assert f.stats()["anomalies"] == 0 and not f.budget_binds(), seed
        centres, seqs, counts = windows_of(f)
        assert np.array_equal(hc, centres) and np.array_equal(hs, seqs) and np.array_equal(hcounts, counts), seed
        done += len(hc)
        f.close()
    assert done > 300


def test_an_unsorted_slab_from_a_direct_caller_is_tallied_right_and_reported():
    """clair_frontend_add_reads with the alignments of a slab in shuffled order: the tiled tally bisects over the starts, so the
    slab goes through the order-independent per-base kernel instead, and CLAIR_FE_UNSORTED is raised as the packers would have raised it."""
    case = fc.synth(seed=77, n_reads=500, ref_len=3000)
    (r, o, e, q), st = host_packed(case)
    assert st["anomalies"] == 0
    lo, hi = case["ref0"] - 64, case["ref0"] + len(case["ref"]) + 64
    f = _capi.Frontend(0, case["ref"], case["ref0"], lo, hi)
    f.add_arrays(r, o, e, q)
    g = _capi.Frontend(0, case["ref"], case["ref0"], lo, hi)
    # the same alignments in shuffled order, the slab laid out the way a packer would lay it out for THAT order (operations and their element
    # prefix follow the alignments; the bases stay where they are, every record names its own offset)
    perm = np.random.default_rng(5).permutation(len(r))
    assert (np.diff(r["pos0"][perm].astype(np.int64)) < 0).any()
    per_op = np.diff(e.astype(np.int64))
    r2, pieces, counts = r[perm].copy(), [], []
    at = 0
    for k, rec in enumerate(r[perm]):
        a, n = int(rec["op0"]), int(rec["n_ops"])
        piece = o[a:a + n].copy()
        piece["read"] = k                      # an operation names its alignment by its index in the slab
        pieces.append(piece); counts.append(per_op[a:a + n])
        r2["op0"][k] = at
        at += n
    o2 = np.concatenate(pieces)
    e2 = np.concatenate([[0], np.cumsum(np.concatenate(counts))]).astype(np.uint32)
    assert len(o2) == len(o) and e2[-1] == e[-1]
    g.add_arrays(r2, o2, e2, q)
    assert f.stats()["anomalies"] == 0 and g.stats()["anomalies"] == fe.A_UNSORTED
    for x in (f, g):
        x.find_candidates(min_coverage=3, threshold=0.1)
    assert len(f.candidates()) > 20 and np.array_equal(f.candidates(), g.candidates())
    f.close(); g.close()


def test_unsorted_text_on_the_device_is_tallied_right_and_reported():
    """The same on the text entry point (clair_frontend_add_text passed `sorted = true` to the tally even after its own packer had
    raised CLAIR_FE_UNSORTED for the slab): alignment lines in shuffled order, in one chunk and in three -- the anomaly word says so and the
    column tallies (hence the candidates and the windows built on them) equal those of the sorted text.  No depth cap here: the cap
    walks the lines in file order and would keep different alignments."""
    case = fc.synth(seed=78, n_reads=500, ref_len=3000)
    lines = case["sam"].split(b"\n")
    assert lines[-1] == b""
    perm = np.random.default_rng(6).permutation(len(lines) - 1)
    shuffled = dict(case, sam=b"\n".join(lines[i] for i in perm) + b"\n")
    f = device_frontend_text(case, chunks=1, dcov=1 << 20)
    assert f.text_stats()["anomalies"] == 0
    n = f.find_candidates(min_coverage=3, threshold=0.1)
    want = f.candidates()
    f.build_windows(min_coverage=0, drop_non_iupac_centre=False)
    want_windows = windows_of(f)
    assert n > 20
    for chunks in (1, 3):
        g = device_frontend_text(shuffled, chunks=chunks, dcov=1 << 20)
        assert g.text_stats()["anomalies"] & fe.A_UNSORTED
        assert g.find_candidates(min_coverage=3, threshold=0.1) == n and np.array_equal(g.candidates(), want)
        g.build_windows(min_coverage=0, drop_non_iupac_centre=False)
        for a, b in zip(windows_of(g), want_windows):
            assert np.array_equal(a, b)
        g.close()
    f.close()


def test_differential_fuzz_with_leading_indels_on_the_device():
    """As tests/test_frontend.py::test_differential_fuzz_with_leading_indels, through both packing paths of the device front end."""
    silent = reported = 0
    for seed in range(300, 318):
        case, pile_kw, evc_kw, region = fc.fuzz_case(seed, lead_indel=0.25, lead_indel_late=bool(seed & 1))
        rng = dict(ctg_start=region[0], ctg_end=region[1]) if region else {}
        want_pos = fc.host_candidates(case, **rng, **{k: v for k, v in evc_kw.items() if k != "bed"})
        hc, hs, hcounts = fc.host_windows(case, candidates=want_pos, pile_region=region, dcov=pile_kw["dcov"], min_mq=pile_kw["min_mq"],
                                          min_coverage=pile_kw["min_coverage"])
        kw = dict(dcov=pile_kw["dcov"], pile_min_mq=pile_kw["min_mq"], evc_min_mq=evc_kw["min_mq"], pile_region=region)
        f, g = device_frontend_text(case, chunks=1 + seed % 4, **kw), device_frontend(case, slabs=1 + seed % 3, **kw)
        bits = g.host_anomalies
        assert f.text_stats()["anomalies"] == bits and bits & ~fe.A_LEAD_INDEL == 0, seed
        for h in (f, g):
            n = h.find_candidates(min_coverage=evc_kw["min_coverage"], threshold=evc_kw["threshold"], ctg_start=rng.get("ctg_start"), ctg_end=rng.get("ctg_end"))
            if not bits:
                assert n == len(want_pos) and np.array_equal(h.candidates(), want_pos), seed
            h.set_candidates(want_pos)
            h.build_windows(min_coverage=pile_kw["min_coverage"], drop_non_iupac_centre=False)
            centres, seqs, counts = windows_of(h)
            assert np.array_equal(hc, centres) and np.array_equal(hs, seqs) and np.array_equal(hcounts, counts), seed
            h.close()
        silent += not bits
        reported += bool(bits)
    assert silent >= 4 and reported >= 2, (silent, reported)


def test_text_on_the_device_reports_what_the_host_packer_reports():
    ok = b"r1\t0\tchrS\t10\t60\t5M\t*\t0\t0\tACGTA\tIIIII\n"
    ref = "ACGT" * 100

    def fresh(**kw):
        f = _capi.Frontend(0, ref, 0, -64, 464)
        f.text_options("chrS", **kw)
        return f
    f = fresh()
    f.add_text(ok + b"r2\t0\tchrS\t5\t60\t5M\t*\t0\t0\tACGTA\tIIIII\n")
    assert f.text_stats()["anomalies"] == fe.A_UNSORTED and f.stats()["anomalies"] & fe.A_UNSORTED
    f = fresh()
    f.add_text(ok)
    f.add_text(b"r2\t0\tchrS\t5\t60\t5M\t*\t0\t0\tACGTA\tIIIII\n")              # ... across two calls
    assert f.text_stats()["anomalies"] == fe.A_UNSORTED
    f = fresh()
    f.add_text(b"r1\t0\tchrS\t10\t60\t3M0I2M\t*\t0\t0\tACGTA\tIIIII\n")
    assert f.text_stats()["anomalies"] == fe.A_ZERO_INDEL
    for cigar in (b"2M200000D3M", b"3000000000S5M"):
        f = fresh()
        f.add_text(b"r1\t0\tchrS\t10\t60\t" + cigar + b"\t*\t0\t0\tACGTA\tIIIII\n")
        assert f.text_stats()["anomalies"] == fe.A_LONG_SPAN
    for bad in (b"r1\t0\tchrS\n", b"r1\tx\tchrS\t10\t60\t5M\t*\t0\t0\tACGTA\tIIIII\n", b"\n", ok + b"  \n" + ok):
        with pytest.raises(_capi.MalformedText):
            fresh().add_text(bad)
    f = fresh(evc_min_mq=10, pile_min_mq=10)
    f.add_text(b"@HD\tVN:1.6\n" + b"r1\t0\tchrS\t10\t5\t5M\t*\t0\t0\tACGTA\tIIIII\n" + ok)
    assert f.text_stats() == dict(lines=3, evc_reads=1, pile_reads=1, anomalies=0) and f.stats()["reads"] == 1
    with pytest.raises(_capi.EngineError, match="line end"):
        fresh().add_text(ok[:-1])
    # an I / D before the first matched base, after another accepted alignment of the same POS: in one chunk, across chunks,
    # behind an alignment only the pileup takes, and the cases that are fine
    from test_frontend import _lead_indel_sam
    for first, later, kw, want in (("30M", "2I30M", {}, fe.A_LEAD_INDEL), ("2I30M", "3D30M", {}, fe.A_LEAD_INDEL), ("30M", "4S2I3D30M", {}, fe.A_LEAD_INDEL),
                                   ("2I30M", "30M", {}, 0), ("30M", "2M2I28M", {}, 0)):
        sam = _lead_indel_sam(first, later)
        for cut in (len(sam), sam.rindex(b"q1")):
            f = fresh(**kw)
            f.add_text(sam[:cut])
            if cut < len(sam):
                f.add_text(sam[cut:])
            p = _hostapi.SamPacker("chrS", **kw)
            p.feed(sam, final=True)
            assert f.text_stats()["anomalies"] == p.stats()["anomalies"] == want, (first, later, cut)
    sam = _lead_indel_sam("30M", "2I30M").replace(b"q0\t0\tchrS\t102\t60", b"q0\t0\tchrS\t102\t5")   # the search skips q0 (--minMQ), the pileup keeps it
    f = fresh(evc_min_mq=10)
    f.add_text(sam)
    assert f.text_stats()["anomalies"] == 0
    # lower-case bases stay lower-case in the slab and count as their upper-case selves (both scripts upper-case SEQ first)
    f, g = fresh(), fresh()
    f.add_text(b"r1\t0\tchrS\t100\t60\t5M\t*\t0\t0\tacgta\tIIIII\n" * 6)
    g.add_text(ok.replace(b"\t10\t", b"\t100\t") * 6)
    for x in (f, g):
        x.find_candidates(min_coverage=1, threshold=0.0)
        x.build_windows(drop_non_iupac_centre=False)
    assert f.stats()["windows"] == g.stats()["windows"] > 0
    for a, b in zip(windows_of(f), windows_of(g)):
        assert np.array_equal(a, b)


def test_text_of_long_reads():
    """Long lines (several KB of CIGAR and SEQ each) through the text kernels: same alignments, same windows."""
    case = fc.synth(9, n_reads=1200, ref_len=30000, read_len=(1500, 6000), cand_step=(1, 30))
    assert len(case["sam"]) / case["sam"].count(b"\n") > 2048
    (r, o, e, q), st = host_packed(case)
    f = device_frontend_text(case, chunks=3)
    got = np.concatenate(f.slab_reads)
    assert f.text_stats() == dict(lines=st["lines"], evc_reads=st["evc_reads"], pile_reads=st["pile_reads"], anomalies=0)
    assert np.array_equal(got["pos0"], r["pos0"]) and np.array_equal(got["flags"], r["flags"]) and np.array_equal(got["n_ops"], r["n_ops"])
    hc, hs, hcounts = fc.host_windows(case)
    f.set_candidates(case["candidates"])
    f.build_windows(drop_non_iupac_centre=False)
    centres, seqs, counts = windows_of(f)
    assert f.stats()["anomalies"] == 0 and np.array_equal(hc, centres) and np.array_equal(hs, seqs) and np.array_equal(hcounts, counts) and len(hc) > 500
