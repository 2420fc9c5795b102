"""Ingest / decode / VCF writer / driver against goldens minted from the REAL reference
(tools/make_ref_goldens.py imports /root/reference/clair/{utils,call_var}.py in the build
container).  Byte-for-byte: every row string must be identical."""
import gzip
import io
import json
import os
from contextlib import redirect_stderr

import numpy as np
import pytest

from clair_amd import call_var as cvar
from clair_amd import task, utils

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CONFIGS = {
    "default": (False, False, False, False, False, None),
    "showref_qual": (True, False, False, False, False, 100),
    "haploid_precision": (False, False, True, False, False, None),
    "haploid_sensitive": (True, False, False, True, False, 50),
    "debug": (False, True, False, False, False, None),
    "ensemble": (False, False, False, False, True, None),
}


@pytest.fixture(scope="module")
def decode_cases():
    with np.load(os.path.join(GOLD, "decode_cases.npz")) as z:
        X = z["x"].astype(np.float32)
        P = z["probs"]
        infos = json.loads(str(z["infos"]))
        tags = json.loads(str(z["tags"]))
    with gzip.open(os.path.join(GOLD, "decode_rows.json.gz"), "rt") as f:
        rows = json.load(f)
    return X, P, infos, tags, rows


def _split(P):
    return [P[:, 0:21], P[:, 21:24], P[:, 24:57], P[:, 57:90]]


@pytest.mark.parametrize("name", sorted(CONFIGS) + ["default_pysam_all"])
def test_decode_rows_byte_identical(decode_cases, name):
    X, P, infos, tags, rows = decode_cases
    cfg = cvar.OutputConfig(*CONFIGS[name.replace("_pysam_all", "")])
    dec = cvar.VariantDecoder(cfg, always_use_bam=name.endswith("pysam_all"), arith="numpy2")
    want = [ln for per in rows[name] for ln in per]
    got = dec.decode_batch(X, infos, _split(P))
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert g == w
    # one candidate at a time gives the same rows (no cross-candidate state)
    for i in range(0, len(infos), 37):
        assert dec.decode_batch(X[i:i + 1], infos[i:i + 1], _split(P[i:i + 1])) == rows[name][i]


def test_every_decode_branch_is_covered(decode_cases):
    X, P, infos, tags, rows = decode_cases
    fams = cvar.OutcomeFamilies(*_split(P), ref_class=np.zeros(len(infos), dtype=np.int64))
    first = fams.flags.argmax(axis=1)
    assert set(range(cvar.N_FAMILIES)) <= set(first.tolist())
    gts = {ln.split("\t")[-1].split(":")[0] for per in rows["showref_qual"] for ln in per}
    assert gts == {"0/0", "1/1", "0/1", "1/2"}
    sizes = [f.shape[1] for f in fams.fam]
    assert sizes == [1, 4, 6, 16, 64, 256, 16, 64, 240, 512] and sum(sizes) == 1179


def test_legacy_arithmetic_differs_only_in_qual_and_af(decode_cases):
    X, P, infos, tags, rows = decode_cases
    cfg = cvar.OutputConfig(*CONFIGS["showref_qual"])
    a = cvar.VariantDecoder(cfg, arith="numpy2").decode_batch(X, infos, _split(P))
    b = cvar.VariantDecoder(cfg, arith="legacy").decode_batch(X, infos, _split(P))
    assert len(a) == len(b)
    ndiff = 0
    for ra, rb in zip(a, b):
        ca, cb = ra.split("\t"), rb.split("\t")
        assert ca[:5] == cb[:5]                                   # CHROM POS ID REF ALT
        assert ca[-1].split(":")[0] == cb[-1].split(":")[0]       # GT
        assert ca[-1].split(":")[2] == cb[-1].split(":")[2]       # DP
        assert abs(int(ca[5]) - int(cb[5])) <= 1                  # QUAL: float32 vs float64 log argument
        assert abs(float(ca[-1].split(":")[3]) - float(cb[-1].split(":")[3])) <= 1.0001e-4
        ndiff += ra != rb
    assert ndiff < len(a) // 10


def legacy_fixture():
    """(cases file, config name, rows) of tests/golden/decode_rows_legacy.json.gz: the reference's writer run with float32 scalars promoted
    as NumPy 1.x did (tools/make_ref_goldens.py: L32 / LArr) -- the cases of decode_cases.npz, plus cases at read depth 160 and calls of
    probability exactly 1, where that arithmetic and NumPy 2's part (or NumPy 2 raises)."""
    with gzip.open(os.path.join(GOLD, "decode_rows_legacy.json.gz"), "rt") as f:
        doc = json.load(f)
    assert "NumPy 1.x" in doc["meta"]["minted_with"]
    out = []
    for fn, names in (("decode_cases.npz", ("default", "showref_qual", "haploid_sensitive", "debug")), ("decode_cases_legacy.npz", ("extra_default", "extra_showref_qual"))):
        with np.load(os.path.join(GOLD, fn)) as z:
            X, P, infos = z["x"].astype(np.float32), z["probs"], json.loads(str(z["infos"]))
        out += [(X, P, infos, name, doc["rows"][name]) for name in names]
    return out


@pytest.mark.parametrize("native", [False, True])
def test_default_arithmetic_reproduces_the_reference_rows_minted_with_numpy1_promotion(native):
    """The shipped default (`--arith legacy`) against the reference itself: call_var.py:568-586 (QUAL) and :1151 (AF) run in float64 from
    the first operation that meets a Python number, as under the pinned NumPy 1.18.  Byte-identical rows, Python and native decode."""
    parted = 0
    for X, P, infos, name, rows in legacy_fixture():
        cfg = cvar.OutputConfig(*CONFIGS[name.replace("extra_", "")])
        want = [ln for per in rows for ln in per]
        got = cvar.VariantDecoder(cfg, arith="legacy", native=native).decode_batch(X, infos, _split(P))
        assert got == want
        if name == "extra_showref_qual":
            keep = [i for i in range(len(infos)) if P[i].max() < 1.0]      # NumPy 2 raises on a call of probability 1 (below)
            a = cvar.VariantDecoder(cfg, arith="numpy2", native=native).decode_batch(X[keep], [infos[i] for i in keep], _split(P[keep]))
            b = [ln for i in keep for ln in rows[i]]
            parted = sum(x != y for x, y in zip(a, b))
            assert any("9096930" in ln.split("\t")[5] or int(ln.split("\t")[5]) > 9000000 for ln in want)      # the certain calls
    assert parted >= 10      # the fixture does tell the two arithmetics apart


def test_legacy_quality_handles_certain_calls():
    g = np.zeros(21, np.float32)
    g[task.GT21_INDEX["AA"]] = 1.0
    z = np.array([1, 0, 0], np.float32)
    assert cvar.quality_score("A", "A", "0/0", g, z, "legacy") == 9096930 or cvar.quality_score("A", "A", "0/0", g, z, "legacy") > 9000000
    with pytest.raises((ValueError, ZeroDivisionError)):
        cvar.quality_score("A", "A", "0/0", g, z, "numpy2")      # the reference under NumPy 2 raises too


@pytest.mark.parametrize("tag", list("abcde"))
def test_ingest_matches_reference(tag):
    with np.load(os.path.join(GOLD, "ingest_cases.npz")) as z:
        gold = {k: z[k] for k in z.files}
    batch = int(gold["%s_batch" % tag])
    err = io.StringIO()
    got = []
    with redirect_stderr(err):
        for X, infos in utils.tensor_generator_from(os.path.join(GOLD, "ingest_%s.txt.gz" % tag), batch):
            got.append((np.array(X, copy=True), [list(i) for i in infos]))
    assert len(got) == int(gold["%s_nbatches" % tag])
    for k, (X, infos) in enumerate(got):
        want = gold["%s_X%d" % (tag, k)]
        assert X.dtype == np.float32 and X.shape == want.shape
        assert X.tobytes() == want.tobytes()
        assert infos == json.loads(str(gold["%s_info%d" % (tag, k)]))
    assert err.getvalue() == str(gold["%s_stderr" % tag])


@pytest.mark.parametrize("tag,sample", [("nofai", "HG002"), ("fai", "HG002")])
def test_header_matches_reference(tmp_path, tag, sample):
    ref = None
    if tag == "fai":
        ref = str(tmp_path / "ref.fa")
        open(ref + ".fai", "w").write(open(os.path.join(GOLD, "header_fai.fai")).read())
    out = str(tmp_path / "h.vcf")
    w = cvar.VcfWriter(out, sample, ref, False)
    w.write_header()
    w.close()
    assert open(out).read() == open(os.path.join(GOLD, "header_%s.vcf" % tag)).read()


class _OracleModel(object):
    """Same stand-in the golden generator used inside the reference driver."""

    def __init__(self, w):
        self.w = w
        self.prediction = None

    def predict(self, batchX):
        from oracle import model_np
        self.prediction = model_np.forward(self.w, batchX)
        return self.prediction


@pytest.mark.parametrize("tag,cfgname", [("default", "default"), ("showref", "showref_qual")])
def test_driver_end_to_end_matches_reference_driver(tmp_path, tag, cfgname):
    """tensor file -> (oracle probabilities) -> VCF must equal what the reference's own
    call_variants wrote with the same probabilities, byte for byte."""
    from clair_amd import weights
    w = weights.synthetic_weights(seed=4242, head_gain=6.0, lstm_bias_scale=0.1)
    out = str(tmp_path / "o.vcf")
    args = cvar.build_parser().parse_args(["--tensor_fn", os.path.join(GOLD, "e2e_230.txt.gz"), "--call_fn", out])
    dec = cvar.VariantDecoder(cvar.OutputConfig(*CONFIGS[cfgname]), arith="numpy2")
    wr = cvar.VcfWriter(out, "SAMPLE", None, False)
    with redirect_stderr(io.StringIO()):
        cvar.call_variants(args, _OracleModel(w), dec, wr, batch_size=100)
    wr.close()
    assert open(out).read() == open(os.path.join(GOLD, "e2e_230_%s.vcf" % tag)).read()


class _CallsModel(_OracleModel):
    """An asynchronous model whose wait() returns call records (include/clair_call.h), as clair_amd.model.Clair does on the GPU: the
    oracle's probabilities, resolved by the decode kernel's CPU twin."""
    n_slots = 2

    def __init__(self, w):
        _OracleModel.__init__(self, w)
        self.slots, self.calls_submits, self.strided, self.buffers = {}, 0, 0, []

    def submit(self, slot, batchX):
        self.slots[slot] = ("probs", np.asarray(batchX), None, False)

    def submit_calls(self, slot, batch, centre, counts=False, with_probabilities=False):
        self.calls_submits += 1
        x = np.asarray(batch)
        if counts:                                  # raw counts (possibly a strided view into a record buffer): what the device would convert
            self.strided += int(not x.flags.c_contiguous)
            x = x.astype(np.float32)
            x[..., 1:] -= x[..., :1]
        self.slots[slot] = ("calls", x, centre, with_probabilities)

    def pinned_buffer(self, nbytes):
        self.buffers.append(np.zeros(nbytes, dtype=np.uint8))
        return self.buffers[-1]

    def wait(self, slot):
        from clair_amd import _hostapi
        kind, x, centre, with_probs = self.slots.pop(slot)
        Y = self.predict(x)
        if kind == "probs":
            return Y
        calls = _hostapi.resolve_calls(x, Y, centre)
        return (calls, Y) if with_probs else calls


@pytest.mark.parametrize("tag,cfgname", [("default", "default"), ("showref", "showref_qual")])
def test_driver_with_the_decode_on_the_device_side_writes_the_same_vcf(tmp_path, monkeypatch, tag, cfgname):
    """call_variants hands a model that offers submit_calls the centre bytes of each batch and formats the call records it gets
    back: the same VCF as the reference driver's, byte for byte; CLAIR_AMD_DEVICE_DECODE=0 keeps the probabilities path."""
    from clair_amd import weights
    w = weights.synthetic_weights(seed=4242, head_gain=6.0, lstm_bias_scale=0.1)
    for env, expect_calls in ((None, True), ("0", False)):
        if env is None:
            monkeypatch.delenv("CLAIR_AMD_DEVICE_DECODE", raising=False)
        else:
            monkeypatch.setenv("CLAIR_AMD_DEVICE_DECODE", env)
        out = str(tmp_path / "o.vcf")
        args = cvar.build_parser().parse_args(["--tensor_fn", os.path.join(GOLD, "e2e_230.txt.gz"), "--call_fn", out])
        dec = cvar.VariantDecoder(cvar.OutputConfig(*CONFIGS[cfgname]), arith="numpy2")
        wr = cvar.VcfWriter(out, "SAMPLE", None, False)
        m = _CallsModel(w)
        with redirect_stderr(io.StringIO()):
            cvar.call_variants(args, m, dec, wr, batch_size=100)
        wr.close()
        assert (m.calls_submits > 0) == expect_calls
        assert open(out).read() == open(os.path.join(GOLD, "e2e_230_%s.vcf" % tag)).read()


def test_driver_reads_binary_records_into_the_engines_buffers_and_writes_the_same_vcf(tmp_path):
    """Binary tensor records + decode on the device side: call_variants reads every batch straight into a buffer the model lent it
    (page-locked on the GPU), submits the counts column as a strided view of that buffer, gives the buffer back when the batch has
    left the model, and writes the reference driver's VCF."""
    from clair_amd import tensor_binary, utils, weights
    w = weights.synthetic_weights(seed=4242, head_gain=6.0, lstm_bias_scale=0.1)
    binary = str(tmp_path / "t.bin")
    with redirect_stderr(io.StringIO()):
        batches = list(utils.tensor_generator_from(os.path.join(GOLD, "e2e_230.txt.gz"), 1000))
    with open(binary, "wb") as f:
        f.write(tensor_binary.MAGIC)
        for X, infos in batches:
            raw = np.rint(np.concatenate([X[..., :1], X[..., 1:] + X[..., :1]], axis=-1)).astype(np.int16)
            f.write(tensor_binary.pack_records(infos[0][0], [int(i[1]) for i in infos], [i[2] for i in infos], raw))
    out = str(tmp_path / "o.vcf")
    args = cvar.build_parser().parse_args(["--tensor_fn", binary, "--call_fn", out])
    dec = cvar.VariantDecoder(cvar.OutputConfig(*CONFIGS["default"]), arith="numpy2")
    wr = cvar.VcfWriter(out, "SAMPLE", None, False)
    m = _CallsModel(w)
    with redirect_stderr(io.StringIO()):
        cvar.call_variants(args, m, dec, wr, batch_size=50)
    wr.close()
    assert len(m.buffers) == 2 * m.n_slots + 4 and m.calls_submits == 5 and m.strided == 5      # 230 records in batches of 50, all from the pool
    assert open(out).read() == open(os.path.join(GOLD, "e2e_230_default.vcf")).read()


def test_ensemble_roundtrip_through_input_probabilities(decode_cases, tmp_path):
    """--output_for_ensemble lines fed back through --input_probabilities give the rows the decode
    produces from the %.6f-rounded probabilities (call_var.py:950-1000, 1276-1309)."""
    X, P, infos, tags, rows = decode_cases
    lines = [ln for per in rows["ensemble"][:200] for ln in per]
    out = str(tmp_path / "o.vcf")
    cfg = cvar.OutputConfig(*CONFIGS["showref_qual"])
    dec = cvar.VariantDecoder(cfg, arith="legacy")
    wr = cvar.VcfWriter(out, "SAMPLE", None, False)
    cvar.call_variants_with_probabilities_input(None, dec, wr, stream=io.StringIO("\n".join(lines) + "\n"))
    wr.close()
    body = [ln for ln in open(out).read().splitlines() if not ln.startswith("#")]
    keep = [i for i in range(200) if rows["ensemble"][i]]
    Pr = np.array([[float(v) for v in rows["ensemble"][i][0].split("\t")[3 + 1056:]] for i in keep], dtype=np.float32)
    want = dec.decode_batch(X[keep], [infos[i] for i in keep], _split(Pr))
    assert body == want


def test_argparse_surface():
    """Every flag of the reference CLI (call_var.py:1370-1429) is accepted with the same defaults."""
    p = cvar.build_parser()
    a = p.parse_args([])
    assert (a.tensor_fn, a.chkpnt_fn, a.call_fn, a.bam_fn, a.qual, a.sampleName) == ("PIPE", None, None, "bam.bam", None, "SAMPLE")
    assert (a.showRef, a.debug, a.ref_fn, a.threads, a.activation_only, a.max_plot, a.log_path) == (False, False, None, None, False, 10, None)
    assert (a.parallel_level, a.fast_plotting, a.workers, a.pysam_for_all_indel_bases) == (2, False, 8, False)
    assert (a.haploid_precision, a.haploid_sensitive, a.input_probabilities, a.output_for_ensemble) == (False, False, False, False)
    a = p.parse_args("--tensor_fn t --chkpnt_fn c --call_fn o --bam_fn b --qual 7 --sampleName S --showRef --debug "
                     "--ref_fn r --threads 3 --activation_only --max_plot 2 --log_path l -p 0 -w 2 --fast_plotting "
                     "--pysam_for_all_indel_bases --haploid_precision --haploid_sensitive --input_probabilities "
                     "--output_for_ensemble".split())
    assert a.qual == 7 and a.threads == 3 and a.parallel_level == 0 and a.workers == 2


# ---- BAM look-ups (call_var.py:102-170, 498-524, 540-565, 805-823) against rows the reference wrote with the same fake pysam ----
@pytest.fixture
def fake_pysam(monkeypatch):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import fake_pysam as fp
    monkeypatch.setitem(sys.modules, "pysam", fp)
    return fp


def _pysam_case():
    z = np.load(os.path.join(GOLD, "pysam_cases.npz"))
    x = z["x"].astype(np.float32)
    infos = json.loads(str(z["infos"]))
    P = z["probs"]
    with gzip.open(os.path.join(GOLD, "pysam_rows.json.gz"), "rt") as f:
        rows = json.load(f)
    return x, infos, [P[:, 0:21], P[:, 21:24], P[:, 24:57], P[:, 57:90]], rows


@pytest.mark.parametrize("native", [False, True])
@pytest.mark.parametrize("mode", ["default", "pysam_all"])
def test_bam_lookups_match_reference_rows_minted_with_the_same_fake_pysam(fake_pysam, native, mode):
    x, infos, Y, rows = _pysam_case()
    lookup = cvar.AlignmentLookup(os.path.join(GOLD, "pysam_bam.json"), os.path.join(GOLD, "pysam_ref.json"))
    assert lookup.sam is not None and lookup.fasta is not None
    dec = cvar.VariantDecoder(cvar.OutputConfig(False, False, False, False, False, None), lookup,
                              always_use_bam=(mode == "pysam_all"), arith="numpy2", native=native)
    got = dec.decode_batch(x, infos, Y)
    want = [ln for r in rows[mode] for ln in r]
    assert got == want
    assert sum(1 for ln in want if max(len(a) for a in (ln.split("\t")[3] + "," + ln.split("\t")[4]).split(",")) > 16) > 50
    if native and mode == "default":
        # only the candidates that pass a look-up point went through pysam: far fewer pile-ups than candidates
        assert 0 < lookup.sam.queries < len(infos)
    lookup.close()
    assert lookup.sam.closed and lookup.fasta.closed


def test_native_decoder_flags_exactly_the_candidates_a_bam_can_change(fake_pysam):
    from clair_amd import _hostapi
    x, infos, Y, rows = _pysam_case()
    no_bam = cvar.VariantDecoder(cvar.OutputConfig(False, False, False, False, False, None), arith="numpy2", native=False)
    with_bam = cvar.VariantDecoder(cvar.OutputConfig(False, False, False, False, False, None),
                                   cvar.AlignmentLookup(os.path.join(GOLD, "pysam_bam.json"), os.path.join(GOLD, "pysam_ref.json")),
                                   arith="numpy2", native=False)
    _, status = _hostapi.decode_rows(x, infos, Y, False, False, False, None, True, with_status=True)
    changed = 0
    for i in range(len(infos)):
        a = no_bam.decode_batch_py(x[i:i + 1], infos[i:i + 1], [y[i:i + 1] for y in Y])
        b = with_bam.decode_batch_py(x[i:i + 1], infos[i:i + 1], [y[i:i + 1] for y in Y])
        if a != b:
            changed += 1
            assert status[i] & 2, "candidate %d: the BAM changes its row but the native decoder did not flag it" % i
        assert bool(status[i] & 1) == bool(a)
    assert changed > 20 and int((status & 2).astype(bool).sum()) < len(infos)


def test_a_failing_stage_on_a_helper_thread_fails_call_variants():
    """call_variants runs the batch source and the decode on helper threads: a sys.exit / exception raised there (a failing
    `samtools view` inside callVarBam's generator, a malformed record, a decode error) must surface in the caller instead of
    reading as end of input (the reference checks its stages' exit codes, clair/callVarBam.py:218-233)."""
    import sys
    import types

    class Sink(object):
        rows = 0

        def write_header(self):
            pass

        def write_rows(self, rows):
            self.rows += len(rows)

    class Model(object):
        def predict(self, x):
            return [np.full((len(x), k), 1.0 / k, np.float32) for k in (21, 3, 33, 33)]

    def failing_source():
        yield np.zeros((2, 33, 8, 4), np.float32), [["c", "1", "A" * 33], ["c", "2", "A" * 33]]
        sys.exit("[ERROR] `samtools view` failed")

    dec = cvar.VariantDecoder(cvar.OutputConfig(True, False, False, False, False, None))
    with pytest.raises(SystemExit) as ei:
        cvar.call_variants(types.SimpleNamespace(tensor_fn=None), Model(), dec, Sink(), 2, generator=failing_source())
    assert "samtools view" in str(ei.value)

    class BadDecoder(object):
        def decode_batch(self, *a):
            raise ValueError("decode blew up")

    def two_batches():
        for _ in range(2):
            yield np.zeros((2, 33, 8, 4), np.float32), [["c", "1", "A" * 33], ["c", "2", "A" * 33]]

    with pytest.raises(ValueError):
        cvar.call_variants(types.SimpleNamespace(tensor_fn=None), Model(), BadDecoder(), Sink(), 2, generator=two_batches())


def test_binary_records_read_by_several_threads_are_the_single_readers_batches(tmp_path):
    """tensor_binary.read_batches_into on a regular file: batch k is read at its own offset by one of `readers` threads (round 4: one
    thread copying out of the page cache capped the 4096-batch pipeline); same batches, same order, ragged tail, dropped centres, and
    a consumer that stops early gets its buffers back."""
    import io
    from clair_amd import synth, tensor_binary as tb
    raw, infos = synth.synthetic_candidates(1030, "ont", seed=5)
    seqs = [i[2] for i in infos]
    seqs[7] = seqs[7][:16] + "-" + seqs[7][17:]           # not an IUPAC centre: dropped from its batch
    blob = tb.MAGIC + tb.pack_records(infos[0][0], [int(i[1]) for i in infos], seqs, raw)
    path = tmp_path / "t.bin"
    path.write_bytes(blob)

    def batches(readers, stop_after=None):
        pool = tb.BufferPool([np.zeros(128 * tb.RECORD.itemsize, np.uint8) for _ in range(6)])
        out = []
        with open(path, "rb", buffering=1 << 16) as f:
            assert f.read(len(tb.MAGIC)) == tb.MAGIC
            gen = tb.read_batches_into(f, 128, pool, readers=readers)
            for _, inf, counts, buf in gen:
                out.append((inf.rows(), np.array(counts)))
                pool.put(buf)
                if stop_after and len(out) == stop_after:
                    gen.close()
                    break
        return out, pool

    one, _ = batches(1)
    three, pool = batches(3)
    assert [len(b[0]) for b in one] == [127] + [128] * 7 + [6]
    assert len(one) == len(three) and all(a[0] == b[0] and np.array_equal(a[1], b[1]) for a, b in zip(one, three))
    assert pool._q.qsize() == 6
    part, pool = batches(3, stop_after=2)
    assert len(part) == 2 and pool._q.qsize() == 6            # the reads in flight gave their buffers back
    # a pipe (no offsets to read at) takes the sequential path
    pool = tb.BufferPool([np.zeros(128 * tb.RECORD.itemsize, np.uint8) for _ in range(3)])
    piped = []
    for _, inf, counts, buf in tb.read_batches_into(io.BytesIO(blob[len(tb.MAGIC):]), 128, pool, readers=3):
        piped.append((inf.rows(), np.array(counts)))
        pool.put(buf)
    assert len(piped) == len(one) and all(a[0] == b[0] and np.array_equal(a[1], b[1]) for a, b in zip(one, piped))
