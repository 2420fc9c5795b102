"""Host-side native helpers (include/clair_host.h, libclair_host.so): ingest parser against the line-by-line Python restatement
of the reference (which tests/test_decode.py pins to fixtures minted from the real reference)."""
import gzip
import io
import os
import re
from contextlib import redirect_stderr

import numpy as np
import pytest

from clair_amd import _hostapi, synth, utils

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_header_symbols_are_exported():
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "clair_host.h")).read(), flags=re.S)
    declared = sorted(set(re.findall(r"\b(clair_host_[a-z0-9_]+)\s*\(", text)))
    lib = _hostapi.load()
    assert declared and sorted(_hostapi.SYMBOLS) == declared
    for name in declared:
        assert hasattr(lib, name)
    assert lib.clair_host_abi_version() == 6


def _collect(gen, path, batch):
    err = io.StringIO()
    out = []
    with redirect_stderr(err):
        for X, infos in gen(path, batch):
            out.append((np.array(X, copy=True), [list(i) for i in infos]))
    return out, err.getvalue()


@pytest.mark.parametrize("n,batch", [(0, 8), (1, 8), (8, 8), (9, 8), (500, 64), (1500, 1000)])
def test_native_ingest_equals_python_ingest(tmp_path, n, batch):
    raw, infos = synth.synthetic_candidates(max(n, 1), "illumina", seed=90 + n)
    lines = list(synth.tensor_records(raw, infos))[:n]
    # sprinkle the irregular cases the format allows: non-IUPAC centre bases (dropped rows), tabs, doubled blanks, '+' signs,
    # a float-looking value, no trailing newline
    for k in range(0, n, 7):
        cols = lines[k].split()
        cols[2] = cols[2][:16] + "Z" + cols[2][17:]
        lines[k] = " ".join(cols)
    for k in range(3, n, 11):
        cols = lines[k].split()
        cols[5] = "+" + cols[5].lstrip("-")
        cols[9] = "7.0"
        lines[k] = "\t".join(cols[:6]) + "  " + " ".join(cols[6:])
    path = str(tmp_path / "t.txt.gz")
    with gzip.open(path, "wt") as f:
        f.write("\n".join(l.rstrip("\n") for l in lines))      # last line without '\n'
    got, err_g = _collect(utils.tensor_generator_from, path, batch)
    want, err_w = _collect(utils.tensor_generator_from_py, path, batch)
    assert err_g == err_w
    assert len(got) == len(want)
    for (xg, ig), (xw, iw) in zip(got, want):
        assert xg.dtype == np.float32 and xg.tobytes() == xw.tobytes()
        assert ig == iw
    # the same records as an uncompressed file (read directly instead of through `gzip -fdc`'s pass-through): the same batches
    plain = str(tmp_path / "t.txt")
    with open(plain, "w") as f:
        f.write("\n".join(l.rstrip("\n") for l in lines))
    got_plain, err_p = _collect(utils.tensor_generator_from, plain, batch)
    assert err_p == err_w and len(got_plain) == len(want)
    for (xg, ig), (xw, iw) in zip(got_plain, want):
        assert xg.tobytes() == xw.tobytes() and ig == iw


@pytest.mark.parametrize("mutation", ["short", "extra_head", "two_head", "bad_value", "short_seq", "blank"])
def test_native_ingest_rejects_what_the_reference_rejects(tmp_path, mutation):
    raw, infos = synth.synthetic_candidates(3, "ont", seed=5)
    lines = [l.rstrip("\n") for l in synth.tensor_records(raw, infos)]
    cols = lines[1].split()
    if mutation == "short":
        cols = cols[:-1]
    elif mutation == "extra_head":
        cols = ["x"] + cols
    elif mutation == "two_head":
        cols = cols[1:]
    elif mutation == "bad_value":
        cols[40] = "1x"
    elif mutation == "short_seq":
        cols[2] = cols[2][:10]
    lines[1] = "" if mutation == "blank" else " ".join(cols)
    path = str(tmp_path / "bad.txt.gz")
    with gzip.open(path, "wt") as f:
        f.write("\n".join(lines) + "\n")
    with pytest.raises(Exception):
        _collect(utils.tensor_generator_from_py, path, 8)      # ValueError / IndexError in the reference's code path
    with pytest.raises(ValueError):
        _collect(utils.tensor_generator_from, path, 8)


# ---------------------------------------------------------------------------------------------------------------------
# native decode (clair_host_decode_rows) against the Python decoder, which tests/test_decode.py pins to the reference
# ---------------------------------------------------------------------------------------------------------------------
def _random_probs(rng, n, k, peak):
    logits = rng.standard_normal((n, k)).astype(np.float32) * peak
    tie = rng.random(n) < 0.15                       # exact ties between two classes
    logits[tie, 1] = logits[tie, 0]
    e = np.exp(logits - logits.max(axis=1, keepdims=True)).astype(np.float32)
    p = (e / e.sum(axis=1, keepdims=True)).astype(np.float32)
    p[rng.random((n, k)) < 0.05] = 0.0               # exact zeros
    return p


@pytest.mark.parametrize("platform,peak", [("ont", 0.5), ("illumina", 2.0), ("pacbio_ccs", 6.0), ("ont", 12.0)])
@pytest.mark.parametrize("arith", ["legacy", "numpy2"])
def test_native_decode_equals_python_decode(platform, peak, arith):
    from clair_amd import call_var as cvar
    rng = np.random.default_rng(int(peak * 10) + len(platform))
    n = 1500
    x, infos = synth.synthetic_input(n, platform, seed=int(peak * 7) + 3)
    infos = [list(i) for i in infos]
    for k in range(0, n, 97):
        infos[k][2] = infos[k][2][:16] + "N" + infos[k][2][17:]      # not callable: no row
    for k in range(5, n, 131):
        infos[k][2] = infos[k][2][:16] + "U" + infos[k][2][17:]
    for k in range(9, n, 211):
        x[k, 16, :, :] = 0                                           # read depth zero: no row
    Y = [_random_probs(rng, n, 21, peak), _random_probs(rng, n, 3, peak), _random_probs(rng, n, 33, peak),
         _random_probs(rng, n, 33, peak)]
    configs = [cvar.OutputConfig(True, False, False, False, False, None), cvar.OutputConfig(False, False, False, False, False, 30),
               cvar.OutputConfig(True, False, True, False, False, None), cvar.OutputConfig(False, False, False, True, False, 100)]
    for cfg in configs:
        native = cvar.VariantDecoder(cfg, arith=arith)
        python = cvar.VariantDecoder(cfg, arith=arith, native=False)
        try:
            want = python.decode_batch(x, infos, Y)
        except (ValueError, ZeroDivisionError):
            with pytest.raises(ValueError):                          # numpy2 quality score at p == 1: both raise
                native.decode_batch(x, infos, Y)
            continue
        assert native.decode_batch(x, infos, Y) == want


def test_native_decode_is_bypassed_where_it_does_not_apply():
    from clair_amd import call_var as cvar
    x, infos = synth.synthetic_input(8, "ont", seed=1)
    rng = np.random.default_rng(0)
    Y = [_random_probs(rng, 8, k, 2.0) for k in (21, 3, 33, 33)]
    debug = cvar.VariantDecoder(cvar.OutputConfig(True, True, False, False, False, None))
    assert debug.decode_batch(x, infos, Y) == debug.decode_batch_py(x, infos, Y)          # --debug rows come from Python
    ens = cvar.VariantDecoder(cvar.OutputConfig(True, False, False, False, True, None))
    assert ens.decode_batch(x, infos, Y) == ens.decode_batch_py(x, infos, Y)
    assert cvar.VariantDecoder(cvar.OutputConfig(True, False, False, False, False, None)).decode_batch(x[:0], [], [y[:0] for y in Y]) == []


class _Dribble(io.RawIOBase):
    """A pipe-like stream: read(n) hands out at most `step` bytes."""

    def __init__(self, data, step):
        self.data, self.at, self.step = data, 0, step

    def read(self, n=-1):
        n = self.step if n < 0 else min(n, self.step)
        out = self.data[self.at:self.at + n]
        self.at += len(out)
        return out


def test_binary_record_reader_batches_like_the_text_reader_and_feeds_the_native_decoder_its_columns():
    """tensor_binary.read_batches: batch_size records TAKEN per batch, non-IUPAC / missing centre bases dropped from it, ragged
    last batch, short reads, a caller-supplied prefix, truncated tails rejected -- and the InfoTable it yields is the same list of
    [ctg, pos, seq] the text reader builds, whose columns the native decoder takes as they are (byte-identical rows)."""
    from clair_amd import call_var as cvar, tensor_binary
    n = 1000
    raw, infos = synth.synthetic_candidates(n, "ont", seed=5)
    infos = [list(i) for i in infos]
    for k in range(0, n, 37):
        infos[k][2] = infos[k][2][:16] + "Z" + infos[k][2][17:]      # not an IUPAC code: dropped
    for k in range(3, n, 101):
        infos[k][2] = infos[k][2][:12]                               # no centre base at all: dropped
    for k in range(7, n, 53):
        infos[k][2] = infos[k][2][:16] + "N" + infos[k][2][17:]      # kept by the reader (IUPAC), skipped by the decoder
    infos[11][1] = "123456789012"                                    # a long position
    blob = b"".join(tensor_binary.pack_records("chr%d" % (k % 3), [int(infos[k][1])], [infos[k][2]], raw[k:k + 1]) for k in range(n))
    kept = [k for k in range(n) if len(infos[k][2]) > 16 and infos[k][2][16] in "ACGTURYSWKMBDHVN"]
    want_infos = [["chr%d" % (k % 3), infos[k][1], infos[k][2]] for k in kept]
    for batch, step, first in ((64, 1 << 30, b""), (64, 999, b""), (300, 7000, blob[:5000]), (2000, 1 << 30, b"")):
        with redirect_stderr(io.StringIO()):
            got = list(tensor_binary.read_batches(_Dribble(blob[len(first):], step), batch, first=first))
        taken = 0
        for x, inf, counts in got:
            lo = sum(1 for k in kept if k < taken)
            taken += batch
            hi = sum(1 for k in kept if k < taken)
            assert len(inf) == hi - lo and list(inf) == want_infos[lo:hi] and inf[0] == want_infos[lo] and inf[1:3] == want_infos[lo + 1:lo + 3]
            assert counts.dtype == np.int16 and np.array_equal(counts, raw[kept[lo:hi]])
            assert np.array_equal(x, synth.to_model_input(raw[kept[lo:hi]]))
        assert sum(len(g[1]) for g in got) == len(kept)
    with pytest.raises(ValueError), redirect_stderr(io.StringIO()):
        list(tensor_binary.read_batches(io.BytesIO(blob + b"tail"), 64))
    # the native decoder over the record columns == over the list of strings
    with redirect_stderr(io.StringIO()):
        x, table, _ = next(tensor_binary.read_batches(io.BytesIO(blob), n))
    rng = np.random.default_rng(3)
    Y = [_random_probs(rng, len(table), k, 4.0) for k in (21, 3, 33, 33)]
    dec = cvar.VariantDecoder(cvar.OutputConfig(True, False, False, False, False, None))
    rows = dec.decode_batch(x, table, Y)
    assert rows == dec.decode_batch(x, [list(i) for i in table], Y) and len(rows) > 800
    assert rows == cvar.VariantDecoder(cvar.OutputConfig(True, False, False, False, False, None), native=False).decode_batch(x, table, Y)


def test_native_counts_to_input_equals_the_reference_arithmetic():
    """clair_host_counts_to_input_*: float32 conversion, then channels 1..3 -= channel 0 (clair/utils.py:81-83, 96-98), for int16
    and int32 counts including the extremes of the range."""
    rng = np.random.default_rng(4)
    for dtype, lo, hi in ((np.int16, -32768, 32767), (np.int32, -100000, 100000)):
        c = rng.integers(lo, hi, size=(300, 33, 8, 4), endpoint=True).astype(dtype)
        c[0] = hi
        c[1] = lo
        want = c.astype(np.float32)
        want[:, :, :, 1:] -= want[:, :, :, 0:1]
        assert np.array_equal(_hostapi.counts_to_input(c), want)
    assert _hostapi.counts_to_input(np.zeros((0, 33, 8, 4), np.int16)).shape == (0, 33, 8, 4)


def test_in_process_inflate_reads_what_gzip_fdc_writes(tmp_path):
    """utils._InflateReader (gz tensor files are inflated in-process) against the `gzip -fdc` child it replaces: one member, several
    members (an empty one among them), bytes behind the last member (passed through by -f), an empty member alone; odd read sizes."""
    import subprocess
    rng = np.random.default_rng(0)
    base = "".join("line %d %s\n" % (i, "x" * int(rng.integers(0, 200))) for i in range(20000)).encode()
    cases = {"single": gzip.compress(base),
             "multi": gzip.compress(base[:100000]) + gzip.compress(base[100000:300000]) + gzip.compress(b"") + gzip.compress(base[300000:]),
             "trailing": gzip.compress(base[:5000]) + b"\0\0\0\0garbage", "trailing1": gzip.compress(base[:5000]) + b"x",
             "trailing_big": gzip.compress(base[:5000]) + base[:1500000], "empty_member": gzip.compress(b"")}
    for name, blob in cases.items():
        path = str(tmp_path / (name + ".gz"))
        with open(path, "wb") as f:
            f.write(blob)
        want = subprocess.run(["gzip", "-fdc", path], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout
        for n in (7, 1000, 1 << 23):
            r = utils._InflateReader(path)
            got = b""
            while True:
                c = r.read(n)
                if not c:
                    break
                assert len(c) <= max(n, 2) or name.startswith("trailing")      # pass-through pieces come as read from the file
                got += c
            r.close()
            assert got == want, (name, n)


def test_text_reader_hands_the_native_decoder_the_parsed_fields_as_bytes(tmp_path):
    """utils.tensor_generator_from yields a MetaInfoTable (the ctg / pos / seq fields as the parser found them): list-like, and the
    native decoder gives the same rows over it as over the list of strings -- across parse-buffer boundaries (batches spanning two
    reads of the file) and with dropped rows."""
    from clair_amd import call_var as cvar
    n = 3000
    raw, infos = synth.synthetic_candidates(n, "pacbio_ccs", seed=12)
    lines = list(synth.tensor_records(raw, infos))
    for k in range(0, n, 13):
        cols = lines[k].split()
        cols[2] = cols[2][:16] + "Z" + cols[2][17:]
        lines[k] = " ".join(cols)
    path = str(tmp_path / "t.txt")
    with open(path, "w") as f:
        f.write("\n".join(l.rstrip("\n") for l in lines) + "\n")
    rng = np.random.default_rng(8)
    dec = cvar.VariantDecoder(cvar.OutputConfig(True, False, False, False, False, None))
    total = 0
    with redirect_stderr(io.StringIO()):
        for x, table in utils.tensor_generator_from(path, 700):
            assert isinstance(table, _hostapi.MetaInfoTable) and len(table) == len(x)
            as_list = [list(i) for i in table]
            assert table == as_list and table[0] == as_list[0] and table[2:5] == as_list[2:5]
            Y = [_random_probs(rng, len(x), k, 3.0) for k in (21, 3, 33, 33)]
            rows = dec.decode_batch(x, table, Y)
            assert rows == dec.decode_batch(x, as_list, Y) and len(rows) > 0.9 * len(x)
            total += len(x)
    assert total == n - len(range(0, n, 13))


# ---------------------------------------------------------------------------------------------------------------------
# the decode in two halves around the 32-byte call record (include/clair_call.h): resolve (arithmetic) | format (text)
# ---------------------------------------------------------------------------------------------------------------------
def decode_cases(n=3000, seed=77, platform="ont", peak=6.0):
    """Candidates that walk every branch of the decode: peaky random probabilities with exact ties and exact zeros, non-callable
    and 'U' centres, a zero-depth window, truncated reference windows (deletions that cannot be written down)."""
    rng = np.random.default_rng(seed)
    x, infos = synth.synthetic_input(n, platform, seed=seed + 1)
    infos = [list(i) for i in infos]
    for k in range(0, n, 97):
        infos[k][2] = infos[k][2][:16] + "N" + infos[k][2][17:]
    for k in range(5, n, 131):
        infos[k][2] = infos[k][2][:16] + "U" + infos[k][2][17:]
    for k in range(9, n, 61):
        infos[k][2] = infos[k][2][:17 + (k % 9)]                       # 17 .. 25 characters: short or empty deletion text
    x[7] = 0.0
    Y = [_random_probs(rng, n, k, peak) for k in (21, 3, 33, 33)]
    hot = rng.random(n) < 0.5                                          # make indel outcomes win often, long ones included
    Y[1][hot] = np.float32([0.02, 0.49, 0.49])
    for a in (Y[2], Y[3]):
        a[hot, 16] *= np.float32(0.01)
        a[hot, 32] += np.float32(0.3) * (rng.random(int(hot.sum())) < 0.2)
        a[hot, 0] += np.float32(0.3) * (rng.random(int(hot.sum())) < 0.2)
    return x, infos, Y


@pytest.mark.parametrize("arith", ["legacy", "numpy2"])
def test_resolve_then_format_is_the_native_decode(arith):
    x, infos, Y = decode_cases()
    centre = _hostapi.centre_bytes(infos)
    assert centre.shape == (len(infos), 2) and centre[0, 1] == 33 and chr(centre[0, 0]) == infos[0][2][16]
    calls = _hostapi.resolve_calls(x, Y, centre)
    for cfg in ((True, False, False, None), (False, False, False, 100), (True, True, False, None), (True, False, True, 50)):
        want, st_w = _hostapi.decode_rows(x, infos, Y, *cfg, arith == "numpy2", with_status=True)
        got, st_g = _hostapi.format_calls(calls, infos, *cfg, arith == "numpy2", with_status=True)
        assert got == want and np.array_equal(st_g, st_w)
    # the record itself: every family wins somewhere, fall-throughs happen, flags carry the ties, the probability is a product of two inputs
    ok = calls["status"] & 1 == 1
    assert set(np.unique(calls["family"][ok])) == set(range(10))
    assert (calls["rounds"][ok] > 1).any() and (calls["status"] & 2).any() and (calls["status"] & 8).any()
    multi_flag = np.array([bin(int(f)).count("1") for f in calls["flags"][ok]])
    assert (multi_flag >= 1).all() and (multi_flag > 1).any()
    assert not ok[7] and not ok[0] and ok[5]                            # zero depth / 'N' centre / 'U' centre
    i = int(np.flatnonzero(ok & (calls["gi"] < 21))[0])
    zi = {0: 0, 1: 1, 2: 2, 3: 2}[int(calls["gt"][i])]
    assert calls["p_call"][i] == Y[0][i, calls["gi"][i]] * Y[1][i, zi]
    assert calls.dtype.itemsize == 32


def test_format_rejects_a_record_whose_class_contradicts_its_strings():
    x, infos, Y = decode_cases(n=200)
    calls = _hostapi.resolve_calls(x, Y, _hostapi.centre_bytes(infos))
    i = int(np.flatnonzero((calls["status"] & 1 == 1) & (calls["family"] == 1) & (calls["status"] & 8 == 0))[0])
    calls["gi"][i] = (calls["gi"][i] + 1) % 21
    with pytest.raises(ValueError, match="gt21 class"):
        _hostapi.format_calls(calls, infos, True, False, False, None, False)


def test_formatting_from_record_columns_equals_formatting_from_the_text_table():
    """A batch of binary tensor records hands its columns to the formatter as they are (clair_host_format_calls_records) and the rows
    come back as the bytes that go into the file: the same bytes as the list-of-strings path over the [[ctg, pos, seq], ...] table."""
    import io
    from contextlib import redirect_stderr
    from clair_amd import tensor_binary
    x, infos, Y = decode_cases(n=700)
    keep = [i for i, inf in enumerate(infos) if len(inf[2]) == 33 and inf[2][16] in "ACGTURYSWKMBDHVN"]
    x, infos, Y = x[keep], [infos[i] for i in keep], [a[keep] for a in Y]
    raw = np.rint(np.concatenate([x[..., :1], x[..., 1:] + x[..., :1]], axis=-1)).astype(np.int16)          # back to raw counts
    buf = tensor_binary.pack_records(infos[0][0], [int(i[1]) for i in infos], [i[2] for i in infos], raw)
    with redirect_stderr(io.StringIO()):
        (_, table, counts), = list(tensor_binary.read_batches(io.BytesIO(buf), 4096, with_input=False))
    assert list(table) == [list(i) for i in infos] and np.array_equal(_hostapi.centre_bytes(table), _hostapi.centre_bytes(infos))
    calls = _hostapi.resolve_calls(x, Y, _hostapi.centre_bytes(infos))
    for cfg in ((True, False, False, None), (False, False, True, 40)):
        rows = _hostapi.format_calls(calls, infos, *cfg, False)
        text, status = _hostapi.format_calls(calls, table, *cfg, False, with_status=True, as_text=True)
        assert text == ("\n".join(rows) + "\n").encode() and int((status & 1).sum()) == len(rows)
