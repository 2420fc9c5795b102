"""GPU decode (clair_amd/csrc/decode.hip.h) against its CPU twin (clair_host_resolve_calls) and against the full native decode.

Bit-exact: the call record is integers plus three float32 values that are sums / products of inputs in a fixed order."""
import numpy as np
import pytest

from clair_amd import _hostapi, synth

pytestmark = pytest.mark.gpu


def _cases(n, seed, platform="ont", peak=6.0):
    from test_host import decode_cases
    return decode_cases(n=n, seed=seed, platform=platform, peak=peak)


def _same_records(got, want):
    for name in want.dtype.names:
        if name.startswith("reserved"):
            continue
        a, b = got[name], want[name]
        same = a.view(np.uint32) == b.view(np.uint32) if a.dtype.kind == "f" else a == b
        assert same.all(), "%s differs at %s: device %s, host %s" % (name, np.flatnonzero(~same)[:8], a[~same][:8], b[~same][:8])


@pytest.mark.parametrize("platform,peak,seed", [("ont", 6.0, 77), ("illumina", 2.0, 5), ("pacbio_ccs", 12.0, 9), ("ont", 0.5, 11)])
def test_device_records_equal_the_host_records_bit_for_bit(engine, platform, peak, seed):
    """Crafted probabilities -- exact ties between classes, exact zeros, indel-heavy rows, truncated reference windows, non-callable and
    'U' centres, a zero-depth window -- through clair_decode: every field of every record equals clair_host_resolve_calls'."""
    x, infos, Y = _cases(1000, seed, platform, peak)
    centre = _hostapi.centre_bytes(infos)
    want = _hostapi.resolve_calls(x, Y, centre)
    got = engine.decode(x, Y, centre)
    _same_records(got, want)
    ok = want["status"] & 1 == 1
    assert ok.sum() > 900 and (want["rounds"][ok] > 1).any() and len(np.unique(want["family"][ok])) >= 8
    # and the rows: format(device records) == the native decode of the same probabilities
    for cfg in ((True, False, False, None), (False, True, False, 30)):
        assert _hostapi.format_calls(got, infos, *cfg, False) == _hostapi.decode_rows(x, infos, Y, *cfg, False)


def test_device_records_format_to_the_reference_rows_minted_with_numpy1_promotion(engine):
    """The default arithmetic end to end on the device side: clair_decode's records -> clair_host_format_calls (float64 QUAL / AF) == the
    reference's rows under NumPy 1.x's scalar promotion (tests/golden/decode_rows_legacy.json.gz), read depth 160 and certain calls included."""
    from test_decode import CONFIGS, legacy_fixture
    for X, P, infos, name, rows in legacy_fixture():
        if name.endswith("debug"):      # the debug lines are the Python writer's
            continue
        show_ref, _debug, hp, hs, _ens, qual = CONFIGS[name.replace("extra_", "")]
        Y = [np.ascontiguousarray(P[:, 0:21]), np.ascontiguousarray(P[:, 21:24]), np.ascontiguousarray(P[:, 24:57]), np.ascontiguousarray(P[:, 57:90])]
        got = engine.decode(X, Y, _hostapi.centre_bytes(infos))
        assert _hostapi.format_calls(got, infos, show_ref, hp, hs, qual, False) == [ln for per in rows for ln in per]


def test_products_that_underflow_keep_their_denormal_bits(engine):
    """Tiny probabilities: products in the float32 denormal range (and below: exact zeros by underflow) must tie and order exactly as
    on the host, where SSE arithmetic keeps denormals."""
    rng = np.random.default_rng(3)
    n = 512
    x, infos = synth.synthetic_input(n, "ont", seed=21)
    Y = []
    for k in (21, 3, 33, 33):
        e = rng.uniform(-45.0, 0.0, size=(n, k))
        Y.append(np.power(10.0, e).astype(np.float32))          # 1e-45 .. 1: pairs and triples of these underflow
    centre = _hostapi.centre_bytes(infos)
    want = _hostapi.resolve_calls(x, Y, centre)
    _same_records(engine.decode(x, Y, centre), want)
    p = want["p_call"][want["status"] & 1 == 1]
    assert ((p > 0) & (p < 1.2e-38)).any() or (p == 0).any()    # the set really reaches the denormal range


def test_nan_probabilities_resolve_to_no_call_on_both_sides(engine):
    x, infos = synth.synthetic_input(8, "ont", seed=2)
    Y = [np.full((8, k), np.nan, np.float32) for k in (21, 3, 33, 33)]
    centre = _hostapi.centre_bytes(infos)
    _same_records(engine.decode(x, Y, centre), _hostapi.resolve_calls(x, Y, centre))


@pytest.mark.parametrize("counts", [False, True])
def test_submit_ex_returns_the_records_of_its_own_probabilities(engine, synth_weights, counts):
    """Forward pass + decode in one submit: the records equal the host resolution of the probabilities the same submit returns, and
    without the probabilities (records only) they are the same records."""
    raw, infos = synth.synthetic_candidates(1000, "ont", seed=31)
    x = synth.to_model_input(raw)
    centre = _hostapi.centre_bytes(infos)
    batch = raw.astype(np.int16) if counts else x
    engine.submit_calls(0, batch, centre, counts=counts, with_probabilities=True)
    calls, Y = engine.wait(0)
    _same_records(calls, _hostapi.resolve_calls(x, Y, centre))
    engine.submit_calls(1, batch, centre, counts=counts)
    only = engine.wait(1)
    _same_records(only, calls)
    assert _hostapi.format_calls(only, infos, True, False, False, None, False) == _hostapi.decode_rows(x, infos, Y, True, False, False, None, False)


def test_call_records_survive_a_fused_launch_failure(synth_weights, monkeypatch):
    """clair_submit_ex on a handle whose fused layer-2 launch reports a placement failure (CLAIR_AMD_FUSED_FAULT): the engine re-runs the
    forward pass on the two-launch path AND the decode behind it; the records are the ones an undisturbed handle returns."""
    from clair_amd import _capi
    raw, infos = synth.synthetic_candidates(1024, "ont", seed=41)
    counts, centre = raw.astype(np.int16), _hostapi.centre_bytes(infos)
    want = None
    monkeypatch.setenv("CLAIR_AMD_LSTM2_FUSED", "1")        # the fused launch is opt-in since round 5
    for fault in (None, "2"):
        if fault is None:
            monkeypatch.delenv("CLAIR_AMD_FUSED_FAULT", raising=False)
        else:
            monkeypatch.setenv("CLAIR_AMD_FUSED_FAULT", fault)
        eng = _capi.Engine(device=0, max_batch=1024, n_slots=2)
        try:
            eng.load_weights(synth_weights)
            got = []
            for rep in range(3):
                eng.submit_calls(rep % 2, counts, centre, counts=True)
                got.append(eng.wait(rep % 2))
            if fault is None:
                want = got[0]
                assert eng.counter("fused_recoveries") == 0
            else:
                assert eng.counter("fused_recoveries") == 1
            for g in got:
                _same_records(g, want)
        finally:
            eng.close()


def test_counts_in_the_engines_pinned_buffer_go_to_the_device_from_where_they_lie(engine, synth_weights):
    """clair_submit_ex on a strided view INTO a buffer of clair_pinned_alloc (binary tensor records read in place): a 2-D DMA of the
    counts column, no staging copy -- the same records as the dense, pageable submit."""
    from clair_amd import tensor_binary
    raw, infos = synth.synthetic_candidates(900, "ont", seed=51)
    centre = _hostapi.centre_bytes(infos)
    engine.submit_calls(0, raw.astype(np.int16), centre, counts=True)
    want = engine.wait(0)
    packed = np.frombuffer(tensor_binary.pack_records(infos[0][0], [int(i[1]) for i in infos], [i[2] for i in infos], raw), dtype=np.uint8)
    buf = engine.pinned_buffer(1024 * tensor_binary.RECORD.itemsize)
    buf[:packed.size] = packed
    rec = buf[:packed.size].view(tensor_binary.RECORD)
    assert not rec["counts"].flags.c_contiguous and rec["counts"].strides[0] == 2192
    for rep in range(2):
        engine.submit_calls(rep, rec["counts"], centre, counts=True)
        _same_records(engine.wait(rep), want)
