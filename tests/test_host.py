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
    assert lib.clair_host_abi_version() == 1


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
