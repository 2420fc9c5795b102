"""The C-ABI library loads on a GPU-less host and exports every symbol include/clair_amd.h declares."""
import ctypes
import os
import re

import numpy as np
import pytest

from clair_amd import _capi, weights

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "clair_amd.h")


def _declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(clair_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported():
    lib = _capi.load()
    declared = _declared_functions()
    assert declared, "no declarations parsed from the header"
    for name in declared:
        assert hasattr(lib, name), "%s declared in clair_amd.h but not exported" % name
    assert sorted(_capi.SYMBOLS) == declared


def test_abi_version_and_tensor_table():
    lib = _capi.load()
    assert lib.clair_abi_version() == 6
    text = open(HEADER).read()
    ids = dict((m.group(1), int(m.group(2))) for m in re.finditer(r"CLAIR_T_([A-Z0-9_]+)\s*=\s*(\d+)", text))
    assert ids.pop("COUNT") == len(weights.TENSOR_TABLE) == 22
    for key, tid in weights.TENSOR_IDS.items():
        assert ids[key.upper()] == tid
    assert weights.N_PARAMS == 2377818


def test_create_fails_loudly_without_device():
    lib = _capi.load()
    if lib.clair_device_count() > 0:
        pytest.skip("a HIP device is present")
    with pytest.raises(_capi.EngineError) as ei:
        _capi.Engine(device=0, max_batch=16, n_slots=1)
    assert "no HIP device" in str(ei.value)


def test_create_argument_validation():
    lib = _capi.load()
    h = ctypes.c_void_p()
    assert lib.clair_engine_create(0, 0, 1, ctypes.byref(h)) != 0
    assert b"max_batch" in lib.clair_last_error(None)
    assert lib.clair_engine_create(0, 16, 0, ctypes.byref(h)) != 0
    assert b"n_slots" in lib.clair_last_error(None)
    assert not h.value
