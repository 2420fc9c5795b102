"""Weight container, synthetic initialiser and the TensorFlow-bundle reader (round trip through the
bundled writer; no real TF checkpoint is obtainable offline -- see clair_amd/tf_bundle.py)."""
import os

import numpy as np
import pytest

from clair_amd import tf_bundle, weights


def test_synthetic_weights_are_reproducible_and_shaped():
    a = weights.synthetic_weights(seed=1)
    b = weights.synthetic_weights(seed=1)
    c = weights.synthetic_weights(seed=2)
    weights.check_weights(a)
    for k in a:
        assert a[k].dtype == np.float32 and np.array_equal(a[k], b[k])
    assert not np.array_equal(a["l4_kernel"], c["l4_kernel"])
    # initialisers of the reference graph (clair/model.py:394-398; LSTM: scope default Glorot-uniform)
    assert abs(float(a["l4_kernel"].std()) - np.sqrt(1.3 / 7680) * 0.88) < 2e-4      # truncated normal at 2 sigma
    assert float(np.abs(a["l4_kernel"]).max()) <= 2 * np.sqrt(1.3 / 7680) + 1e-7
    lim = np.sqrt(6.0 / (160 + 512))
    assert float(np.abs(a["lstm1_fw_kernel"]).max()) <= lim
    assert not a["l4_bias"].any() and not a["lstm2_bw_bias"].any()


def test_npz_container_roundtrip(tmp_path):
    w = weights.synthetic_weights(seed=3)
    path = weights.save_weights(str(tmp_path / "model"), w)
    assert path.endswith(".npz")
    for src in (str(tmp_path / "model"), path):
        r = weights.load_weights(src)
        for k in w:
            assert np.array_equal(w[k], r[k])
    with pytest.raises(FileNotFoundError):
        weights.load_weights(str(tmp_path / "missing"))
    bad = dict(w)
    bad["l4_bias"] = np.zeros(5, np.float32)
    with pytest.raises(ValueError):
        weights.check_weights(bad)


def test_tf_variable_name_table_covers_every_parameter():
    names = weights.tf_variable_names()
    assert len(names) == 8 + 512 + 2 + 8 + 8
    covered = {}
    for tf_name, (key, index) in names.items():
        shape = weights.TENSOR_TABLE[key]
        covered[key] = covered.get(key, 0) + int(np.prod(shape if index is None else shape[1:]))
    assert covered == {k: int(np.prod(s)) for k, s in weights.TENSOR_TABLE.items()}
    assert "LSTM1/stack_bidirectional_rnn/cell_0/bidirectional_rnn/fw/cudnn_compatible_lstm_cell/kernel" in names
    assert "Prediction/Y_base_change_logits/kernel" in names and "L3/Unit_255/bias" in names


def test_tf_bundle_roundtrip_under_reference_variable_names(tmp_path):
    w = weights.synthetic_weights(seed=4, lstm_bias_scale=0.1)
    prefix = str(tmp_path / "ckpt" / "model-000020")
    os.makedirs(os.path.dirname(prefix))
    tf_bundle.export_checkpoint(prefix, w)
    assert os.path.isfile(prefix + ".index") and os.path.isfile(prefix + ".data-00000-of-00001")
    entries = tf_bundle.read_index(prefix + ".index")
    assert len(entries) == 538
    assert entries["L4/kernel"]["shape"] == (7680, 192) and entries["L4/kernel"]["dtype"] == tf_bundle.DT_FLOAT
    r = weights.load_weights(prefix)          # restore_parameters() hands over exactly this prefix
    for k in w:
        assert np.array_equal(w[k], r[k]), k


def test_tf_bundle_rejects_garbage(tmp_path):
    p = tmp_path / "x.index"
    p.write_bytes(b"\x00" * 100)
    with pytest.raises(ValueError):
        tf_bundle.read_index(str(p))
