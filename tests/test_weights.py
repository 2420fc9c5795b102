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


# ---- the reader against bytes assembled independently of its own writer (tools/make_tf_bundle_fixture.py) ----
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_crc32c_known_answers():
    assert tf_bundle._crc32c(b"123456789") == 0xE3069283            # RFC 3720 B.4
    assert tf_bundle._crc32c(b"\x00" * 32) == 0x8A9136AA
    big = bytes(range(256)) * 64                                      # >= 4096 bytes: the host-library path
    table_only = 0xFFFFFFFF
    tf_bundle._crc32c(b"x")                                           # builds the table
    for b in big:
        table_only = tf_bundle._CRC_TABLE[(table_only ^ b) & 0xFF] ^ (table_only >> 8)
    assert tf_bundle._crc32c(big) == table_only ^ 0xFFFFFFFF


def test_reader_on_a_bundle_assembled_from_the_format_description():
    import json
    prefix = os.path.join(GOLD, "tf_bundle_small")
    want = json.load(open(prefix + ".json"))
    entries = tf_bundle.read_index(prefix + ".index")
    assert sorted(entries) == sorted(want)
    for name, e in entries.items():
        assert e["dtype"] == want[name]["dtype"] and list(e["shape"]) == want[name]["shape"], name
    tensors = tf_bundle.read_tensors(prefix)
    assert "global_step" not in tensors                               # DT_INT64
    assert "emb" in tensors and "emb/part_0" not in tensors           # /part_N pieces joined along axis 0
    assert tensors["emb"].shape == (5, 5)
    assert tensors["emb"].ravel().tolist() == want["emb/part_0"]["values"] + want["emb/part_1"]["values"]
    for name in ("L4/bias", "L4/bias/Adam", "L5_1/kernel", "Prediction/Y_genotype_logits/bias", "Training_Operation/beta1_power"):
        assert tensors[name].shape == tuple(want[name]["shape"])
        assert np.array_equal(tensors[name].ravel(), np.array(want[name]["values"], dtype=np.float32)), name


def test_reader_detects_corruption(tmp_path):
    import shutil
    for suffix in (".index", ".data-00000-of-00001"):
        shutil.copy(os.path.join(GOLD, "tf_bundle_small" + suffix), str(tmp_path / ("c" + suffix)))
    idx = bytearray((tmp_path / "c.index").read_bytes())
    idx[10] ^= 0x40                                                   # inside the first (snappy) block
    (tmp_path / "bad.index").write_bytes(bytes(idx))
    with pytest.raises(ValueError, match="CRC32C"):
        tf_bundle.read_index(str(tmp_path / "bad.index"))
    data = bytearray((tmp_path / "c.data-00000-of-00001").read_bytes())
    data[5] ^= 1
    (tmp_path / "c.data-00000-of-00001").write_bytes(bytes(data))
    with pytest.raises(ValueError, match="CRC32C"):
        tf_bundle.read_tensors(str(tmp_path / "c"))
    assert "L4/bias" in tf_bundle.read_tensors(str(tmp_path / "c"), verify_crc=False)


def test_snappy_decoder_rejects_bad_streams():
    assert tf_bundle.snappy_decompress(b"\x05" + bytes([4 << 2]) + b"hello") == b"hello"
    assert tf_bundle.snappy_decompress(b"\x0a" + bytes([1 << 2]) + b"ab" + bytes([1 | (4 << 2), 2])) == b"ababababab"   # overlapping copy
    with pytest.raises(ValueError):
        tf_bundle.snappy_decompress(b"\x05" + bytes([4 << 2]) + b"hel")
    with pytest.raises(ValueError):
        tf_bundle.snappy_decompress(b"\x08" + bytes([1 << 2]) + b"ab" + bytes([1 | (0 << 2), 9]))                       # offset beyond the output
    with pytest.raises(ValueError):
        tf_bundle.snappy_decompress(b"\x09" + bytes([4 << 2]) + b"hello")


def test_checkpoint_with_other_variable_names_loads_through_fallback_and_override(tmp_path, monkeypatch):
    """The TF names are recalled, not verified: (1) the LSTM variables are also looked for inside the CudnnLSTM
    layer's own scope; (2) a JSON file maps expected -> actual names; (3) a miss lists what the checkpoint holds."""
    import json
    monkeypatch.delenv("CLAIR_AMD_TF_NAMES", raising=False)
    w = weights.synthetic_weights(seed=8)
    tensors = {}
    for tf_name, (key, index) in weights.tf_variable_names().items():
        a = w[key] if index is None else w[key][index]
        if tf_name.startswith("LSTM"):
            head, _, tail = tf_name.partition("/")
            tf_name = "%s/cudnn_lstm/%s" % (head, tail)
        elif tf_name.startswith("L4/"):
            tf_name = "Dense4/" + tf_name[3:]
        elif tf_name == "Prediction/Y_genotype_logits/bias":
            tf_name = "heads/zygosity_b"
        tensors[tf_name] = a
        if tf_name.endswith("kernel"):
            tensors[tf_name + "/Adam"] = np.zeros_like(a)              # optimizer slots ride along
    prefix = str(tmp_path / "model")
    tf_bundle.write_checkpoint(prefix, tensors)
    with pytest.raises(KeyError) as ei:
        weights.load_weights(prefix)
    msg = str(ei.value)
    assert "L4/kernel" in msg and "Dense4/kernel (7680, 192)" in msg and ".names.json" in msg and "/Adam" not in msg.split("float32 variables")[1]
    json.dump({"rename_prefix": {"L4/": "Dense4/"}, "rename": {"Prediction/Y_genotype_logits/bias": "heads/zygosity_b"}},
              open(prefix + ".names.json", "w"))
    r = weights.load_weights(prefix)
    for k in w:
        assert np.array_equal(w[k], r[k]), k
    json.dump({"renamed": {}}, open(prefix + ".names.json", "w"))
    with pytest.raises(ValueError, match="unknown keys"):
        weights.load_weights(prefix)
    # a variable of the wrong size is reported with its shape
    json.dump({"rename_prefix": {"L4/": "Dense4/"}, "rename": {"Prediction/Y_genotype_logits/bias": "L4/bias"}}, open(prefix + ".names.json", "w"))
    tensors["L4/bias"] = np.zeros(5, np.float32)
    tf_bundle.write_checkpoint(prefix, tensors)
    with pytest.raises(ValueError, match="has shape"):
        weights.load_weights(prefix)


def test_a_miss_names_the_shape_compatible_variables_and_the_override_line(tmp_path, monkeypatch):
    """A checkpoint that calls a tensor something else: the error proposes the unclaimed variables of the same shape and the
    override that would map them."""
    monkeypatch.delenv("CLAIR_AMD_TF_NAMES", raising=False)
    w = weights.synthetic_weights(seed=9)
    tensors = {}
    for tf_name, (key, index) in weights.tf_variable_names().items():
        a = w[key] if index is None else w[key][index]
        if tf_name == "LSTM1/stack_bidirectional_rnn/cell_0/bidirectional_rnn/fw/cudnn_compatible_lstm_cell/kernel":
            tf_name = "rnn_a/forward/weights"
        tensors[tf_name] = a
    prefix = str(tmp_path / "model")
    tf_bundle.write_checkpoint(prefix, tensors)
    with pytest.raises(KeyError) as ei:
        weights.load_weights(prefix)
    msg = str(ei.value)
    assert "unclaimed variables of the same shape (160, 512): rnn_a/forward/weights" in msg      # not the bw kernel: that one is claimed
    assert '"rename": {"LSTM1/stack_bidirectional_rnn/cell_0/bidirectional_rnn/fw/cudnn_compatible_lstm_cell/kernel": "rnn_a/forward/weights"}' in msg


def test_opaque_cudnn_lstm_buffers_are_converted_to_the_canonical_cell_form(tmp_path, monkeypatch):
    """A GPU-trained graph keeps each LSTM layer as ONE flat CudnnLSTM buffer (clair/model.py:281-296).  When a checkpoint carries the
    buffer itself instead of the saveable's canonical tensors, the loader converts it: weights [gate][unit][input] in cuDNN's gate
    order i, f, c, o (input part, then recurrent part, per direction), then two bias sets per direction that add up.  The buffer
    here is assembled from that description, independently of the converter."""
    monkeypatch.delenv("CLAIR_AMD_TF_NAMES", raising=False)
    w = weights.synthetic_weights(seed=10, lstm_bias_scale=0.1)
    rng = np.random.default_rng(3)
    tensors = {}
    for tf_name, (key, index) in weights.tf_variable_names().items():
        if not tf_name.startswith("LSTM"):
            tensors[tf_name] = w[key] if index is None else w[key][index]
    cudnn_gate_of_tf_block = {0: 0, 1: 2, 2: 1, 3: 3}        # TF column block (i, c~, f, o) -> position in cuDNN's (i, f, c, o)
    for layer, input_size in ((1, 32), (2, 256)):
        mats, biases = [], []
        for d in ("fw", "bw"):
            k, b = w["lstm%d_%s_kernel" % (layer, d)], w["lstm%d_%s_bias" % (layer, d)]
            blocks = {cudnn_gate_of_tf_block[j]: k[:, j * 128:(j + 1) * 128] for j in range(4)}
            mats += [blocks[g][:input_size].T.ravel() for g in range(4)] + [blocks[g][input_size:].T.ravel() for g in range(4)]
            bparts = {cudnn_gate_of_tf_block[j]: b[j * 128:(j + 1) * 128] for j in range(4)}
            split = [rng.standard_normal(128).astype(np.float32) for _ in range(4)]        # bW arbitrary, bR = b - bW
            biases += split + [bparts[g] - split[g] for g in range(4)]
        tensors["LSTM%d/cudnn_lstm/opaque_kernel" % layer] = np.concatenate(mats + biases)
    assert tensors["LSTM1/cudnn_lstm/opaque_kernel"].size == 165888
    prefix = str(tmp_path / "model")
    tf_bundle.write_checkpoint(prefix, tensors)
    r = weights.load_weights(prefix)
    for k in w:
        if k.endswith("_bias") and k.startswith("lstm"):
            assert np.abs(w[k] - r[k]).max() < 1e-6, k       # bW + bR: one float32 rounding away from b
        else:
            assert np.array_equal(w[k], r[k]), k
    with pytest.raises(ValueError, match="opaque CudnnLSTM buffer"):
        tf_bundle.cudnn_opaque_to_canonical(np.zeros(100, np.float32), 32, 128)


def test_reader_on_a_checkpoint_written_by_tensorflow_itself_when_present():
    """tools/mint_tf_golden.py --mini-checkpoint, run under TF 1.13, leaves a tf.train.Saver checkpoint of a 4-unit version of the
    graph (real tensor-bundle bytes, real variable names, an optimizer-scope variable and the int64 global step to skip) plus a JSON
    listing of what TF holds.  Until someone commits those files the reader has only ever read bundles written by this repository."""
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prefix = os.path.join(root, "tests", "golden", "tf113_mini")
    if not os.path.isfile(prefix + ".index"):
        pytest.skip("tests/golden/tf113_mini.* absent: no checkpoint written by TensorFlow itself has been read yet -- "
                    "`python tools/mint_tf_golden.py --mini-checkpoint` under tensorflow==1.13.2 mints it")
    check_reader_against_minted_checkpoint(prefix)


def check_reader_against_minted_checkpoint(prefix):
    """What the test above does with the committed files; tests/test_mint_tool.py runs it on a checkpoint minted under the stand-in TensorFlow."""
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    doc = json.load(open(prefix + ".json"))
    got = tf_bundle.read_tensors(prefix)
    floats = {n: v for n, v in doc["variables"].items() if v["dtype"] == "float32"}
    assert set(floats) <= set(got)
    for n, v in floats.items():
        assert list(got[n].shape) == v["shape"] and np.array_equal(got[n].ravel(), np.asarray(v["values"], dtype=np.float32)), n
    h = doc["widths"][0]
    import sys
    sys.path.insert(0, os.path.join(root, "tools"))
    import mint_tf_golden
    assert set(mint_tf_golden.tf_variable_names(h)) <= set(got)          # TF's names ARE the loader's names


def test_reader_on_a_cudnn_checkpoint_written_by_tensorflow_gpu_when_present():
    """tools/mint_tf_golden.py --cudnn-checkpoint, run under tensorflow-gpu 1.13 on a CUDA GPU, saves the two BiLSTM layers of the
    reference's GPU branch (tf.contrib.cudnn_rnn.CudnnLSTM, /root/reference/clair/model.py:281-296) through TF's own CudnnLSTMSaveable AND
    as raw opaque buffers, with the layers' outputs.  Until someone commits those files, the names under which a GPU-trained checkpoint
    stores its LSTM tensors and the opaque -> canonical conversion of clair_amd/tf_bundle.py are [TF-recall]."""
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prefix = os.path.join(root, "tests", "golden", "tf113_cudnn")
    if not os.path.isfile(prefix + ".index"):
        pytest.skip("tests/golden/tf113_cudnn.* absent: no CudnnLSTM checkpoint written by TensorFlow itself has been read yet -- "
                    "`python tools/mint_tf_golden.py --cudnn-checkpoint` under tensorflow-gpu==1.13.x (tools/pin/run.sh --gpu) mints it")
    check_reader_against_cudnn_checkpoint(prefix)


def check_reader_against_cudnn_checkpoint(prefix):
    """(1) every LSTM tensor of the loader's name table is in the saveable-written checkpoint under one of tf_bundle.candidate_names();
    (2) tf_bundle.cudnn_opaque_to_canonical of the raw buffers equals what the saveable wrote; (3) the oracle's BiLSTM layers with those
    tensors reproduce the CudnnLSTM layers' own outputs -- cuDNN and CudnnCompatibleLSTMCell compute the same function, which is what
    lets the reference train on one and call on the other (clair/model.py:299-312)."""
    import json
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import mint_tf_golden
    from oracle import c_oracle
    doc = json.load(open(prefix + ".json"))
    canon, raw = tf_bundle.read_tensors(prefix), tf_bundle.read_tensors(prefix + "_raw")
    assert sorted(raw) == sorted(doc["opaque"]) and len(raw) == 2
    w = {k: np.zeros(shape, np.float32) for k, shape in weights.TENSOR_TABLE.items()}
    for tf_name, (key, idx) in weights.tf_variable_names().items():
        if not tf_name.startswith("LSTM"):
            continue
        found = [n for n in tf_bundle.candidate_names(tf_name, {}, {}) if n in canon]
        assert found, "the saveable stored %s under none of %s; the checkpoint holds %s" % (tf_name, tf_bundle.candidate_names(tf_name, {}, {}), sorted(canon))
        assert idx is None and canon[found[0]].shape == w[key].shape
        w[key] = canon[found[0]]
    for name, values in doc["opaque"].items():
        layer = name.split("/")[0]
        assert np.array_equal(raw[name].ravel(), np.asarray(values, dtype=np.float32))
        parts = tf_bundle.cudnn_opaque_to_canonical(raw[name], 32 if layer == "LSTM1" else 256, 128)
        for (d, kind), value in parts.items():
            assert np.abs(value - w["lstm%s_%s_%s" % (layer[-1], d, kind)]).max() <= 1e-6, (name, d, kind)
    n = doc["n"]
    x = mint_tf_golden.golden_input("illumina300")[:n]
    _, inter = c_oracle.forward(w, x, keep_intermediates=True)
    _, inter64 = c_oracle.forward(w, x, keep_intermediates=True, dtype=np.float64)
    for key in ("a1", "a2"):
        theirs = np.asarray(doc[key], dtype=np.float32).reshape(33, n, 256).transpose(1, 0, 2)
        assert np.abs(inter[key] - theirs).max() <= max(1e-5, 4 * float(np.abs(inter[key] - inter64[key]).max())), key


def test_snappy_decoder_against_googles_own_compressor():
    """clair_amd/tf_bundle.py decodes the snappy blocks of a checkpoint's .index itself (python-snappy is not a dependency).  Its fixtures are
    compressed by this repository's own encoder; here the streams come from Google's snappy library as pyarrow ships it -- an independent,
    real implementation: literals of every length class, copies with 1-, 2- and 4-byte offsets, runs, incompressible data, 1 MB blocks."""
    pa = pytest.importorskip("pyarrow")
    if not pa.Codec.is_available("snappy"):
        pytest.skip("this pyarrow build has no snappy codec")
    rng = np.random.default_rng(7)
    words = [bytes(rng.integers(97, 123, int(n)).astype(np.uint8)) for n in rng.integers(1, 12, 400)]
    cases = [b"", b"a", b"ab" * 5, bytes(rng.integers(0, 256, 70000).astype(np.uint8)),                        # incompressible: literals > 60 and > 256 bytes
             b"".join(words[int(i)] for i in rng.integers(0, 400, 60000)),                                       # text-like: short copies, 1- and 2-byte offsets
             bytes(1 << 20),                                                                                      # one long run
             bytes(rng.integers(0, 256, 3000).astype(np.uint8)) * 300,                                            # period 3000: 2-byte offsets, long copies
             bytes(rng.integers(0, 4, 1 << 20).astype(np.uint8)),                                                 # low entropy
             b"".join(bytes(rng.integers(0, 256, 100).astype(np.uint8)) + bytes(70000) for _ in range(3))]        # copies further than 65535 back
    kinds = set()
    for data in cases:
        blob = pa.compress(data, codec="snappy", asbytes=True)
        assert tf_bundle.snappy_decompress(blob) == data
        at = 0
        while blob[at] & 0x80:                                   # skip the varint length, then note the element kinds present
            at += 1
        at += 1
        while at < len(blob):
            tag = blob[at]
            kind = tag & 3
            kinds.add(kind)
            if kind == 0:
                n = tag >> 2
                extra = max(0, n - 59) if n >= 60 else 0
                n = (int.from_bytes(blob[at + 1:at + 1 + extra], "little") if extra else n) + 1
                at += 1 + extra + n
            else:
                at += {1: 2, 2: 3, 3: 5}[kind]
    assert {0, 1, 2} <= kinds                                     # literals, 1-byte-offset and 2-byte-offset copies all occurred
