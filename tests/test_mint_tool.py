"""tools/mint_tf_golden.py executed end to end -- under tests/fake_tf113.py, a NumPy stand-in for the slice of TensorFlow 1.13's API
the tool touches.  TensorFlow cannot be installed here, so the tool that is to pin the oracle to the real
reference arithmetic had never run; the first person with TF 1.13 should not have to debug it.  These tests prove that the tool
parses, builds the graph of clair/model.py:400-622 in an order the API accepts, finds every variable under the name the loader's
table expects (clair/model.py:712, 1016-1020 restore by exactly these names), slices the recipe weights onto them the way the engine's
tensor ids are laid out, and writes files the three consuming tests can read.  They prove NOTHING about TensorFlow's arithmetic: the
stand-in's rules are the same [TF-recall] the oracle restates, and what it mints goes to a temporary directory, never to tests/golden/."""
import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture
def mint(monkeypatch):
    import fake_tf113
    fake_tf113.reset_default_graph()
    monkeypatch.setitem(sys.modules, "tensorflow", fake_tf113)
    import mint_tf_golden
    return importlib.reload(mint_tf_golden)


def test_the_mint_tool_runs_end_to_end_and_its_file_is_what_the_oracle_test_reads(mint, tmp_path):
    out = str(tmp_path / "nn_tf113_64.npz")
    mint.mint(out)
    with np.load(out) as z:
        assert str(z["recipe"]) == mint.RECIPE and str(z["tf_version"]).endswith("-fake")
        assert z["gt21"].shape == (64, 21) and z["genotype"].shape == (64, 3) and z["len1"].shape == z["len2"].shape == (64, 33)
        assert z["a1_first4"].shape == z["a2_first4"].shape == (33, 4, 256) and z["l3_first4"].shape == (4, 7680) and z["l4"].shape == (64, 192) and z["l5"].shape == (64, 4, 96)
        assert np.allclose(z["gt21"].sum(axis=1), 1.0, atol=1e-5)
    # the consumer of the committed file, on this one: the graph the tool builds and the way it slices the recipe weights onto the
    # variables agree with the oracle's reading of the same tensors (a transposed L3 slice or a swapped L5 branch would show as 1e-1)
    import test_oracle
    test_oracle.check_oracle_against_minted_file(out)
    # round 6: the TRAINED-LIKE variant (LSTM kernels x4, forget bias +1, head gain 6 on 300x Illumina counts) through the same consumer
    out_t = str(tmp_path / "nn_tf113_trained_64.npz")
    mint.mint(out_t, "trained")
    with np.load(out_t) as z:
        assert str(z["variant"]) == "trained" and str(z["recipe"]) == mint.RECIPE + "-trained" and float(np.abs(z["a2_first4"]).max()) > 0.99
    test_oracle.check_oracle_against_minted_file(out_t)


def test_the_mint_tool_finds_every_variable_under_the_loaders_names(mint):
    """Graph construction creates exactly the variables of the name table (clair_amd/weights.py: tf_variable_names), in TensorFlow's
    creation order: LSTM1 fw / bw, LSTM2 fw / bw, the 256 L3 units, L4, L5_1..4, the four heads -- kernel before bias everywhere."""
    import fake_tf113 as tf
    from clair_amd import weights
    tf.reset_default_graph()
    mint.build_graph(tf)
    names = [v.name for v in tf.global_variables()]
    assert all(n.endswith(":0") for n in names)
    names = [n[:-2] for n in names]
    assert set(names) == set(weights.tf_variable_names()) == set(mint.tf_variable_names()) and len(names) == len(set(names)) == 8 + 512 + 2 + 8 + 8
    assert names[:4] == ["LSTM1/stack_bidirectional_rnn/cell_0/bidirectional_rnn/fw/cudnn_compatible_lstm_cell/kernel",
                         "LSTM1/stack_bidirectional_rnn/cell_0/bidirectional_rnn/fw/cudnn_compatible_lstm_cell/bias",
                         "LSTM1/stack_bidirectional_rnn/cell_0/bidirectional_rnn/bw/cudnn_compatible_lstm_cell/kernel",
                         "LSTM1/stack_bidirectional_rnn/cell_0/bidirectional_rnn/bw/cudnn_compatible_lstm_cell/bias"]
    assert names[8:10] == ["L3/Unit_0/kernel", "L3/Unit_0/bias"] and names[-2:] == ["Prediction/Y_indel_length_logits_2/kernel", "Prediction/Y_indel_length_logits_2/bias"]
    shapes = {v.name[:-2]: tuple(v.shape.as_list()) for v in tf.global_variables()}
    assert shapes["LSTM1/stack_bidirectional_rnn/cell_0/bidirectional_rnn/fw/cudnn_compatible_lstm_cell/kernel"] == (160, 512)
    assert shapes["LSTM2/stack_bidirectional_rnn/cell_0/bidirectional_rnn/bw/cudnn_compatible_lstm_cell/kernel"] == (384, 512)
    assert shapes["L3/Unit_255/kernel"] == (33, 30) and shapes["L4/kernel"] == (7680, 192) and shapes["L5_3/kernel"] == (192, 96)
    # a variable the table does not know stops the load with the full listing (the message a real run would print)
    tf.get_variable("stray", initializer=np.float32(1.0))
    with pytest.raises(SystemExit, match="does not know: stray"):
        with tf.Session() as sess:
            mint.load_variables(tf, sess, mint.recipe_weights())


def test_the_mini_checkpoint_is_what_the_reader_test_reads(mint, tmp_path, capsys):
    prefix = str(tmp_path / "tf113_mini")
    mint.mini_checkpoint(prefix)
    assert all(os.path.isfile(prefix + ext) for ext in (".index", ".data-00000-of-00001", ".json"))
    import test_weights
    test_weights.check_reader_against_minted_checkpoint(prefix)
    # --list-checkpoint on it: the 4-unit graph's names against the loader's table for that width
    from clair_amd import tf_bundle
    entries = tf_bundle.read_index(prefix + ".index")
    assert "global_step" in entries and entries["global_step"]["dtype"] == 9 and "Training_Operation/beta1_power" in entries
    mint.list_checkpoint(prefix)
    listing = capsys.readouterr().out
    assert "other  global_step" in listing and "model  L4/kernel" in listing and "variables;" in listing


def test_the_cudnn_checkpoint_recipe_runs_and_is_what_the_reader_test_reads(mint, tmp_path):
    """--cudnn-checkpoint (round 6): the reference's GPU branch -- one CudnnLSTM per BiLSTM layer, saved through the layer's saveable and as
    raw opaque buffers, with the layers' outputs -- executed under the stand-in, then read by the consumer the committed files will go
    through.  Circular by construction (stand-in and reader share the same recollection of cuDNN's parameter order, written twice); what it
    proves is that the recipe runs, names its files and variables as the reader expects, and that the consumer's three checks execute."""
    prefix = str(tmp_path / "tf113_cudnn")
    mint.cudnn_checkpoint(prefix, n=3)
    for ext in (".index", ".data-00000-of-00001", "_raw.index", "_raw.data-00000-of-00001", ".json"):
        assert os.path.isfile(prefix + ext), ext
    from clair_amd import tf_bundle
    assert sorted(tf_bundle.read_index(prefix + "_raw.index")) == ["LSTM1/cudnn_lstm/opaque_kernel", "LSTM2/cudnn_lstm/opaque_kernel"]
    names = sorted(tf_bundle.read_index(prefix + ".index"))
    assert len(names) == 8 and names[0] == "LSTM1/cudnn_lstm/stack_bidirectional_rnn/cell_0/bidirectional_rnn/bw/cudnn_compatible_lstm_cell/bias"
    import test_weights
    test_weights.check_reader_against_cudnn_checkpoint(prefix)
    # the loader proper: the same checkpoint's LSTM tensors through load_checkpoint's own name resolution (the dense layers are absent: refused by name)
    with pytest.raises(Exception, match="L3/Unit_0/kernel|missing|not found"):
        tf_bundle.load_checkpoint(prefix)
