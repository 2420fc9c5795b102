"""CPU tests of the oracle: the NumPy restatement, the C port, an independent torch-CPU
implementation and a float64 evaluation must agree; the committed golden fixture must reproduce.

The reference holds no golden vectors for this path ("parity unpinned"), so the
pin is: two independent float32 implementations + float64 agreeing to <= 2e-6, plus structural
properties of the reference graph (gate order, backward direction, L3 flat layout).
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

from clair_amd import synth, weights
from oracle import c_oracle, model_np

GOLDEN = os.path.join(ROOT, "tests", "golden", "nn_forward_64.npz")


@pytest.fixture(scope="module")
def golden():
    with np.load(GOLDEN) as z:
        g = {k: z[k] for k in z.files}
    w = weights.synthetic_weights(seed=int(g["seed_w"]), head_gain=float(g["head_gain"]),
                                  lstm_bias_scale=float(g["lstm_bias_scale"]))
    return g, w, synth.to_model_input(g["raw"].astype(np.int32))


def test_numpy_oracle_reproduces_golden(golden):
    g, w, x = golden
    outs, inter = model_np.forward(w, x, keep_intermediates=True)
    for got, key in zip(outs, ("gt21", "genotype", "len1", "len2")):
        assert np.abs(got - g[key]).max() <= 1e-6
    assert np.abs(inter["a1"][:, :4] - g["a1_first4"]).max() <= 1e-6
    assert np.abs(inter["a2"][:, :4] - g["a2_first4"]).max() <= 1e-6
    assert np.abs(inter["l3"][:4] - g["l3_first4"]).max() <= 2e-6
    assert np.abs(inter["l4"] - g["l4"]).max() <= 2e-6


def test_c_oracle_matches_golden_and_numpy(golden):
    g, w, x = golden
    outs, inter = c_oracle.forward(w, x, keep_intermediates=True)
    for got, key in zip(outs, ("gt21", "genotype", "len1", "len2")):
        assert got.dtype == np.float32
        assert np.abs(got - g[key]).max() <= 2e-6
        assert np.abs(got.sum(axis=1) - 1).max() < 1e-5
    # C intermediates are batch-major [n,33,256]
    assert np.abs(inter["a1"][:4].transpose(1, 0, 2) - g["a1_first4"]).max() <= 2e-6
    assert np.abs(inter["a2"][:4].transpose(1, 0, 2) - g["a2_first4"]).max() <= 2e-6
    assert np.abs(inter["l3"][:4].reshape(4, 30, 256) - g["l3_first4"]).max() <= 4e-6
    assert np.abs(inter["l4"] - g["l4"]).max() <= 4e-6


def test_torch_and_float64_agree(golden):
    import torch_ref
    g, w, x = golden
    outs_t, _ = torch_ref.forward(w, x)
    outs64 = model_np.forward(w, x, dtype=np.float64)
    for key, a, b in zip(("gt21", "genotype", "len1", "len2"), outs_t, outs64):
        assert np.abs(a - g[key]).max() <= 2e-6
        assert np.abs(b - g[key]).max() <= 2e-6


def test_c_oracle_thread_count_does_not_change_results():
    w = weights.synthetic_weights(seed=3)
    x, _ = synth.synthetic_input(37, "pacbio_ccs", seed=5)   # ragged last block (37 = 4*8 + 5)
    a = c_oracle.forward(w, x, threads=1)
    b = c_oracle.forward(w, x, threads=3)
    for p, q in zip(a, b):
        assert np.array_equal(p, q)


def test_empty_and_single_candidate():
    w = weights.synthetic_weights(seed=3)
    x, _ = synth.synthetic_input(1, "ont", seed=1)
    outs = c_oracle.forward(w, x)
    assert [o.shape for o in outs] == [(1, 21), (1, 3), (1, 33), (1, 33)]
    ref = model_np.forward(w, x)
    for p, q in zip(outs, ref):
        assert np.abs(p - q).max() <= 2e-6
    empty = c_oracle.forward(w, np.zeros((0, 33, 8, 4), np.float32))
    assert [o.shape for o in empty] == [(0, 21), (0, 3), (0, 33), (0, 33)]


def test_candidates_are_independent():
    """Each position is classified independently (docs/POST_PROCESSING.md:17): batch composition
    must not change a candidate's result -- this is what makes the path shard across GPUs."""
    w = weights.synthetic_weights(seed=8)
    x, _ = synth.synthetic_input(24, "ont", seed=2)
    full = c_oracle.forward(w, x)
    part = c_oracle.forward(w, x[5:14])
    for p, q in zip(full, part):
        assert np.array_equal(p[5:14], q)


def test_gate_order_and_backward_direction():
    """Structural checks of the LSTM restatement (clair/model.py:299-312):
    (1) the backward cell equals the forward cell run on the time-reversed input, reversed back;
    (2) column blocks are (i, c~, f, o): with only the c~ block driven the cell state moves, with
        only the f block driven from a zero state nothing moves."""
    rng = np.random.default_rng(0)
    inp = rng.standard_normal((33, 3, 32)).astype(np.float32)
    k = (rng.standard_normal((160, 512)) * 0.1).astype(np.float32)
    b = (rng.standard_normal(512) * 0.1).astype(np.float32)
    fw_on_reversed = model_np.lstm_direction(inp[::-1].copy(), k, b, False)[::-1]
    bw = model_np.lstm_direction(inp, k, b, True)
    assert np.array_equal(fw_on_reversed, bw)
    only_f = np.zeros_like(k)
    only_f[:, 256:384] = k[:, 256:384]
    assert np.abs(model_np.lstm_direction(inp, only_f, np.zeros(512, np.float32), False)).max() == 0.0
    only_g = np.zeros_like(k)
    only_g[:, 128:256] = k[:, 128:256]
    assert np.abs(model_np.lstm_direction(inp, only_g, np.zeros(512, np.float32), False)).max() > 0.0


def test_l3_flat_layout_is_u_times_256_plus_c():
    """clair/model.py:474-478: L3 [n,30,256] flattened row-major -> index u*256 + c.  With one-hot
    L4 weights the L4 pre-activation picks exactly that L3 element."""
    w = weights.synthetic_weights(seed=11)
    x, _ = synth.synthetic_input(2, "ont", seed=4)
    _, inter = model_np.forward(w, x, keep_intermediates=True)
    u, c = 7, 133
    w2 = dict(w)
    k4 = np.zeros_like(w["l4_kernel"])
    k4[u * 256 + c, 5] = 1.0
    w2["l4_kernel"] = k4
    _, inter2 = model_np.forward(w2, x, keep_intermediates=True)
    assert np.allclose(inter2["l4"][:, 5], model_np.selu(inter["l3"][:, u, c]), atol=1e-7)


@pytest.mark.parametrize("n,platform", [(1, "ont"), (47, "ont"), (48, "illumina"), (100, "pacbio_ccs")])
def test_blocked_cpu_port_matches_the_checker(n, platform):
    """oracle/clair_cpu_port.c (bench.py's timed CPU baseline: 48-candidate blocks, register-tiled GEMMs, polynomial exp) against
    the checker, ragged last block included; same tolerance as the HIP path."""
    from clair_amd import synth, weights
    from oracle import c_oracle
    w = weights.synthetic_weights(seed=11, head_gain=4.0, lstm_bias_scale=0.1)
    x, _ = synth.synthetic_input(n, platform, seed=50 + n)
    want = c_oracle.forward(w, x)
    got = c_oracle.port_forward(w, x, threads=2)
    for g, t in zip(got, want):
        assert g.shape == t.shape and np.isfinite(g).all()
        assert np.abs(g - t).max() <= 1e-5
        assert np.abs(g.sum(axis=1) - 1).max() < 1e-5


# ---- the TensorFlow-1.13 golden vectors (tools/mint_tf_golden.py) ---------------------------------------------------------------------
TF_GOLDEN = os.path.join(ROOT, "tests", "golden", "nn_tf113_64.npz")
TF_GOLDEN_ABSENT = ("tests/golden/nn_tf113_64.npz is NOT in the repository: the network arithmetic is still checked only against this "
                    "repository's own restatement of TensorFlow 1.13 (parity unpinned).  Anyone with tensorflow==1.13.2 can pin it: "
                    "`python tools/mint_tf_golden.py`, commit the file, and this test compares the oracle with the real reference.")


def _mint():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import mint_tf_golden
    return mint_tf_golden


def test_tf_golden_recipe_is_pinned():
    """The weights and candidates the TF-1.13 golden file is minted on are defined by an integer hash, not by a NumPy random
    stream: the same bits under the NumPy 1.16 a TF-1.13 environment has and under the NumPy 2 of this image.  Pinned here so
    that a change of the recipe cannot go unnoticed (the minted file records the same checksum)."""
    m = _mint()
    assert np.array_equal(m._uniform(1, 5), np.array([0.7666216, 0.13312304, 0.18237936, -0.7730994, -0.13708842], dtype=np.float32))
    w = m.recipe_weights()
    assert [k for k, _ in m.tensor_shapes()] == list(weights.TENSOR_TABLE) and all(w[k].shape == tuple(s) for k, s in weights.TENSOR_TABLE.items())
    assert abs(sum(float(np.abs(v.astype(np.float64)).sum()) for v in w.values()) - 117751.35826626392) < 1e-6
    assert m.tf_variable_names() == weights.tf_variable_names()          # the minting script loads TF's variables by the loader's names
    x = m.golden_input()
    assert x.shape == (64, 33, 8, 4) and x.dtype == np.float32
    outs = c_oracle.forward(w, x)
    assert abs(float(outs[0][0, 2]) - 0.27634575963020325) < 1e-5          # informative, peaky outputs (not a flat softmax)
    assert 0.5 < float(outs[0].max(axis=1).mean()) < 0.99
    # round 6, the TRAINED-LIKE variant: LSTM kernels x4, +1 on the forget rows, head gain 6, on the 300x Illumina counts of
    # tests/golden/nn_illumina300_64.npz (tools/make_nn_golden.py --illumina300), which also holds this oracle's outputs on it
    wt = m.recipe_weights(trained=True)
    assert np.array_equal(wt["lstm2_fw_kernel"], w["lstm2_fw_kernel"] * np.float32(4)) and np.array_equal(wt["l4_kernel"], w["l4_kernel"])
    assert np.array_equal(wt["lstm1_bw_bias"][256:384], w["lstm1_bw_bias"][256:384] + np.float32(1)) and np.array_equal(wt["lstm1_bw_bias"][:256], w["lstm1_bw_bias"][:256])
    assert abs(sum(float(np.abs(v.astype(np.float64)).sum()) for v in wt.values()) - 251055.0149521771) < 1e-6
    xi = m.golden_input("illumina300")
    assert xi.shape == (64, 33, 8, 4) and float(np.abs(xi).max()) > 150           # depths of the 300x profile
    with np.load(os.path.join(ROOT, "tests", "golden", "nn_illumina300_64.npz")) as z:
        assert str(z["recipe"]) == m.VARIANTS["trained"][3] and 1e-6 < float(z["f32_f64_distance"]) < 5e-5
        o32, o64 = c_oracle.forward(wt, xi), c_oracle.forward(wt, xi, dtype=np.float64)
        for a, b, key in zip(o32, o64, ("gt21", "genotype", "len1", "len2")):
            assert np.abs(a - z[key]).max() <= 1e-7 and np.abs(b - z[key + "_f64"]).max() <= 1e-12, key
        assert float(z["a2_absmax"]) > 0.99                                          # gates that saturate: the regime fresh init never reaches


@pytest.mark.parametrize("variant", ["fresh", "trained"])
def test_oracle_matches_the_tf113_golden_vectors_when_present(variant):
    """THE pin of the oracle: outputs and intermediates TensorFlow 1.13 itself computed for the recipe weights on 64
    golden candidates -- "fresh": fresh-init-like weights on ONT counts; "trained": trained-like weights (LSTM kernels x4, forget
    bias +1, head gain 6) on 300x Illumina counts.  Tolerances: 1e-5 on probabilities, 5e-6 on activations, each widened to four times
    the oracle's own float32 / float64 distance on that tensor (two float32 evaluations with different summation orders cannot
    agree better than either agrees with float64: 4e-6 on the fresh set, 8e-6 on the trained one)."""
    m = _mint()
    path = os.path.join(ROOT, "tests", "golden", m.VARIANTS[variant][2])
    if not os.path.isfile(path):
        pytest.skip(TF_GOLDEN_ABSENT.replace("nn_tf113_64.npz", m.VARIANTS[variant][2]))
    check_oracle_against_minted_file(path)


def check_oracle_against_minted_file(path):
    """What the test above does with the committed file; tests/test_mint_tool.py runs it on files minted under the stand-in TensorFlow."""
    m = _mint()
    with np.load(path) as z:
        variant = str(z["variant"]) if "variant" in z.files else "fresh"
        trained, profile, _, recipe = m.VARIANTS[variant]
        w, x = m.recipe_weights(trained=trained), m.golden_input(profile)
        assert str(z["recipe"]) == recipe
        assert abs(float(z["weights_checksum"]) - sum(float(np.abs(v.astype(np.float64)).sum()) for v in w.values())) < 1e-6
        outs, inter = c_oracle.forward(w, x, keep_intermediates=True)
        outs64, inter64 = c_oracle.forward(w, x, keep_intermediates=True, dtype=np.float64)

        def tol(base, a32, a64):
            return max(base, 4.0 * float(np.abs(a32 - a64).max()))
        for got, g64, key in zip(outs, outs64, ("gt21", "genotype", "len1", "len2")):
            assert np.abs(got - z[key]).max() <= tol(1e-5, got, g64), key
            assert np.abs(g64 - z[key]).max() <= tol(1e-5, got, g64), key            # and TensorFlow is as close to float64 as the oracle is
        for key, mine, mine64, theirs in (("a1", inter["a1"][:4].transpose(1, 0, 2), inter64["a1"][:4].transpose(1, 0, 2), z["a1_first4"]),
                                          ("a2", inter["a2"][:4].transpose(1, 0, 2), inter64["a2"][:4].transpose(1, 0, 2), z["a2_first4"]),
                                          ("l3", inter["l3"][:4], inter64["l3"][:4], z["l3_first4"]), ("l4", inter["l4"], inter64["l4"], z["l4"])):
            assert np.abs(mine - theirs).max() <= tol(5e-6, mine, mine64), key
