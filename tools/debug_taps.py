"""Localise a parity problem: LSTM1 / LSTM2 outputs (debug taps) against the oracle's intermediates, per step and direction."""
import os
import sys
os.environ['CLAIR_AMD_TAP_L3'] = '1'
import numpy as np
sys.path.insert(0, ".")
from clair_amd import _capi, synth, weights
from oracle import c_oracle

w = weights.synthetic_weights(seed=7)
for n, platform in ((40, "ont"), (100, "illumina"), (1024, "ont")):
    eng = _capi.Engine(device=0, max_batch=1024, n_slots=1)
    eng.load_weights(w)
    x, _ = synth.synthetic_input(n, platform, seed=1000 + n)
    got = eng.predict(x)
    want, inter = c_oracle.forward(w, x, keep_intermediates=True)
    n_pad = (n + 31) // 32 * 32
    a1 = eng.debug_read(0, 1, (33, n_pad, 256)).transpose(1, 0, 2)[:n]
    a2 = eng.debug_read(0, 2, (33, n_pad, 256)).transpose(1, 0, 2)[:n]
    e1, e2 = np.abs(a1 - inter["a1"]), np.abs(a2 - inter["a2"])
    print("n=%d %s: probs %.3g | a1 %.3g (fw %.3g bw %.3g) | a2 %.3g (fw %.3g bw %.3g)" % (
        n, platform, max(np.abs(g - w_).max() for g, w_ in zip(got, want)),
        e1.max(), e1[:, :, :128].max(), e1[:, :, 128:].max(), e2.max(), e2[:, :, :128].max(), e2[:, :, 128:].max()))
    print("   a1 err by t:", " ".join("%.1e" % v for v in e1.max(axis=(0, 2))))
    print("   a1 err by unit%32 block:", " ".join("%.1e" % e1[:, :, u::32].max() for u in range(0, 32, 4)))
    print("   a1 err by cand%32:", " ".join("%.1e" % e1[c::32].max() for c in range(0, 32, 4)))
    eng.close()

# the bench's own batch: which candidates carry the largest probability error, and do their LSTM outputs differ?
w = weights.synthetic_weights(seed=20250928, head_gain=4.0)
eng = _capi.Engine(device=0, max_batch=1024, n_slots=1)
eng.load_weights(w)
x, _ = synth.synthetic_input(8 * 1024, "ont", seed=20250928)
x = x[:1024]
for rep in range(3):
    got = eng.predict(x)
    want, inter = c_oracle.forward(w, x, keep_intermediates=True)
    a1 = eng.debug_read(0, 1, (33, 1024, 256)).transpose(1, 0, 2)
    a2 = eng.debug_read(0, 2, (33, 1024, 256)).transpose(1, 0, 2)
    part = eng.debug_read(0, 3, (8, 1024, 192)).astype(np.float64).sum(axis=0) + w["l4_bias"].astype(np.float64)
    selu = lambda v: 1.0507009873554804934193349852946 * np.where(v >= 0, v, 1.6732632423543772848170429916717 * np.expm1(v))
    e4 = np.abs(selu(part) - inter["l4"])
    print("   l4 (selu of summed partials) err vs oracle: %.3g; mean %.3g; by cand%%32:" % (e4.max(), e4.mean()), " ".join("%.1e" % e4[c::32].max() for c in range(32)))
    print("   l4 err by unit block of 16:", " ".join("%.1e" % e4[:, u:u + 16].max() for u in range(0, 192, 16)))
    # per channel-group partial against a float64 evaluation of the same slice from the oracle's l3
    l3 = inter["l3"].astype(np.float64).reshape(1024, 30, 256)
    W4 = w["l4_kernel"].astype(np.float64).reshape(30, 256, 192)
    raw = eng.debug_read(0, 3, (8, 1024, 192))
    print("   partial err by split (32 channels):", " ".join("%.1e" % np.abs(raw[g] - np.einsum("nuc,ucj->nj", l3[:, :, g * 32:(g + 1) * 32], W4[:, g * 32:(g + 1) * 32])).max() for g in range(8)))
    l3g = eng.debug_read(0, 4, (1024, 7680))
    e3 = np.abs(l3g - inter["l3"])
    bad = np.argwhere(e3 > 2e-5)
    print("   l3 err max %.3g mean %.3g; elements with err > 2e-5: %d" % (e3.max(), e3.mean(), len(bad)))
    for n_, k_ in bad[:12]:
        print("      cand %d (%%32=%d) u %d c %d (cg %d, w %d, cc %d): gpu %.8f oracle %.8f diff %.3g" % (n_, n_ % 32, k_ // 256, k_ % 256, (k_ % 256) // 16, ((k_ % 256) % 16) // 4, k_ % 4, l3g[n_, k_], inter["l3"][n_, k_], l3g[n_, k_] - inter["l3"][n_, k_]))
    perr = np.max([np.abs(g - w_).max(axis=1) for g, w_ in zip(got, want)], axis=0)
    worst = np.argsort(perr)[-5:][::-1]
    print("rep", rep, "worst candidates", worst.tolist(), "prob err", ["%.2e" % perr[i] for i in worst],
          "a1 err", ["%.2e" % np.abs(a1[i] - inter["a1"][i]).max() for i in worst],
          "a2 err", ["%.2e" % np.abs(a2[i] - inter["a2"][i]).max() for i in worst], "overall a1 %.2e a2 %.2e" % (np.abs(a1 - inter["a1"]).max(), np.abs(a2 - inter["a2"]).max()))
