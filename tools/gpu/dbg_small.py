import faulthandler, os, sys
faulthandler.enable()
os.environ["CLAIR_AMD_TAP_L3"] = "1"
import numpy as np
sys.path.insert(0, ".")
from clair_amd import _capi, synth, weights
from oracle import c_oracle
w = weights.synthetic_weights(seed=20250928, head_gain=4.0)
for n in (64, 1024):
    x, _ = synth.synthetic_input(n, "ont", seed=1000 + n)
    eng = _capi.Engine(device=0, max_batch=1024, n_slots=2)
    eng.load_weights(w)
    got = eng.predict(x)
    want, inter = c_oracle.forward(w, x, keep_intermediates=True)
    n_pad = (n + 31) // 32 * 32
    a2 = eng.debug_read(0, 2, (33, n_pad, 256)).transpose(1, 0, 2)[:n]
    l3 = eng.debug_read(0, 4, (n_pad, 7680))[:n].reshape(n, 30, 256)
    part = eng.debug_read(0, 3, (8, n_pad, 192))[:, :n]
    o3 = inter["l3"].reshape(n, 30, 256)
    bad3 = ~np.isfinite(l3)
    print(n, "probs", [float(np.abs(g - t).max()) for g, t in zip(got, want)], "a2 %.2e" % np.abs(a2 - inter["a2"]).max(), flush=True)
    print("   l3: nan count %d of %d; finite-part err %.2e; nan by channel%%8 %s; by u %s; by cand%%64 (first 16) %s" % (
        bad3.sum(), bad3.size, np.abs(np.where(bad3, 0, l3 - o3)).max(), bad3.sum(axis=(0, 1)).reshape(32, 8).sum(axis=0).tolist(),
        bad3.sum(axis=(0, 2)).tolist(), bad3.sum(axis=(1, 2))[:16].tolist()))
    badp = ~np.isfinite(part)
    print("   partials: nan count %d of %d; by group %s" % (badp.sum(), badp.size, badp.sum(axis=(1, 2)).tolist()))
    eng.close()
