"""The recurrent kernels' gate arithmetic v2 (clair_amd/csrc/lstm32.hip.h: CLAIR_GATE_GAP), restated operation for operation in NumPy
float32 and checked against a float64 evaluation of the reference's cell (clair/model.py:299-312 -> LSTMBlockCell: c = sig(f) c +
sig(i) tanh(g), h = sig(o) tanh(c)): 8 transcendentals per hidden unit and step instead of 10 --

    K2 sig(i) tanh(g) = K2 (eg - 1) / ((1 + ei)(1 + eg))        h = sig(o) tanh(c) = (ec - 1) / ((1 + eo)(1 + ec))

with eg clamped at 2^63 and every other exponential allowed to overflow.  What this pins on the CPU: no NaN for any finite or infinite
pre-activation, errors no larger than the ten-transcendental form's (LABNOTES.md A1).  The kernels themselves are checked against the
oracle on the GPU (tests/test_parity_gpu.py); v_exp_f32 / v_rcp_f32 are ~1 ulp, NumPy's exp2 and division stand in for them here."""
import numpy as np

f = np.float32
K2 = f(2 * 1.4426950408889634)
L2E = f(1.4426950408889634)
BIG = f(2.0 ** 63)


def _exps(i, g, fg, o):          # the MFMA hands over pre-scaled arguments: 2^z = e^-i, e^2g, e^-f, e^-o
    with np.errstate(over="ignore"):
        return (np.exp2((-L2E * i).astype(f)).astype(f), np.exp2((K2 * g).astype(f)).astype(f),
                np.exp2((-L2E * fg).astype(f)).astype(f), np.exp2((-L2E * o).astype(f)).astype(f))


def gates_v2(i, g, fg, o, c):    # c is c' = K2 c; returns (c', h)
    ei, eg, ef, eo = _exps(i, g, fg, o)
    with np.errstate(over="ignore", invalid="ignore", divide="ignore"):
        eg = np.minimum(eg, BIG)                                   # MN
        ng = (eg * K2 - K2).astype(f)                              # NG (one fma in the kernel)
        p = ((eg + f(1)) * (ei + f(1))).astype(f)                  # A, A, P
        tt = (ng * (f(1) / p).astype(f)).astype(f)                 # R, T
        rf = (f(1) / (ef + f(1))).astype(f)                        # A, R
        c2 = (rf * c + tt).astype(f)                               # C
        ec = np.exp2(c2).astype(f)                                 # X
        q = ((ec + f(1)) * (eo + f(1))).astype(f)                  # AC, A, Q
        h = ((ec - f(1)) * (f(1) / q).astype(f)).astype(f)         # NC, R, H
    return c2, h


def gates_v1(i, g, fg, o, c):    # rounds 1-4: four reciprocals, k = K2 - 2 K2 rg, h = ro - 2 ro rc
    ei, eg, ef, eo = _exps(i, g, fg, o)
    with np.errstate(over="ignore", invalid="ignore", divide="ignore"):
        rg, ri, rf, ro = ((f(1) / (e + f(1))).astype(f) for e in (eg, ei, ef, eo))
        c2 = (rf * c + (ri * (rg * f(-2) * K2 + K2).astype(f)).astype(f)).astype(f)
        rc = (f(1) / (np.exp2(c2).astype(f) + f(1))).astype(f)
        h = (rc * (f(-2) * ro) + ro).astype(f)
    return c2, h


def gates_f64(i, g, fg, o, c):
    i, g, fg, o, c = (x.astype(np.float64) for x in (i, g, fg, o, c))
    with np.errstate(over="ignore"):
        sig = lambda x: 1.0 / (1.0 + np.exp(-x))      # noqa: E731
        cc = sig(fg) * (c / float(K2)) + sig(i) * np.tanh(g)
        return cc * float(K2), sig(o) * np.tanh(cc)


def test_v2_is_as_accurate_as_v1_on_random_gates():
    rng = np.random.default_rng(1)
    n = 400000
    for scale in (1.0, 5.0, 30.0, 100.0, 1e4):
        i, g, fg, o = ((rng.standard_normal(n) * scale).astype(f) for _ in range(4))
        c = (rng.standard_normal(n) * 2).astype(f) if scale == 1.0 else (rng.uniform(-33, 33, n) * float(K2)).astype(f)
        (c2, h2), (c1, h1), (cr, hr) = gates_v2(i, g, fg, o, c), gates_v1(i, g, fg, o, c), gates_f64(i, g, fg, o, c)
        assert not np.isnan(c2).any() and not np.isnan(h2).any(), scale
        assert np.abs(h2 - hr).max() <= max(5e-7, 1.05 * np.abs(h1 - hr).max()), (scale, np.abs(h2 - hr).max(), np.abs(h1 - hr).max())
        assert np.abs(c2 - cr).max() <= 1.05 * np.abs(c1 - cr).max() + 1e-7, scale          # float32 rounding of c' itself (|c'| up to 95)


def test_v2_has_no_nan_and_the_right_limits_at_the_extremes():
    ext = np.array([-np.inf, -1e30, -1e5, -200, -90, -45, -20, -1e-30, 0, 1e-30, 20, 45, 90, 200, 1e5, 1e30, np.inf], dtype=f)
    i, g, fg, o = (a.ravel() for a in np.meshgrid(ext, ext, ext, ext, indexing="ij"))
    for cval in (0.0, -95.0, 95.0):
        c = np.full(i.shape, cval, dtype=f)
        c2, h2 = gates_v2(i, g, fg, o, c)
        cr, hr = gates_f64(i, g, fg, o, c)
        assert not np.isnan(c2).any() and not np.isnan(h2).any(), cval
        assert np.abs(c2 - cr).max() < 3e-6 and np.abs(h2 - hr).max() < 2e-7, (cval, np.abs(c2 - cr).max(), np.abs(h2 - hr).max())
        assert np.abs(h2).max() <= 1.0 and np.abs(c2).max() <= abs(cval) + float(K2) * (1 + 1e-6)
    # the clamp: without it eg = inf gives (inf - 1) / inf
    with np.errstate(invalid="ignore", over="ignore"):
        assert np.isnan((np.float32(np.inf) - 1) / ((1 + np.float32(1.0)) * (1 + np.float32(np.inf))))
    c2, _ = gates_v2(np.array([0], f), np.array([1e30], f), np.array([np.inf], f), np.array([0], f), np.array([0], f))
    assert abs(float(c2[0]) - float(K2) * 0.5) < 1e-6          # sig(0) tanh(+huge) = 0.5, forget gate wide open on a zero state
