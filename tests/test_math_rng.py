"""The deterministic exp/log and the Philox stream that the oracle's parallel rung and
the gfx950 kernels share as a contract."""
import ctypes as C

import numpy as np

import orc


def _ulp_err(got, x, fn):
    import mpmath
    mpmath.mp.prec = 200
    worst = 0.0
    for g, xi in zip(got, x):
        exact = fn(mpmath.mpf(float(xi)))
        if exact == 0:
            continue
        ulp = mpmath.mpf(2) ** (mpmath.floor(mpmath.log(abs(exact), 2)) - 52)
        worst = max(worst, float(abs(mpmath.mpf(float(g)) - exact) / ulp))
    return worst


def test_exp_log_accuracy_vs_mpmath():
    import mpmath
    rng = np.random.RandomState(0)
    L = orc.orc_lib()
    xs = np.concatenate([rng.uniform(-40, 0, 1500), rng.uniform(-1e-3, 1e-3, 300),
                         -np.exp(rng.uniform(-30, 6, 700)), rng.uniform(0, 30, 300)])
    e, l = np.zeros_like(xs), np.zeros_like(xs)
    L.orc_kat_exp_log_array(orc._p(xs, C.c_double), len(xs), orc._p(e, C.c_double), orc._p(l, C.c_double))
    assert _ulp_err(e, xs, mpmath.exp) < 1.0
    ys = np.concatenate([rng.uniform(0, 1, 1500), 1.0 - np.exp(rng.uniform(-36, -1, 500)),
                         np.exp(rng.uniform(-700, 700, 500)), rng.uniform(0.5, 2.0, 500)])
    e, l = np.zeros_like(ys), np.zeros_like(ys)
    L.orc_kat_exp_log_array(orc._p(ys, C.c_double), len(ys), orc._p(e, C.c_double), orc._p(l, C.c_double))
    assert _ulp_err(l, ys, mpmath.log) < 1.0


def test_exp_log_special_values():
    L = orc.orc_lib()
    assert L.orc_kat_exp(0.0) == 1.0 and L.orc_kat_log(1.0) == 0.0
    assert L.orc_kat_exp(-1e4) == 0.0 and L.orc_kat_exp(1e4) == np.inf
    assert L.orc_kat_log(0.0) == -np.inf and np.isnan(L.orc_kat_log(-1.0))
    assert L.orc_kat_log(np.inf) == np.inf and np.isnan(L.orc_kat_exp(np.nan))
    assert 0 < L.orc_kat_exp(-740.0) < 1e-300                      # subnormal range
    assert abs(L.orc_kat_log(5e-324) - np.log(5e-324)) < 1e-12 * 745
    # monotone across the 2^k seams
    xs = np.linspace(-3, 3, 20001)
    v = np.array([L.orc_kat_exp(float(x)) for x in xs])
    assert np.all(np.diff(v) >= 0)


def test_philox4x32_10_known_answers():
    """Random123 known-answer vectors for philox4x32-10 (Salmon et al., SC'11)"""
    L = orc.orc_lib()
    kats = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
            ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
            ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
             (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, exp in kats:
        c, k, o = np.array(ctr, np.uint32), np.array(key, np.uint32), np.zeros(4, np.uint32)
        L.orc_kat_philox(orc._p(c, C.c_uint32), orc._p(k, C.c_uint32), orc._p(o, C.c_uint32))
        assert tuple(int(x) for x in o) == exp


def test_keyed_block_layout():
    L = orc.orc_lib()
    d = np.zeros(2)
    seen = set()
    for args in [(7, 1, 2, 3, 4, 5, 6), (7, 1, 2, 3, 4, 5, 7), (7, 1, 2, 3, 4, 6, 6), (7, 1, 2, 3, 5, 5, 6),
                 (7, 1, 2, 4, 4, 5, 6), (7, 1, 3, 3, 4, 5, 6), (7, 2, 2, 3, 4, 5, 6), (8, 1, 2, 3, 4, 5, 6),
                 ((1 << 32) + 7, 1, 2, 3, 4, 5, 6)]:
        L.orc_kat_keyed_block(*args, orc._p(d, C.c_double))
        assert 0.0 <= d[0] < 1.0 and 0.0 <= d[1] < 1.0
        seen.add((d[0], d[1]))
    assert len(seen) == 9        # every address field (incl. the seed's high word) matters
    # uniformity of the 53-bit doubles
    u = []
    for t in range(4000):
        L.orc_kat_keyed_block(99, 5, 0, 1, 0, t, 0, orc._p(d, C.c_double))
        u += [d[0], d[1]]
    u = np.array(u)
    assert abs(u.mean() - 0.5) < 0.01 and abs(u.var() - 1 / 12.0) < 0.005
