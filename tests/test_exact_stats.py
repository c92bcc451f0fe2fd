"""The parallel rung's sufficient statistics are exact integers (J counts, fixed-point dwell times):
oracle-level checks on the CPU -- they equal the reference-order fp64 sums to rounding, they do not
depend on how the genome is cut, and the scale rule keeps every sum inside int64."""
import ctypes as C

import numpy as np
import pytest

import orc
from common import simulate


def _rows(o, first, row_sites, n_rows, lo, hi):
    out = np.zeros((n_rows, o.B, 16), np.int64)
    o.L.orc_suffstats_rows(o.h, first, row_sites, n_rows, lo, hi, orc._p(out, C.c_int64))
    return out


def _scales(o):
    s = np.zeros(o.B + 1)
    o.L.orc_stat_scales(o.h, orc._p(s, C.c_double))
    return s


@pytest.mark.parametrize("cfg,n", [("tree", 5000), ("pair", 3000), ("bal16", 700)])
def test_exact_statistics_match_sequential_sums(cfg, n):
    model, tree, fp = simulate(cfg, n, seed=4)
    a = orc.Oracle(tree, model, fp, "A")
    b = orc.Oracle(tree, model, fp, "B", cap=64)
    Ja, Da = a.suffstats()          # the reference's order: sequential fp64 (rung A == linked reference)
    Jb, Db = b.suffstats()          # integers, then one conversion
    assert np.array_equal(Ja, Jb)
    np.testing.assert_allclose(Db, Da, rtol=1e-12, atol=0)
    # total time is conserved exactly up to the quantum: sum_ctx D = (n - 2) T per branch
    sc = _scales(b)
    for node in range(1, tree.n_nodes):
        tot = Db[(node - 1) * 8:node * 8].sum()
        assert abs(tot - (n - 2) * tree.branches[node]) <= (n - 2) / sc[node] + 1e-12 * tot


def test_integer_rows_add_up_whatever_the_cut():
    model, tree, fp = simulate("tree", 3000, seed=9)
    o = orc.Oracle(tree, model, fp, "B", cap=32)
    whole = _rows(o, 0, 4096, 1, 0, 2 ** 62)[0]
    for row_sites in (1, 7, 256, 1000):
        n_rows = -(-3000 // row_sites)
        parts = _rows(o, 0, row_sites, n_rows, 0, 2 ** 62)
        assert np.array_equal(parts.sum(axis=0), whole)
    # owned ranges partition the sites: their sums add up too
    a = _rows(o, 0, 4096, 1, 0, 1234)[0]
    b = _rows(o, 0, 4096, 1, 1235, 2 ** 62)[0]
    assert np.array_equal(a + b, whole)
    # and the doubles the oracle reports are those integers, scaled by a power of two
    J, D = o.suffstats()
    sc = _scales(o)
    assert np.array_equal(J.reshape(-1, 8), whole[:, :8].astype(np.float64))
    assert np.array_equal(D.reshape(-1, 8), whole[:, 8:].astype(np.float64) / sc[1:, None])


def test_scale_rule_bounds():
    """a genome's sum stays below 2^61, a single term below 2^50, for any length and branch"""
    model, tree, fp = simulate("pair", 64, seed=1)
    o = orc.Oracle(tree, model, fp, "B", cap=32)
    for n_global in (64, 10 ** 6, 10 ** 7, 2 ** 32 - 1):
        o.L.orc_set_shard(o.h, 0, n_global)
        k = np.log2(_scales(o)[1])
        T = tree.branches[1]
        assert k == int(k)
        assert T * 2.0 ** k < 2.0 ** 50 and n_global * T * 2.0 ** k < 2.0 ** 61
        assert T * 2.0 ** (k + 1) >= 2.0 ** 50 or n_global * T * 2.0 ** (k + 1) >= 2.0 ** 61   # as fine as allowed
