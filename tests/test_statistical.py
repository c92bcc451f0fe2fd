"""Rung C of the oracle ladder: the parallel-schedule chain (rung B: 3-colour, Philox,
deterministic exp/log) against the reference-schedule chain (rung A) and against the
reference's closed forms -- statistically, since the two chains use different random
streams and visiting orders."""
import ctypes as C
import os

import numpy as np
import pytest

import orc
from common import GOLDEN, simulate


def test_end_conditioned_sampler_means_match_reference_closed_forms():
    """expectation_J / expectation_D of the reference (ContinuousTimeMarkovModel.cpp:168-226,
    values stored in golden/kat.npz) are the exact means of ANY correct end-conditioned
    sampler; check forward rejection (the reference's hot path, rung A) and the Nielsen
    sampler the parallel rung and the GPU use for state changes (EndCondSampling.cpp:583-617),
    the latter also where forward rejection needs ~1/(r T) trials per sample."""
    g = np.load(os.path.join(GOLDEN, "kat.npz"))
    L = orc.orc_lib()
    n = 40000
    for (r0, r1, T), e in zip(g["exp_grid"], g["exp_JD"]):
        J0, J1, D0 = e[0:4].reshape(2, 2), e[4:8].reshape(2, 2), e[8:12].reshape(2, 2)
        for a in (0, 1):
            for b in (0, 1):
                for rng, math in ((orc.RNG_PHILOX, orc.MATH_EPV), (orc.RNG_MT, orc.MATH_LIBM),
                                  (orc.RNG_MT | 0x200, orc.MATH_LIBM)):
                    if rng == orc.RNG_MT and a != b and min(r0, r1) * T < 0.02:
                        continue      # forward rejection: ~1/(r T) trials per sample, keep the suite quick
                    out = np.zeros(3)
                    L.orc_kat_end_cond_means(rng, math, 5, r0, r1, a, b, T, n, orc._p(out, C.c_double))
                    exp = np.array([J0[a, b], J1[a, b], D0[a, b]])
                    # standard errors: jumps ~ Poisson-ish, dwell bounded by T
                    tol = 5.0 * np.array([np.sqrt(max(exp[0], 0.05) / n) + 1e-3,
                                          np.sqrt(max(exp[1], 0.05) / n) + 1e-3, T / np.sqrt(n)])
                    assert np.all(np.abs(out - exp) < tol), (r0, r1, T, a, b, out, exp)


@pytest.mark.parametrize("cfg,n", [("tree", 30000), ("pair", 12000)])
def test_rung_b_chain_matches_rung_a_chain_statistically(cfg, n):
    model, tree, fp = simulate(cfg, n, seed=13)
    res = {}
    for rung in ("A", "B"):
        o = orc.Oracle(tree, model, fp, rung, cap=32 if rung == "B" else 0, seed=1)
        o.reset()
        J, D, nacc, acc = o.run_mcmc(4, 12)
        res[rung] = (J, D, acc)
    JA, DA, accA = res["A"]
    JB, DB, accB = res["B"]
    assert abs(accA - accB) < 0.01
    # J per (branch, context): Poisson counts averaged over 12 correlated sweeps
    sd = np.sqrt(np.maximum(JA, 1.0))
    assert np.all(np.abs(JA - JB) < 6.0 * sd + 2.0)
    # total dwell time is conserved exactly; its split over contexts fluctuates
    B = tree.n_nodes - 1
    np.testing.assert_allclose(DA.reshape(B, 8).sum(1), DB.reshape(B, 8).sum(1), rtol=1e-9)
    assert np.all(np.abs(DA - DB) < 0.05 * DA.reshape(B, 8).sum(1, keepdims=True).repeat(8, 1).reshape(-1) + 1.0)


def test_parallel_rung_is_insensitive_to_the_exp_log_pair():
    """the deterministic exp/log differ from glibc by < 1 ulp: no accept/reject or state
    decision flips, and jump times agree to ~1e-13"""
    model, tree, fp = simulate("tree", 20000, seed=5)
    outs = []
    for math in (orc.MATH_EPV, orc.MATH_LIBM):
        o = orc.Oracle(tree, model, fp, (orc.RNG_PHILOX, math, orc.SCHED_3COLOUR, orc.REDUCE_EXACT), cap=16, seed=3)
        o.reset()
        nacc = sum(o.sweep(w) for w in range(3))
        outs.append((nacc, o.paths()))
    assert outs[0][0] == outs[1][0]
    a, b = outs[0][1], outs[1][1]
    assert np.array_equal(a.init, b.init) and np.array_equal(a.offsets, b.offsets)
    np.testing.assert_allclose(a.jumps, b.jumps, rtol=1e-12, atol=1e-15)


def test_three_colour_schedule_is_a_valid_parallel_update():
    """within one colour phase the updates commute: visiting the sites of a colour in
    reverse order gives the identical state (they read and write disjoint data)"""
    model, tree, fp = simulate("pair", 3000, seed=2)
    o1 = orc.Oracle(tree, model, fp, "B", cap=32, seed=8)
    o2 = orc.Oracle(tree, model, fp, "B", cap=32, seed=8)
    o1.reset()
    o2.reset()
    o1.sweep(0)
    n = fp.n_sites
    for c in range(3):
        for s in range(n - 2, 0, -1):
            if s % 3 == c:
                o2.mh_site(s, 0)
    assert orc.paths_equal(o1.paths(), o2.paths())
    assert np.array_equal(o1.tri_llh(), o2.tri_llh())


@pytest.mark.gpu
@pytest.mark.parametrize("cfg,n", [("tree", 40000), ("pair", 12000), ("cat6", 10000)])
def test_gpu_chain_matches_rung_a_chain_statistically(cfg, n):
    """the same comparison with the GPU itself as the parallel chain: rung A is the reference's
    chain (bit-identical to the linked library, tests/test_oracle_golden.py)"""
    from epievo_amd.sampler import DeviceSampler
    model, tree, fp = simulate(cfg, n, seed=13)
    a = orc.Oracle(tree, model, fp, "A", seed=1)
    a.reset()
    JA, DA, _, accA = a.run_mcmc(4, 12)
    d = DeviceSampler(0)
    d.set_tree(tree)
    d.set_model(model)
    d.upload_paths(fp, 32)
    d.reset()
    JB, DB, nacc = d.run_mcmc(4, 12, 1)
    accB = nacc / float(12 * (n - 2))
    assert abs(accA - accB) < 0.01
    sd = np.sqrt(np.maximum(JA, 1.0))
    assert np.all(np.abs(JA - JB) < 6.0 * sd + 2.0)
    B = tree.n_nodes - 1
    np.testing.assert_allclose(DA.reshape(B, 8).sum(1), DB.reshape(B, 8).sum(1), rtol=1e-9)
    assert np.all(np.abs(DA - DB) < 0.05 * DA.reshape(B, 8).sum(1, keepdims=True).repeat(8, 1).reshape(-1) + 1.0)
