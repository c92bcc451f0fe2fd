"""Pins the oracle: the CPU restatement in reference-schedule mode (rung A: sequential
sweep, mt19937 + libstdc++ distribution semantics, glibc exp/log, sequential J/D sums)
against the golden vectors in tests/golden/, which tests/golden/make_golden.py captured
from the UNMODIFIED reference library.  Everything is compared bit-for-bit."""
import glob
import os

import numpy as np
import pytest

import orc
from common import GOLDEN
from epievo_amd import host

CASES = sorted(glob.glob(os.path.join(GOLDEN, "*_n*_s*.npz")))


def _load(path):
    g = np.load(path)
    tree = host.Tree(g["subtree"], g["parent"], g["branches"])
    model = host.Model(g["rates"], g["T"], np.zeros(4))
    fp = host.FlatPaths(int(g["n_sites"]), tree.n_nodes, g["init"], g["offsets"], g["jumps"])
    return g, tree, model, fp


def test_fixture_set_is_complete():
    assert len(CASES) == 15


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(c)[:-4] for c in CASES])
def test_rung_a_matches_reference(path):
    g, tree, model, fp = _load(path)
    o = orc.Oracle(tree, model, fp, "A", seed=int(g["seed"]))
    o.reset()
    assert np.array_equal(o.tri_llh(), g["tri_llh"])
    done = 0
    for k in (1, 3):
        nacc = sum(o.sweep(0) for _ in range(k - done))
        done = k
        assert nacc == int(g["nacc_%d" % k])
        p = o.paths()
        assert np.array_equal(p.init, g["init_%d" % k])
        assert np.array_equal(p.offsets, g["offsets_%d" % k])
        assert np.array_equal(p.jumps, g["jumps_%d" % k])
        assert np.array_equal(o.tri_llh(), g["tri_%d" % k])
    J, D, nacc, acc = o.run_mcmc(1, 2)
    assert np.array_equal(J, g["J"]) and np.array_equal(D, g["D"])
    assert acc == float(g["acc"])
    p = o.paths()
    assert np.array_equal(p.init, g["init_f"])
    assert np.array_equal(p.offsets, g["offsets_f"])
    assert np.array_equal(p.jumps, g["jumps_f"])
    Js, Ds = o.suffstats()
    assert np.array_equal(Js, g["J_stat"]) and np.array_equal(Ds, g["D_stat"])
    o.scale_jump_times(tree.branches * 1.25)
    assert np.array_equal(o.paths().jumps, g["scaled_jumps"])


def test_kat_ctmc_segments_suffstats_rng():
    import ctypes as C
    g = np.load(os.path.join(GOLDEN, "kat.npz"))
    L = orc.orc_lib()
    dbl, u64 = C.c_double, C.c_uint64
    for (r0, r1, t), P, G in zip(g["ctmc_grid"], g["ctmc_P"], g["ctmc_G"]):
        p = np.zeros(4)
        L.orc_kat_trans_prob_mat(orc.MATH_LIBM, r0, r1, t, orc._p(p, dbl))
        assert np.array_equal(p, P)
        got = [L.orc_kat_get_trans_prob(orc.MATH_LIBM, r0, r1, t, a, b) for a in (0, 1) for b in (0, 1)]
        assert np.array_equal(np.array(got), G)
    rates = g["model_scaled_rates"]
    for i in range(int(g["n_seg_cases"])):
        lj, rj = g["seg%d_lj" % i], g["seg%d_rj" % i]
        K = len(lj) + len(rj) + 1
        r0, r1, ln = np.zeros(K), np.zeros(K), np.zeros(K)
        t0, t1 = np.zeros(K, np.uint64), np.zeros(K, np.uint64)
        lje, rje = np.concatenate([lj, [0.0]]), np.concatenate([rj, [0.0]])
        k = L.orc_kat_segments(orc._p(rates, dbl), int(g["seg%d_li" % i]), len(lj), orc._p(lje, dbl),
                               int(g["seg%d_ri" % i]), len(rj), orc._p(rje, dbl), 1.0,
                               orc._p(r0, dbl), orc._p(r1, dbl), orc._p(t0, u64), orc._p(t1, u64),
                               orc._p(ln, dbl))
        assert k == K
        for name, arr in (("r0", r0), ("r1", r1), ("t0", t0), ("t1", t1), ("len", ln)):
            assert np.array_equal(arr, g["seg%d_%s" % (i, name)])
        mj = g["st%d_mj" % i]
        lj, rj = g["st%d_lj" % i], g["st%d_rj" % i]
        lje, mje, rje = (np.concatenate([x, [0.0]]) for x in (lj, mj, rj))
        J, D = np.zeros(8), np.zeros(8)
        L.orc_kat_suffstats(int(g["st%d_li" % i]), len(lj), orc._p(lje, dbl), int(g["st%d_mi" % i]),
                            len(mj), orc._p(mje, dbl), int(g["st%d_ri" % i]), len(rj),
                            orc._p(rje, dbl), 1.0, orc._p(J, dbl), orc._p(D, dbl))
        assert np.array_equal(J, g["st%d_J" % i]) and np.array_equal(D, g["st%d_D" % i])
    for seed in (1, 42, 4294967295):
        d = np.zeros(64)
        L.orc_kat_mt_canonical(seed, 64, orc._p(d, dbl))
        assert np.array_equal(d, g["mt_%d" % seed])


@pytest.mark.skipif(not orc.have_ref(), reason="oracle/_ref not built (no /root/reference here)")
@pytest.mark.parametrize("cfg,n,seed", [("tree", 3000, 7), ("pair", 2500, 9), ("bal16", 600, 3),
                                        ("tree", 3, 1), ("tree", 4, 1), ("pair", 5, 2)])
def test_rung_a_vs_linked_reference_live(cfg, n, seed):
    """beyond the stored vectors: fresh inputs, compared live against the linked reference"""
    from common import simulate
    model, tree, fp = simulate(cfg, n, seed=seed)
    o = orc.Oracle(tree, model, fp, "A", seed=seed)
    R = orc.Reference(tree, model, fp, seed=seed)
    o.reset()
    R.reset(2, 3)
    assert np.array_equal(o.tri_llh(), R.tri_llh())
    Jo, Do, nacc, acc = o.run_mcmc(2, 3)
    Jr, Dr, accr = R.run_mcmc()
    assert np.array_equal(Jo, Jr) and np.array_equal(Do, Dr) and acc == accr
    assert orc.paths_equal(o.paths(), R.paths())
    assert np.array_equal(o.tri_llh(), R.tri_llh())


@pytest.mark.skipif(not orc.have_ref(), reason="oracle/_ref not built (no /root/reference here)")
@pytest.mark.parametrize("n,seed", [(2000, 1), (37, 5), (3, 2)])
def test_init_paths_indep_rung_a_vs_linked_reference(n, seed):
    """initialize_paths_indep (epievo_sim_pairwise.cpp:62-110): the oracle in
    reference-schedule mode against the glue around the LINKED forward-rejection sampler"""
    from common import simulate
    model, tree, fp = simulate("pair", n, seed=seed)
    root = fp.init
    leaf = fp.init ^ (fp.counts() & 1).astype(np.uint8)
    a = orc.init_paths_indep("orc", seed, model.rates, root, leaf, 1.0, "A")
    r = orc.init_paths_indep("ref", seed, model.rates, root, leaf, 1.0)
    assert orc.paths_equal(a, r)


def test_end_cond_samplers_match_linked_reference():
    """forward rejection (EndCondSampling.cpp:512-542) and end_cond_sampling_Nielsen (:583-617):
    the oracle's restatement with mt19937 + glibc gives the linked functions' jump times
    bit-for-bit over 200 consecutive samples of each grid point"""
    import ctypes as C
    g = np.load(os.path.join(GOLDEN, "kat.npz"))
    L = orc.orc_lib()
    for sampler, bits in ((0, 0x100), (1, 0x200)):
        for i, (r0, r1, a, b, T) in enumerate(g["ec_grid"]):
            want_c, want_t = g["ec%d_%d_counts" % (sampler, i)], g["ec%d_%d_times" % (sampler, i)]
            cnt, tt = np.zeros(len(want_c), np.uint32), np.zeros(max(len(want_t), 1))
            tot = L.orc_kat_end_cond_paths(orc.RNG_MT | bits, orc.MATH_LIBM, 7, r0, r1, int(a), int(b), T,
                                           len(want_c), orc._p(cnt, C.c_uint32), orc._p(tt, C.c_double), len(tt))
            assert tot == len(want_t) and np.array_equal(cnt, want_c)
            assert np.array_equal(tt[:tot], want_t)
