"""The reference's own end-to-end methodology for the sampler (src/harnesses/MCMC_test.cpp):
compare the MCMC's sufficient statistics with EXACT posterior draws obtained by rejecting
whole-sequence forward simulations that miss the observed leaf sequence.  This checks the
stationary law of the 3-colour chain independently of the oracle ladder."""
import ctypes as C

import numpy as np
import pytest

import orc
from common import ref_test_model
from epievo_amd import host

T_BRANCH = 0.4
ROOT = np.array([0, 1, 1, 0, 1, 0, 0, 1], np.uint8)
LEAF = np.array([0, 1, 0, 0, 1, 1, 0, 1], np.uint8)      # two interior flips, ends unchanged


def _exact(model, want=20000):
    L = orc.orc_lib()
    Jm, Dm, J2, D2 = (np.zeros(8) for _ in range(4))
    kept = L.orc_exact_posterior(orc._p(model.rates, C.c_double), len(ROOT), orc._p(ROOT, C.c_uint8),
                                 orc._p(LEAF, C.c_uint8), T_BRANCH, 3, want, 400000000,
                                 orc._p(Jm, C.c_double), orc._p(Dm, C.c_double), orc._p(J2, C.c_double),
                                 orc._p(D2, C.c_double))
    assert kept == want
    return Jm, Dm, np.sqrt(np.maximum(J2 - Jm ** 2, 1e-12) / want), np.sqrt(np.maximum(D2 - Dm ** 2, 1e-12) / want)


def _initial_paths():
    # any valid history with the right end points: one jump at T/2 where root and leaf differ
    cnt = (ROOT != LEAF).astype(np.int64)
    off = np.zeros(len(ROOT) + 1, np.uint64)
    off[1:] = np.cumsum(cnt)
    return host.FlatPaths(len(ROOT), 2, ROOT.copy(), off, np.full(int(cnt.sum()), T_BRANCH / 2))


def _check(Jc, Dc, n_eff, exact):
    Jm, Dm, Jse, Dse = exact
    # chain averages: allow for autocorrelation with a generous effective-sample factor
    tolJ = 6.0 * (Jse * np.sqrt(20000.0 / n_eff) + 1e-3) + 0.01
    tolD = 6.0 * (Dse * np.sqrt(20000.0 / n_eff) + 1e-3) + 0.005
    assert np.all(np.abs(Jc - Jm) < tolJ), (Jc, Jm)
    assert np.all(np.abs(Dc - Dm) < tolD), (Dc, Dm)
    assert abs(Dc.sum() - (len(ROOT) - 2) * T_BRANCH) < 1e-9


def test_oracle_chains_match_exact_posterior():
    model = ref_test_model()
    exact = _exact(model)
    tree = host.Tree.single_branch(T_BRANCH)
    for rung in ("A", "B"):
        o = orc.Oracle(tree, model, _initial_paths(), rung, cap=48 if rung == "B" else 0, seed=17)
        o.reset()
        J, D, nacc, acc = o.run_mcmc(500, 30000)
        _check(J, D, 3000.0, exact)


@pytest.mark.gpu
def test_gpu_chain_matches_exact_posterior():
    from epievo_amd.sampler import DeviceSampler
    model = ref_test_model()
    exact = _exact(model)
    d = DeviceSampler(0)
    d.set_tree(host.Tree.single_branch(T_BRANCH))
    d.set_model(model)
    d.upload_paths(_initial_paths(), 48)
    d.reset()
    J, D, nacc = d.run_mcmc(500, 20000, 99)
    _check(J, D, 2000.0, exact)


# ---------------------------------------------------------------- the same on a tree
# tree.nwk's topology (((C,D)E,F)G), 14 sites (12 interior), leaf data with two flips on two of
# the three leaves; the exact sampler simulates the whole sequence down the tree and keeps the
# histories that end in the observed leaves (oracle: orc_exact_posterior_tree)
TROOT = np.array([0, 0, 0, 0, 1, 1, 1, 1, 1, 0, 0, 0, 1, 1], np.uint8)
TLEAF = {"C": [], "D": [3], "F": [9]}          # sites where the leaf differs from the root: two block
                                               # boundaries move (contexts with rates of 3.5-3.7)


def _tree_case():
    from common import tree_nwk
    tree = tree_nwk()
    n, N = len(TROOT), tree.n_nodes
    leaf = np.tile(TROOT, (N, 1))
    for name, flips in TLEAF.items():
        leaf[tree.node_names.index(name), flips] ^= 1
    # a valid start: every leaf flip is one jump in the middle of the leaf's own branch
    init = np.tile(TROOT, (N - 1, 1))
    jumps, cnt = [], np.zeros((N - 1, n), np.int64)
    for b in range(1, N):
        for s in range(n):
            if tree.subtree_sizes[b] == 1 and leaf[b, s] != TROOT[s]:
                cnt[b - 1, s] = 1
                jumps.append(tree.branches[b] / 2)
    off = np.zeros((N - 1) * n + 1, np.uint64)
    off[1:] = np.cumsum(cnt.reshape(-1))
    fp = host.FlatPaths(n, N, init.reshape(-1).copy(), off, np.array(jumps))
    return tree, leaf, fp


def _exact_tree(model, tree, leaf, want=20000):
    L = orc.orc_lib()
    B = tree.n_nodes - 1
    u8p, u32p, dp = C.POINTER(C.c_uint8), C.POINTER(C.c_uint32), C.POINTER(C.c_double)
    L.orc_exact_posterior_tree.restype = C.c_uint64
    L.orc_exact_posterior_tree.argtypes = [dp, C.c_uint64, C.c_int, u32p, u32p, dp, u8p, u8p, C.c_uint64, C.c_uint64,
                                           C.c_uint64, dp, dp, dp, dp]
    Jm, Dm, J2, D2 = (np.zeros(B * 8) for _ in range(4))
    flat = np.ascontiguousarray(leaf.reshape(-1))
    kept = L.orc_exact_posterior_tree(orc._p(model.rates, C.c_double), leaf.shape[1], tree.n_nodes,
                                      orc._p(tree.parent_ids, C.c_uint32), orc._p(tree.subtree_sizes, C.c_uint32),
                                      orc._p(tree.branches, C.c_double), orc._p(TROOT, C.c_uint8),
                                      orc._p(flat, C.c_uint8), 7, want, 400000000, orc._p(Jm, C.c_double),
                                      orc._p(Dm, C.c_double), orc._p(J2, C.c_double), orc._p(D2, C.c_double))
    assert kept == want
    return Jm, Dm, np.sqrt(np.maximum(J2 - Jm ** 2, 1e-12) / want), np.sqrt(np.maximum(D2 - Dm ** 2, 1e-12) / want)


def _check_tree(Jc, Dc, n_eff, exact, tree, want=20000):
    Jm, Dm, Jse, Dse = exact
    tolJ = 6.0 * (Jse * np.sqrt(want / n_eff) + 1e-3) + 0.01
    tolD = 6.0 * (Dse * np.sqrt(want / n_eff) + 3e-4) + 0.002
    assert np.all(np.abs(Jc - Jm) < tolJ), (Jc - Jm, tolJ)
    assert np.all(np.abs(Dc - Dm) < tolD), (Dc - Dm, tolD)
    B = tree.n_nodes - 1
    np.testing.assert_allclose(Dc.reshape(B, 8).sum(1), (len(TROOT) - 2) * tree.branches[1:], rtol=1e-9)


def test_oracle_chains_match_exact_posterior_on_the_tree():
    model = ref_test_model()
    tree, leaf, fp = _tree_case()
    exact = _exact_tree(model, tree, leaf)
    assert exact[0].sum() > 1.9          # at least the two observed flips, on average a bit more
    for rung, seed in (("A", 3), ("B", 4), ("B", 5)):
        o = orc.Oracle(tree, model, fp, rung, cap=32 if rung == "B" else 0, seed=seed)
        o.reset()
        J, D, nacc, acc = o.run_mcmc(500, 20000)
        _check_tree(J, D, 2000.0, exact, tree)


@pytest.mark.gpu
@pytest.mark.parametrize("seed,opts", [(11, {}), (12, {}), (13, {}), (14, {"forward_rejection": True}),
                                       (15, {"reference_proposal_ratio": True})])
def test_gpu_chain_matches_exact_posterior_on_the_tree(seed, opts):
    """several seeds of the GPU chain -- the default kernels, forward rejection for every segment,
    and the reference's proposal-ratio arithmetic -- against exact posterior draws"""
    from epievo_amd.sampler import DeviceSampler
    model = ref_test_model()
    tree, leaf, fp = _tree_case()
    exact = _exact_tree(model, tree, leaf)
    d = DeviceSampler(0)
    d.set_tree(tree)
    d.set_model(model)
    d.upload_paths(fp, 32)
    d.set_options(**opts)
    d.reset()
    J, D, nacc = d.run_mcmc(300, 12000, seed)
    _check_tree(J, D, 1200.0, exact, tree)
    # the leaves still carry the observed data
    p = d.paths()
    B, n = tree.n_nodes - 1, len(TROOT)
    es = (p.init.reshape(B, n) ^ (p.counts().reshape(B, n) & 1).astype(np.uint8))
    for b in range(1, tree.n_nodes):
        if tree.subtree_sizes[b] == 1:
            assert np.array_equal(es[b - 1], leaf[b])
