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
