"""Several shards on one GPU (epievo_amd.parallel.LocalGroup) reproduce the single-context run
bit for bit -- paths, accept counts, cached log-likelihoods, J AND D."""
import numpy as np
import pytest

import orc
from common import simulate
from epievo_amd.parallel import LocalGroup

pytestmark = pytest.mark.gpu


def _single(tree, model, fp, cap):
    from epievo_amd.sampler import DeviceSampler
    d = DeviceSampler(0)
    d.set_tree(tree); d.set_model(model); d.upload_paths(fp, cap)
    return d


@pytest.mark.parametrize("cfg,n,k", [("tree", 20011, 2), ("tree", 20011, 3), ("pair", 9000, 2),
                                     ("cat6", 7000, 2), ("tree", 1500, 2)])
def test_group_equals_single_context(cfg, n, k):
    model, tree, fp = simulate(cfg, n, seed=4)
    cap = int(max(16, 2 * fp.counts().max() + 8))
    d = _single(tree, model, fp, cap)
    g = LocalGroup(0, k)
    g.set_tree(tree); g.set_model(model); g.upload_paths(fp, cap)
    assert len(g.subs) == (k if n >= k * 1536 else 1)
    d.reset(); g.reset()
    assert np.array_equal(g.tri_llh(), d.tri_llh())
    Jd, Dd, nd = d.run_mcmc(2, 3, 99, sweep_base=7)
    Jg, Dg, ng = g.run_mcmc(2, 3, 99, sweep_base=7)
    assert nd == ng
    assert np.array_equal(Jd, Jg)
    assert np.array_equal(Dd, Dg)              # bit-identical D: shared block partials
    assert orc.paths_equal(g.paths(), d.paths())
    # a second E-step after an M-step-like change of branch lengths
    nb = tree.branches * 1.1
    d.scale_jump_times(nb); g.scale_jump_times(nb)
    d.reset(); g.reset()
    Jd, Dd, nd = d.run_mcmc(1, 2, 5, sweep_base=20)
    Jg, Dg, ng = g.run_mcmc(1, 2, 5, sweep_base=20)
    assert nd == ng and np.array_equal(Jd, Jg) and np.array_equal(Dd, Dg)
    assert orc.paths_equal(g.paths(), d.paths())
    # plain sweeps, more than one internal halo lasts (forces an internal refresh)
    assert g.sweep(100, 3, sweep_base=40) == d.sweep(100, 3, sweep_base=40)
    assert orc.paths_equal(g.paths(), d.paths())
    g.close()


def test_group_capacity_overflow_follows_single_context():
    """a deliberately tiny capacity: both absorb the overflow (auto_grow) and stay identical"""
    model, tree, fp = simulate("pair", 6000, seed=8)
    cap = int(fp.counts().max())
    d = _single(tree, model, fp, cap)
    d.auto_grow = True
    g = LocalGroup(0, 2)
    g.set_tree(tree); g.set_model(model); g.upload_paths(fp, cap)
    g.auto_grow = True
    assert len(g.subs) == 2
    for it in range(3):
        d.reset(); g.reset()
        Jd, Dd, nd = d.run_mcmc(1, 2, 4, sweep_base=3 * it)
        Jg, Dg, ng = g.run_mcmc(1, 2, 4, sweep_base=3 * it)
        assert nd == ng and np.array_equal(Jd, Jg) and np.array_equal(Dd, Dg)
        assert orc.paths_equal(g.paths(), d.paths())
    g.reset()
    assert d.capacity() == g.capacity() > cap and d.capacity_events and g.capacity_events
    g.close()
