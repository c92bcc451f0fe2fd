"""Site-independent model of epievo_initialization (IndepSite.cpp) on the GPU against the
oracle's parallel rung: conditional expectations, path counts and update_paths_indep."""
import numpy as np
import pytest

import orc
from common import simulate

pytestmark = pytest.mark.gpu
RATES = np.array([0.7, 1.9])


def _dev(tree, model, fp, cap=32):
    from epievo_amd.sampler import DeviceSampler
    d = DeviceSampler(0)
    d.set_tree(tree)
    d.set_model(model)
    d.upload_paths(fp, cap)
    return d


# bal16 / cat6 run the 64-lane launch shape with B % 4 != 0 (30 and 10 branches): its level-0
# partials are nb*V16 doubles, more than the 8-context statistics need (round-1 overrun)
@pytest.mark.parametrize("cfg,n", [("tree", 5000), ("pair", 3000), ("bal16", 700), ("tree", 3), ("tree", 100001),
                                   ("bal16", 1000), ("bal16", 100001), ("cat6", 5000), ("multi", 70000)])
def test_indep_expectation_and_counts_bit_exact(cfg, n):
    model, tree, fp = simulate(cfg, n, seed=6)
    d = _dev(tree, model, fp)
    o = orc.Oracle(tree, model, fp, "B", cap=32)
    Jd, Dd = d.indep_expectation(RATES)
    Jo, Do = o.indep_expectation(RATES)
    assert np.array_equal(Jd, Jo) and np.array_equal(Dd, Do)
    Jd, Dd = d.indep_suffstats()
    Jo, Do = o.indep_suffstats()
    assert np.array_equal(Jd, Jo) and np.array_equal(Dd, Do)
    # and they are close to the reference-schedule rung (sequential sums, glibc exp)
    a = orc.Oracle(tree, model, fp, "A")
    Ja, Da = a.indep_expectation(RATES)
    np.testing.assert_allclose(Jd * 0 + d.indep_expectation(RATES)[0], Ja, rtol=1e-10)


@pytest.mark.parametrize("cfg,n", [("tree", 5000), ("pair", 3000), ("bal16", 700), ("tree", 3)])
def test_indep_update_paths_bit_exact(cfg, n):
    model, tree, fp = simulate(cfg, n, seed=7)
    d = _dev(tree, model, fp)
    o = orc.Oracle(tree, model, fp, "B", cap=32, seed=123)
    for w in range(3):
        d.indep_update_paths(RATES, 123, sweep=0xF0000000 + w)
        o.indep_update_paths(RATES, 0xF0000000 + w)
        assert orc.paths_equal(d.paths(), o.paths())
    # leaves keep their states, the 8-context statistics still work on the new paths
    B = tree.n_nodes - 1
    p = d.paths()
    leaves = [b for b in range(B) if tree.subtree_sizes[b + 1] == 1]
    es = lambda q: (q.init.reshape(B, -1) ^ (q.counts().reshape(B, -1) & 1).astype(np.uint8))
    assert np.array_equal(es(p)[leaves], es(fp)[leaves])
    Jd, Dd = d.suffstats()
    Jo, Do = o.suffstats()
    assert np.array_equal(Jd, Jo) and np.array_equal(Dd, Do)
