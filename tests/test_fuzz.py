"""Randomised parity: random tree topologies (2..12 leaves, multifurcations, branch lengths
over two orders of magnitude), random models and seeds -- the HIP path must reproduce the
parallel rung of the oracle bit for bit on every one (paths, accept counts, cached
log-likelihoods, J, D)."""
import numpy as np
import pytest

import orc
from common import _tmp
from epievo_amd import host



def _random_newick(rng, n_leaves):
    """random rooted tree as Newick; inner nodes get 2 or 3 children"""
    names = iter("n%d" % i for i in range(1000))
    nodes = ["%s:%.4g" % (next(names), 10 ** rng.uniform(-2.0, 0.0)) for _ in range(n_leaves)]
    while len(nodes) > 1:
        k = min(len(nodes), int(rng.choice([2, 2, 3])))
        idx = sorted(rng.choice(len(nodes), size=k, replace=False), reverse=True)
        kids = [nodes.pop(i) for i in idx]
        if nodes:
            nodes.append("(%s)%s:%.4g" % (",".join(kids), next(names), 10 ** rng.uniform(-2.0, 0.0)))
        else:
            nodes.append("(%s)%s:0.0" % (",".join(kids), next(names)))
    return nodes[0] + ";\n"


def _random_case(case):
    rng = np.random.RandomState(1000 + case)
    n_leaves = int(rng.randint(2, 13))
    tree = host.Tree.read(_tmp("fuzz%d.nwk" % case, _random_newick(rng, n_leaves)))
    st0, st1 = rng.uniform(0.6, 0.95, size=2)
    b0, b1 = -rng.uniform(0.2, 1.5), -rng.uniform(0.5, 2.5)
    model = host.Model.read(_tmp("fuzz%d.param" % case,
                                 "stationary\t%.6f\t%.6f\nbaseline\t%.6f\t%.6f\n" % (st0, st1, b0, b1)), scale=True)
    n = int(rng.choice([3, 17, 500, 2001]))
    fp = host.simulate(model, tree, n, int(rng.randint(1, 1 << 30)))
    cap = int(max(16, 2 * fp.counts().max() + 8))
    seed = int(rng.randint(1, 1 << 62))
    return tree, model, fp, cap, seed


@pytest.mark.skipif(not orc.have_ref(), reason="oracle/_ref not built (no /root/reference here)")
@pytest.mark.parametrize("case", range(12))
def test_random_case_rung_a_vs_linked_reference(case):
    """the same random inputs, reference-schedule rung against the LINKED reference, live"""
    tree, model, fp, cap, seed = _random_case(case)
    o = orc.Oracle(tree, model, fp, "A", seed=seed & 0xffffffff)
    R = orc.Reference(tree, model, fp, seed=seed & 0xffffffff)
    o.reset()
    R.reset(1, 2)
    assert np.array_equal(o.tri_llh(), R.tri_llh())
    Jo, Do, nacc, acc = o.run_mcmc(1, 2)
    Jr, Dr, accr = R.run_mcmc()
    assert np.array_equal(Jo, Jr) and np.array_equal(Do, Dr) and acc == accr
    assert orc.paths_equal(o.paths(), R.paths())


@pytest.mark.gpu
@pytest.mark.parametrize("case", range(12))
def test_random_tree_model_seed(case):
    from epievo_amd.sampler import DeviceSampler
    tree, model, fp, cap, seed = _random_case(case)
    d = DeviceSampler(0)
    d.set_tree(tree)
    d.set_model(model)
    d.upload_paths(fp, cap)
    d.reset()
    o = orc.Oracle(tree, model, fp, "B", cap=cap, seed=seed)
    o.reset()
    assert np.array_equal(d.tri_llh(), o.tri_llh())
    for w in range(2):
        assert d.sweep(1, seed, sweep_base=w) == o.sweep(w)
        assert orc.paths_equal(d.paths(), o.paths())
        assert np.array_equal(d.tri_llh(), o.tri_llh())
    Jd, Dd, nd = d.run_mcmc(1, 2, seed, sweep_base=5)
    Jo, Do, no, _ = o.run_mcmc(1, 2, sweep_base=5)
    assert nd == no and np.array_equal(Jd, Jo) and np.array_equal(Dd, Do)
    assert d.counters()["overflow"] == o.counters()["overflow"]
