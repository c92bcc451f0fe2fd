"""The proposal ratio q(old)/q(new) of Metropolis_Hastings_site (SingleSiteSampler.cpp:503-507).

With SAMPLE_ROOT false (hard-wired, :441) it is exactly 1: per segment the reference accumulates
log P(end | start, data) - log PT(start -> end) = log(p[k+1][end] / p[k][start]), which telescopes
along a branch and over the tree to the log of the proposal's normalising constant -- a function
of the neighbours and the leaf data, not of the path.  The reference evaluates the two sums
numerically; the GPU (and the parallel rung of the oracle) use the exact 0 by default and keep
the reference's arithmetic as an option.  Pinned here:
  * rung A -- bit-identical to the linked reference (test_oracle_golden.py) -- never sees a
    ratio further than rounding from 0;
  * the parallel rung gives the same paths, accept counts and statistics in both modes;
  * the GPU equals the parallel rung bit for bit in either mode (-m gpu), and in
    forward-rejection mode (EPV_OPT_FORWARD_REJECTION) equals the rung with the reference's
    hot-path sampler, so that the Nielsen sampler is the only difference left to statistics."""
import numpy as np
import pytest

import orc
from common import simulate

CASES = [("tree", 6000), ("pair", 2500), ("bal16", 600), ("multi", 1500), ("star4", 1500)]


@pytest.mark.parametrize("cfg,n", CASES)
def test_reference_ratio_is_rounding_noise(cfg, n):
    model, tree, fp = simulate(cfg, n, seed=3)
    a = orc.Oracle(tree, model, fp, "A", seed=5)
    a.reset()
    for w in range(4):
        a.sweep(w)
    assert 0.0 < a.max_qdiff() < 1e-10        # evaluated (non-zero) and nothing but rounding
    if orc.have_ref():                       # ... on the chain the reference itself runs
        r = orc.Reference(tree, model, fp, seed=5)
        r.reset(0, 1)
        r.sweeps(4)
        assert orc.paths_equal(a.paths(), r.paths())


@pytest.mark.parametrize("cfg,n", CASES)
def test_modes_give_the_same_chain(cfg, n):
    model, tree, fp = simulate(cfg, n, seed=4)
    cap = int(max(16, 2 * fp.counts().max() + 8))
    ref = orc.Oracle(tree, model, fp, "B", cap=cap, seed=9)
    ref.set_proposal_mode(True)
    tel = orc.Oracle(tree, model, fp, "B", cap=cap, seed=9)      # the rung's default: telescoped
    ref.reset(); tel.reset()
    Jr, Dr, nr, _ = ref.run_mcmc(2, 4, sweep_base=0)
    Jt, Dt, nt, _ = tel.run_mcmc(2, 4, sweep_base=0)
    assert nr == nt and np.array_equal(Jr, Jt) and np.array_equal(Dr, Dt)
    assert orc.paths_equal(ref.paths(), tel.paths())
    assert np.array_equal(ref.tri_llh(), tel.tri_llh())
    assert 0.0 < ref.max_qdiff() < 1e-10 and tel.max_qdiff() == 0.0


def _dev(tree, model, fp, cap, **opts):
    from epievo_amd.sampler import DeviceSampler
    d = DeviceSampler(0)
    d.set_tree(tree)
    d.set_model(model)
    d.upload_paths(fp, cap)
    d.set_options(**opts)
    return d


@pytest.mark.gpu
@pytest.mark.parametrize("cfg,n", [("tree", 30000), ("pair", 6000), ("bal16", 1500), ("multi", 4000)])
def test_gpu_reference_ratio_mode_bit_exact(cfg, n):
    model, tree, fp = simulate(cfg, n, seed=4)
    cap = int(max(16, 2 * fp.counts().max() + 8))
    d = _dev(tree, model, fp, cap, reference_proposal_ratio=True)
    o = orc.Oracle(tree, model, fp, "B", cap=cap, seed=77)
    o.set_proposal_mode(True)
    d.reset(); o.reset()
    Jd, Dd, nd = d.run_mcmc(2, 3, 77, sweep_base=5)
    Jo, Do, no, _ = o.run_mcmc(2, 3, sweep_base=5)
    assert nd == no and np.array_equal(Jd, Jo) and np.array_equal(Dd, Do)
    assert orc.paths_equal(d.paths(), o.paths())
    assert np.array_equal(d.tri_llh(), o.tri_llh())
    # ... and the default mode walks the same chain
    t = _dev(tree, model, fp, cap)
    t.reset()
    Jt, Dt, nt = t.run_mcmc(2, 3, 77, sweep_base=5)
    assert nt == nd and np.array_equal(Jt, Jd) and np.array_equal(Dt, Dd)
    assert orc.paths_equal(t.paths(), d.paths())


@pytest.mark.gpu
@pytest.mark.parametrize("cfg,n,refq", [("tree", 3000, False), ("pair", 1500, False), ("multi", 1200, True)])
def test_gpu_forward_rejection_mode_bit_exact(cfg, n, refq):
    """every segment by forward rejection, as the reference's hot path does: GPU == the parallel
    rung with ORC_SAMPLER_FORWARD (small n: a flip on a short branch needs ~1/P(a->b) trials)"""
    model, tree, fp = simulate(cfg, n, seed=6)
    cap = int(max(16, 2 * fp.counts().max() + 8))
    d = _dev(tree, model, fp, cap, forward_rejection=True, reference_proposal_ratio=refq)
    o = orc.Oracle(tree, model, fp, "B", cap=cap, seed=31)
    o.set_sampler(True)
    o.set_proposal_mode(refq)
    d.reset(); o.reset()
    for w in range(3):
        assert d.sweep(1, 31, sweep_base=w) == o.sweep(w)
    assert orc.paths_equal(d.paths(), o.paths())
    Jd, Dd = d.suffstats()
    Jo, Do = o.suffstats()
    assert np.array_equal(Jd, Jo) and np.array_equal(Dd, Do)
    # the Nielsen default draws other jump times for flips but targets the same law: after the
    # same sweeps the two GPU modes agree on the leaf states and on most paths
    nd = _dev(tree, model, fp, cap)
    nd.reset()
    nd.sweep(3, 31, sweep_base=0)
    pf, pn = d.paths(), nd.paths()
    es = lambda q: (q.init ^ (q.counts() & 1).astype(np.uint8))
    B = tree.n_nodes - 1
    leaves = [b for b in range(B) if tree.subtree_sizes[b + 1] == 1]
    assert np.array_equal(es(pf).reshape(B, -1)[leaves], es(pn).reshape(B, -1)[leaves])
