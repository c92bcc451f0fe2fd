"""Host-side C++ library (model, M-step, formats) against the reference's golden outputs."""
import glob
import os

import numpy as np
import pytest

from common import GOLDEN, ref_test_model, tree_nwk, _tmp
from epievo_amd import host


def test_read_model_matches_reference():
    g = np.load(os.path.join(GOLDEN, "kat.npz"))
    m = ref_test_model()
    assert np.array_equal(m.rates, g["model_scaled_rates"])
    assert np.array_equal(m.T, g["model_scaled_T"])
    assert np.array_equal(m.baseline, g["model_scaled_bl"])
    from common import TEST_PARAM_TEXT
    u = host.Model.read(_tmp("test.param", TEST_PARAM_TEXT), scale=False)
    assert np.array_equal(u.rates, g["model_unscaled_rates"])
    # SURVEY.md section 8d quotes the scaled rates to 6 digits
    np.testing.assert_allclose(m.rates, [0.236332, 3.65369, 10.201, 3.45555, 3.65369, 4.19543,
                                         3.45555, 0.0869415], rtol=2e-6)
    assert abs(host.lib().epvh_rate_scaling_factor(
        m.rates.ctypes.data_as(__import__("ctypes").POINTER(__import__("ctypes").c_double))) - 1.0) < 1e-12


def test_triplet_param_form_roundtrip():
    m = ref_test_model()
    text = "\n".join("%s\t%.17g" % (format(i, "03b"), r) for i, r in enumerate(m.rates)) + "\n"
    t = host.Model.read(_tmp("triplet.param", text), scale=False)
    np.testing.assert_allclose(t.rates, m.rates, rtol=1e-12)
    np.testing.assert_allclose(t.T, m.T, rtol=1e-9)


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "*_n1000_s*.npz"))))
def test_m_step_matches_reference(path):
    g = np.load(path)
    model = host.Model(g["rates"], g["T"], np.zeros(4))
    for tag, opt in (("mr", False), ("mb", True)):
        m2, br, llh, text = host.m_step(model, g["branches"], g["J"], g["D"], optimize_branches=opt)
        assert np.array_equal(m2.rates, g[tag + "_rates"])
        assert np.array_equal(m2.T, g[tag + "_T"])
        assert np.array_equal(m2.baseline, g[tag + "_baseline"])
        assert np.array_equal(br, g[tag + "_branches"])
        assert llh == float(g[tag + "_llh"])
        assert text == str(g[tag + "_text"])


def test_tree_parse_and_newick():
    t = tree_nwk()
    assert list(t.subtree_sizes) == [5, 3, 1, 1, 1]
    assert list(t.parent_ids) == [0, 0, 1, 1, 0]
    assert np.array_equal(t.branches, [0.0, 0.02, 0.03, 0.06, 0.1])
    assert t.node_names == ["G", "E", "C", "D", "F"]
    u = host.Tree.read(_tmp("unnamed.nwk", "((:0.1,:0.2):0.3,B:0.4);"))
    assert list(u.subtree_sizes) == [5, 3, 1, 1, 1]
    # the reference passes the counter by value into children: siblings share a name
    assert u.node_names == ["node_0", "node_1", "node_2", "node_2", "B"]
    with pytest.raises(RuntimeError):
        host.Tree.read(_tmp("bad.nwk", "((A:0.1,B:0.2);"))


def test_paths_file_roundtrip(tmp_path):
    from common import simulate
    model, tree, fp = simulate("tree", 300, seed=4)
    f = str(tmp_path / "x.local_paths")
    host.write_paths(f, tree.node_names, tree.branches, fp)
    fp2, names, tt = host.read_paths(f)
    assert names == tree.node_names
    assert np.array_equal(tt, tree.branches)
    assert np.array_equal(fp2.init, fp.init) and np.array_equal(fp2.offsets, fp.offsets)
    assert np.array_equal(fp2.jumps, fp.jumps)          # max_digits10 round-trips exactly
    lines = open(f).read().split("\n")
    assert lines[0] == "NODE:G" and lines[1] == "NODE:E"
    assert lines[2].startswith("0\t%d\t0.02" % fp.init[0]) and lines[2].endswith("\t")


def test_simulator_statistics():
    from common import simulate
    model, tree, fp = simulate("tree", 200000, seed=42)
    kbar = len(fp.jumps) / float(fp.n_sites * 4)
    assert 0.045 < kbar < 0.06           # 0.052 jumps per path on tree.nwk
    model, tree, fp = simulate("pair", 100000, seed=42)
    assert 0.9 < len(fp.jumps) / 1e5 < 1.1   # one expected change per site per unit time


def _states_from_paths(tree, fp):
    B, n = tree.n_nodes - 1, fp.n_sites
    es = fp.init.reshape(B, n) ^ (fp.counts().reshape(B, n) & 1).astype(np.uint8)
    st = np.zeros((tree.n_nodes, n), np.uint8)
    st[0] = fp.init.reshape(B, n)[0]
    st[1:] = es
    return st


def test_indep_m_steps_match_linked_reference():
    """estimate_rates_indep / estimate_rates_and_branches_indep (IndepSite.cpp:299-360)"""
    import orc
    if not orc.have_ref():
        pytest.skip("oracle/_ref not built")
    from common import simulate
    for cfg in ("tree", "bal16"):
        model, tree, fp = simulate(cfg, 800, seed=3)
        R = orc.Reference(tree, model, fp)
        J, D = R.indep_suffstats()
        for opt in (False, True):
            R2 = orc.Reference(tree, model, fp)
            r_ref, br_ref = R2.indep_m_step(opt, J, D, np.array([0.4, 0.9]), tree.n_nodes)
            r, br = host.indep_m_step(np.array([0.4, 0.9]), tree.branches, J, D, optimize_branches=opt)
            assert np.array_equal(r, r_ref) and np.array_equal(br, br_ref)
            if opt:   # the reference also rescaled its paths: the same as scale_jump_times
                o = orc.Oracle(tree, model, fp, "A")
                o.scale_jump_times(br)
                assert orc.paths_equal(o.paths(), R2.paths())


def test_heuristic_initial_paths_invariants():
    """initialize_paths of epievo_initialization.cpp:141-185 is a static function of a main
    that cannot be built here (smithlab_cpp): parity UNPINNED; its defining properties are
    checked instead -- leaves untouched, a child starts in its parent's state, one jump
    exactly where the two ends of a branch differ, inside the branch."""
    from common import simulate
    model, tree, fp = simulate("tree", 2000, seed=8)
    st = _states_from_paths(tree, fp)
    leaves = [i for i in range(tree.n_nodes) if tree.subtree_sizes[i] == 1]
    st0 = st.copy()
    st[[i for i in range(tree.n_nodes) if i not in leaves]] = 0
    p = host.initialize_paths_heuristic(5, tree, st)
    B, n = tree.n_nodes - 1, 2000
    assert np.array_equal(st[leaves], st0[leaves])
    init, cnt = p.init.reshape(B, n), p.counts().reshape(B, n)
    for b in range(B):
        node, par = b + 1, tree.parent_ids[b + 1]
        assert np.array_equal(init[b], st[par])
        assert np.array_equal(cnt[b], (st[par] != st[node]).astype(np.int64))
    assert p.jumps.min() > 0 and np.all(cnt <= 1)
    # internal states are copies of one of the children's states
    assert np.all((st[1] == st[2]) | (st[1] == st[3]))
