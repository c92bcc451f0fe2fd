"""Error behaviour and edge cases of the C ABI on the GPU (the boundary never throws: it
returns EPV_ERR_* and a message)."""
import numpy as np
import pytest

import orc
from common import simulate
from epievo_amd import host
from epievo_amd.sampler import DeviceSampler, EpvError, CapacityError, EPV_ERR_STATE, EPV_ERR_ARG

pytestmark = pytest.mark.gpu


def test_call_order_and_argument_errors():
    model, tree, fp = simulate("tree", 100, seed=1)
    d = DeviceSampler(0)
    with pytest.raises(EpvError) as e:
        d.upload_paths(fp, 16)                      # tree first
    assert e.value.code == EPV_ERR_STATE
    d.set_tree(tree)
    d.upload_paths(fp, 16)
    with pytest.raises(EpvError) as e:
        d.reset()                                   # model missing
    assert e.value.code == EPV_ERR_STATE
    d.set_model(model)
    with pytest.raises(EpvError) as e:
        d.sweep(1, 1)                               # reset missing
    assert e.value.code == EPV_ERR_STATE and "epv_reset" in str(e.value)
    bad = host.Model(model.rates * np.array([1, 1, 1, 1, 1, 1, 1, -1.0]), model.T, model.baseline)
    with pytest.raises(EpvError) as e:
        d.set_model(bad)
    assert e.value.code == EPV_ERR_ARG
    bad_tree = host.Tree(tree.subtree_sizes, tree.parent_ids, tree.branches * np.array([0, 1, 1, 0, 1.0]))
    with pytest.raises(EpvError):
        DeviceSampler(0).set_tree(bad_tree)         # zero-length branch
    with pytest.raises(EpvError):
        d.set_update_range(0, 50)
    d2 = DeviceSampler(0)
    m2, t2, f2 = simulate("pair", 2000, seed=1)
    d2.set_tree(t2)
    with pytest.raises(CapacityError):
        d2.upload_paths(f2, 1)                      # an input path has more jumps than capacity


def test_scale_invalidates_reset_and_model_change_needs_reset():
    model, tree, fp = simulate("tree", 500, seed=2)
    d = DeviceSampler(0)
    d.set_tree(tree)
    d.set_model(model)
    d.upload_paths(fp, 16)
    d.reset()
    d.sweep(1, 3)
    d.scale_jump_times(tree.branches * 1.5)
    with pytest.raises(EpvError):
        d.sweep(1, 3)                               # cached log-likelihoods are stale
    d.reset()
    d.sweep(1, 3)
    d.set_model(model)
    with pytest.raises(EpvError):
        d.run_mcmc(1, 1, 3)


def test_no_jump_inputs_and_all_equal_states():
    """degenerate inputs: no jumps anywhere / every site in the same state"""
    model, tree, fp = simulate("tree", 300, seed=3)
    B, n = tree.n_nodes - 1, 300
    for state in (0, 1):
        empty = host.FlatPaths(n, tree.n_nodes, np.full(B * n, state, np.uint8), np.zeros(B * n + 1, np.uint64),
                               np.zeros(0))
        d = DeviceSampler(0)
        d.set_tree(tree)
        d.set_model(model)
        d.upload_paths(empty, 16)
        d.reset()
        o = orc.Oracle(tree, model, empty, "B", cap=16, seed=9)
        o.reset()
        assert np.array_equal(d.tri_llh(), o.tri_llh())
        for w in range(3):
            assert d.sweep(1, 9, w) == o.sweep(w)
        assert orc.paths_equal(d.paths(), o.paths())
        Jd, Dd = d.suffstats()
        Jo, Do = o.suffstats()
        assert np.array_equal(Jd, Jo) and np.array_equal(Dd, Do)


def test_paths_at_full_capacity_survive_roundtrip_and_sweeps():
    """ragged input: a few paths filled to the capacity limit, ties between neighbours"""
    model, tree, fp = simulate("pair", 400, seed=4)
    cap = 12
    init = fp.init.copy()
    cnt = np.zeros(400, np.int64)
    cnt[[5, 6, 7, 100, 250]] = cap
    cnt[[8, 9]] = 3
    jl = []
    for s in range(400):
        if cnt[s] == cap:
            jl.append(np.arange(1, cap + 1) / (cap + 1.0))      # identical times in neighbours: ties
        elif cnt[s]:
            jl.append(np.array([1, 2, 3]) / (cap + 1.0))
    off = np.zeros(401, np.uint64)
    off[1:] = np.cumsum(cnt)
    full = host.FlatPaths(400, 2, init, off, np.concatenate(jl))
    d = DeviceSampler(0)
    d.set_tree(tree)
    d.set_model(model)
    d.upload_paths(full, cap)
    assert orc.paths_equal(d.paths(), full)
    d.reset()
    o = orc.Oracle(tree, model, full, "B", cap=cap, seed=2)
    o.reset()
    assert np.array_equal(d.tri_llh(), o.tri_llh())
    for w in range(4):
        try:
            d.sweep(1, 2, w)
        except CapacityError:
            pass
        o.sweep(w)
        assert orc.paths_equal(d.paths(), o.paths())
    assert d.counters()["overflow"] == o.counters()["overflow"]
