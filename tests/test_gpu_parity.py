"""GPU parity: the HIP path (through the C ABI) against the CPU oracle's parallel-schedule
rung B on the same seeded inputs -- bit-exact on paths, states, accept counts, J, D and
the cached triple log-likelihoods."""
import numpy as np
import pytest

import orc
from common import simulate

pytestmark = pytest.mark.gpu


def _dev(tree, model, fp, capacity=16):
    from epievo_amd.sampler import DeviceSampler
    d = DeviceSampler(0)
    d.set_tree(tree)
    d.set_model(model)
    d.upload_paths(fp, capacity)
    return d


@pytest.mark.parametrize("cfg,n", [("tree", 64), ("tree", 1000), ("pair", 1000), ("tree", 20011),
                                   ("pair", 20011), ("bal16", 3000), ("tree", 3), ("tree", 4),
                                   ("tree", 5), ("pair", 7), ("star4", 3000), ("multi", 3000),
                                   ("cat6", 3000)])
def test_reset_and_roundtrip(cfg, n):
    model, tree, fp = simulate(cfg, n, seed=3)
    d = _dev(tree, model, fp)
    assert orc.paths_equal(d.paths(), fp)          # upload -> download round trip
    d.reset()
    o = orc.Oracle(tree, model, fp, "B", cap=16)
    o.reset()
    assert np.array_equal(d.tri_llh(), o.tri_llh())
    Jd, Dd = d.suffstats()
    Jo, Do = o.suffstats()
    assert np.array_equal(Jd, Jo)
    assert np.array_equal(Dd, Do)


@pytest.mark.parametrize("cfg,n,sweeps", [("tree", 64, 3), ("tree", 1000, 3), ("pair", 1000, 3),
                                          ("tree", 20011, 2), ("pair", 20011, 2),
                                          ("bal16", 3000, 2), ("bal16", 3, 2), ("bal16", 5, 2), ("bal16", 67, 2),
                                          ("cat20", 200, 2), ("tree", 3, 2), ("tree", 4, 2),
                                          ("tree", 5, 2), ("pair", 6, 2), ("star4", 3000, 3),
                                          ("multi", 3000, 3), ("cat6", 3000, 3)])
def test_sweeps_bit_exact(cfg, n, sweeps):
    model, tree, fp = simulate(cfg, n, seed=5)
    seed = 0x1234567890abcdef
    d = _dev(tree, model, fp)
    d.reset()
    o = orc.Oracle(tree, model, fp, "B", cap=16, seed=seed)
    o.reset()
    for w in range(sweeps):
        na_d = d.sweep(1, seed, sweep_base=w)
        na_o = o.sweep(w)
        assert na_d == na_o, "accept count differs in sweep %d" % w
        assert orc.paths_equal(d.paths(), o.paths()), "paths differ after sweep %d" % w
        assert np.array_equal(d.tri_llh(), o.tri_llh())
    assert d.counters()["overflow"] == o.counters()["overflow"] == 0


@pytest.mark.parametrize("cfg,n", [("tree", 5000), ("pair", 5000), ("multi", 4000), ("cat6", 4000)])
def test_run_mcmc_bit_exact(cfg, n):
    model, tree, fp = simulate(cfg, n, seed=11)
    d = _dev(tree, model, fp)
    d.reset()
    o = orc.Oracle(tree, model, fp, "B", cap=16, seed=99)
    o.reset()
    Jd, Dd, nacc_d = d.run_mcmc(2, 3, 99, sweep_base=7)
    Jo, Do, nacc_o, _ = o.run_mcmc(2, 3, sweep_base=7)
    assert nacc_d == nacc_o
    assert np.array_equal(Jd, Jo)
    assert np.array_equal(Dd, Do)
    assert orc.paths_equal(d.paths(), o.paths())


def test_capacity_overflow_matches_oracle():
    """a deliberately tiny capacity: overflowing proposals are rejected and counted the
    same way on both sides, and the ABI reports EPV_ERR_CAPACITY"""
    from epievo_amd.sampler import CapacityError
    model, tree, fp = simulate("pair", 2000, seed=8)
    cap = int(fp.counts().max())
    d = _dev(tree, model, fp, capacity=cap)
    d.reset()
    o = orc.Oracle(tree, model, fp, "B", cap=cap, seed=4)
    o.reset()
    n_ovf = 0
    for w in range(3):
        try:
            d.sweep(1, 4, sweep_base=w)
        except CapacityError:
            n_ovf += 1
        o.sweep(w)
        assert orc.paths_equal(d.paths(), o.paths())
    assert o.counters()["overflow"] > 0 and n_ovf > 0
    assert d.counters()["overflow"] == o.counters()["overflow"]


def test_scale_jump_times():
    model, tree, fp = simulate("tree", 3000, seed=2)
    d = _dev(tree, model, fp)
    o = orc.Oracle(tree, model, fp, "B", cap=16)
    nb = tree.branches * np.array([1.0, 1.1, 0.7, 1.3, 0.9])
    d.scale_jump_times(nb)
    o.scale_jump_times(nb)
    assert orc.paths_equal(d.paths(), o.paths())


def test_columns_roundtrip():
    model, tree, fp = simulate("tree", 500, seed=2)
    d = _dev(tree, model, fp)
    d.reset()
    d.sweep(2, 5)
    before, tri = d.paths(), d.tri_llh()
    buf = d.get_columns(100, 7)
    d.put_columns(100, 7, buf)
    assert orc.paths_equal(d.paths(), before)
    assert np.array_equal(d.tri_llh(), tri)


def test_full_size_properties():
    """BASELINE config 3 size (n = 1e6, tree.nwk): properties that need no oracle run --
    leaf states are invariant under MCMC, jumps stay sorted inside (0, branch), J equals
    the total number of jumps on interior sites, D sums to (n-2) * branch length."""
    model, tree, fp = simulate("tree", 1000000, seed=42)
    d = _dev(tree, model, fp)
    d.reset()
    B, n = tree.n_nodes - 1, fp.n_sites

    def end_states(p):
        return (p.init.reshape(B, n) ^ (p.counts().reshape(B, n) & 1).astype(np.uint8))

    e0 = end_states(fp)
    J, D, nacc = d.run_mcmc(1, 2, 42)
    p = d.paths()
    leaves = [b for b in range(B) if tree.subtree_sizes[b + 1] == 1]
    assert np.array_equal(end_states(p)[leaves], e0[leaves])
    assert np.array_equal(p.init.reshape(B, n)[0], fp.init.reshape(B, n)[0])  # root fixed
    # child branches start in the parent's end state
    es = end_states(p)
    for b in range(B):
        par = tree.parent_ids[b + 1]
        if par:
            assert np.array_equal(p.init.reshape(B, n)[b], es[par - 1])
    cnt = p.counts().reshape(B, n)
    off = p.offsets[:-1].reshape(B, n)
    for b in range(B):
        seg = p.jumps[int(off[b, 0]):int(off[b, -1] + cnt[b, -1])]
        assert seg.min() > 0.0 and seg.max() < tree.branches[b + 1]
        ids = np.repeat(np.arange(n), cnt[b])
        same = ids[1:] == ids[:-1]
        assert np.all(np.diff(seg)[same] >= 0.0)
    Jn, Dn = d.suffstats()
    assert np.allclose(Jn.reshape(B, 8).sum(1), cnt[:, 1:-1].sum(1))
    assert np.allclose(Dn.reshape(B, 8).sum(1), (n - 2) * tree.branches[1:], rtol=1e-12)
    assert 0.9 < nacc / (2.0 * (n - 2)) < 1.0


@pytest.mark.parametrize("n,seed", [(5000, 3), (200001, 9), (3, 1), (4, 2)])
def test_init_paths_indep_bit_exact(n, seed):
    """epievo_sim_pairwise's initial paths on the device == the oracle's parallel rung"""
    from epievo_amd import host
    from epievo_amd.sampler import DeviceSampler
    model, tree, fp = simulate("pair", n, seed=seed)
    root = fp.init
    leaf = fp.init ^ (fp.counts() & 1).astype(np.uint8)
    d = DeviceSampler(0)
    d.set_tree(host.Tree.single_branch(1.0))
    d.set_model(model)
    d.init_paths_indep(root, leaf, seed=77, capacity=32)
    exp = orc.init_paths_indep("orc", 77, model.rates, root, leaf, 1.0, "B")
    assert orc.paths_equal(d.paths(), exp)
    # and the MCMC can start from them
    d.reset()
    o = orc.Oracle(tree, model, exp, "B", cap=32, seed=5)
    o.reset()
    assert d.sweep(2, 5) == o.sweep(0) + o.sweep(1)
    assert orc.paths_equal(d.paths(), o.paths())


def test_set_capacity_and_auto_grow_match_oracle():
    """epv_set_capacity re-strides the resident jump planes on the device; with auto_grow the
    Python/C++ mirrors absorb an overflow by doubling the slots, and the chain stays the
    oracle's chain at the same sequence of capacities"""
    from epievo_amd.sampler import CapacityError
    model, tree, fp = simulate("pair", 2000, seed=8)
    cap = int(fp.counts().max())
    d = _dev(tree, model, fp, capacity=cap)
    d.reset()
    before = d.paths()
    assert d.capacity() == cap
    d.set_capacity(cap + 5)                          # grow: nothing changes but the stride
    assert d.capacity() == cap + 5 and orc.paths_equal(d.paths(), before)
    with pytest.raises(CapacityError):
        d.set_capacity(cap - 1)                      # a resident path would not fit
    d.set_capacity(cap)                              # shrink back (every path fits)
    assert orc.paths_equal(d.paths(), before)
    d.auto_grow = True
    o = orc.Oracle(tree, model, fp, "B", cap=cap, seed=4)
    o.reset()
    caps = [cap]
    for w in range(4):
        assert d.sweep(1, 4, sweep_base=w) == o.sweep(w)   # no CapacityError: absorbed
        assert orc.paths_equal(d.paths(), o.paths())
        caps.append(d.capacity())
        o.set_rung("B", caps[-1])
    assert caps[-1] >= 2 * cap and d.capacity_events and "rejected" in d.capacity_events[0]
    assert d.counters()["overflow"] == o.counters()["overflow"] > 0
    # tri_llh cache survived the re-striding
    assert np.array_equal(d.tri_llh(), o.tri_llh())


@pytest.mark.parametrize("T,n", [(20.0, 2000), (150.0, 400)])
def test_long_branch_beyond_127_jumps(T, n):
    """Path::jumps is an unbounded std::vector in the reference (Path.hpp:52).  The device keeps
    fixed-stride slots whose count is a 15-bit field: a branch of length 20 (tens of jumps per
    path) and one of length 150 (paths with more than 127 jumps, beyond the 7-bit field of round 1)
    run without a single rejection by capacity, bit-identical to the oracle."""
    from epievo_amd import host
    from common import ref_test_model
    model = ref_test_model()
    tree = host.Tree.single_branch(T)
    fp = host.simulate(model, tree, n, 9)
    maxj = int(fp.counts().max())
    if T > 100:
        assert maxj > 127
    d = _dev(tree, model, fp, 0)                   # the library's default: max(16, 2 maxj + 8)
    cap = d.capacity()
    assert cap == max(16, 2 * maxj + 8)
    assert orc.paths_equal(d.paths(), fp)
    o = orc.Oracle(tree, model, fp, "B", cap=cap, seed=41)
    d.reset(); o.reset()
    assert np.array_equal(d.tri_llh(), o.tri_llh())
    for w in range(2):
        assert d.sweep(1, 41, sweep_base=w) == o.sweep(w)
    assert d.counters()["overflow"] == o.counters()["overflow"] == 0
    assert orc.paths_equal(d.paths(), o.paths())
    Jd, Dd = d.suffstats()
    Jo, Do = o.suffstats()
    assert np.array_equal(Jd, Jo) and np.array_equal(Dd, Do)
    # growing on demand past the old ceiling: a tiny start capacity is doubled until it fits
    g = _dev(tree, model, fp, maxj)
    g.auto_grow = True
    g.reset()
    for w in range(3):
        g.sweep(1, 41, sweep_base=w)
    assert g.capacity() > maxj and g.capacity_events


_SEPARATE = {"EPV_FUSED_PHASE": "0"}      # small launches take the fused phase kernel by default
@pytest.mark.parametrize("cfg,n,env", [("tree", 20011, dict(_SEPARATE, EPV_SEG_JUMPS="1")),
                                       ("tree", 20011, dict(_SEPARATE, EPV_SEG_JUMPS="0")),
                                       ("pair", 9000, dict(_SEPARATE, EPV_SEG_JUMPS="0")),
                                       ("pair", 9000, dict(_SEPARATE, EPV_SEG_JUMPS="1")),
                                       ("cat6", 3000, dict(_SEPARATE, EPV_SEG_JUMPS="1")),
                                       ("tree", 20011, dict(_SEPARATE, EPV_PROPOSE_V1="1")),
                                       ("pair", 9000, dict(_SEPARATE, EPV_PROPOSE_V1="1")),
                                       ("bal16", 2000, dict(_SEPARATE, EPV_PROPOSE_V2_GLOBAL="1", EPV_FORCE_GLOBAL_POOL="1")),
                                       ("tree", 5000, dict(_SEPARATE, EPV_FORCE_GLOBAL_POOL="1", EPV_PROPOSE_V2_GLOBAL="1",
                                                           EPV_SEG_JUMPS="1")),
                                       ("bal16", 2000, dict(_SEPARATE, EPV_PROPOSE_V3="0")),
                                       ("bal16", 2000, dict(_SEPARATE, EPV_PROPOSE_V3="1")),
                                       ("bal16", 2000, dict(_SEPARATE, EPV_PROPOSE_V3="1", EPV_P3_MIN_LIST="1")),
                                       ("bal16", 9000, dict(_SEPARATE, EPV_P3_SLAB_POOL="2")),
                                       ("bal32", 1500, dict(_SEPARATE, EPV_PROPOSE_V3="1")), ("bal64", 1200, dict(_SEPARATE, EPV_PROPOSE_V3="1")),
                                       ("bal64", 1200, dict(_SEPARATE, EPV_PROPOSE_V3="1", EPV_P3_MIN_LIST="1", EPV_P3_SLAB_POOL="2")), ("cat20", 2000, dict(_SEPARATE, EPV_PROPOSE_V3="1")),
                                       ("cat20", 2000, dict(_SEPARATE, EPV_PROPOSE_V3="1", EPV_P3_MIN_LIST="1", EPV_P3_SLAB_POOL="2")),
                                       ("tree", 20011, dict(_SEPARATE, EPV_PROPOSE_V3="1")),
                                       ("bal16", 2000, dict(_SEPARATE, EPV_ACCEPT_V3="0")),
                                       ("tree", 20011, dict(_SEPARATE, EPV_ACCEPT_V3="1")),
                                       ("cat6", 3000, dict(_SEPARATE, EPV_ACCEPT_V3="1", EPV_SEG_JUMPS="1")),
                                       ("pair", 9000, dict(_SEPARATE, EPV_ACCEPT_V3="1")),
                                       ("cat6", 3000, dict(_SEPARATE, EPV_PROPOSE_V3="1")),
                                       ("star4", 3000, dict(_SEPARATE, EPV_PROPOSE_V3="1")),
                                       ("pair", 9000, dict(_SEPARATE, EPV_PROPOSE_V3="1", EPV_P3_MIN_LIST="1")),
                                       ("tree", 20011, {"EPV_FUSED_PHASE": "1"}), ("pair", 9000, {"EPV_FUSED_PHASE": "1"}),
                                       ("cat6", 3000, {"EPV_FUSED_PHASE": "1"}),
                                       ("tree", 20011, {"EPV_FUSED_PHASE": "1", "EPV_ACCEPT_NO_CACHE": "1"})])
def test_every_kernel_path_is_bit_exact(cfg, n, env):
    """the library picks its kernels by workload (proposal kernel generation, where its record
    pool lives, sequential or segment-parallel jump sampling); every combination must give the
    oracle's bits, so each is forced here on a workload that would not select it by itself"""
    import os
    import subprocess
    import sys
    code = r'''
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np
import orc
from common import simulate
from epievo_amd.sampler import DeviceSampler
model, tree, fp = simulate(%r, %d, seed=6)
cap = int(max(16, 2 * fp.counts().max() + 8))
d = DeviceSampler(0); d.set_tree(tree); d.set_model(model); d.upload_paths(fp, cap); d.reset()
o = orc.Oracle(tree, model, fp, "B", cap=cap, seed=19); o.reset()
Jd, Dd, nd = d.run_mcmc(2, 3, 19, sweep_base=4)
Jo, Do, no, _ = o.run_mcmc(2, 3, sweep_base=4)
assert nd == no and np.array_equal(Jd, Jo) and np.array_equal(Dd, Do)
assert orc.paths_equal(d.paths(), o.paths()) and np.array_equal(d.tri_llh(), o.tri_llh())
# a tiny capacity: the overflow decisions must be the sequential sampler's too
cap2 = int(fp.counts().max())
d2 = DeviceSampler(0); d2.set_tree(tree); d2.set_model(model); d2.upload_paths(fp, cap2); d2.reset()
o2 = orc.Oracle(tree, model, fp, "B", cap=cap2, seed=23); o2.reset()
for w in range(2):
    try:
        na = d2.sweep(1, 23, sweep_base=w)
    except Exception as e:
        na = None
    nb = o2.sweep(w)
    assert orc.paths_equal(d2.paths(), o2.paths())
assert d2.counters()["overflow"] == o2.counters()["overflow"]
print("ok")
''' % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)), cfg, n)
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]


def test_phase_mode_follows_the_workload():
    """epv_phase_mode (include/epievo_mi355x.h): small launches take the fused colour phase, the
    reference's proposal arithmetic the first proposal kernel, large trees the third, a large capacity
    (more than 64 segments per branch possible) the separate kernels; the choice never changes a
    number (test_every_kernel_path_is_bit_exact forces each against rung B)"""
    from common import simulate
    from epievo_amd.sampler import DeviceSampler
    model, tree, fp = simulate("tree", 20011, seed=6)
    d = DeviceSampler(0); d.set_tree(tree); d.set_model(model); d.upload_paths(fp, 16); d.reset()
    assert d.phase_mode() == 3
    J0, D0, n0 = d.run_mcmc(1, 2, 5)
    d.set_options(reference_proposal_ratio=True)
    assert d.phase_mode() == 0
    d.set_options()
    d.set_capacity(40)                      # 2 C + 1 > 64: the fused phase's hand-over word is too short
    assert d.phase_mode() in (1, 2)
    d.reset()
    d2 = DeviceSampler(0); d2.set_tree(tree); d2.set_model(model); d2.upload_paths(fp, 40); d2.reset()
    J1, D1, n1 = d2.run_mcmc(1, 2, 5)
    assert n1 == n0 and np.array_equal(J1, J0) and np.array_equal(D1, D0)
    model, tree, fp = simulate("bal16", 2000, seed=6)
    d3 = DeviceSampler(0); d3.set_tree(tree); d3.set_model(model); d3.upload_paths(fp, 16); d3.reset()
    assert d3.phase_mode() == 4             # record pool does not fit LDS: the large-tree proposal kernel
    d3.set_options(reference_proposal_ratio=True)
    assert d3.phase_mode() == 0             # the reference's sums: first proposal kernel, pool in global memory


@pytest.mark.parametrize("scale,n", [(4.0, 30000), (3.0, 20000)])
def test_fused_phase_with_many_heavy_branches_and_several_rounds(scale, n, monkeypatch):
    """long branches on the tree.nwk topology (0.5 - 2 jumps per path): most (site, branch) pairs are
    heavy, a wave's records exceed its LDS pool, so waves run the update in several ROUNDS over
    prefixes of their lanes and later rounds list their heavy pairs over the node-table rows of lanes
    that have finished -- the stages behind the rounds (segment search, assembly, acceptance) must not
    depend on those rows.  Fused phase forced; bit-exact against rung B"""
    from common import tree_nwk
    from epievo_amd import host
    from epievo_amd.sampler import DeviceSampler
    monkeypatch.setenv("EPV_FUSED_PHASE", "1")
    model = simulate("tree", 100, seed=1)[0]
    t0 = tree_nwk()
    tree = host.Tree(t0.subtree_sizes, t0.parent_ids, t0.branches * scale)
    fp = host.simulate(model, tree, n, 77)
    assert fp.counts().mean() > 0.15 * scale / 4.0
    cap = int(max(16, min(31, 2 * fp.counts().max() + 8)))      # 2 C + 1 <= 64: the fused phase stays eligible
    d = DeviceSampler(0); d.set_tree(tree); d.set_model(model); d.upload_paths(fp, cap); d.reset()
    assert d.phase_mode() == 3
    o = orc.Oracle(tree, model, fp, "B", cap=cap, seed=41); o.reset()
    for w in range(3):
        try:
            na = d.sweep(1, 41, sweep_base=w)
        except CapacityError:
            na = None
        nb = o.sweep(w)
        assert na is None or na == nb
        assert orc.paths_equal(d.paths(), o.paths())
    assert np.array_equal(d.tri_llh(), o.tri_llh())
    assert d.counters()["overflow"] == o.counters()["overflow"]
    d.close()
