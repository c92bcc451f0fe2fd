"""SURVEY section 8f row 3 on the device: epievo_sim's forward simulation, site-parallel by thinning
(epievo_amd/csrc/epv_forward.h).  Ladder: the sequential host restatement is bit-identical to the
linked TripletSampler (tests/test_forward_sim.py, rung A); the thinning rung of the oracle
(orc_forward_thinning) is THE SAME PROCESS statistically (here, on the CPU); the GPU equals the
thinning rung bit for bit (here, -m gpu)."""
import ctypes as C
import os
import subprocess
import time

import numpy as np
import pytest

import orc
from common import config, ref_test_model, TEST_PARAM_TEXT, TREE_NWK_TEXT
from epievo_amd import _build, host
from test_forward_sim import _run as run_sequential


def _local_from_global(tree, n, seqs, off, tt, pp):
    """global jumps (time order per node) -> FlatPaths, as global_jumps_to_paths does"""
    N, B = tree.n_nodes, tree.n_nodes - 1
    seqs = seqs.reshape(N, n)
    init = seqs[tree.parent_ids[1:]].reshape(-1).copy()
    cnt = np.zeros((B, n), np.int64)
    jumps = []
    for b in range(B):
        lo, hi = int(off[b + 1]), int(off[b + 2])
        pos = pp[lo:hi].astype(np.int64)
        order = np.argsort(pos, kind="stable")
        cnt[b] = np.bincount(pos, minlength=n)
        jumps.append(tt[lo:hi][order])
    offsets = np.concatenate([[0], np.cumsum(cnt.reshape(-1))]).astype(np.uint64)
    return host.FlatPaths(n, N, init, offsets, np.concatenate(jumps) if jumps else np.zeros(0))


def _check_valid(fp, states, tree, n):
    B = tree.n_nodes - 1
    init, cnt = fp.init.reshape(B, n), fp.counts().reshape(B, n)
    assert np.array_equal(init, states[tree.parent_ids[1:]])              # a child starts where its parent ended
    assert np.array_equal(init ^ (cnt & 1).astype(np.uint8), states[1:])  # ... and ends in init ^ parity
    assert cnt[:, 0].sum() == 0 and cnt[:, n - 1].sum() == 0              # the end sites never change
    for b in range(B):
        lo, hi = int(fp.offsets[b * n]), int(fp.offsets[(b + 1) * n])
        jb = fp.jumps[lo:hi]
        if jb.size:
            assert jb.min() > 0 and jb.max() < tree.branches[b + 1]
            up = np.diff(jb) > 0
            first = (fp.offsets[b * n:(b + 1) * n] - lo).astype(np.int64)
            first = first[(first > 0) & (first < jb.size)]
            up[first - 1] = True
            assert up.all()


@pytest.mark.parametrize("cfg,n", [("tree", 2000), ("pair", 500), ("bal16", 300), ("tree", 3)])
def test_thinning_rung_makes_valid_histories(cfg, n):
    m, tree = ref_test_model(), config(cfg)
    fp, states = orc.forward_thinning(m, tree, n, seed=5)
    _check_valid(fp, states, tree, n)
    # a given root sequence is kept; the same seed gives the same histories
    root = (np.arange(n) % 3 == 0).astype(np.uint8)
    fp2, st2 = orc.forward_thinning(m, tree, n, seed=5, root=root)
    assert np.array_equal(st2[0], root)
    _check_valid(fp2, st2, tree, n)
    fp3, _ = orc.forward_thinning(m, tree, n, seed=5, root=root)
    assert orc.paths_equal(fp2, fp3)


def test_thinning_rung_is_the_reference_process():
    """event counts and dwell times per context (J, D of the complete histories -- the sufficient
    statistics of the process), and the pair frequencies at the leaves, against the sequential
    simulator that is bit-identical to the reference's TripletSampler loop: several seeds each"""
    m, tree, n = ref_test_model(), config("tree"), 30000
    B = tree.n_nodes - 1

    def stats(fp):
        o = orc.Oracle(tree, m, fp, "A")
        J, D = o.suffstats()
        es = fp.init.reshape(B, n) ^ (fp.counts().reshape(B, n) & 1).astype(np.uint8)
        leaf = es[[b for b in range(B) if tree.subtree_sizes[b + 1] == 1]]
        pairs = np.array([np.mean((leaf[:, :-1] == a) & (leaf[:, 1:] == b)) for a in (0, 1) for b in (0, 1)])
        return J.reshape(B, 8).sum(0), D.reshape(B, 8).sum(0), pairs

    seq = [stats(_local_from_global(tree, n, *run_sequential(host.lib().epvh_forward_sim, s, m, tree, n))) for s in (1, 2, 3)]
    thin = [stats(orc.forward_thinning(m, tree, n, seed=s)[0]) for s in (11, 12, 13)]
    Js, Jt = sum(x[0] for x in seq), sum(x[0] for x in thin)
    # Poisson counts: |difference| within 5 standard deviations in every context
    assert np.all(np.abs(Js - Jt) < 5 * np.sqrt(Js + Jt + 1)), (Js, Jt)
    # J_c / D_c estimates the rate of context c in both
    Ds, Dt = sum(x[1] for x in seq), sum(x[1] for x in thin)
    for c in range(8):
        if Jt[c] > 50:
            assert abs(Jt[c] / Dt[c] - m.rates[c]) < 5 * m.rates[c] / np.sqrt(Jt[c])
            assert abs(Js[c] / Ds[c] - m.rates[c]) < 5 * m.rates[c] / np.sqrt(Js[c])
    ps, pt = np.mean([x[2] for x in seq], 0), np.mean([x[2] for x in thin], 0)
    assert np.all(np.abs(ps - pt) < 0.01), (ps, pt)


# ------------------------------------------------------------------ the device (-m gpu)
@pytest.mark.gpu
@pytest.mark.parametrize("cfg,n,seed", [("tree", 3, 1), ("tree", 64, 2), ("tree", 5000, 3), ("tree", 100001, 4),
                                        ("pair", 4000, 5), ("bal16", 2000, 6), ("cat6", 3000, 7)])
def test_device_forward_simulation_equals_the_thinning_rung(cfg, n, seed):
    from epievo_amd.sampler import DeviceSampler
    m, tree = ref_test_model(), config(cfg)
    exp, states = orc.forward_thinning(m, tree, n, seed=seed)
    d = DeviceSampler(0)
    d.set_tree(tree); d.set_model(m)
    tot = d.forward_simulate(n, seed)
    assert tot == len(exp.jumps)
    assert orc.paths_equal(d.paths(), exp)
    # with a given root sequence
    root = (np.arange(n) % 5 < 2).astype(np.uint8)
    exp2, _ = orc.forward_thinning(m, tree, n, seed=seed + 100, root=root)
    d.forward_simulate(n, seed + 100, root=root)
    assert orc.paths_equal(d.paths(), exp2)
    d.close()


@pytest.mark.gpu
def test_device_forward_simulation_long_branch_and_capacity():
    """T = 3: ~30 candidates per site, dependency chains over many rounds and several launches; slots
    too narrow at first (the wrapper widens them: same histories)"""
    from epievo_amd.sampler import DeviceSampler
    m, tree, n = ref_test_model(), host.Tree.single_branch(3.0), 3000
    exp, _ = orc.forward_thinning(m, tree, n, seed=9)
    assert exp.counts().max() > 16
    d = DeviceSampler(0)
    d.set_tree(tree); d.set_model(m)
    d.forward_simulate(n, 9, capacity=4)
    assert d.capacity() >= exp.counts().max()
    assert orc.paths_equal(d.paths(), exp)
    d.close()


@pytest.mark.gpu
def test_simulated_histories_are_resident_for_the_sampler():
    from epievo_amd.sampler import DeviceSampler
    m, tree, n = ref_test_model(), config("tree"), 20000
    d = DeviceSampler(0)
    d.set_tree(tree); d.set_model(m)
    d.forward_simulate(n, 21)
    fp = d.paths()
    d.reset()
    J, D, nacc = d.run_mcmc(1, 2, 77)
    o = orc.Oracle(tree, m, fp, "B", cap=16, seed=77)
    o.reset()
    Jo, Do, nacc_o, _ = o.run_mcmc(1, 2)
    assert nacc == nacc_o and np.array_equal(J, Jo) and np.array_equal(D, Do)
    assert orc.paths_equal(d.paths(), o.paths())
    d.close()


@pytest.mark.gpu
def test_epievo_sim_gpu_mode_files(tmp_path):
    """epievo_sim -g 0: the reference's files from the device simulation; global_jumps_to_paths turns
    them into the local paths the device holds (-P writes those directly)"""
    d = str(tmp_path)
    open(d + "/p.param", "w").write(TEST_PARAM_TEXT)
    open(d + "/t.nwk", "w").write(TREE_NWK_TEXT)
    n = 3000
    r = subprocess.run([os.path.join(_build.BIN_DIR, "epievo_sim"), "-v", "-g", "0", "-n", str(n), "-s", "42",
                        "-p", d + "/g.jumps", "-P", d + "/direct.paths", "-t", d + "/t.nwk", d + "/p.param", d + "/x.states"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    m, tree = ref_test_model(), config("tree")
    exp, states = orc.forward_thinning(m, tree, n, seed=42)
    ev = int([l for l in r.stderr.split("\n") if "TOTAL SAMPLED EVENTS" in l][0].split(":")[1].strip(" ]"))
    assert ev == len(exp.jumps)
    st = np.loadtxt(d + "/x.states", dtype=np.int64, skiprows=1)[:, 1:].T.astype(np.uint8)
    assert np.array_equal(st, states)
    direct, names, tt = host.read_paths(d + "/direct.paths")
    assert orc.paths_equal(direct, exp) and np.array_equal(tt, tree.branches)
    r = subprocess.run([os.path.join(_build.BIN_DIR, "global_jumps_to_paths"), "-t", d + "/t.nwk", d + "/x.states",
                        d + "/g.jumps", d + "/conv.paths"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    conv, _, _ = host.read_paths(d + "/conv.paths")
    assert orc.paths_equal(conv, exp)


@pytest.mark.gpu
def test_config5_inputs_generated_on_the_device_in_seconds():
    """SURVEY 8f-3's point: inputs for configs 4-5 (n = 1e7) took the sequential generator 20-34 s;
    the device generates the 16-leaf tree's 3e8 site-branch histories in about a second"""
    from epievo_amd.sampler import DeviceSampler
    m, tree, n = ref_test_model(), config("bal16"), 10_000_000
    d = DeviceSampler(0)
    d.set_tree(tree); d.set_model(m)
    d.forward_simulate(100000, 1)           # warm the kernels
    t0 = time.time()
    tot = d.forward_simulate(n, 5, capacity=8)
    wall = time.time() - t0
    alloc_ms, sim_ms = d.forward_last_ms()
    el = sim_ms * 1e-3                      # (the rest is 38 GB of device memory being freed / allocated)
    B = tree.n_nodes - 1
    # expected events: (n - 2) sites x 30 branches x 0.05 x (mean rate = 1 per site and unit time)
    assert abs(tot / ((n - 2) * B * 0.05) - 1.0) < 0.02
    J, D = d.suffstats()
    np.testing.assert_allclose(D.reshape(B, 8).sum(1), (n - 2) * tree.branches[1:], rtol=1e-10)
    assert J.sum() == tot
    print("device forward simulation, 16-leaf tree, n = 1e7: %.2f s simulating + %.2f s of device memory "
          "management (call: %.2f s), %d events" % (el, alloc_ms * 1e-3, wall, tot))
    assert el < 2.0, el
    d.close()
