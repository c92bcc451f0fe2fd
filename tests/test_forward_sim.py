"""TripletSampler / epievo_sim / global_jumps_to_paths, restated on the
host (epv_forward.cpp).  Same std::mt19937 and libstdc++ distributions in the same order as
the reference, so a seed gives bit-identical sequences and global jumps."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import orc
from common import GOLDEN, config, ref_test_model, TEST_PARAM_TEXT, TREE_NWK_TEXT
from epievo_amd import _build, host

dp, u8p, u32p, u64p = (C.POINTER(C.c_double), C.POINTER(C.c_uint8), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64))


def _run(fn, seed, model, tree, n):
    N, cap = tree.n_nodes, 40 * n + 100
    seqs, off = np.zeros(N * n, np.uint8), np.zeros(N + 1, np.uint64)
    tt, pp = np.zeros(cap), np.zeros(cap, np.uint64)
    fn.restype = C.c_uint64
    fn.argtypes = [C.c_uint64, dp, dp, C.c_int, u32p, dp, C.c_uint64, u8p, u64p, dp, u64p, C.c_uint64]
    tot = fn(seed, orc._p(model.rates, C.c_double), orc._p(model.T, C.c_double), N,
             orc._p(tree.parent_ids, C.c_uint32), orc._p(tree.branches, C.c_double), n,
             orc._p(seqs, C.c_uint8), orc._p(off, C.c_uint64), orc._p(tt, C.c_double), orc._p(pp, C.c_uint64), cap)
    return seqs, off, tt[:tot], pp[:tot]


def test_forward_sim_matches_golden():
    g = np.load(os.path.join(GOLDEN, "kat.npz"))
    m = ref_test_model()
    for cfg, n in (("tree", 600), ("pair", 300)):
        seqs, off, tt, pp = _run(host.lib().epvh_forward_sim, 42, m, config(cfg), n)
        assert np.array_equal(seqs, g["fwd_%s_seqs" % cfg]) and np.array_equal(off, g["fwd_%s_off" % cfg])
        assert np.array_equal(tt, g["fwd_%s_t" % cfg]) and np.array_equal(pp, g["fwd_%s_p" % cfg])


@pytest.mark.skipif(not orc.have_ref(), reason="oracle/_ref not built (no /root/reference here)")
@pytest.mark.parametrize("cfg,n,seed", [("tree", 5000, 1), ("pair", 3000, 7), ("bal16", 800, 3), ("tree", 3, 5)])
def test_forward_sim_matches_linked_triplet_sampler(cfg, n, seed):
    m, tree = ref_test_model(), config(cfg)
    a = _run(host.lib().epvh_forward_sim, seed, m, tree, n)
    b = _run(orc.ref_lib().ref_forward_sim, seed, m, tree, n)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))


def test_sim_and_convert_clis(tmp_path):
    """epievo_sim -> global_jumps_to_paths, the first two steps of the README pipeline
    (README.md:134-140), with the reference's flags and file formats"""
    d = str(tmp_path)
    open(d + "/p.param", "w").write(TEST_PARAM_TEXT)
    open(d + "/t.nwk", "w").write(TREE_NWK_TEXT)
    n = 700
    r = subprocess.run([os.path.join(_build.BIN_DIR, "epievo_sim"), "-v", "-n", str(n), "-s", "42", "-p", d + "/g.jumps",
                        "-t", d + "/t.nwk", d + "/p.param", d + "/x.states"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([os.path.join(_build.BIN_DIR, "global_jumps_to_paths"), "-t", d + "/t.nwk", d + "/x.states",
                        d + "/g.jumps", d + "/x.paths"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    m, tree = ref_test_model(), config("tree")
    seqs, off, tt, pp = _run(host.lib().epvh_forward_sim, 42, m, tree, n)
    seqs = seqs.reshape(tree.n_nodes, n)
    # states file: header of all node names, one row per site
    rows = open(d + "/x.states").read().split("\n")
    assert rows[0] == "#G\tE\tC\tD\tF" and rows[1] == "0\t" + "\t".join(str(x) for x in seqs[:, 0])
    got = np.array([[int(v) for v in row.split("\t")[1:]] for row in rows[1:n + 1]], np.uint8).T
    assert np.array_equal(got, seqs)
    # global jumps file: ROOT line, root bits, NODE blocks of "time<TAB>position" at max_digits10
    lines = open(d + "/g.jumps").read().split("\n")
    assert lines[0] == "ROOT:G" and lines[1] == "".join(str(x) for x in seqs[0]) and lines[2] == "NODE:E"
    k = int(off[1])
    assert lines[3] == "%.17g\t%d" % (tt[k], pp[k])
    # local paths: a child starts in its parent's state; jumps land on the right sites in order
    fp, names, tot = host.read_paths(d + "/x.paths")
    assert names == ["G", "E", "C", "D", "F"] and np.array_equal(tot, tree.branches)
    B = tree.n_nodes - 1
    assert np.array_equal(fp.init.reshape(B, n), seqs[tree.parent_ids[1:]])
    cnt = fp.counts().reshape(B, n)
    for b in range(B):
        lo, hi = int(off[b + 1]), int(off[b + 2])
        assert np.array_equal(cnt[b], np.bincount(pp[lo:hi].astype(np.int64), minlength=n))
        order = np.argsort(pp[lo:hi], kind="stable")
        o0 = int(fp.offsets[b * n])
        assert np.array_equal(fp.jumps[o0:o0 + hi - lo], tt[lo:hi][order])
    # end states of the leaves equal the simulated leaf sequences
    es = fp.init.reshape(B, n) ^ (cnt & 1).astype(np.uint8)
    assert np.array_equal(es, seqs[1:])
    # usage errors follow the reference: message, EXIT_SUCCESS
    r = subprocess.run([os.path.join(_build.BIN_DIR, "epievo_sim"), d + "/p.param", d + "/y.states"],
                       capture_output=True, text=True)
    assert r.returncode == 0 and "specify exactly one of: tree or time" in r.stderr


def test_parallel_forward_sim_mode(tmp_path):
    """epievo_sim -j N (new): sibling subtrees on their own threads, one std::mt19937 per branch
    seeded from (seed, node).  The output depends on the seed and the tree only -- not on N -- and
    is another realisation of the same process as the sequential mode (which stays pinned to the
    linked TripletSampler above); a fixed-seed digest pins the mode against regressions."""
    import hashlib
    d = str(tmp_path)
    open(d + "/p.param", "w").write(TEST_PARAM_TEXT)
    open(d + "/t.nwk", "w").write(TREE_NWK_TEXT)
    n = 20000
    ev = {}
    for j in (1, 2, 8):
        r = subprocess.run([os.path.join(_build.BIN_DIR, "epievo_sim"), "-v", "-n", str(n), "-s", "42", "-j", str(j),
                            "-p", d + "/g%d.jumps" % j, "-t", d + "/t.nwk", d + "/p.param", d + "/x%d.states" % j],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        ev[j] = int([l for l in r.stderr.split("\n") if "TOTAL SAMPLED EVENTS" in l][0].split(":")[1].strip(" ]"))
    g = {j: open(d + "/g%d.jumps" % j, "rb").read() for j in (1, 2, 8)}
    x = {j: open(d + "/x%d.states" % j, "rb").read() for j in (1, 2, 8)}
    assert g[2] == g[8] and x[2] == x[8]                 # independent of the number of threads
    assert g[1] != g[2]                                   # not the sequential stream
    assert abs(ev[2] - ev[1]) < 6 * np.sqrt(ev[1])        # the same process: event counts agree (Poisson)
    assert hashlib.sha256(g[2]).hexdigest()[:16] == PARALLEL_DIGEST, hashlib.sha256(g[2]).hexdigest()[:16]
    # the parallel output is a valid history: global_jumps_to_paths reproduces the states file
    r = subprocess.run([os.path.join(_build.BIN_DIR, "global_jumps_to_paths"), "-t", d + "/t.nwk", d + "/x2.states",
                        d + "/g2.jumps", d + "/x2.paths"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    fp, names, tt = host.read_paths(d + "/x2.paths")
    tree = config("tree")
    B = tree.n_nodes - 1
    end = fp.init.reshape(B, n) ^ (fp.counts().reshape(B, n) & 1).astype(np.uint8)
    st = np.loadtxt(d + "/x2.states", dtype=np.int64, skiprows=1)[:, 1:].T.astype(np.uint8)   # [node][site]
    for b in range(1, tree.n_nodes):
        assert np.array_equal(end[b - 1], st[b]) and np.array_equal(fp.init.reshape(B, n)[b - 1], st[tree.parent_ids[b]])


PARALLEL_DIGEST = "e0cae3586c452ba4"
