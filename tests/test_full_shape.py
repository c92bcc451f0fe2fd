"""BASELINE.json's configurations at their stated shapes (-m gpu), checked through
size-independent properties where the oracle cannot follow in seconds:

  config 2  single branch T = 1.0, n = 1e5, epievo_sim_pairwise -L 100 through the CLI:
            bit-compared with the oracle's parallel rung (it can: 1e7 resamples)
  config 3/4  tree.nwk at n = 1e7 -- config 4's whole genome on ONE GPU: two contexts equal one
            context bit for bit (J, D, accept count, cached log-likelihoods: a checksum of
            checksums), dwell times add up to (n - 2) x branch length, counts are integers,
            reset() is idempotent on the cached log-likelihoods, leaves keep their data
  config 5  16-leaf balanced tree, n = 1.25e6 (= 1e7 / 8, one GPU's share), -b through the CLI:
            leaf data kept, every jump inside its branch, the same run sharded over four
            rehearsal slots is byte-identical

  config 5+ 16-leaf balanced tree at n = 4.6e6 with 16 jump slots: 4.4e9 jump slots, i.e. element
            offsets beyond 2^32 (the 64-bit index paths jump_idx / meta_idx / jbaseL of the kernels):
            the same properties, two contexts against one

The timings of these runs are kept under profiles/ by tools/full_shape_artifacts.py."""
import os
import subprocess

import numpy as np
import pytest

import orc
from common import simulate, TEST_PARAM_TEXT
from epievo_amd import _build, host

pytestmark = pytest.mark.gpu
BIN = _build.BIN_DIR


def test_config2_sim_pairwise_as_stated(tmp_path):
    n, L, seed = 100000, 100, 9
    model, tree, fp = simulate("pair", n, seed=3)
    d = str(tmp_path)
    open(d + "/p.param", "w").write(TEST_PARAM_TEXT)
    root = fp.init
    leaf = fp.init ^ (fp.counts() & 1).astype(np.uint8)
    with open(d + "/obs.states", "w") as f:
        f.write("#root\tleaf\n")
        f.write("".join("%d\t%d\t%d\n" % (i, root[i], leaf[i]) for i in range(n)))
    r = subprocess.run([os.path.join(BIN, "epievo_sim_pairwise"), "-L", str(L), "-T", "1.0", "-s", str(seed),
                        "-o", d + "/out.paths", "-v", d + "/p.param", d + "/obs.states"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    out, names, tt = host.read_paths(d + "/out.paths")
    t1 = host.Tree.single_branch(1.0)
    exp0 = orc.init_paths_indep("orc", seed, model.rates, root, leaf, 1.0, "B")
    o = orc.Oracle(t1, model, exp0, "B", cap=32, seed=seed)
    o.reset()
    nacc = sum(o.sweep(w) for w in range(L))
    assert o.counters()["overflow"] == 0
    assert orc.paths_equal(out, o.paths())
    acc = [l for l in r.stderr.split("\n") if l.startswith("acceptance rate")]
    assert acc and abs(float(acc[0].split(":")[1]) - nacc / float(L * (n - 2))) < 1e-6      # printed at 6 digits


def test_config4_genome_on_one_gpu_properties():
    from epievo_amd.parallel import LocalGroup
    from epievo_amd.sampler import DeviceSampler
    n, burn, batch, seed = 10_000_000, 2, 3, 77
    model, tree, fp = simulate("tree", n, seed=42)
    B = tree.n_nodes - 1
    es0 = fp.init.reshape(B, n) ^ (fp.counts().reshape(B, n) & 1).astype(np.uint8)
    d = DeviceSampler(0)
    d.set_tree(tree); d.set_model(model); d.upload_paths(fp, 16)
    d.reset()
    J, D, nacc = d.run_mcmc(burn, batch, seed, average=False)
    assert np.array_equal(J, np.round(J))                       # J are counts, summed exactly
    np.testing.assert_allclose(D.reshape(B, 8).sum(1), batch * (n - 2) * tree.branches[1:], rtol=1e-10)
    assert 0.9 < nacc / float(batch * (n - 2)) <= 1.0
    tri = d.tri_llh()
    d.reset()                                                   # recomputed from the paths ...
    assert np.array_equal(tri, d.tri_llh())                     # ... equal to what the sweeps cached
    p = d.paths()
    leaves = [b for b in range(B) if tree.subtree_sizes[b + 1] == 1]
    es = p.init.reshape(B, n) ^ (p.counts().reshape(B, n) & 1).astype(np.uint8)
    assert np.array_equal(es[leaves], es0[leaves])
    assert np.array_equal(p.init.reshape(B, n)[:, [0, n - 1]], fp.init.reshape(B, n)[:, [0, n - 1]])
    d.close()
    del p, es
    # two concurrent contexts on the same genome: every number the same
    g = LocalGroup(0, 2, burn + batch)
    g.set_tree(tree); g.set_model(model); g.upload_paths(fp, 16)
    assert len(g.subs) == 2
    g.reset()
    Jg, Dg, ng = g.run_mcmc(burn, batch, seed, average=False)
    assert ng == nacc and np.array_equal(Jg, J) and np.array_equal(Dg, D)
    assert np.array_equal(g.tri_llh(), tri)
    g.close()


def test_config5_shard_through_the_cli(tmp_path):
    n = 1_250_000
    model, tree, fp = simulate("bal16", n, seed=5)
    B, N = tree.n_nodes - 1, tree.n_nodes
    d = str(tmp_path)
    open(d + "/p.param", "w").write(TEST_PARAM_TEXT)
    from test_cli import _newick
    open(d + "/t.nwk", "w").write(_newick(tree))
    host.write_paths(d + "/in.paths", tree.node_names, tree.branches, fp)
    es0 = fp.init.reshape(B, n) ^ (fp.counts().reshape(B, n) & 1).astype(np.uint8)
    leaves = [b for b in range(B) if tree.subtree_sizes[b + 1] == 1]
    outs = {}
    for tag, env in (("one", {"EPV_DEVICES": "0"}), ("four", {"EPV_DEVICES": "0,0,0,0"})):
        r = subprocess.run([os.path.join(BIN, "epievo_est_params_histories"), "-i", "1", "-B", "4", "-L", "2", "-b",
                            "-s", "3", "-o", d + "/%s.paths" % tag, "-p", d + "/%s.param" % tag, "-t", d + "/%s.nwk" % tag,
                            "-v", d + "/p.param", d + "/t.nwk", d + "/in.paths"], capture_output=True, text=True,
                           env=dict(os.environ, **env))
        assert r.returncode == 0, r.stderr
        outs[tag] = (open(d + "/%s.param" % tag).read(), open(d + "/%s.nwk" % tag).read(),
                     [l for l in r.stderr.split("\n") if l and l[0].isdigit()])
    assert outs["one"] == outs["four"]
    # the 1 GB paths files: byte-identical between the two layouts
    import hashlib
    h = []
    for tag in ("one", "four"):
        m = hashlib.sha256()
        with open(d + "/%s.paths" % tag, "rb") as f:
            for chunk in iter(lambda: f.read(1 << 24), b""):
                m.update(chunk)
        h.append(m.hexdigest())
    assert h[0] == h[1]
    os.remove(d + "/four.paths")
    out, names, tt = host.read_paths(d + "/one.paths")
    assert out.n_sites == n and names == tree.node_names
    es = out.init.reshape(B, n) ^ (out.counts().reshape(B, n) & 1).astype(np.uint8)
    assert np.array_equal(es[leaves], es0[leaves])
    t2 = host.Tree.read(d + "/one.nwk")
    for b in range(1, N):
        jb = out.jumps[int(out.offsets[(b - 1) * n]):int(out.offsets[b * n])]
        assert jb.size == 0 or (jb.min() > 0 and jb.max() < tt[b])
        assert abs(t2.branches[b] - tt[b]) < 1e-5 * tt[b]
    assert np.all(np.isfinite(host.Model.read(d + "/one.param", scale=True).rates))


def test_config3_fused_contexts_equal_one_context_of_separate_kernels():
    """the bench's own configuration (tree.nwk, n = 1e6, -L 10 -B 50): three contexts, each small
    enough for the fused colour phase, against ONE context, whose 5209-wave phases take the separate
    proposal / jump / accept kernels -- every number the same, bit for bit"""
    from epievo_amd.parallel import LocalGroup
    from epievo_amd.sampler import DeviceSampler
    n, burn, batch, seed = 1_000_000, 10, 50, 42
    model, tree, fp = simulate("tree", n, seed=42)
    d = DeviceSampler(0)
    d.set_tree(tree); d.set_model(model); d.upload_paths(fp, 16)
    assert d.phase_mode() == 1
    d.reset()
    J, D, nacc = d.run_mcmc(burn, batch, seed)
    tri, p = d.tri_llh(), d.paths()
    d.close()
    g = LocalGroup(0, 3, burn + batch)
    g.set_tree(tree); g.set_model(model); g.upload_paths(fp, 16)
    assert len(g.subs) == 3 and g.phase_mode() == 3
    g.reset()
    Jg, Dg, ng = g.run_mcmc(burn, batch, seed)
    assert ng == nacc and np.array_equal(Jg, J) and np.array_equal(Dg, D)
    assert np.array_equal(g.tri_llh(), tri)
    assert orc.paths_equal(g.paths(), p)
    g.close()


def _check_paths_are_valid(p, tree, n):
    """every jump inside its branch, ascending inside its path"""
    B = tree.n_nodes - 1
    starts = p.offsets[:-1]
    for b in range(B):
        lo, hi = int(p.offsets[b * n]), int(p.offsets[(b + 1) * n])
        jb = p.jumps[lo:hi]
        if jb.size == 0:
            continue
        assert jb.min() > 0.0 and jb.max() < tree.branches[b + 1]
        up = np.diff(jb) > 0.0
        first = (starts[b * n:(b + 1) * n] - lo).astype(np.int64)      # index of each path's first jump
        first = first[(first > 0) & (first < jb.size)]
        up[first - 1] = True                                           # a new path may start lower
        assert up.all()


def test_config5_beyond_32_bit_jump_offsets():
    """2 * 30 * 16 * n jump slots > 2^32 from n = 4.47e6 on: indices into the second path buffer of
    the high branches no longer fit 32 bits.  ~35 GB of HBM"""
    from epievo_amd.parallel import LocalGroup
    from epievo_amd.sampler import DeviceSampler
    n, burn, batch, seed, cap = 4_600_000, 1, 2, 21, 16
    model, tree, fp = simulate("bal16", n, seed=11)
    B = tree.n_nodes - 1
    assert 2 * B * cap * n > 2 ** 32
    leaves = [b for b in range(B) if tree.subtree_sizes[b + 1] == 1]
    es0 = (fp.init.reshape(B, n) ^ (fp.counts().reshape(B, n) & 1).astype(np.uint8))[leaves]
    d = DeviceSampler(0)
    d.set_tree(tree); d.set_model(model); d.upload_paths(fp, cap)
    assert d.capacity() == cap
    d.reset()
    tri0 = d.tri_llh()
    J, D, nacc = d.run_mcmc(burn, batch, seed, average=False)
    assert np.array_equal(J, np.round(J))
    # dwell times add up: integer statistics lose at most one quantum (2^-k_b) per term
    np.testing.assert_allclose(D.reshape(B, 8).sum(1), batch * (n - 2) * tree.branches[1:], rtol=1e-11)
    assert 0.5 < nacc / float(batch * (n - 2)) <= 1.0
    tri = d.tri_llh()
    assert not np.array_equal(tri, tri0)
    d.reset()
    assert np.array_equal(tri, d.tri_llh())                     # reset() is idempotent on what the sweeps cached
    p = d.paths()
    es = (p.init.reshape(B, n) ^ (p.counts().reshape(B, n) & 1).astype(np.uint8))[leaves]
    assert np.array_equal(es, es0)                              # the leaves keep their data
    _check_paths_are_valid(p, tree, n)
    # the high slots are in use: accepted proposals live in buffer 1, whose last branches start beyond 2^32
    changed = (p.counts().reshape(B, n)[B - 1] != fp.counts().reshape(B, n)[B - 1]).sum()
    assert changed > 1000
    d.close()
    del es
    g = LocalGroup(0, 2, burn + batch)
    g.set_tree(tree); g.set_model(model); g.upload_paths(fp, cap)
    assert len(g.subs) == 2
    g.reset()
    Jg, Dg, ng = g.run_mcmc(burn, batch, seed, average=False)
    assert ng == nacc and np.array_equal(Jg, J) and np.array_equal(Dg, D)
    assert np.array_equal(g.tri_llh(), tri)
    assert orc.paths_equal(g.paths(), p)
    g.close()
