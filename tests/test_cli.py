"""The drop-in CLIs on the GPU: epievo_est_params_histories / epievo_sim_pairwise keep the
reference's flags and file formats, and their outputs equal an EM loop driven through the
CPU oracle's parallel rung (bit-identical param text and local_paths bytes)."""
import os
import subprocess

import numpy as np
import pytest

import orc
from common import simulate, TEST_PARAM_TEXT, TREE_NWK_TEXT
from epievo_amd import _build, host

pytestmark = pytest.mark.gpu
BIN = _build.BIN_DIR


def _write_expected(path, tree, tot_times, fp):
    """the EXPECTED local_paths bytes: written by the reference's own operator<<(Path) through
    oracle/_ref wherever that exists (this container and the GPU box), so that the CLI's writer
    is compared with the reference's and not with itself"""
    if orc.have_ref():
        from test_file_formats import ref_write_paths
        ref_write_paths(path, tree, tot_times, fp)
    else:
        host.write_paths(path, tree.node_names, tot_times, fp)


def _oracle_em(model, tree, fp, iters, burn, batch, seed, optimize):
    o = orc.Oracle(tree, model, fp, "B", cap=max(16, 2 * int(fp.counts().max()) + 8), seed=seed)
    branches = tree.branches.copy()
    text = ""
    for it in range(iters):
        o.set_model(model)
        o.reset()
        J, D, nacc, acc = o.run_mcmc(burn, batch, sweep_base=it * (burn + batch))
        model, branches, llh, text = host.m_step(model, branches, J, D, optimize_branches=optimize)
        o.scale_jump_times(branches)
    return model, branches, text, o.paths()


def _newick(tree):
    """Newick text of a pre-order array tree (all nodes named)"""
    def rec(i):
        kids, c = [], 1
        while c < tree.subtree_sizes[i]:
            kids.append(rec(i + c))
            c += tree.subtree_sizes[i + c]
        return ("(" + ",".join(kids) + ")" if kids else "") + "%s:%.17g" % (tree.node_names[i], tree.branches[i])
    return rec(0) + ";\n"


@pytest.mark.parametrize("cfg,optimize", [("tree", False), ("tree", True), ("bal16", True)])
def test_est_params_histories_matches_oracle_em(tmp_path, cfg, optimize):
    """BASELINE configs 3 and 5 in miniature (tree.nwk; a balanced 16-leaf tree with -b)"""
    model, tree, fp = simulate(cfg, 4000 if cfg == "tree" else 800, seed=21)
    d = str(tmp_path)
    open(d + "/p.param", "w").write(TEST_PARAM_TEXT)
    open(d + "/t.nwk", "w").write(TREE_NWK_TEXT if cfg == "tree" else _newick(tree))
    _write_expected(d + "/in.paths", tree, tree.branches, fp)     # the CLI reads a reference-written file
    cmd = [os.path.join(BIN, "epievo_est_params_histories"), "-i", "2", "-B", "3", "-L", "2", "-s", "77",
           "-o", d + "/out.paths", "-p", d + "/out.param", "-t", d + "/out.nwk", "-v"]
    if optimize:
        cmd.append("-b")
    cmd += [d + "/p.param", d + "/t.nwk", d + "/in.paths"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    lines = [l for l in r.stderr.split("\n") if l and l[0].isdigit()]
    assert len(lines) == 3 and len(lines[2].split("\t")) == 7      # itr T00 T11 b00 b11 acc llh
    m2, br, text, paths = _oracle_em(model, tree, fp, 2, 2, 3, 77, optimize)
    assert open(d + "/out.param").read() == text + "\n"
    _write_expected(d + "/exp.paths", tree, br, paths)
    assert open(d + "/out.paths", "rb").read() == open(d + "/exp.paths", "rb").read()
    if optimize:
        t2 = host.Tree.read(d + "/out.nwk")
        np.testing.assert_allclose(t2.branches, br, rtol=1e-5)       # default stream precision


def test_sim_pairwise_runs_and_keeps_end_states(tmp_path):
    model, tree, fp = simulate("pair", 2000, seed=3)
    d = str(tmp_path)
    open(d + "/p.param", "w").write(TEST_PARAM_TEXT)
    root = fp.init
    leaf = fp.init ^ (fp.counts() & 1).astype(np.uint8)
    with open(d + "/obs.states", "w") as f:
        f.write("#root\tleaf\n")
        for i in range(fp.n_sites):
            f.write("%d\t%d\t%d\n" % (i, root[i], leaf[i]))
    r = subprocess.run([os.path.join(BIN, "epievo_sim_pairwise"), "-L", "5", "-T", "1.0", "-s", "9",
                        "-o", d + "/out.paths", d + "/p.param", d + "/obs.states"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    out, names, tt = host.read_paths(d + "/out.paths")
    assert names == ["root", "leaf"] and tt[1] == 1.0
    assert np.array_equal(out.init, root)
    assert np.array_equal(out.init ^ (out.counts() & 1).astype(np.uint8), leaf)
    # bit-identical to the oracle: device init (initialize_paths_indep) + 5 sweeps
    t1 = host.Tree.single_branch(1.0)
    exp0 = orc.init_paths_indep("orc", 9, model.rates, root, leaf, 1.0, "B")
    o = orc.Oracle(t1, model, exp0, "B", cap=32, seed=9)
    o.reset()
    for w in range(5):
        o.sweep(w)
    assert orc.paths_equal(out, o.paths())
    # error behaviour: bad file -> message on stderr, EXIT_FAILURE (main's catch block)
    r = subprocess.run([os.path.join(BIN, "epievo_sim_pairwise"), "-o", d + "/x", d + "/nope", d + "/obs.states"],
                       capture_output=True, text=True)
    assert r.returncode == 1 and "Could not open file" in r.stderr
    # missing required -o -> message, EXIT_SUCCESS (as the reference does)
    r = subprocess.run([os.path.join(BIN, "epievo_sim_pairwise"), d + "/p.param", d + "/obs.states"],
                       capture_output=True, text=True)
    assert r.returncode == 0 and "required" in r.stderr


@pytest.mark.parametrize("optimize", [False, True])
def test_initialization_matches_oracle_pipeline(tmp_path, optimize):
    """epievo_initialization end to end (states at the leaves -> initial param + local_paths)
    against the same pipeline driven through the CPU oracle's parallel rung"""
    model, tree, fp = simulate("tree", 4000, seed=31)
    d = str(tmp_path)
    open(d + "/t.nwk", "w").write(TREE_NWK_TEXT)
    B, n, N = tree.n_nodes - 1, fp.n_sites, tree.n_nodes
    es = fp.init.reshape(B, n) ^ (fp.counts().reshape(B, n) & 1).astype(np.uint8)
    leaves = [i for i in range(N) if tree.subtree_sizes[i] == 1]
    with open(d + "/obs.states", "w") as f:
        f.write("#" + "\t".join(tree.node_names[i] for i in leaves) + "\n")
        for s in range(n):
            f.write("%d\t%s\n" % (s, "\t".join(str(es[i - 1, s]) for i in leaves)))
    cmd = [os.path.join(BIN, "epievo_initialization"), "-i", "4", "-B", "3", "-s", "5", "-p", d + "/out.param",
           "-o", d + "/out.paths", "-t", d + "/out.nwk", "-v"]
    if optimize:
        cmd.append("-b")
    r = subprocess.run(cmd + [d + "/t.nwk", d + "/obs.states"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr

    # the same pipeline: host heuristics + host M-steps + oracle rung B for the O(n) parts
    st = np.zeros((N, n), np.uint8)
    st[leaves] = es[[i - 1 for i in leaves]]
    p0 = host.initialize_paths_heuristic(5, tree, st)
    dummy = host.Model(np.ones(8), np.full(4, 0.5), np.zeros(4))
    o = orc.Oracle(tree, dummy, p0, "B", cap=32, seed=5)
    rates, br = np.zeros(2), tree.branches.copy()
    J, D = o.indep_suffstats()
    for it in range(4):
        rates, br_new = host.indep_m_step(rates, br, J, D, optimize_branches=optimize)
        if optimize:
            o.scale_jump_times(br_new)
            br = br_new
        J, D = o.indep_expectation(rates)
    Jt, Dt = np.zeros(B * 8), np.zeros(B * 8)
    for i in range(3):
        o.indep_update_paths(rates, 0xF0000000 + i)
        J1, D1 = o.suffstats()
        Jt += J1
        Dt += D1
    Jt /= 3
    Dt /= 3
    m2, br2, llh, text = host.m_step(host.model_from_indep_rates(rates), br, Jt, Dt, optimize_branches=optimize)
    o.scale_jump_times(br2)
    assert open(d + "/out.param").read() == text + "\n"
    _write_expected(d + "/exp.paths", tree, br2, o.paths())
    assert open(d + "/out.paths", "rb").read() == open(d + "/exp.paths", "rb").read()
    # the fitted model is sane: positive rates, leaves kept
    out, names, tt = host.read_paths(d + "/out.paths")
    es2 = out.init.reshape(B, n) ^ (out.counts().reshape(B, n) & 1).astype(np.uint8)
    assert np.array_equal(es2[[i - 1 for i in leaves]], es[[i - 1 for i in leaves]])


@pytest.mark.parametrize("optimize", [False, True])
def test_est_complete_matches_host_m_step(tmp_path, optimize):
    """epievo_est_complete: GPU sufficient statistics + host M-step on complete histories"""
    model, tree, fp = simulate("tree", 5000, seed=12)
    d = str(tmp_path)
    open(d + "/p.param", "w").write(TEST_PARAM_TEXT)
    open(d + "/t.nwk", "w").write(TREE_NWK_TEXT)
    host.write_paths(d + "/in.paths", tree.node_names, tree.branches, fp)
    cmd = [os.path.join(BIN, "epievo_est_complete"), "-o", d + "/out.param", "-t", d + "/out.nwk"]
    if optimize:
        cmd.append("-b")
    r = subprocess.run(cmd + [d + "/p.param", d + "/t.nwk", d + "/in.paths"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    o = orc.Oracle(tree, model, fp, "B", cap=16)
    if optimize:
        o.scale_jump_times(np.concatenate([[0.0], np.ones(tree.n_nodes - 1)]))
    J, D = o.suffstats()
    m2, br, llh, text = host.m_step(model, tree.branches, J, D, optimize_branches=optimize)
    assert open(d + "/out.param").read() == text + "\n"
    # the estimates are close to the truth the data were simulated from
    est = host.Model.read(d + "/out.param", scale=True)
    np.testing.assert_allclose(est.T, model.T, atol=0.1)   # n = 5000 sites: a noisy estimate
    if optimize:
        t2 = host.Tree.read(d + "/out.nwk")
        np.testing.assert_allclose(t2.branches, br, rtol=1e-5)


def _run_em(d, env_extra, tag, args, tree_text=TREE_NWK_TEXT):
    env = dict(os.environ, **env_extra)
    cmd = [os.path.join(BIN, "epievo_est_params_histories")] + args + \
          ["-o", d + "/%s.paths" % tag, "-p", d + "/%s.param" % tag, "-t", d + "/%s.nwk" % tag, "-v",
           d + "/p.param", d + "/t.nwk", d + "/in.paths"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    tsv = [l for l in r.stderr.split("\n") if l and l[0].isdigit()]
    layout = [l for l in r.stderr.split("\n") if l.startswith("[GPU LAYOUT")]
    return (open(d + "/%s.paths" % tag, "rb").read(), open(d + "/%s.param" % tag).read(), tsv,
            layout[0] if layout else "")


@pytest.mark.parametrize("cfg,optimize", [("tree", False), ("bal16", True)])
def test_em_cli_multi_gpu_rehearsal_is_byte_identical(tmp_path, cfg, optimize):
    """The C++ sharded EM driver (epv_sampler.cpp): device slots x contexts, halo exchange and
    the statistics all-gather through the exchange layer.  On the 1-GPU box the slots share
    device 0 (loopback transport) -- paths file, param file and the -v lines (acceptance rate,
    log-likelihood) must equal the one-context run byte for byte; one slot with EPV_FORCE_COMM
    drives the same path through a one-rank RCCL communicator (ncclCommInitAll, ncclAllGather)."""
    n = 14000 if cfg == "tree" else 9000
    model, tree, fp = simulate(cfg, n, seed=23)
    d = str(tmp_path)
    open(d + "/p.param", "w").write(TEST_PARAM_TEXT)
    open(d + "/t.nwk", "w").write(TREE_NWK_TEXT if cfg == "tree" else _newick(tree))
    host.write_paths(d + "/in.paths", tree.node_names, tree.branches, fp)
    args = ["-i", "3", "-B", "4", "-L", "3", "-s", "5"] + (["-b"] if optimize else [])
    one = _run_em(d, {"EPV_CONTEXTS_PER_GPU": "1", "EPV_DEVICES": "0"}, "one", args)
    assert "1 context" in one[3]
    # 3 slots x 2 contexts on device 0, rows of 4 blocks (cut points on multiples of 1024 sites)
    reh = _run_em(d, {"EPV_CONTEXTS_PER_GPU": "2", "EPV_DEVICES": "0,0,0", "EPV_ROW_BLOCKS": "4"}, "reh", args)
    assert "3 GPU slot(s)" in reh[3] and "loopback" in reh[3] and "= 6 parts" in reh[3]
    assert reh[:3] == one[:3]
    # the -g flag instead of the environment, 2 slots x 1 context
    flag = _run_em(d, {"EPV_CONTEXTS_PER_GPU": "1", "EPV_ROW_BLOCKS": "2"}, "flag", args + ["-g", "0,0"])
    assert "2 GPU slot(s)" in flag[3] and flag[:3] == one[:3]
    # RCCL itself, one rank: communicator set-up, all-gather, tear-down on hardware
    rccl = _run_em(d, {"EPV_CONTEXTS_PER_GPU": "2", "EPV_DEVICES": "0", "EPV_FORCE_COMM": "1", "EPV_ROW_BLOCKS": "4"},
                   "rccl", args)
    assert "RCCL" in rccl[3] and rccl[:3] == one[:3]


def test_em_cli_long_chain_two_contexts(tmp_path):
    """-L 50 -B 50 (100 sweeps between halo refreshes): the internal halo is sized from the chain
    length, so the default two-context mode runs it and equals the one-context run"""
    model, tree, fp = simulate("tree", 6000, seed=29)
    d = str(tmp_path)
    open(d + "/p.param", "w").write(TEST_PARAM_TEXT)
    open(d + "/t.nwk", "w").write(TREE_NWK_TEXT)
    host.write_paths(d + "/in.paths", tree.node_names, tree.branches, fp)
    args = ["-i", "1", "-B", "50", "-L", "50", "-s", "11"]
    one = _run_em(d, {"EPV_CONTEXTS_PER_GPU": "1"}, "one", args)
    two = _run_em(d, {"EPV_CONTEXTS_PER_GPU": "2"}, "two", args)
    assert "halo 768" in two[3] and "= 2 parts" in two[3]
    assert two[:3] == one[:3]


@pytest.mark.parametrize("optimize", [False, True])
def test_initialization_output_feeds_the_em_driver(tmp_path, optimize):
    """the README pipeline: epievo_initialization ... tree obs -> epievo_est_params_histories
    init.param tree init.paths.  Without -b the paths carry rate-scaled tot_times next to the
    unscaled tree; with -b the updated tree is printed at 6 significant digits.  Both are
    ordinary inputs (the reference compares nothing): the driver rescales and runs."""
    model, tree, fp = simulate("tree", 3000, seed=33)
    d = str(tmp_path)
    open(d + "/t.nwk", "w").write(TREE_NWK_TEXT)
    B, n, N = tree.n_nodes - 1, fp.n_sites, tree.n_nodes
    es = fp.init.reshape(B, n) ^ (fp.counts().reshape(B, n) & 1).astype(np.uint8)
    leaves = [i for i in range(N) if tree.subtree_sizes[i] == 1]
    with open(d + "/obs.states", "w") as f:
        f.write("#" + "\t".join(tree.node_names[i] for i in leaves) + "\n")
        for s in range(n):
            f.write("%d\t%s\n" % (s, "\t".join(str(es[i - 1, s]) for i in leaves)))
    cmd = [os.path.join(BIN, "epievo_initialization"), "-i", "3", "-B", "2", "-s", "5", "-p", d + "/init.param",
           "-o", d + "/init.paths", "-t", d + "/init.nwk"] + (["-b"] if optimize else [])
    r = subprocess.run(cmd + [d + "/t.nwk", d + "/obs.states"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    tree_in = d + "/init.nwk" if optimize else d + "/t.nwk"
    r = subprocess.run([os.path.join(BIN, "epievo_est_params_histories"), "-i", "2", "-B", "3", "-L", "2", "-s", "7",
                        "-o", d + "/out.paths", "-p", d + "/out.param", "-v", d + "/init.param", tree_in,
                        d + "/init.paths"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "RESCALING PATHS" in r.stderr
    out, names, tt = host.read_paths(d + "/out.paths")
    es2 = out.init.reshape(B, n) ^ (out.counts().reshape(B, n) & 1).astype(np.uint8)
    assert np.array_equal(es2[[i - 1 for i in leaves]], es[[i - 1 for i in leaves]])   # leaf data kept
    t_in = host.Tree.read(tree_in)
    for b in range(1, N):          # every jump inside its (final) branch
        jb = out.jumps[int(out.offsets[(b - 1) * n]):int(out.offsets[b * n])]
        assert jb.size == 0 or (jb.min() > 0 and jb.max() < tt[b])
    assert np.all(np.isfinite(host.Model.read(d + "/out.param", scale=True).rates))
    assert len(t_in.branches) == N


def test_em_cli_failure_in_a_later_iteration_leaves_a_complete_paths_file(tmp_path):
    """the paths file of iteration i is written by a background thread while iteration i + 1 runs;
    when that iteration fails (here: the param file cannot be rewritten) the program must report the
    error and exit with EXIT_FAILURE only AFTER the pending file is complete -- the synchronous
    reference would have left a whole file behind too"""
    model, tree, fp = simulate("tree", 60000, seed=5)
    d = str(tmp_path)
    open(d + "/p.param", "w").write(TEST_PARAM_TEXT)
    open(d + "/t.nwk", "w").write(TREE_NWK_TEXT)
    _write_expected(d + "/in.paths", tree, tree.branches, fp)
    os.mkdir(d + "/pdir")
    link = d + "/pdir/out.param"
    cmd = [os.path.join(BIN, "epievo_est_params_histories"), "-i", "3", "-B", "2", "-L", "1", "-s", "7",
           "-o", d + "/out.paths", "-p", link, d + "/p.param", d + "/t.nwk", d + "/in.paths"]
    env = dict(os.environ, EPV_TEST_FAIL_PARAM_AT="2")   # the 2nd rewrite of -p fails (test hook)
    r = subprocess.run(cmd, capture_output=True, text=True, env=env)
    assert r.returncode == 1, (r.returncode, r.stderr)
    assert "bad output param file" in r.stderr
    # the file of iteration 1 is whole: it parses, has every site of every node, and equals the
    # oracle's paths after one iteration
    out, names, tt = host.read_paths(d + "/out.paths")
    assert out.n_sites == fp.n_sites and out.n_nodes == tree.n_nodes
    m2, br, text, paths = _oracle_em(model, tree, fp, 1, 1, 2, 7, False)
    assert orc.paths_equal(out, paths)


def test_em_cli_paths_every_k_leaves_the_same_final_files(tmp_path):
    """-e k (extension): the paths file is rewritten every k-th iteration and after the last one
    instead of after every iteration -- the final paths, parameters and -v lines are the same"""
    model, tree, fp = simulate("tree", 3000, seed=13)
    d = str(tmp_path)
    open(d + "/p.param", "w").write(TEST_PARAM_TEXT)
    open(d + "/t.nwk", "w").write(TREE_NWK_TEXT)
    _write_expected(d + "/in.paths", tree, tree.branches, fp)
    outs = {}
    for tag, extra in (("every", []), ("third", ["-e", "3"]), ("last", ["-e", "100"])):
        r = subprocess.run([os.path.join(BIN, "epievo_est_params_histories"), "-i", "5", "-B", "2", "-L", "1", "-s", "3", "-b",
                            "-o", d + "/%s.paths" % tag, "-p", d + "/%s.param" % tag, "-t", d + "/%s.nwk" % tag, "-v"] + extra +
                           [d + "/p.param", d + "/t.nwk", d + "/in.paths"], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        outs[tag] = (open(d + "/%s.paths" % tag, "rb").read(), open(d + "/%s.param" % tag).read(),
                     open(d + "/%s.nwk" % tag).read(), [l for l in r.stderr.split("\n") if l and l[0].isdigit()])
    assert outs["every"] == outs["third"] == outs["last"]
