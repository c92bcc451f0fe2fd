"""The text formats of the drop-in surface pinned against the REFERENCE's
own writers and readers, not against round trips through the product's code:

  local_paths   rows = the linked operator<<(ostream&, const Path&) (Path.cpp:62-71) inside the
                three lines of glue of epievo_est_params_histories.cpp:56-75; reader read_paths
                (Path.cpp:123-148)
  global_jumps  write_root_to_pathfile_global / append_to_pathfile_global / read_pathfile_global
                (GlobalJump.cpp:71-140)
  states        read_states_file (epievo_utils.cpp:90-125) reads what the product writes
  Newick        PhyloTree operator>> + Newick_format (PhyloTree.cpp:110-122,189-203), TreeHelper
  param         EpiEvoModel::format_for_param_file (EpiEvoModel.cpp:192-200) -- see test_host_model.py

Two layers: live against oracle/_ref where it exists (this container and the GPU box, which
receives the built .so), and against tests/golden/text/* -- files WRITTEN BY THE REFERENCE
(tests/golden/make_golden.py) -- everywhere."""
import ctypes as C
import os

import numpy as np
import pytest

import orc
from common import GOLDEN, simulate, config, ref_test_model, TREE_NWK_TEXT, EXTRA_TREES
from epievo_amd import host

TEXT = os.path.join(GOLDEN, "text")
need_ref = pytest.mark.skipif(not orc.have_ref(), reason="oracle/_ref not built (no /root/reference here)")
dp, u8p, u32p, u64p = (C.POINTER(C.c_double), C.POINTER(C.c_uint8), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64))


def awkward_paths(cfg="tree", n=257, seed=5):
    """simulated paths plus hand-made awkward values: subnormal-ish, 17-digit, integer-valued and
    very large jump times, an empty path, a path with many jumps"""
    model, tree, fp = simulate(cfg, n, seed=seed)
    B = tree.n_nodes - 1
    cnt = fp.counts().astype(np.int64)
    jumps = [list(fp.jumps[int(fp.offsets[e]):int(fp.offsets[e + 1])]) for e in range(B * n)]
    jumps[0] = [1e-300, 4.9406564584124654e-324, 0.1, 1.0 / 3.0, 2.0 / 3.0, 0.30000000000000004]
    jumps[1] = []
    jumps[2] = [1.0, 2.0, 1e15, 1.2345678901234567e+22]
    jumps[3] = list(np.linspace(0.001, 0.002, 40))
    off = np.zeros(B * n + 1, np.uint64)
    off[1:] = np.cumsum([len(j) for j in jumps])
    flat = np.array([t for j in jumps for t in j], dtype=np.float64)
    return tree, host.FlatPaths(n, tree.n_nodes, fp.init.copy(), off, flat)


def ref_write_paths(path, tree, tot_times, fp):
    L = orc.ref_lib()
    L.ref_write_local_paths.argtypes = [C.c_char_p, C.c_int, C.c_uint64, C.c_char_p, dp, u8p, u64p, dp]
    tt = np.ascontiguousarray(tot_times, np.float64)
    j = fp.jumps if len(fp.jumps) else np.zeros(1)
    assert L.ref_write_local_paths(path.encode(), fp.n_nodes, fp.n_sites, "\n".join(tree.node_names).encode(),
                                   orc._p(tt, C.c_double), orc._p(fp.init, C.c_uint8),
                                   orc._p(fp.offsets, C.c_uint64), orc._p(j, C.c_double)) == 0


def ref_read_paths(path):
    L = orc.ref_lib()
    L.ref_read_local_paths.argtypes = [C.c_char_p, C.POINTER(C.c_int), u64p, u64p]
    L.ref_local_paths_copy.argtypes = [u8p, dp, u64p, dp, C.c_char_p, C.c_uint64]
    nn, ns, tot = C.c_int(0), C.c_uint64(0), C.c_uint64(0)
    assert L.ref_read_local_paths(path.encode(), C.byref(nn), C.byref(ns), C.byref(tot)) == 0
    B, n = nn.value - 1, ns.value
    init, tt = np.zeros(B * n, np.uint8), np.zeros(nn.value)
    off, jumps = np.zeros(B * n + 1, np.uint64), np.zeros(max(tot.value, 1))
    names = C.create_string_buffer(1 << 16)
    assert L.ref_local_paths_copy(orc._p(init, C.c_uint8), orc._p(tt, C.c_double), orc._p(off, C.c_uint64),
                                  orc._p(jumps, C.c_double), names, len(names)) == 0
    return host.FlatPaths(n, nn.value, init, off, jumps[:tot.value]), names.value.decode().split("\n"), tt


@need_ref
@pytest.mark.parametrize("cfg,n", [("tree", 257), ("pair", 100), ("cat6", 50)])
def test_local_paths_writer_and_reader_against_the_reference(tmp_path, cfg, n):
    tree, fp = awkward_paths(cfg, n)
    tt = tree.branches * 1.0000000000000002 + 1e-17        # values that need all 17 digits
    d = str(tmp_path)
    ref_write_paths(d + "/ref.paths", tree, tt, fp)
    host.write_paths(d + "/own.paths", tree.node_names, tt, fp)
    assert open(d + "/own.paths", "rb").read() == open(d + "/ref.paths", "rb").read()
    # each reader on the OTHER side's file: the same values, bit for bit
    own, names, t1 = host.read_paths(d + "/ref.paths")
    assert names == tree.node_names and np.array_equal(t1[1:], tt[1:]) and orc.paths_equal(own, fp)
    ref, names2, t2 = ref_read_paths(d + "/own.paths")
    assert names2 == tree.node_names and np.array_equal(t2[1:], tt[1:]) and orc.paths_equal(ref, fp)


def test_local_paths_against_reference_written_fixture(tmp_path):
    """tests/golden/text/tree_n40.paths was written by the reference's operator<<: the product's
    reader recovers the stored values, its writer reproduces the bytes"""
    z = np.load(os.path.join(TEXT, "tree_n40.npz"))
    fp = host.FlatPaths(int(z["n_sites"]), int(z["n_nodes"]), z["init"], z["offsets"], z["jumps"])
    own, names, tt = host.read_paths(os.path.join(TEXT, "tree_n40.paths"))
    assert orc.paths_equal(own, fp) and np.array_equal(tt[1:], z["tot_times"][1:])
    assert names == [str(x) for x in z["names"]]
    host.write_paths(str(tmp_path / "w.paths"), names, z["tot_times"], fp)
    assert open(str(tmp_path / "w.paths"), "rb").read() == open(os.path.join(TEXT, "tree_n40.paths"), "rb").read()


def _fwd(seed, tree, n):
    import test_forward_sim as tf
    return tf._run(host.lib().epvh_forward_sim, seed, ref_test_model(), tree, n)


def own_write_global(path, tree, n, seqs, off, tt, pp):
    L = host.lib()
    L.epvh_write_global_jumps.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_uint64, u8p, u64p, dp, u64p]
    root = np.ascontiguousarray(seqs[:n])
    assert L.epvh_write_global_jumps(path.encode(), tree.n_nodes, "\n".join(tree.node_names).encode(), n,
                                     orc._p(root, C.c_uint8), orc._p(off, C.c_uint64), orc._p(tt, C.c_double),
                                     orc._p(pp, C.c_uint64)) == 0


def own_read_global(path):
    L = host.lib()
    L.epvh_read_global_jumps.argtypes = [C.c_char_p, C.POINTER(C.c_int), u64p, u64p]
    L.epvh_read_global_jumps.restype = C.c_void_p
    L.epvh_global_jumps_copy.argtypes = [C.c_void_p, u8p, u64p, dp, u64p, C.c_char_p, C.c_int]
    nn, ns, tot = C.c_int(0), C.c_uint64(0), C.c_uint64(0)
    h = L.epvh_read_global_jumps(path.encode(), C.byref(nn), C.byref(ns), C.byref(tot))
    assert h, L.epvh_last_error()
    root, off = np.zeros(ns.value, np.uint8), np.zeros(nn.value + 1, np.uint64)
    tt, pp = np.zeros(max(tot.value, 1)), np.zeros(max(tot.value, 1), np.uint64)
    names = C.create_string_buffer(1 << 16)
    L.epvh_global_jumps_copy(h, orc._p(root, C.c_uint8), orc._p(off, C.c_uint64), orc._p(tt, C.c_double),
                             orc._p(pp, C.c_uint64), names, len(names))
    return root, off, tt[:tot.value], pp[:tot.value], names.value.decode().split("\n")


@need_ref
def test_global_jumps_writer_and_reader_against_the_reference(tmp_path):
    tree, n = config("tree"), 300
    seqs, off, tt, pp = _fwd(11, tree, n)
    d = str(tmp_path)
    own_write_global(d + "/own.jumps", tree, n, seqs, off, tt, pp)
    L = orc.ref_lib()
    L.ref_write_global_jumps.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_uint64, u8p, u64p, dp, u64p]
    root = np.ascontiguousarray(seqs[:n])
    assert L.ref_write_global_jumps((d + "/ref.jumps").encode(), tree.n_nodes, "\n".join(tree.node_names).encode(),
                                    n, orc._p(root, C.c_uint8), orc._p(off, C.c_uint64), orc._p(tt, C.c_double),
                                    orc._p(pp, C.c_uint64)) == 0
    assert open(d + "/own.jumps", "rb").read() == open(d + "/ref.jumps", "rb").read()
    r2, o2, t2, p2, names = own_read_global(d + "/ref.jumps")
    assert np.array_equal(r2, root) and np.array_equal(o2, off) and np.array_equal(t2, tt) and np.array_equal(p2, pp)
    assert names == tree.node_names
    # the reference's reader on the product's file
    L.ref_read_global_jumps.argtypes = [C.c_char_p, C.POINTER(C.c_int), u64p, u64p]
    L.ref_global_jumps_copy.argtypes = [u8p, u64p, dp, u64p, C.c_char_p, C.c_uint64]
    nn, ns, tot = C.c_int(0), C.c_uint64(0), C.c_uint64(0)
    assert L.ref_read_global_jumps((d + "/own.jumps").encode(), C.byref(nn), C.byref(ns), C.byref(tot)) == 0
    assert (nn.value, ns.value, tot.value) == (tree.n_nodes, n, len(tt))
    r3, o3 = np.zeros(n, np.uint8), np.zeros(nn.value + 1, np.uint64)
    t3, p3 = np.zeros(max(tot.value, 1)), np.zeros(max(tot.value, 1), np.uint64)
    nb = C.create_string_buffer(1 << 16)
    assert L.ref_global_jumps_copy(orc._p(r3, C.c_uint8), orc._p(o3, C.c_uint64), orc._p(t3, C.c_double),
                                   orc._p(p3, C.c_uint64), nb, len(nb)) == 0
    assert np.array_equal(r3, root) and np.array_equal(o3, off) and np.array_equal(t3[:tot.value], tt)
    assert np.array_equal(p3[:tot.value], pp) and nb.value.decode().split("\n") == tree.node_names


def test_global_jumps_against_reference_written_fixture(tmp_path):
    z = np.load(os.path.join(TEXT, "tree_n60_global.npz"))
    r, o, t, p, names = own_read_global(os.path.join(TEXT, "tree_n60.jumps"))
    assert np.array_equal(r, z["root"]) and np.array_equal(o, z["off"]) and np.array_equal(t, z["t"])
    assert np.array_equal(p, z["p"]) and names == [str(x) for x in z["names"]]
    tree = config("tree")
    own_write_global(str(tmp_path / "w.jumps"), tree, len(r), r, o, t, p)
    assert open(str(tmp_path / "w.jumps"), "rb").read() == open(os.path.join(TEXT, "tree_n60.jumps"), "rb").read()


@need_ref
@pytest.mark.parametrize("only_leaves", [0, 1])
def test_states_file_is_read_by_the_reference(tmp_path, only_leaves):
    tree, n = config("tree"), 123
    seqs = _fwd(3, tree, n)[0].reshape(tree.n_nodes, n)
    L = host.lib()
    L.epvh_write_states.argtypes = [C.c_char_p, C.c_int, C.c_int, u32p, C.c_char_p, C.c_uint64, u8p]
    path = str(tmp_path / "x.states")
    flat = np.ascontiguousarray(seqs.reshape(-1))
    assert L.epvh_write_states(path.encode(), only_leaves, tree.n_nodes, orc._p(tree.subtree_sizes, C.c_uint32),
                               "\n".join(tree.node_names).encode(), n, orc._p(flat, C.c_uint8)) == 0
    R = orc.ref_lib()
    R.ref_read_states.argtypes = [C.c_char_p, C.POINTER(C.c_int), u64p]
    R.ref_states_copy.argtypes = [u8p, C.c_char_p, C.c_uint64]
    ns, nsites = C.c_int(0), C.c_uint64(0)
    assert R.ref_read_states(path.encode(), C.byref(ns), C.byref(nsites)) == 0
    keep = [i for i in range(tree.n_nodes) if not only_leaves or tree.subtree_sizes[i] == 1]
    assert (ns.value, nsites.value) == (len(keep), n)
    out = np.zeros(len(keep) * n, np.uint8)
    nb = C.create_string_buffer(1 << 12)
    assert R.ref_states_copy(orc._p(out, C.c_uint8), nb, len(nb)) == 0
    assert np.array_equal(out.reshape(len(keep), n), seqs[keep])
    assert nb.value.decode().split("\n") == [tree.node_names[i] for i in keep]


NEWICKS = [TREE_NWK_TEXT] + list(EXTRA_TREES.values()) + [
    "((A:0.5,B:1e-3):0.25,(C:2,D:0.125):1.5);\n",                 # unnamed internal nodes, no root length
    "(A:0.1,(B:0.2,(C:0.3,(D:0.4,E:0.5)I1:0.6)I2:0.7)I3:0.8)R:0.0;\n",
]


@need_ref
@pytest.mark.parametrize("text", NEWICKS)
def test_newick_parse_and_print_against_the_reference(tmp_path, text):
    R = orc.ref_lib()
    R.ref_newick_roundtrip.argtypes = [C.c_char_p, C.c_char_p, C.c_uint64, C.POINTER(C.c_int), u32p, u32p, dp,
                                       C.c_char_p, C.c_uint64]
    out, names = C.create_string_buffer(1 << 14), C.create_string_buffer(1 << 14)
    nn = C.c_int(0)
    pa, st, br = np.zeros(256, np.uint32), np.zeros(256, np.uint32), np.zeros(256)
    assert R.ref_newick_roundtrip(text.encode(), out, len(out), C.byref(nn), orc._p(pa, C.c_uint32),
                                  orc._p(st, C.c_uint32), orc._p(br, C.c_double), names, len(names)) == 0
    p = str(tmp_path / "t.nwk")
    open(p, "w").write(text)
    t = host.Tree.read(p)
    N = nn.value
    assert t.n_nodes == N
    assert np.array_equal(t.parent_ids, pa[:N]) and np.array_equal(t.subtree_sizes, st[:N])
    assert np.array_equal(t.branches, br[:N])
    assert t.node_names == names.value.decode().split("\n")
    L = host.lib()
    L.epvh_tree_newick.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
    own = C.create_string_buffer(1 << 14)
    assert L.epvh_tree_newick(p.encode(), own, len(own)) == 0
    assert own.value == out.value


def test_local_paths_io_on_several_threads_is_the_sequential_io(tmp_path, monkeypatch):
    """files of 65536 rows or more are parsed / formatted in pieces on all host cores (pread / pwrite at
    the pieces' offsets): the same bytes and the same arrays as one thread, and as the reference"""
    model, tree, fp = simulate("tree", 70001, seed=8)
    d = str(tmp_path)
    monkeypatch.setenv("EPV_IO_THREADS", "1")
    host.write_paths(d + "/one.paths", tree.node_names, tree.branches, fp)
    one, names1, tt1 = host.read_paths(d + "/one.paths")
    for threads in ("3", "8"):
        monkeypatch.setenv("EPV_IO_THREADS", threads)
        host.write_paths(d + "/many.paths", tree.node_names, tree.branches, fp)
        assert open(d + "/many.paths", "rb").read() == open(d + "/one.paths", "rb").read()
        many, names, tt = host.read_paths(d + "/one.paths")
        assert names == names1 == tree.node_names and np.array_equal(tt, tt1)
        assert orc.paths_equal(many, one) and orc.paths_equal(many, fp)
    if orc.have_ref():
        ref_write_paths(d + "/ref.paths", tree, tree.branches, fp)
        assert open(d + "/ref.paths", "rb").read() == open(d + "/one.paths", "rb").read()
    # rows that take the reader's slow path inside a piece: blank lines, spaces, a row per node whose
    # tot_time is spelled differently (same value)
    txt = open(d + "/one.paths").read().split("\n")
    k = len(txt) // 2 + 7
    assert not txt[k].startswith("NODE")
    site, st, ttok, rest = txt[k].split("\t", 3)
    txt[k] = "%s %s\t%s0\t%s" % (site, st, ttok, rest) if "." in ttok else txt[k]
    txt.insert(k, "")
    open(d + "/odd.paths", "w").write("\n".join(txt))
    odd, _, _ = host.read_paths(d + "/odd.paths")
    assert orc.paths_equal(odd, fp)
