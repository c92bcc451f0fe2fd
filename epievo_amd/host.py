"""ctypes binding of libepv_host.so -- the C++ host library (model, M-step, file
formats, synthetic-input simulator).  No GPU code, no oracle code."""
import ctypes as C
import os

import numpy as np

from . import _build

_lib = None


def lib():
    global _lib
    if _lib is None:
        path = _build.HOST_SO
        if not os.path.exists(path):
            _build.build_host()
        L = C.CDLL(path)
        dp, u8p, u32p, u64p = (C.POINTER(C.c_double), C.POINTER(C.c_uint8),
                               C.POINTER(C.c_uint32), C.POINTER(C.c_uint64))
        L.epvh_last_error.restype = C.c_char_p
        L.epvh_model_read.argtypes = [C.c_char_p, C.c_int, dp, dp, dp]
        L.epvh_rate_scaling_factor.argtypes = [dp]
        L.epvh_rate_scaling_factor.restype = C.c_double
        L.epvh_m_step.argtypes = [C.c_int, C.c_int, dp, dp, dp, dp, dp, dp, dp, C.c_char_p, C.c_int]
        L.epvh_simulate.argtypes = [dp, dp, C.c_int, u32p, dp, C.c_uint64, C.c_uint64]
        L.epvh_simulate.restype = C.c_void_p
        L.epvh_paths_total_jumps.argtypes = [C.c_void_p]
        L.epvh_paths_total_jumps.restype = C.c_uint64
        L.epvh_paths_n_sites.argtypes = [C.c_void_p]
        L.epvh_paths_n_sites.restype = C.c_uint64
        L.epvh_paths_n_nodes.argtypes = [C.c_void_p]
        L.epvh_paths_copy.argtypes = [C.c_void_p, u8p, u64p, dp]
        L.epvh_paths_free.argtypes = [C.c_void_p]
        L.epvh_read_paths.argtypes = [C.c_char_p, C.c_char_p, C.c_int, dp, C.c_int]
        L.epvh_read_paths.restype = C.c_void_p
        L.epvh_write_paths.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_uint64, dp, u8p, u64p, dp]
        L.epvh_read_tree.argtypes = [C.c_char_p, C.c_int, u32p, u32p, dp, C.c_char_p, C.c_int]
        L.epvh_indep_m_step.argtypes = [C.c_int, C.c_int, dp, dp, dp, dp]
        L.epvh_model_from_indep_rates.argtypes = [dp, dp, dp, dp]
        L.epvh_initialize_paths_heuristic.argtypes = [C.c_uint64, C.c_int, u32p, u32p, dp, C.c_uint64, u8p]
        L.epvh_initialize_paths_heuristic.restype = C.c_void_p
        _lib = L
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


class Model:
    """rates[8], T[4] (row-major 2x2), baseline[4]"""

    def __init__(self, rates, T, baseline):
        self.rates = np.ascontiguousarray(rates, dtype=np.float64)
        self.T = np.ascontiguousarray(T, dtype=np.float64)
        self.baseline = np.ascontiguousarray(baseline, dtype=np.float64)

    @staticmethod
    def read(param_file, scale=True):
        rates, T, bl = np.zeros(8), np.zeros(4), np.zeros(4)
        if lib().epvh_model_read(param_file.encode(), int(scale), _p(rates, C.c_double),
                                 _p(T, C.c_double), _p(bl, C.c_double)):
            raise RuntimeError(lib().epvh_last_error().decode())
        return Model(rates, T, bl)


class Tree:
    def __init__(self, subtree_sizes, parent_ids, branches, node_names=None):
        self.subtree_sizes = np.ascontiguousarray(subtree_sizes, dtype=np.uint32)
        self.parent_ids = np.ascontiguousarray(parent_ids, dtype=np.uint32)
        self.branches = np.ascontiguousarray(branches, dtype=np.float64)
        self.n_nodes = len(self.subtree_sizes)
        self.node_names = node_names or ["node_%d" % i for i in range(self.n_nodes)]

    @staticmethod
    def read(tree_file, max_nodes=4096):
        st, pa, br = (np.zeros(max_nodes, np.uint32), np.zeros(max_nodes, np.uint32),
                      np.zeros(max_nodes))
        buf = C.create_string_buffer(64 * max_nodes)
        n = lib().epvh_read_tree(tree_file.encode(), max_nodes, _p(st, C.c_uint32),
                                 _p(pa, C.c_uint32), _p(br, C.c_double), buf, len(buf))
        if n < 0:
            raise RuntimeError(lib().epvh_last_error().decode())
        return Tree(st[:n].copy(), pa[:n].copy(), br[:n].copy(), buf.value.decode().split("\n"))

    @staticmethod
    def single_branch(evo_time):
        return Tree([2, 1], [0, 0], [0.0, evo_time], ["root", "leaf"])

    @staticmethod
    def balanced(n_leaves, branch_len):
        """synthetic balanced binary tree in pre-order (BASELINE config 5)"""
        sizes, parents, br = [], [], []

        def rec(leaves, parent):
            me = len(sizes)
            sizes.append(1)
            parents.append(parent)
            br.append(0.0 if parent < 0 else branch_len)
            if leaves > 1:
                rec(leaves // 2, me)
                rec(leaves - leaves // 2, me)
                sizes[me] = len(sizes) - me
        rec(n_leaves, -1)
        parents[0] = 0
        return Tree(sizes, parents, br)


class FlatPaths:
    """node-major flat local paths: entry (b-1)*n_sites + site, b = 1..n_nodes-1"""

    def __init__(self, n_sites, n_nodes, init, offsets, jumps):
        self.n_sites, self.n_nodes = int(n_sites), int(n_nodes)
        self.init = np.ascontiguousarray(init, dtype=np.uint8)
        self.offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        self.jumps = np.ascontiguousarray(jumps, dtype=np.float64)

    @staticmethod
    def _from_handle(h):
        L = lib()
        n, N, tot = L.epvh_paths_n_sites(h), L.epvh_paths_n_nodes(h), L.epvh_paths_total_jumps(h)
        init = np.zeros((N - 1) * n, np.uint8)
        off = np.zeros((N - 1) * n + 1, np.uint64)
        jumps = np.zeros(max(tot, 1), np.float64)
        L.epvh_paths_copy(h, _p(init, C.c_uint8), _p(off, C.c_uint64), _p(jumps, C.c_double))
        L.epvh_paths_free(h)
        return FlatPaths(n, N, init, off, jumps[:tot])

    def counts(self):
        return np.diff(self.offsets).astype(np.int64)

    def slice_sites(self, lo, hi):
        """sub-range of sites [lo, hi) as a new FlatPaths"""
        n, B = self.n_sites, self.n_nodes - 1
        cnt = self.counts().reshape(B, n)[:, lo:hi]
        init = self.init.reshape(B, n)[:, lo:hi]
        offs = self.offsets[:-1].reshape(B, n)[:, lo:hi]
        pieces = [self.jumps[int(offs[b, 0]):int(offs[b, -1] + cnt[b, -1])] for b in range(B)]
        new_off = np.zeros(B * (hi - lo) + 1, np.uint64)
        new_off[1:] = np.cumsum(cnt.reshape(-1))
        return FlatPaths(hi - lo, self.n_nodes, init.reshape(-1).copy(), new_off,
                         np.concatenate(pieces) if pieces else np.zeros(0))


def simulate(model, tree, n_sites, seed):
    h = lib().epvh_simulate(_p(model.rates, C.c_double), _p(model.T, C.c_double), tree.n_nodes,
                            _p(tree.parent_ids, C.c_uint32), _p(tree.branches, C.c_double),
                            int(n_sites), int(seed))
    if not h:
        raise RuntimeError(lib().epvh_last_error().decode())
    return FlatPaths._from_handle(h)


def read_paths(path_file, max_nodes=4096):
    buf = C.create_string_buffer(64 * max_nodes)
    tt = np.zeros(max_nodes)
    h = lib().epvh_read_paths(path_file.encode(), buf, len(buf), _p(tt, C.c_double), max_nodes)
    if not h:
        raise RuntimeError(lib().epvh_last_error().decode())
    fp = FlatPaths._from_handle(h)
    return fp, buf.value.decode().split("\n"), tt[:fp.n_nodes].copy()


def write_paths(path_file, node_names, tot_times, fp):
    tt = np.ascontiguousarray(tot_times, dtype=np.float64)
    jumps = fp.jumps if len(fp.jumps) else np.zeros(1)
    if lib().epvh_write_paths(path_file.encode(), "\n".join(node_names).encode(), fp.n_nodes,
                              fp.n_sites, _p(tt, C.c_double), _p(fp.init, C.c_uint8),
                              _p(fp.offsets, C.c_uint64), _p(jumps, C.c_double)):
        raise RuntimeError(lib().epvh_last_error().decode())


def m_step(model, tree_branches, J, D, optimize_branches=False):
    """The EM driver's M-step (est_params_histories.cpp:253-263).  Returns
    (new Model, new branches, llh, param-file text)."""
    n_nodes = len(tree_branches)
    rates, T, bl = model.rates.copy(), np.zeros(4), np.zeros(4)
    br = np.ascontiguousarray(tree_branches, dtype=np.float64).copy()
    J = np.ascontiguousarray(J, dtype=np.float64)
    D = np.ascontiguousarray(D, dtype=np.float64)
    llh = C.c_double(0.0)
    buf = C.create_string_buffer(512)
    if lib().epvh_m_step(int(optimize_branches), n_nodes, _p(J, C.c_double), _p(D, C.c_double),
                         _p(rates, C.c_double), _p(T, C.c_double), _p(bl, C.c_double),
                         _p(br, C.c_double), C.byref(llh), buf, len(buf)):
        raise RuntimeError(lib().epvh_last_error().decode())
    return Model(rates, T, bl), br, llh.value, buf.value.decode()


def indep_m_step(rates, branches, J, D, optimize_branches=False):
    """M-step of the site-independent model (IndepSite.cpp:299-360) -> (rates[2], branches)"""
    r = np.ascontiguousarray(rates, np.float64).copy()
    br = np.ascontiguousarray(branches, np.float64).copy()
    J = np.ascontiguousarray(J, np.float64)
    D = np.ascontiguousarray(D, np.float64)
    if lib().epvh_indep_m_step(int(optimize_branches), len(br), _p(J, C.c_double), _p(D, C.c_double),
                               _p(r, C.c_double), _p(br, C.c_double)):
        raise RuntimeError(lib().epvh_last_error().decode())
    return r, br


def model_from_indep_rates(rates2):
    r2 = np.ascontiguousarray(rates2, np.float64)
    rates, T, bl = np.zeros(8), np.zeros(4), np.zeros(4)
    if lib().epvh_model_from_indep_rates(_p(r2, C.c_double), _p(rates, C.c_double), _p(T, C.c_double),
                                         _p(bl, C.c_double)):
        raise RuntimeError(lib().epvh_last_error().decode())
    return Model(rates, T, bl)


def initialize_paths_heuristic(seed, tree, states):
    """epievo_initialization's heuristic start; states [n_nodes][n_sites] uint8 (leaves filled,
    internal nodes overwritten) -> FlatPaths"""
    st = np.ascontiguousarray(states, np.uint8)
    h = lib().epvh_initialize_paths_heuristic(int(seed), tree.n_nodes, _p(tree.subtree_sizes, C.c_uint32),
                                              _p(tree.parent_ids, C.c_uint32), _p(tree.branches, C.c_double),
                                              st.shape[1], _p(st, C.c_uint8))
    if not h:
        raise RuntimeError(lib().epvh_last_error().decode())
    states[...] = st
    return FlatPaths._from_handle(h)
