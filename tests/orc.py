"""ctypes bindings of the TEST-ONLY parity oracle (oracle/liborc.so) and, where it has
been built (this container; the prebuilt .so travels to the GPU box), of the linked
reference (oracle/_ref/libepievo_ref.so).  Only tests/, smoke() and bench.py's
cpu_baseline leg import this module."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORC_SO = os.path.join(ROOT, "oracle", "liborc.so")
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libepievo_ref.so")

RNG_MT, RNG_PHILOX = 0, 1
MATH_LIBM, MATH_EPV = 0, 1
SAMPLER_FORWARD, SAMPLER_NIELSEN = 0, 1
SCHED_SEQ, SCHED_3COLOUR = 0, 1
REDUCE_SEQ, REDUCE_EXACT = 0, 1

dp, u8p, u32p, u64p = (C.POINTER(C.c_double), C.POINTER(C.c_uint8), C.POINTER(C.c_uint32),
                       C.POINTER(C.c_uint64))


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


_orc = None
_ref = None


def orc_lib():
    global _orc
    if _orc is None:
        src = [os.path.join(ROOT, "oracle", f) for f in ("epv_oracle.c", "orc_math.h", "orc_rng.h")]
        if (not os.path.exists(ORC_SO)) or any(
                os.path.exists(s) and os.path.getmtime(s) > os.path.getmtime(ORC_SO) for s in src):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")],
                                  stdout=subprocess.DEVNULL)
        L = C.CDLL(ORC_SO)
        L.orc_create.argtypes = [C.c_uint64, C.c_int, u32p, u32p, dp, dp, dp, u8p, u64p, dp]
        L.orc_create.restype = C.c_void_p
        L.orc_destroy.argtypes = [C.c_void_p]
        L.orc_set_modes.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint32]
        L.orc_seed_mt.argtypes = [C.c_void_p, C.c_uint64]
        L.orc_seed_philox.argtypes = [C.c_void_p, C.c_uint64]
        L.orc_set_model.argtypes = [C.c_void_p, dp, dp]
        L.orc_reset.argtypes = [C.c_void_p]
        L.orc_get_tri_llh.argtypes = [C.c_void_p, dp]
        L.orc_sweep.argtypes = [C.c_void_p, C.c_uint32]
        L.orc_sweep.restype = C.c_uint64
        L.orc_mh_site.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32]
        L.orc_suffstats.argtypes = [C.c_void_p, dp, dp]
        L.orc_suffstats_range.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, dp, dp]
        L.orc_suffstats_rows.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.POINTER(C.c_int64)]
        L.orc_stat_scales.argtypes = [C.c_void_p, dp]
        L.orc_set_proposal_mode.argtypes = [C.c_void_p, C.c_int]
        L.orc_set_sample_root.argtypes = [C.c_void_p, C.c_int]
        L.orc_set_sampler.argtypes = [C.c_void_p, C.c_int]
        L.orc_get_max_qdiff.argtypes = [C.c_void_p]
        L.orc_get_max_qdiff.restype = C.c_double
        L.orc_set_shard.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64]
        L.orc_sweep_phase.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32] + [C.c_uint64] * 4
        L.orc_sweep_phase.restype = C.c_uint64
        L.orc_set_site.argtypes = [C.c_void_p, C.c_uint64, u8p, u32p, dp]
        L.orc_set_tri.argtypes = [C.c_void_p, C.c_uint64, C.c_double]
        L.orc_run_mcmc.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32, dp, dp, u64p, dp]
        L.orc_scale_jump_times.argtypes = [C.c_void_p, dp]
        L.orc_total_jumps.argtypes = [C.c_void_p]
        L.orc_total_jumps.restype = C.c_uint64
        L.orc_get_paths.argtypes = [C.c_void_p, u8p, u64p, dp]
        L.orc_get_counters.argtypes = [C.c_void_p, u64p]
        L.orc_kat_trans_prob_mat.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double, dp]
        L.orc_kat_get_trans_prob.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int]
        L.orc_kat_get_trans_prob.restype = C.c_double
        L.orc_kat_segments.argtypes = [dp, C.c_int, C.c_uint32, dp, C.c_int, C.c_uint32, dp,
                                       C.c_double, dp, dp, u64p, u64p, dp]
        L.orc_kat_suffstats.argtypes = [C.c_int, C.c_uint32, dp, C.c_int, C.c_uint32, dp, C.c_int,
                                        C.c_uint32, dp, C.c_double, dp, dp]
        L.orc_kat_end_cond_means.argtypes = [C.c_int, C.c_int, C.c_uint64, C.c_double, C.c_double, C.c_int, C.c_int, C.c_double, C.c_uint64, dp]
        L.orc_init_paths_indep.argtypes = [C.c_int, C.c_int, C.c_uint64, dp, C.c_uint64, u8p, u8p, C.c_double, u8p, u64p, dp, C.c_uint64]
        L.orc_init_paths_indep.restype = C.c_uint64
        L.orc_indep_expectation.argtypes = [C.c_void_p, dp, dp, dp]
        L.orc_indep_suffstats.argtypes = [C.c_void_p, dp, dp]
        L.orc_indep_update_paths.argtypes = [C.c_void_p, dp, C.c_uint32]
        L.orc_exact_posterior.argtypes = [dp, C.c_uint64, u8p, u8p, C.c_double, C.c_uint64, C.c_uint64, C.c_uint64, dp, dp, dp, dp]
        L.orc_exact_posterior.restype = C.c_uint64
        L.orc_kat_end_cond_paths.restype = C.c_uint64
        L.orc_kat_end_cond_paths.argtypes = [C.c_int, C.c_int, C.c_uint64, C.c_double, C.c_double, C.c_int, C.c_int,
                                             C.c_double, C.c_uint64, C.POINTER(C.c_uint32), dp, C.c_uint64]
        L.orc_set_sampler.argtypes = [C.c_void_p, C.c_int]
        L.orc_forward_thinning.argtypes = [dp, dp, C.c_int, u32p, dp, C.c_uint64, C.c_uint64, u8p, u8p, u64p, dp,
                                           C.c_uint64, u8p]
        L.orc_forward_thinning.restype = C.c_uint64
        L.orc_kat_exp.argtypes = [C.c_double]
        L.orc_kat_exp.restype = C.c_double
        L.orc_kat_log.argtypes = [C.c_double]
        L.orc_kat_log.restype = C.c_double
        L.orc_kat_exp_log_array.argtypes = [dp, C.c_uint64, dp, dp]
        L.orc_kat_philox.argtypes = [u32p, u32p, u32p]
        L.orc_kat_keyed_block.argtypes = [C.c_uint64] + [C.c_uint32] * 6 + [dp]
        L.orc_kat_mt_canonical.argtypes = [C.c_uint64, C.c_uint64, dp]
        _orc = L
    return _orc


def have_ref():
    return os.path.exists(REF_SO)


def ref_lib():
    global _ref
    if _ref is None:
        L = C.CDLL(REF_SO)
        L.ref_create.argtypes = [C.c_uint64, C.c_int, u32p, u32p, dp, dp, dp, u8p, u64p, dp]
        L.ref_create.restype = C.c_void_p
        L.ref_destroy.argtypes = [C.c_void_p]
        L.ref_set_model.argtypes = [C.c_void_p, dp, dp]
        L.ref_seed.argtypes = [C.c_void_p, C.c_uint64]
        L.ref_reset.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64]
        L.ref_tri_llh.argtypes = [C.c_void_p, dp]
        L.ref_sweeps.argtypes = [C.c_void_p, C.c_uint64]
        L.ref_sweeps.restype = C.c_uint64
        L.ref_mh_site.argtypes = [C.c_void_p, C.c_uint64]
        L.ref_run_mcmc.argtypes = [C.c_void_p, dp, dp, dp]
        L.ref_suffstats.argtypes = [C.c_void_p, dp, dp]
        L.ref_scale_jump_times.argtypes = [C.c_void_p, dp]
        L.ref_total_jumps.argtypes = [C.c_void_p]
        L.ref_total_jumps.restype = C.c_uint64
        L.ref_get_paths.argtypes = [C.c_void_p, u8p, u64p, dp]
        L.ref_read_model.argtypes = [C.c_char_p, C.c_int, dp, dp, dp]
        L.ref_m_step.argtypes = [C.c_int, C.c_int, dp, dp, dp, dp, dp, dp, C.c_char_p, C.c_int]
        L.ref_m_step.restype = C.c_double
        L.ref_kat_segments.argtypes = [dp, C.c_int, C.c_uint32, dp, C.c_int, C.c_uint32, dp,
                                       C.c_double, dp, dp, u64p, u64p, dp]
        L.ref_kat_trans_prob_mat.argtypes = [C.c_double, C.c_double, C.c_double, dp]
        L.ref_kat_get_trans_prob.argtypes = [C.c_double, C.c_double, C.c_double, C.c_int, C.c_int]
        L.ref_kat_get_trans_prob.restype = C.c_double
        L.ref_kat_suffstats.argtypes = [C.c_int, C.c_uint32, dp, C.c_int, C.c_uint32, dp, C.c_int,
                                        C.c_uint32, dp, C.c_double, dp, dp]
        L.ref_kat_expectations.argtypes = [C.c_double, C.c_double, C.c_double, dp]
        L.ref_kat_mt_canonical.argtypes = [C.c_uint64, C.c_uint64, dp]
        if hasattr(L, 'ref_set_sample_root'):
            L.ref_set_sample_root.argtypes = [C.c_int]
        if hasattr(L, 'ref_indep_expectation'):
            L.ref_indep_expectation.argtypes = [C.c_void_p, dp, dp, dp]
            L.ref_indep_suffstats.argtypes = [C.c_void_p, dp, dp]
            L.ref_indep_update_paths.argtypes = [C.c_void_p, dp]
            L.ref_indep_m_step.argtypes = [C.c_void_p, C.c_int, dp, dp, dp, dp]
        if hasattr(L, 'ref_init_paths_indep'):
            L.ref_init_paths_indep.argtypes = [C.c_uint64, dp, C.c_uint64, u8p, u8p, C.c_double, u8p, u64p, dp, C.c_uint64]
            L.ref_init_paths_indep.restype = C.c_uint64
        _ref = L
    return _ref


class _Engine:
    """Common face of the oracle and the linked reference over flat paths."""

    def __init__(self, lib, prefix, tree, model, fp):
        self.L, self.px = lib, prefix
        self.n_sites, self.n_nodes = fp.n_sites, tree.n_nodes
        self.B = tree.n_nodes - 1
        jumps = fp.jumps if len(fp.jumps) else np.zeros(1)
        self.h = getattr(lib, prefix + "_create")(
            fp.n_sites, tree.n_nodes, _p(tree.parent_ids, C.c_uint32),
            _p(tree.subtree_sizes, C.c_uint32), _p(tree.branches, C.c_double),
            _p(model.rates, C.c_double), _p(model.T, C.c_double), _p(fp.init, C.c_uint8),
            _p(fp.offsets, C.c_uint64), _p(jumps, C.c_double))
        assert self.h

    def close(self):
        if self.h:
            getattr(self.L, self.px + "_destroy")(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def set_model(self, model):
        getattr(self.L, self.px + "_set_model")(self.h, _p(model.rates, C.c_double),
                                                _p(model.T, C.c_double))

    def suffstats(self):
        J, D = np.zeros(self.B * 8), np.zeros(self.B * 8)
        getattr(self.L, self.px + "_suffstats")(self.h, _p(J, C.c_double), _p(D, C.c_double))
        return J, D

    def scale_jump_times(self, new_branches):
        nb = np.ascontiguousarray(new_branches, dtype=np.float64)
        getattr(self.L, self.px + "_scale_jump_times")(self.h, _p(nb, C.c_double))

    def paths(self):
        from epievo_amd.host import FlatPaths
        tot = getattr(self.L, self.px + "_total_jumps")(self.h)
        init = np.zeros(self.B * self.n_sites, np.uint8)
        off = np.zeros(self.B * self.n_sites + 1, np.uint64)
        jumps = np.zeros(max(tot, 1))
        getattr(self.L, self.px + "_get_paths")(self.h, _p(init, C.c_uint8), _p(off, C.c_uint64),
                                                _p(jumps, C.c_double))
        return FlatPaths(self.n_sites, self.n_nodes, init, off, jumps[:tot])


def _indep_methods(cls, px):
    def indep_expectation(self, rates):
        r = np.ascontiguousarray(rates, np.float64)
        J, D = np.zeros(self.B * 2), np.zeros(self.B * 2)
        getattr(self.L, px + "_indep_expectation")(self.h, _p(r, C.c_double), _p(J, C.c_double), _p(D, C.c_double))
        return J, D

    def indep_suffstats(self):
        J, D = np.zeros(self.B * 2), np.zeros(self.B * 2)
        getattr(self.L, px + "_indep_suffstats")(self.h, _p(J, C.c_double), _p(D, C.c_double))
        return J, D
    cls.indep_expectation, cls.indep_suffstats = indep_expectation, indep_suffstats
    return cls


class Oracle(_Engine):
    def __init__(self, tree, model, fp, rung="A", cap=0, seed=0):
        super().__init__(orc_lib(), "orc", tree, model, fp)
        self.set_rung(rung, cap)
        self.seed(seed)

    def set_rung(self, rung, cap=0):
        if rung == "A":
            self.L.orc_set_modes(self.h, RNG_MT, MATH_LIBM, SCHED_SEQ, REDUCE_SEQ, 0)
        elif rung == "B":
            self.L.orc_set_modes(self.h, RNG_PHILOX, MATH_EPV, SCHED_3COLOUR, REDUCE_EXACT, cap)
        else:
            self.L.orc_set_modes(self.h, *rung, cap)
        self.rung = rung

    def seed(self, seed):
        self.L.orc_seed_mt(self.h, seed)
        self.L.orc_seed_philox(self.h, seed)

    def set_proposal_mode(self, reference):
        """True: q(old)/q(new) by the reference's sums; False: the exact (telescoped) 0"""
        self.L.orc_set_proposal_mode(self.h, 0 if reference else 1)

    def set_sample_root(self, on):
        """SingleSiteSampler::SAMPLE_ROOT (forces the reference's proposal-ratio arithmetic)"""
        self.L.orc_set_sample_root(self.h, 1 if on else 0)

    def set_sampler(self, forward):
        """True: forward rejection for every segment; False: Nielsen for state changes"""
        self.L.orc_set_sampler(self.h, 0 if forward else 1)

    def max_qdiff(self):
        return float(self.L.orc_get_max_qdiff(self.h))

    def reset(self):
        self.L.orc_reset(self.h)

    def tri_llh(self):
        out = np.zeros(self.n_sites)
        self.L.orc_get_tri_llh(self.h, _p(out, C.c_double))
        return out

    def sweep(self, sweep_index=0):
        return int(self.L.orc_sweep(self.h, sweep_index))

    def mh_site(self, site, sweep_index=0):
        return int(self.L.orc_mh_site(self.h, site, sweep_index))

    def run_mcmc(self, burn_in, batch, sweep_base=0):
        J, D = np.zeros(self.B * 8), np.zeros(self.B * 8)
        nacc, acc = C.c_uint64(0), C.c_double(0)
        self.L.orc_run_mcmc(self.h, burn_in, batch, sweep_base, _p(J, C.c_double),
                            _p(D, C.c_double), C.byref(nacc), C.byref(acc))
        return J, D, int(nacc.value), acc.value

    def indep_update_paths(self, rates, sweep=0):
        r = np.ascontiguousarray(rates, np.float64)
        self.L.orc_indep_update_paths(self.h, _p(r, C.c_double), sweep)

    def counters(self):
        out = np.zeros(4, np.uint64)
        self.L.orc_get_counters(self.h, _p(out, C.c_uint64))
        return dict(zip(("overflow", "trials", "draws", "segments"), (int(x) for x in out)))


class Reference(_Engine):
    def __init__(self, tree, model, fp, seed=0):
        super().__init__(ref_lib(), "ref", tree, model, fp)
        self.L.ref_seed(self.h, seed)

    def seed(self, seed):
        self.L.ref_seed(self.h, seed)

    def reset(self, burn_in=0, batch=1):
        self.L.ref_reset(self.h, burn_in, batch)

    def tri_llh(self):
        out = np.zeros(self.n_sites)
        self.L.ref_tri_llh(self.h, _p(out, C.c_double))
        return out

    def sweeps(self, n):
        return int(self.L.ref_sweeps(self.h, n))

    def mh_site(self, site):
        return int(self.L.ref_mh_site(self.h, site))

    def run_mcmc(self):
        J, D = np.zeros(self.B * 8), np.zeros(self.B * 8)
        acc = C.c_double(0)
        self.L.ref_run_mcmc(self.h, _p(J, C.c_double), _p(D, C.c_double), C.byref(acc))
        return J, D, acc.value


_indep_methods(Oracle, "orc")
_indep_methods(Reference, "ref")


def _ref_indep_update(self, rates):
    r = np.ascontiguousarray(rates, np.float64)
    self.L.ref_indep_update_paths(self.h, _p(r, C.c_double))


def _ref_indep_m_step(self, optimize, J, D, rates, n_nodes):
    r = np.ascontiguousarray(rates, np.float64).copy()
    br = np.zeros(n_nodes)
    J = np.ascontiguousarray(J, np.float64)
    D = np.ascontiguousarray(D, np.float64)
    self.L.ref_indep_m_step(self.h, int(optimize), _p(J, C.c_double), _p(D, C.c_double),
                            _p(r, C.c_double), _p(br, C.c_double))
    return r, br


Reference.indep_update_paths = _ref_indep_update
Reference.indep_m_step = _ref_indep_m_step


def paths_equal(a, b):
    return (a.n_sites == b.n_sites and a.n_nodes == b.n_nodes and np.array_equal(a.init, b.init)
            and np.array_equal(a.offsets, b.offsets) and np.array_equal(a.jumps, b.jumps))


def init_paths_indep(engine, seed, rates, root, leaf, T, rung="B"):
    """initialize_paths_indep through the oracle (engine='orc', rung A or B) or the linked
    reference (engine='ref') -> FlatPaths of the two-node tree"""
    from epievo_amd.host import FlatPaths
    n = len(root)
    root = np.ascontiguousarray(root, np.uint8)
    leaf = np.ascontiguousarray(leaf, np.uint8)
    rates = np.ascontiguousarray(rates, np.float64)
    init, off = np.zeros(n, np.uint8), np.zeros(n + 1, np.uint64)
    cap = 8 * n + 64
    jumps = np.zeros(cap)
    if engine == "ref":
        tot = ref_lib().ref_init_paths_indep(seed, _p(rates, C.c_double), n, _p(root, C.c_uint8),
                                             _p(leaf, C.c_uint8), T, _p(init, C.c_uint8),
                                             _p(off, C.c_uint64), _p(jumps, C.c_double), cap)
    else:
        modes = (RNG_MT, MATH_LIBM) if rung == "A" else (RNG_PHILOX, MATH_EPV)
        tot = orc_lib().orc_init_paths_indep(modes[0], modes[1], seed, _p(rates, C.c_double), n,
                                             _p(root, C.c_uint8), _p(leaf, C.c_uint8), T,
                                             _p(init, C.c_uint8), _p(off, C.c_uint64),
                                             _p(jumps, C.c_double), cap)
    assert tot <= cap
    return FlatPaths(n, 2, init, off, jumps[:tot])


def forward_thinning(model, tree, n, seed, root=None):
    """the forward simulator's parallel rung on the CPU (thinning with keyed randomness, candidates in
    global time order) -> (FlatPaths, states[n_nodes][n])"""
    from epievo_amd.host import FlatPaths
    B = tree.n_nodes - 1
    init, off = np.zeros(B * n, np.uint8), np.zeros(B * n + 1, np.uint64)
    states = np.zeros((tree.n_nodes, n), np.uint8)
    rp = None
    if root is not None:
        root = np.ascontiguousarray(root, np.uint8)
        rp = _p(root, C.c_uint8)
    cap = max(1024, int(4 * n * B * (0.5 + tree.branches[1:].max() * model.rates.max())))
    while True:
        jumps = np.zeros(cap)
        tot = orc_lib().orc_forward_thinning(_p(model.rates, C.c_double), _p(model.T, C.c_double), tree.n_nodes,
                                             _p(tree.parent_ids, C.c_uint32), _p(tree.branches, C.c_double), n, seed, rp,
                                             _p(init, C.c_uint8), _p(off, C.c_uint64), _p(jumps, C.c_double), cap,
                                             _p(states, C.c_uint8))
        if tot != 2 ** 64 - 1:
            return FlatPaths(n, tree.n_nodes, init, off, jumps[:tot]), states
        cap *= 4
