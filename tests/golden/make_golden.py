#!/usr/bin/env python3
"""Generate the golden vectors in tests/golden/ by RUNNING THE UNMODIFIED REFERENCE
(/root/reference/src/libepievo compiled by oracle/Makefile into oracle/_ref/, driven
through oracle/ref_shim.cpp).  Only data is written: inputs and the reference's outputs.

  python tests/golden/make_golden.py          (needs oracle/_ref/libepievo_ref.so)

Contents of each <case>.npz  (case = <tree>_n<sites>_s<seed>):
  inputs : n_sites, subtree, parent, branches, rates, T, init, offsets, jumps
  ref    : tri_llh (after reset); for k in {1,3} sequential sweeps with mt19937(seed):
           paths_k (init/offsets/jumps), nacc_k, tri_k; run_mcmc(L=1,B=2) after those 3
           sweeps: J, D, acc; paths afterwards; J/D from get_sufficient_statistics;
           M-step (rates only, and rates+branches) from that J/D: rates, T, baseline,
           branches, llh, param text.
kat.npz: per-function known answers on fixed grids (collect_segment_info,
         continuous_time_trans_prob_mat, get_trans_prob, add_sufficient_statistics,
         expectation_J/D, libstdc++ uniform draws of mt19937, read_model of test.param).
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import orc  # noqa: E402
from common import simulate, ref_test_model  # noqa: E402

dbl = C.c_double


def m_step(L, optimize, n_nodes, J, D, rates, branches):
    r, T, bl, br = rates.copy(), np.zeros(4), np.zeros(4), branches.copy()
    buf = C.create_string_buffer(256)
    llh = L.ref_m_step(int(optimize), n_nodes, orc._p(J, dbl), orc._p(D, dbl), orc._p(r, dbl),
                       orc._p(T, dbl), orc._p(bl, dbl), orc._p(br, dbl), buf, len(buf))
    return r, T, bl, br, llh, buf.value.decode()


def case(cfg, n, seed):
    model, tree, fp = simulate(cfg, n, seed=seed)
    R = orc.Reference(tree, model, fp, seed=seed)
    out = dict(n_sites=n, subtree=tree.subtree_sizes, parent=tree.parent_ids,
               branches=tree.branches, rates=model.rates, T=model.T, init=fp.init,
               offsets=fp.offsets, jumps=fp.jumps, seed=seed)
    R.reset(1, 2)
    out["tri_llh"] = R.tri_llh()
    done = 0
    for k in (1, 3):
        nacc = R.sweeps(k - done)
        done = k
        p = R.paths()
        out["nacc_%d" % k] = nacc
        out["init_%d" % k], out["offsets_%d" % k], out["jumps_%d" % k] = p.init, p.offsets, p.jumps
        out["tri_%d" % k] = R.tri_llh()
    J, D, acc = R.run_mcmc()
    out["J"], out["D"], out["acc"] = J, D, acc
    p = R.paths()
    out["init_f"], out["offsets_f"], out["jumps_f"] = p.init, p.offsets, p.jumps
    Js, Ds = R.suffstats()
    out["J_stat"], out["D_stat"] = Js, Ds
    L = orc.ref_lib()
    # the reference asserts T[0]+T[1] == 1.0 EXACTLY inside the M-step
    # (EpiEvoModel.cpp:131), which tiny data sets can trip: M-step vectors only for n >= 1000
    for tag, opt in (("mr", 0), ("mb", 1)) if n >= 1000 else ():
        r, T, bl, br, llh, txt = m_step(L, opt, tree.n_nodes, J, D, model.rates, tree.branches)
        out[tag + "_rates"], out[tag + "_T"], out[tag + "_baseline"] = r, T, bl
        out[tag + "_branches"], out[tag + "_llh"], out[tag + "_text"] = br, llh, txt
    nb = tree.branches * 1.25
    R.scale_jump_times(nb)
    out["scaled_jumps"] = R.paths().jumps
    np.savez_compressed(os.path.join(HERE, "%s_n%d_s%d.npz" % (cfg, n, seed)), **out)
    print("wrote", cfg, n, seed, "jumps", len(fp.jumps))


def kat():
    L = orc.ref_lib()
    rng = np.random.RandomState(123)
    model = ref_test_model()
    out = {}
    # continuous_time_trans_prob_mat / get_trans_prob on a grid
    grid = [(r0, r1, t) for r0 in (0.0869, 0.236, 3.65, 10.2) for r1 in (0.0869, 3.45, 4.19)
            for t in (1e-9, 1e-4, 0.02, 0.1, 1.0, 7.5, 300.0)]
    P, G = [], []
    for r0, r1, t in grid:
        p = np.zeros(4)
        L.ref_kat_trans_prob_mat(r0, r1, t, orc._p(p, dbl))
        P.append(p)
        G.append([L.ref_kat_get_trans_prob(r0, r1, t, a, b) for a in (0, 1) for b in (0, 1)])
    out["ctmc_grid"], out["ctmc_P"], out["ctmc_G"] = np.array(grid), np.array(P), np.array(G)
    # expectation_J / expectation_D
    eg = [(r0, r1, t) for r0 in (0.236, 3.65) for r1 in (3.45, 10.2) for t in (0.02, 0.1, 1.0)]
    E = []
    for r0, r1, t in eg:
        e = np.zeros(16)
        L.ref_kat_expectations(r0, r1, t, orc._p(e, dbl))
        E.append(e)
    out["exp_grid"], out["exp_JD"] = np.array(eg), np.array(E)
    # collect_segment_info and add_sufficient_statistics on random paths (with ties)
    segs, stats = [], []
    u64 = C.c_uint64
    for trial in range(40):
        T = 1.0
        def rp():
            k = rng.randint(0, 5)
            t = np.sort(rng.choice(np.arange(1, 20) / 20.0, size=k, replace=False)) if trial % 3 == 0 \
                else np.sort(rng.uniform(0, T, size=k))
            return int(rng.randint(0, 2)), np.ascontiguousarray(t)
        (li, lj), (mi, mj), (ri, rj) = rp(), rp(), rp()
        K = len(lj) + len(rj) + 1
        r0, r1, ln = np.zeros(K), np.zeros(K), np.zeros(K)
        t0, t1 = np.zeros(K, np.uint64), np.zeros(K, np.uint64)
        lje, rje, mje = (np.concatenate([x, [0.0]]) for x in (lj, rj, mj))
        k = L.ref_kat_segments(orc._p(model.rates, dbl), li, len(lj), orc._p(lje, dbl), ri, len(rj),
                               orc._p(rje, dbl), T, orc._p(r0, dbl), orc._p(r1, dbl),
                               orc._p(t0, u64), orc._p(t1, u64), orc._p(ln, dbl))
        assert k == K
        J, D = np.zeros(8), np.zeros(8)
        L.ref_kat_suffstats(li, len(lj), orc._p(lje, dbl), mi, len(mj), orc._p(mje, dbl), ri,
                            len(rj), orc._p(rje, dbl), T, orc._p(J, dbl), orc._p(D, dbl))
        segs.append(dict(li=li, lj=lj, ri=ri, rj=rj, r0=r0, r1=r1, t0=t0, t1=t1, len=ln))
        stats.append(dict(li=li, lj=lj, mi=mi, mj=mj, ri=ri, rj=rj, J=J, D=D))
    for i, s in enumerate(segs):
        for k, v in s.items():
            out["seg%d_%s" % (i, k)] = v
    for i, s in enumerate(stats):
        for k, v in s.items():
            out["st%d_%s" % (i, k)] = v
    out["n_seg_cases"] = len(segs)
    # libstdc++ uniform_real_distribution<double>(0,1) over mt19937(seed)
    for seed in (1, 42, 4294967295):
        d = np.zeros(64)
        L.ref_kat_mt_canonical(seed, 64, orc._p(d, dbl))
        out["mt_%d" % seed] = d
    # read_model(test.param) scaled / unscaled
    from common import _tmp, TEST_PARAM_TEXT
    pf = _tmp("test.param", TEST_PARAM_TEXT).encode()
    for tag, sc in (("scaled", 1), ("unscaled", 0)):
        r, T, bl = np.zeros(8), np.zeros(4), np.zeros(4)
        assert L.ref_read_model(pf, sc, orc._p(r, dbl), orc._p(T, dbl), orc._p(bl, dbl)) == 0
        out["model_%s_rates" % tag], out["model_%s_T" % tag], out["model_%s_bl" % tag] = r, T, bl
    # forward simulation through the LINKED TripletSampler (ref_forward_sim)
    from common import config
    u8p, u32p, u64p, dp = (C.POINTER(C.c_uint8), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), C.POINTER(C.c_double))
    L.ref_forward_sim.restype = C.c_uint64
    L.ref_forward_sim.argtypes = [C.c_uint64, dp, dp, C.c_int, u32p, dp, C.c_uint64, u8p, u64p, dp, u64p, C.c_uint64]
    for cfg, n in (("tree", 600), ("pair", 300)):
        tree = config(cfg)
        N, cap = tree.n_nodes, 40 * n
        seqs, off = np.zeros(N * n, np.uint8), np.zeros(N + 1, np.uint64)
        tt, pp = np.zeros(cap), np.zeros(cap, np.uint64)
        tot = L.ref_forward_sim(42, orc._p(model.rates, dbl), orc._p(model.T, dbl), N, orc._p(tree.parent_ids, C.c_uint32),
                                orc._p(tree.branches, dbl), n, orc._p(seqs, C.c_uint8), orc._p(off, C.c_uint64),
                                orc._p(tt, dbl), orc._p(pp, C.c_uint64), cap)
        out["fwd_%s_seqs" % cfg], out["fwd_%s_off" % cfg] = seqs, off
        out["fwd_%s_t" % cfg], out["fwd_%s_p" % cfg] = tt[:tot], pp[:tot]
    # the LINKED end-conditioned samplers alone: forward rejection (the hot path's) and
    # end_cond_sampling_Nielsen (the one the parallel rung / the GPU use for state changes)
    L.ref_kat_end_cond_paths.restype = C.c_uint64
    L.ref_kat_end_cond_paths.argtypes = [C.c_int, C.c_uint64, dbl, dbl, C.c_int, C.c_int, dbl, C.c_uint64,
                                         u32p, dp, C.c_uint64]
    ec = [(0.236, 10.2, 0, 1, 0.05), (3.65, 4.19, 1, 0, 0.3), (0.0869, 3.45, 1, 1, 1.0),
          (10.2, 0.236, 0, 1, 2.0), (3.65, 3.45, 0, 0, 0.02), (0.0869, 0.236, 1, 0, 0.001)]
    out["ec_grid"] = np.array(ec)
    for sampler in (0, 1):
        for i, (r0, r1, a, b, T) in enumerate(ec):
            n, cap = 200, 50000
            cnt, tt = np.zeros(n, np.uint32), np.zeros(cap)
            tot = L.ref_kat_end_cond_paths(sampler, 7, r0, r1, a, b, T, n, orc._p(cnt, C.c_uint32),
                                           orc._p(tt, dbl), cap)
            out["ec%d_%d_counts" % (sampler, i)], out["ec%d_%d_times" % (sampler, i)] = cnt, tt[:tot]
    np.savez_compressed(os.path.join(HERE, "kat.npz"), **out)
    print("wrote kat")


def text_fixtures():
    """tests/golden/text/: files WRITTEN BY THE REFERENCE (its operator<<(Path) inside the mains'
    three lines of glue; write_root_to_pathfile_global / append_to_pathfile_global), next to the
    values they hold -- what tests/test_file_formats.py pins the product's reader and writer to
    where oracle/_ref is absent."""
    import test_file_formats as t
    from common import config
    d = os.path.join(HERE, "text")
    os.makedirs(d, exist_ok=True)
    tree, fp = t.awkward_paths("tree", 40)
    tt = tree.branches * 1.0000000000000002 + 1e-17
    t.ref_write_paths(os.path.join(d, "tree_n40.paths"), tree, tt, fp)
    np.savez(os.path.join(d, "tree_n40.npz"), n_sites=fp.n_sites, n_nodes=fp.n_nodes, init=fp.init,
             offsets=fp.offsets, jumps=fp.jumps, tot_times=tt, names=np.array(tree.node_names))
    tree, n = config("tree"), 60
    seqs, off, tj, pp = t._fwd(11, tree, n)
    L = orc.ref_lib()
    L.ref_write_global_jumps.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_uint64, t.u8p, t.u64p, t.dp, t.u64p]
    root = np.ascontiguousarray(seqs[:n])
    assert L.ref_write_global_jumps(os.path.join(d, "tree_n60.jumps").encode(), tree.n_nodes,
                                    "\n".join(tree.node_names).encode(), n, orc._p(root, C.c_uint8),
                                    orc._p(off, C.c_uint64), orc._p(tj, C.c_double), orc._p(pp, C.c_uint64)) == 0
    np.savez(os.path.join(d, "tree_n60_global.npz"), root=root, off=off, t=tj, p=pp, names=np.array(tree.node_names))
    print("wrote text fixtures")


if __name__ == "__main__":
    assert orc.have_ref(), "build oracle/_ref first: make -C oracle ref"
    text_fixtures()
    for cfg in ("pair", "tree"):
        for n in (16, 64, 1000):
            for seed in (1, 42):
                case(cfg, n, seed)
    for cfg in ("star4", "multi", "cat6"):     # multifurcations, a long branch, a deep tree
        case(cfg, 200, 7)
    kat()
