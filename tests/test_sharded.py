"""The N>1 path: site shards with wide halos (epievo_amd/parallel.py) must reproduce the
unsharded run bit-for-bit on paths, J, D and the acceptance rate (shards are cut on whole rows of
the statistics tree, so every rank sums the same balanced tree as one context does).
CPU: 2 and 3 gloo ranks over the oracle-backed device double.  GPU: 2 gloo ranks sharing
the one MI355X through the real HIP path."""
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

import orc
from common import simulate
from epievo_amd.parallel import concat_sites, halo_width, shard_cuts
from epievo_amd.host import FlatPaths

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _run_ranks(world, backend, cfg, n_global, burn, batch, iters, row_blocks=1):
    port = _free_port()      # a fixed port can still be in TIME_WAIT from the previous run
    out = tempfile.mkdtemp(prefix="epv_shard_")
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), LOCAL_RANK=str(r), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen(
            [sys.executable, os.path.join(HERE, "dist_worker.py"), backend, cfg, str(n_global),
             str(burn), str(batch), str(iters), out, str(row_blocks)], env=env))
    for p in procs:
        assert p.wait(timeout=600) == 0
    return [np.load(os.path.join(out, "rank%d.npz" % r)) for r in range(world)]


def _unsharded(engine, cfg, n, burn, batch, iters):
    model, tree, fp = simulate(cfg, n, seed=17)
    res = []
    eng = engine(tree, model, fp)
    for it in range(iters):
        eng.reset()
        J, D, nacc = eng.run_mcmc(burn, batch, 1234, it * (burn + batch))
        res.append((J, D, nacc / float(batch * (n - 2))))
        tree.branches[:] = tree.branches * (1.0 + 0.01 * (it + 1))
        eng.scale_jump_times(tree.branches)
    return res, eng.paths(), tree


class _OracleEngine:
    def __init__(self, tree, model, fp):
        self.o = orc.Oracle(tree, model, fp, "B", cap=16)

    def reset(self):
        self.o.reset()

    def run_mcmc(self, burn, batch, seed, base):
        self.o.seed(seed)
        J, D, nacc, _ = self.o.run_mcmc(burn, batch, base)
        return J, D, nacc

    def scale_jump_times(self, nb):
        self.o.scale_jump_times(nb)

    def paths(self):
        return self.o.paths()


def _check(ranks, ref_res, ref_paths, n_nodes, cuts):
    parts = [FlatPaths(cuts[i + 1] - cuts[i], n_nodes, r["init"], r["offsets"], r["jumps"])
             for i, r in enumerate(ranks)]
    assert orc.paths_equal(concat_sites(parts), ref_paths)
    for it, (J, D, acc) in enumerate(ref_res):
        for r in ranks:
            assert np.array_equal(r["J%d" % it], J)
            assert np.array_equal(r["D%d" % it], D)       # same balanced tree, same bits
            assert float(r["acc%d" % it]) == acc


# shards of unequal length (the cut points sit on whole rows), 1- and 4-block rows
@pytest.mark.parametrize("world,cfg,n,row_blocks", [(2, "tree", 1500, 1), (3, "pair", 2300, 1), (2, "tree", 4000, 4)])
def test_sharded_oracle_gloo(world, cfg, n, row_blocks):
    burn, batch, iters = 1, 3, 2
    ranks = _run_ranks(world, "oracle", cfg, n, burn, batch, iters, row_blocks)
    ref_res, ref_paths, tree = _unsharded(_OracleEngine, cfg, n, burn, batch, iters)
    _check(ranks, ref_res, ref_paths, tree.n_nodes, shard_cuts(n, world, row_blocks))


def test_shard_cuts_and_halo_width():
    assert halo_width(60) == 512 and halo_width(1) == 256 and halo_width(100) == 768
    cuts = shard_cuts(10 ** 7, 8)
    assert cuts[0] == 0 and cuts[-1] == 10 ** 7 and all(c % 16384 == 0 for c in cuts[1:-1])
    lens = np.diff(cuts)
    assert lens.max() - lens.min() <= 2 * 16384
    with pytest.raises(ValueError):
        shard_cuts(20000, 4)


@pytest.mark.gpu
def test_sharded_hip_two_ranks_one_gpu():
    from epievo_amd.sampler import DeviceSampler

    class _HipEngine:
        def __init__(self, tree, model, fp):
            self.d = DeviceSampler(0)
            self.d.set_tree(tree)
            self.d.set_model(model)
            self.d.upload_paths(fp, 16)

        def reset(self):
            self.d.reset()

        def run_mcmc(self, burn, batch, seed, base):
            return self.d.run_mcmc(burn, batch, seed, base)

        def scale_jump_times(self, nb):
            self.d.scale_jump_times(nb)

        def paths(self):
            return self.d.paths()

    burn, batch, iters, n, rb = 2, 3, 2, 10000, 4
    ranks = _run_ranks(2, "hip", "tree", n, burn, batch, iters, rb)
    ref_res, ref_paths, tree = _unsharded(_HipEngine, "tree", n, burn, batch, iters)
    cuts = shard_cuts(n, 2, rb)
    _check(ranks, ref_res, ref_paths, tree.n_nodes, cuts)
    # the same with two concurrent contexts per rank (LocalGroup inside each rank's shard)
    ranks = _run_ranks(2, "hipgroup", "tree", n, burn, batch, iters, rb)
    _check(ranks, ref_res, ref_paths, tree.n_nodes, cuts)
