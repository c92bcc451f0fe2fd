"""The N>1 path: site shards with wide halos (epievo_amd/parallel.py) must reproduce the
unsharded run bit-for-bit on paths and J (D up to cross-shard summation order).
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
from epievo_amd.parallel import concat_sites
from epievo_amd.host import FlatPaths

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _run_ranks(world, backend, cfg, n_own, burn, batch, iters, port=None):
    port = _free_port()      # a fixed port can still be in TIME_WAIT from the previous run
    out = tempfile.mkdtemp(prefix="epv_shard_")
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), LOCAL_RANK=str(r), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen(
            [sys.executable, os.path.join(HERE, "dist_worker.py"), backend, cfg, str(n_own),
             str(burn), str(batch), str(iters), out], env=env))
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


def _check(ranks, ref_res, ref_paths, n_nodes, n_own):
    parts = [FlatPaths(n_own, n_nodes, r["init"], r["offsets"], r["jumps"]) for r in ranks]
    assert orc.paths_equal(concat_sites(parts), ref_paths)
    for it, (J, D, acc) in enumerate(ref_res):
        for r in ranks:
            assert np.array_equal(r["J%d" % it], J)
            np.testing.assert_allclose(r["D%d" % it], D, rtol=1e-12)
            assert float(r["acc%d" % it]) == acc


@pytest.mark.parametrize("world,cfg,n_own", [(2, "tree", 400), (3, "pair", 300)])
def test_sharded_oracle_gloo(world, cfg, n_own):
    burn, batch, iters = 1, 3, 2
    ranks = _run_ranks(world, "oracle", cfg, n_own, burn, batch, iters, 29611 + world)
    ref_res, ref_paths, tree = _unsharded(_OracleEngine, cfg, n_own * world, burn, batch, iters)
    _check(ranks, ref_res, ref_paths, tree.n_nodes, n_own)


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

    burn, batch, iters, n_own = 2, 3, 2, 5000
    ranks = _run_ranks(2, "hip", "tree", n_own, burn, batch, iters, 29655)
    ref_res, ref_paths, tree = _unsharded(_HipEngine, "tree", 2 * n_own, burn, batch, iters)
    _check(ranks, ref_res, ref_paths, tree.n_nodes, n_own)
    # the same with two concurrent contexts per rank (LocalGroup inside each rank's shard)
    ranks = _run_ranks(2, "hipgroup", "tree", n_own, burn, batch, iters, 29657)
    _check(ranks, ref_res, ref_paths, tree.n_nodes, n_own)
