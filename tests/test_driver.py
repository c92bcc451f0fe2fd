"""The product's C++ EM driver (epv::SingleSiteSampler, epievo_amd/csrc/host/epv_sampler.cpp) through
its flat C face (include/epievo_mi355x_driver.h): what the drop-in CLIs and `bench.py --gpus N` run.
Several GPU slots rehearsed on ONE GPU (loopback transport), one slot per process (ncclCommInitRank
with a world of one), and the bench entry itself -- all against the CPU oracle's parallel rung."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import orc
from common import simulate

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_driver_library_exports_every_declared_symbol():
    """no GPU needed: the library loads and has the symbols include/epievo_mi355x_driver.h declares"""
    import re
    from epievo_amd import driver
    L = driver.lib()
    declared = set(re.findall(r"\b(epvd_[a-z_]+)\s*\(", open(os.path.join(ROOT, "include", "epievo_mi355x_driver.h")).read()))
    assert declared == set(driver.DRIVER_SYMBOLS)
    assert not [s for s in declared if not hasattr(L, s)]
    # the cut table is pure host arithmetic: whole 16384-site rows, every slot fed
    cuts = driver.shard_cuts(10 ** 7, 8, 10, 50)
    assert len(cuts) == 9 and cuts[0] == 0 and cuts[-1] == 10 ** 7
    assert all(c % 16384 == 0 for c in cuts[1:-1]) and min(np.diff(cuts)) > 1.2e6
    assert driver.shard_cuts(20000, 8, 10, 50) == [0, 16384, 20000]     # too short for eight slots: two
    assert driver.shard_cuts(10000, 8, 10, 50) == [0, 10000]


def _oracle(model, tree, fp, burn, batch, seed, iters=1):
    o = orc.Oracle(tree, model, fp, "B", cap=16, seed=seed)
    out = []
    for it in range(iters):
        o.reset()
        J, D, nacc, acc = o.run_mcmc(burn, batch, sweep_base=it * (burn + batch))
        out.append((J, D, acc))
    return out, o.paths()


@pytest.mark.gpu
@pytest.mark.parametrize("devices,contexts", [([0, 0, 0, 0], 3), ([0, 0], 1), ([0], 2)])
def test_cpp_driver_rehearsal_slots_match_the_oracle(devices, contexts, monkeypatch):
    from epievo_amd import driver
    monkeypatch.setenv("EPV_ROW_BLOCKS", "4")
    monkeypatch.setenv("EPV_CONTEXTS_PER_GPU", str(contexts))
    model, tree, fp = simulate("tree", 40000, seed=12)
    s = driver.CppSampler(1, 2, devices=devices, capacity=16)
    s.reset(model, tree, fp)
    lay = s.layout()
    assert lay["slots_here"] == len(devices) and lay["parts_here"] == len(devices) * contexts
    assert lay["rccl"] is False or len(devices) == 1
    exp, exp_paths = _oracle(model, tree, fp, 1, 2, 99, iters=2)
    for it in range(2):
        if it:
            s.reset(model)
        J, D, acc = s.run_mcmc(99, it)
        assert np.array_equal(J, exp[it][0]) and np.array_equal(D, exp[it][1]) and acc == exp[it][2]
    assert orc.paths_equal(s.paths(), exp_paths)
    s.close()


@pytest.mark.gpu
def test_cpp_driver_one_slot_per_process_world_of_one():
    """the torchrun-style constructor: ncclCommInitRank, the halo and all-gather legs with a world of
    one rank (RCCL refuses two ranks on the one GPU of this pool)"""
    from epievo_amd import driver
    model, tree, fp = simulate("tree", 30000, seed=3)
    s = driver.CppSampler(1, 2, capacity=16, rank=(0, 1, 0, driver.unique_id()))
    s.reset(model, tree, fp, n_global=fp.n_sites)
    exp, exp_paths = _oracle(model, tree, fp, 1, 2, 5)
    J, D, acc = s.run_mcmc(5, 0)
    assert np.array_equal(J, exp[0][0]) and np.array_equal(D, exp[0][1]) and acc == exp[0][2]
    assert orc.paths_equal(s.paths(), exp_paths)
    with pytest.raises(driver.DriverError):
        s.reset(model, tree, fp.slice_sites(0, 100), n_global=fp.n_sites)     # not the columns shard_cuts assigns
    s.close()


def _bench(args, env_extra):
    env = dict(os.environ, **env_extra)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.split("\n") if l.strip()]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_plain_launch_runs_the_cpp_driver_over_rehearsal_slots():
    """`python bench.py --gpus 4` called plainly, four slots rehearsed on GPU 0"""
    quick = ["--steps", "1", "--warmup", "1", "--sites", "200000", "--no-cpu-baseline", "--no-reference-leg"]
    d = _bench(["--gpus", "4"] + quick, {"EPV_DEVICES": "0,0,0,0"})
    assert d["n_gpus"] == 4 and d["config"]["transport"] == "loopback"
    assert "epv::SingleSiteSampler" in d["config"]["driver"] and d["config"]["sites_total"] == 800000
    assert d["value"] > 0 and d["roofline"]["launches_timed"] > 0 and d["roofline"]["traffic_source"]
    one = _bench(["--gpus", "1"] + quick, {})
    assert one["n_gpus"] == 1 and one["config"]["transport"].startswith("none") and one["config"]["shards_per_gpu"] == 3


@pytest.mark.gpu
def test_bench_under_a_launcher_joins_by_rank():
    """the torch.distributed.run form with one rank: gloo control plane, ncclCommInitRank data plane"""
    env = {"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29617",
           "EPV_BENCH_FORCE_DIST": "1"}
    d = _bench(["--gpus", "1", "--steps", "1", "--warmup", "0", "--sites", "200000", "--no-cpu-baseline",
                "--no-reference-leg"], env)
    assert d["n_gpus"] == 1 and d["config"]["launch"].startswith("torch.distributed.run")


@pytest.mark.gpu
def test_cpp_driver_slots_agree_on_a_grown_capacity(monkeypatch):
    """a slot that absorbs a capacity overflow widens its jump slots; the width rides in the tail of
    the statistics all-gather and every slot follows before the next halo exchange (whose column
    size depends on it) -- also across processes, where the slots cannot look at each other"""
    from epievo_amd import driver
    monkeypatch.setenv("EPV_ROW_BLOCKS", "4")
    monkeypatch.setenv("EPV_CONTEXTS_PER_GPU", "1")
    model, tree, fp = simulate("pair", 30000, seed=2)          # T = 1: ~1 jump per path, up to ~10
    cap0 = int(fp.counts().max())
    s = driver.CppSampler(1, 2, devices=[0, 0, 0], capacity=cap0)
    s.reset(model, tree, fp)
    for it in range(4):                                        # overflows happen, are absorbed, the run goes on
        if it:
            s.reset(model)
        J, D, acc = s.run_mcmc(7, it)
        assert np.all(np.isfinite(D)) and 0.3 < acc <= 1.0
    p = s.paths()
    assert p.n_sites == fp.n_sites and p.counts().max() <= 2047
    s.close()
