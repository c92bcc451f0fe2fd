"""RCCL on the one GPU of the test box: the nccl (= RCCL) process group of torch.distributed with
the library's device buffers -- zero-copy views through the CUDA array interface, an in-place
all-gather, a send/receive pair, and ShardedSampler over TorchComm equal to the plain device call
(tools/nccl_selfcheck.py, launched through torch.distributed.run as bench.py is).  More than one
rank needs more than one GPU: that leg is only covered by the gloo tests (tests/test_sharded.py)."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_rccl_one_rank_selfcheck():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                        "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(ROOT, "tools", "nccl_selfcheck.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "nccl selfcheck ok: world 1" in r.stdout


@pytest.mark.gpu
def test_bench_two_ranks_share_the_gpu():
    """bench.py as the driver launches it for N = 2 (torch.distributed.run, one rank per process),
    with the gloo backend so that both ranks can use the one GPU of the test box: the whole N > 1
    path -- row-aligned shards, the halo exchange of 512 columns, the all-gather of the statistic
    rows -- runs on the HIP kernels and rank 0 prints the one JSON line"""
    import json
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--driver", "torch", "--backend", "gloo", "--sites", "200000", "--steps", "2",
                        "--warmup", "1", "--no-cpu-baseline"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.split("\n") if l.strip()]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["scaling"] == "weak" and j["value"] > 1e8
    assert "512-column redundant halos" in j["config"]["sharding"] and "cpu_baseline" not in j
    assert j["config"]["driver"].startswith("Python driver")


@pytest.mark.gpu
def test_bench_launcher_default_falls_back_together_when_rccl_cannot_form():
    """the default (C++ driver, one rank per GPU through ncclCommInitRank) launched with two ranks on
    the ONE GPU of this box: RCCL refuses, every rank notices, all switch to the Python driver over
    the gloo control plane and the line says so -- no hang, no half-switched job"""
    import json
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--sites", "200000", "--steps", "1", "--warmup", "1", "--no-cpu-baseline",
                        "--no-reference-leg"], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, NCCL_DEBUG="WARN"))
    assert r.returncode == 0, r.stderr[-3000:]
    j = json.loads([l for l in r.stdout.split("\n") if l.strip()][-1])
    assert j["n_gpus"] == 2 and "C++ driver unavailable" in j["config"]["driver"] and j["value"] > 1e8
