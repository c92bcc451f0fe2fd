"""RCCL rehearsal on a 1-GPU box: launch with
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 \
      --master-port 29517 tools/nccl_selfcheck.py
Checks, with the nccl (= RCCL) process group of torch.distributed in the same process as the HIP
library's own streams, every leg of the one-process-per-GPU path that one rank can exercise:
  * torch sees the library's device buffers (DevBuf) through the CUDA array interface without a
    copy, and RCCL moves them: an in-place all-gather and a send/receive pair to the own rank;
  * ShardedSampler over TorchComm (statistics rows + accept count in ONE all-gather) gives the
    numbers of the plain DeviceSampler.run_mcmc, bit for bit."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from epievo_amd.workloads import ref_test_model, config          # noqa: E402
from epievo_amd import host                                      # noqa: E402
from epievo_amd.parallel import ShardedSampler, TorchComm, shard_cuts   # noqa: E402
from epievo_amd.sampler import DeviceSampler                     # noqa: E402

lr = int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(lr)
dist.init_process_group("nccl", device_id=torch.device("cuda", lr))
comm = TorchComm(dist, torch.device("cuda", lr))
model, tree = ref_test_model(), config("tree")
n = 40000
fp = host.simulate(model, tree, n, 5)

# --- device buffers through torch and RCCL
d = DeviceSampler(lr)
d.set_tree(tree); d.set_model(model); d.upload_paths(fp, 16)
a, b = d.alloc(4096), d.alloc(4096)
d.write(a, 0, np.arange(512, dtype=np.float64))
ta, tb = comm._t(a), comm._t(b)
assert ta.data_ptr() == a.ptr and ta.is_cuda and ta.numel() == 4096          # a view, not a copy
dist.all_gather_into_tensor(ta, ta)                                           # world 1: in place
ops = [dist.P2POp(dist.isend, ta, comm.rank), dist.P2POp(dist.irecv, tb, comm.rank)]
try:
    for w in dist.batch_isend_irecv(ops):
        w.wait()
    torch.cuda.synchronize()
    assert np.array_equal(d.read(b, 0, 512), np.arange(512.0)), "send/recv to self moved wrong bytes"
    p2p = "send/recv to self ok"
except RuntimeError as e:                                                     # some RCCL builds refuse self-sends
    p2p = "send/recv to self refused by RCCL (%s)" % str(e).split("\n")[0][:80]
# packed halo columns written by the kernels straight into such a buffer, and read back
cols = d.alloc(64 * d.column_bytes())
d.pack_columns(100, 64, cols)
d.unpack_columns(100, 64, cols)

# --- the sharded driver on one rank == the plain device call
d.reset()
J0, D0, n0 = d.run_mcmc(2, 3, 11)
ss = ShardedSampler(comm, device=lr)
ss.setup(model, tree, fp, shard_cuts(n, comm.world), capacity=16, sweeps_per_refresh=5)
ss.reset()
J, D, acc = ss.run_mcmc(2, 3, 11)
assert np.array_equal(J, J0) and np.array_equal(D, D0) and acc == n0 / float(3 * (n - 2))
t = torch.tensor([acc], dtype=torch.float64, device="cuda")
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
print("nccl selfcheck ok: world", comm.world, "|", p2p, "| acc", float(t.item()), "J0", J[:4])
dist.destroy_process_group()
