"""RCCL rehearsal on a 1-GPU box: launch with
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 \
      --master-port 29517 tools/nccl_selfcheck.py
Checks that the nccl (= RCCL) process group comes up in the same process as the HIP library's
own stream, and that TorchComm's all-gather leg and the sampler run side by side."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from epievo_amd.workloads import ref_test_model, config          # noqa: E402
from epievo_amd import host                        # noqa: E402
from epievo_amd.parallel import ShardedSampler, TorchComm   # noqa: E402

lr = int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(lr)
dist.init_process_group("nccl", device_id=torch.device("cuda", lr))
comm = TorchComm(dist, torch.device("cuda", lr))
parts = comm.allgather(np.arange(5, dtype=np.float64) + comm.rank)
assert len(parts) == comm.world and np.array_equal(parts[comm.rank], np.arange(5.) + comm.rank)
model, tree = ref_test_model(), config("tree")
fp = host.simulate(model, tree, 20000, 5)
ss = ShardedSampler(comm, device=lr)
ss.setup(model, tree, fp, 20000 * comm.world, capacity=16)
ss.reset()
J, D, acc = ss.run_mcmc(2, 3, 11)
t = torch.tensor([acc], dtype=torch.float64, device="cuda")
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
print("nccl selfcheck ok: world", comm.world, "acc", float(t.item()), "J0", J[:4])
dist.destroy_process_group()
