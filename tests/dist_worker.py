"""Worker for the world_size>1 tests: runs ShardedSampler on a slice of a genome and
writes its owned paths + J/D so the parent test can compare with the unsharded run.
  python dist_worker.py <backend: oracle|hip|hipgroup> <cfg> <n_global> <burn> <batch> <em_iters> <outdir> [row_blocks]
Rendezvous via env (RANK/WORLD_SIZE/MASTER_ADDR/MASTER_PORT); comm backend is gloo."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))


def main():
    backend, cfg, n_global, burn, batch, iters, outdir = sys.argv[1:8]
    n_global, burn, batch, iters = int(n_global), int(burn), int(batch), int(iters)
    row_blocks = int(sys.argv[8]) if len(sys.argv) > 8 else 1
    import torch.distributed as dist
    from common import simulate
    from epievo_amd.parallel import ShardedSampler, TorchComm, shard_cuts
    dist.init_process_group("gloo")
    if backend == "oracle":
        comm = TorchComm(dist)
    else:   # the HIP path keeps its buffers on the GPU; gloo moves CUDA tensors through the host itself
        import torch
        comm = TorchComm(dist, torch.device("cuda", 0))
    model, tree, fp = simulate(cfg, n_global, seed=17)
    cuts = shard_cuts(n_global, comm.world, row_blocks)
    own = fp.slice_sites(cuts[comm.rank], cuts[comm.rank + 1])
    if backend == "oracle":
        from fake_device import OracleDevice
        ss = ShardedSampler(comm, device_factory=OracleDevice)
    elif backend == "hipgroup":      # two concurrent contexts per rank (LocalGroup) under the rank sharding
        from epievo_amd.parallel import LocalGroup
        ss = ShardedSampler(comm, device=0, device_factory=lambda dev: LocalGroup(dev, 2, burn + batch))
    else:
        ss = ShardedSampler(comm, device=0)
    ss.setup(model, tree, own, cuts, capacity=16, sweeps_per_refresh=burn + batch, row_blocks=row_blocks)
    out = {}
    for it in range(iters):
        ss.reset()
        J, D, acc = ss.run_mcmc(burn, batch, 1234, sweep_base=it * (burn + batch))
        out["J%d" % it], out["D%d" % it], out["acc%d" % it] = J, D, acc
        ss.scale_jump_times(tree.branches * (1.0 + 0.01 * (it + 1)))
        tree.branches[:] = tree.branches * (1.0 + 0.01 * (it + 1))
    p = ss.owned_paths()
    np.savez(os.path.join(outdir, "rank%d.npz" % comm.rank), init=p.init, offsets=p.offsets,
             jumps=p.jumps, **out)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
