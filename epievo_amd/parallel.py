"""Site-sharded multi-GPU driver: one process per GPU, contiguous shards of the genome
with 2-site halos, boundary columns exchanged after every colour phase, one all-reduce
of the per-branch J/D (+ accept count) per run_mcmc.

Why it is correct: one MH update of site i reads the paths of sites
i-2..i+2 and the cached triple log-likelihoods tri[i-1], tri[i+1]; it writes path i and
tri[i-1..i+1].  Within one colour phase only sites congruent mod 3 (GLOBAL index) are
updated, so after phase c each rank ships, for each of its two boundary-most owned
sites whose colour is c, the site's column and tri[s-1..s+1] to the neighbour; nothing
else near the boundary changed in that phase.  The RNG is keyed by the global site
index, so a sharded run reproduces the unsharded one bit-for-bit on paths, states and J
(D differs only in summation order across shards: the per-shard canonical trees are
added in rank order).

The reference has no parallelism at all (single-threaded); this
module is new capability, not a translation.
"""
import numpy as np

from .host import FlatPaths

HALO = 2


def concat_sites(parts):
    """concatenate FlatPaths along the site axis"""
    B = parts[0].n_nodes - 1
    n = sum(p.n_sites for p in parts)
    init = np.concatenate([p.init.reshape(B, p.n_sites) for p in parts], axis=1)
    cnt = np.concatenate([p.counts().reshape(B, p.n_sites) for p in parts], axis=1)
    jumps = []
    for b in range(B):
        for p in parts:
            c = p.counts().reshape(B, p.n_sites)[b]
            o = p.offsets[:-1].reshape(B, p.n_sites)[b]
            jumps.append(p.jumps[int(o[0]):int(o[-1] + c[-1])])
    off = np.zeros(B * n + 1, np.uint64)
    off[1:] = np.cumsum(cnt.reshape(-1))
    return FlatPaths(n, parts[0].n_nodes, init.reshape(-1).copy(), off,
                     np.concatenate(jumps) if jumps else np.zeros(0))


class NullComm:
    rank, world = 0, 1

    def exchange(self, to_left, to_right):
        return None, None

    def allgather(self, arr):
        return [np.asarray(arr)]


class TorchComm:
    """torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the
    CPU tests).  Messages are tiny (two packed columns; (n_nodes-1)*16 doubles), so the
    cost is latency only; they go through device tensors when the backend needs it."""

    def __init__(self, dist, device=None):
        import torch
        self.dist, self.torch = dist, torch
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.device = device if device is not None else torch.device("cpu")

    def _t(self, a):
        return self.torch.from_numpy(np.ascontiguousarray(a)).to(self.device)

    def exchange(self, to_left, to_right):
        """send byte arrays to the left/right neighbour, receive theirs (same sizes)"""
        dist, torch = self.dist, self.torch
        ops, rl, rr = [], None, None
        if self.rank > 0:
            rl = torch.empty(len(to_left), dtype=torch.uint8, device=self.device)
            ops.append(dist.P2POp(dist.isend, self._t(to_left), self.rank - 1))
            ops.append(dist.P2POp(dist.irecv, rl, self.rank - 1))
        if self.rank < self.world - 1:
            rr = torch.empty(len(to_right), dtype=torch.uint8, device=self.device)
            ops.append(dist.P2POp(dist.isend, self._t(to_right), self.rank + 1))
            ops.append(dist.P2POp(dist.irecv, rr, self.rank + 1))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        return (rl.cpu().numpy() if rl is not None else None,
                rr.cpu().numpy() if rr is not None else None)

    def allgather(self, arr):
        t = self._t(np.asarray(arr))
        outs = [self.torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(outs, t)
        return [o.cpu().numpy() for o in outs]


class ShardedSampler:
    """SingleSiteSampler over a site-sharded genome.  `device_factory(device)` builds the
    per-rank engine (the HIP DeviceSampler in the product; the tests inject an
    oracle-backed double to check the sharding logic on CPU with gloo)."""

    def __init__(self, comm, device=0, device_factory=None):
        self.comm = comm
        if device_factory is None:
            from .sampler import DeviceSampler
            device_factory = DeviceSampler
        self.dev = device_factory(device)
        self.first = self.last = 0

    def owned_sites(self):
        return self.last - self.first + 1

    def setup(self, model, tree, fp_own, n_global, capacity=16):
        c = self.comm
        n_own = fp_own.n_sites
        self.n_own, self.n_global, self.B = n_own, n_global, tree.n_nodes - 1
        left = HALO if c.rank > 0 else 0
        right = HALO if c.rank < c.world - 1 else 0
        parts = []
        if left:
            parts.append(fp_own.slice_sites(0, HALO))            # placeholder, overwritten below
        parts.append(fp_own)
        if right:
            parts.append(fp_own.slice_sites(n_own - HALO, n_own))
        fp_loc = concat_sites(parts) if len(parts) > 1 else fp_own
        self.n_loc = fp_loc.n_sites
        self.g0 = c.rank * n_own - left
        self.dev.set_tree(tree)
        self.dev.set_model(model)
        self.dev.upload_paths(fp_loc, capacity, self.g0, n_global)
        self.first = left if left else 1
        self.last = self.n_loc - 1 - right if right else self.n_loc - 2
        self.dev.set_update_range(self.first, self.last)
        self.left, self.right = left, right
        self._exchange(colour=None)   # fill the halos with the neighbours' true edge columns

    # ---- halo exchange
    def _pack_edge(self, sites, colour):
        cb = self.dev.column_bytes()
        buf = np.zeros(2 + HALO * cb, np.uint8)
        for k, s in enumerate(sites):
            if colour is None or (self.g0 + s) % 3 == colour:
                buf[k] = 1
                buf[2 + k * cb: 2 + (k + 1) * cb] = self.dev.get_columns(s, 1)
        return buf

    def _unpack_edge(self, buf, sites):
        cb = self.dev.column_bytes()
        for k, s in enumerate(sites):
            if buf[k]:
                self.dev.put_columns(s, 1, buf[2 + k * cb: 2 + (k + 1) * cb])

    def _exchange(self, colour):
        if self.comm.world == 1:
            return
        own_l = [self.first, self.first + 1]               # my leftmost owned sites
        own_r = [self.last - 1, self.last]                 # my rightmost owned sites
        to_left = self._pack_edge(own_l, colour) if self.left else None
        to_right = self._pack_edge(own_r, colour) if self.right else None
        from_left, from_right = self.comm.exchange(to_left, to_right)
        if from_left is not None:                          # neighbour's rightmost owned -> my left halo
            self._unpack_edge(from_left, [0, 1])
        if from_right is not None:                         # neighbour's leftmost owned -> my right halo
            self._unpack_edge(from_right, [self.n_loc - 2, self.n_loc - 1])

    # ---- SingleSiteSampler interface
    def set_model(self, model):
        self.dev.set_model(model)

    def reset(self):
        self.dev.reset()

    def sweep(self, seed, sweep_index):
        nacc = 0
        for colour in range(3):
            nacc += self.dev.sweep_phase(colour, seed, sweep_index)
            self._exchange(colour)
        return nacc

    def run_mcmc(self, burn_in, batch, seed, sweep_base=0):
        """-> (J, D, acc_rate): batch averages over the WHOLE genome on every rank"""
        if self.comm.world == 1:
            J, D, nacc = self.dev.run_mcmc(burn_in, batch, seed, sweep_base)
            return J, D, nacc / float(batch * (self.n_global - 2))
        sweep = sweep_base
        for _ in range(burn_in):
            self.sweep(seed, sweep)
            sweep += 1
        J, D, nacc = np.zeros(self.B * 8), np.zeros(self.B * 8), 0
        for _ in range(batch):
            nacc += self.sweep(seed, sweep)
            sweep += 1
            J1, D1 = self.dev.suffstats()
            J += J1
            D += D1
        # one exchange per EM iteration: [J | D | n_acc], summed in rank order
        parts = self.comm.allgather(np.concatenate([J, D, [float(nacc)]]))
        tot = np.zeros_like(parts[0])
        for p in parts:
            tot = tot + p
        nb = float(batch)
        return (tot[:self.B * 8] / nb, tot[self.B * 8:self.B * 16] / nb,
                tot[-1] / float(batch * (self.n_global - 2)))

    def scale_jump_times(self, new_branches):
        self.dev.scale_jump_times(new_branches)

    def owned_paths(self):
        return self.dev.paths().slice_sites(self.first if self.left else 0,
                                            self.last + 1 if self.right else self.n_loc)
