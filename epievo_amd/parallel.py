"""Site-sharded multi-GPU driver: one process per GPU, contiguous shards of the genome
with WIDE halos that are updated redundantly, refreshed once per run_mcmc, and one
exchange of the per-branch J/D (+ accept count) per run_mcmc.

Why it is correct: one MH update of site i reads the paths of sites
i-2..i+2 and the cached triple log-likelihoods tri[i-1], tri[i+1]; it writes path i and
tri[i-1..i+1].  The RNG and the 3-colouring are keyed by the GLOBAL site index, so a
rank that holds copies of a neighbour's edge columns can update them itself and obtain
exactly what the owner computes.  Each colour phase, the two outermost still-valid halo
columns at every shard-internal edge lose a neighbour and go stale, so a halo of H
columns lasts H/2 phases = H/6 sweeps.  With H = 6*(burn_in + batch) + 2 a whole
run_mcmc needs NO communication inside it: the halos are refreshed once before
reset(), and J/D/accepts are combined once after it -- instead of 3 exchanges per
sweep.  The redundant work is 2H/n of a shard (0.07 % at n = 1e6, -L 10 -B 50).
A sharded run reproduces the unsharded one bit-for-bit on paths, states and J; D
differs only in summation order across shards (per-shard canonical trees added in
rank order).

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
        # a capacity overflow widens this shard's jump slots and the run carries on (as the
        # reference's vectors would); refresh_halos() brings all shards to the same width
        # before columns travel
        if hasattr(self.dev, "auto_grow"):
            self.dev.auto_grow = True
        self.halo = 0

    def owned_sites(self):
        return self.n_own - (1 if self.comm.rank == 0 else 0) - \
            (1 if self.comm.rank == self.comm.world - 1 else 0)

    def setup(self, model, tree, fp_own, n_global, capacity=16, sweeps_per_refresh=60):
        """fp_own: this rank's owned columns (every rank owns the same number of sites).
        The halo is sized for `sweeps_per_refresh` sweeps between refreshes."""
        c = self.comm
        n_own = fp_own.n_sites
        self.n_own, self.n_global, self.B = n_own, n_global, tree.n_nodes - 1
        H = 6 * sweeps_per_refresh + 2 if c.world > 1 else 0
        if H > n_own:
            raise ValueError("shards of %d sites are too small for a %d-column halo" % (n_own, H))
        self.halo = H
        left = H if c.rank > 0 else 0
        right = H if c.rank < c.world - 1 else 0
        parts = []
        if left:
            parts.append(fp_own.slice_sites(0, H))               # placeholder, refreshed below
        parts.append(fp_own)
        if right:
            parts.append(fp_own.slice_sites(n_own - H, n_own))
        fp_loc = concat_sites(parts) if len(parts) > 1 else fp_own
        self.n_loc = fp_loc.n_sites
        self.left, self.right = left, right
        self.g0 = c.rank * n_own - left
        self.dev.set_tree(tree)
        self.dev.set_model(model)
        self.dev.upload_paths(fp_loc, capacity, self.g0, n_global)
        self.dev.set_halo(left, right)
        self.refresh_halos()

    def refresh_halos(self):
        """ship my H outermost owned columns to each neighbour; take theirs as my halos"""
        if self.comm.world == 1:
            return
        if hasattr(self.dev, "capacity"):
            # packed columns have capacity-dependent size: agree on the widest
            caps = [int(x[0]) for x in self.comm.allgather(np.array([float(self.dev.capacity())]))]
            if max(caps) != self.dev.capacity():
                self.dev.set_capacity(max(caps))
        H = self.halo
        to_left = self.dev.get_columns(self.left, H) if self.left else None
        to_right = self.dev.get_columns(self.n_loc - self.right - H, H) if self.right else None
        from_left, from_right = self.comm.exchange(to_left, to_right)
        if from_left is not None:
            self.dev.put_columns(0, H, from_left)
        if from_right is not None:
            self.dev.put_columns(self.n_loc - H, H, from_right)
        self.dev.set_halo(self.left, self.right)     # marks the halos fresh

    # ---- SingleSiteSampler interface
    def set_model(self, model):
        self.dev.set_model(model)

    def reset(self):
        """refresh the halos (they are stale after the previous run_mcmc), then cache the
        triple log-likelihoods as SingleSiteSampler::reset does"""
        self.refresh_halos()
        self.dev.reset()

    def sweeps(self, n_sweeps, seed, sweep_base=0):
        """n plain sweeps (the epievo_sim_pairwise loop), refreshing halos as needed"""
        nacc, done = 0, 0
        while done < n_sweeps:
            k = min(n_sweeps - done, self.dev.halo_phases_left() // 3) if self.comm.world > 1 \
                else n_sweeps - done
            if k == 0:
                self.refresh_halos()
                self.dev.reset()
                continue
            nacc += self.dev.sweep(k, seed, sweep_base + done)
            done += k
        return nacc

    def run_mcmc(self, burn_in, batch, seed, sweep_base=0):
        """-> (J, D, acc_rate): batch averages over the WHOLE genome, identical on every rank"""
        if self.comm.world > 1 and self.dev.halo_phases_left() < 3 * (burn_in + batch):
            raise RuntimeError("halo too narrow for %d sweeps: call reset() first or set up with "
                               "a larger sweeps_per_refresh" % (burn_in + batch))
        if self.comm.world == 1:
            J, D, nacc = self.dev.run_mcmc(burn_in, batch, seed, sweep_base)
        else:
            # shards return batch SUMS; the one exchange per EM iteration adds them in rank
            # order ([J | D | n_acc]; J and n_acc are integers, hence exact) and the division
            # by the batch size happens once, as in the unsharded run
            J, D, nacc = self.dev.run_mcmc(burn_in, batch, seed, sweep_base, average=False)
            parts = self.comm.allgather(np.concatenate([J, D, [float(nacc)]]))
            tot = np.zeros_like(parts[0])
            for p in parts:
                tot = tot + p
            J, D, nacc = (tot[:self.B * 8] / float(batch), tot[self.B * 8:self.B * 16] / float(batch),
                          tot[-1])
        return J, D, nacc / float(batch * (self.n_global - 2))

    def scale_jump_times(self, new_branches):
        self.dev.scale_jump_times(new_branches)

    def owned_paths(self):
        return self.dev.paths().slice_sites(self.left, self.n_loc - self.right)


class LocalGroup:
    """Two or three shards on ONE GPU behind the DeviceSampler interface.

    The three kernels of a colour phase depend on each other, so on one stream every launch
    pays its own ramp and tail.  Two contexts on the same device, each owning half of the
    (local) genome plus redundant halos exactly like shards on different GPUs, run on their own
    streams from their own host threads and fill each other's gaps: +17 % on one MI355X
    (tools/probe_streams.py).  Unlike shards on different GPUs, the group reproduces the
    single-context run bit-for-bit INCLUDING D: the shards own whole 256-site blocks of the
    canonical reduction tree (cut points and the internal halo width are multiples of 256)
    and write their level-0 block partials into one shared buffer that is reduced once
    (epv_run_mcmc_blocks / epv_reduce_blocks).  Drop-in for DeviceSampler inside
    ShardedSampler, so it composes with the multi-GPU sharding."""

    H_INT = 512        # internal halo columns: >= 6 * 60 + 2 and a multiple of 256
    BLOCK = 256

    def __init__(self, device=0, shards=2):
        from concurrent.futures import ThreadPoolExecutor
        from .sampler import DeviceSampler
        self.device, self.k_req = device, max(1, int(shards))
        self.subs = [DeviceSampler(device) for _ in range(self.k_req)]
        self.pool = ThreadPoolExecutor(max_workers=self.k_req)
        self.n_sites = self.n_nodes = self.B = 0
        self.capacity_events = []
        self._auto_grow = False
        self._blocks, self._blocks_batch = None, 0
        self.outer = (0, 0)
        self.halo_mode = False

    # ---- plumbing
    @property
    def auto_grow(self):
        return self._auto_grow

    @auto_grow.setter
    def auto_grow(self, v):
        self._auto_grow = bool(v)
        for s in self.subs:
            s.auto_grow = bool(v)

    def close(self):
        if getattr(self, "_closed", False):
            return
        self._closed = True
        try:
            if self._blocks is not None and self.subs and self.subs[0].h:
                self.subs[0].dev_free(self._blocks)
        finally:
            self._blocks = None
            for s in self.subs:
                s.close()
            self.pool.shutdown(wait=True)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _each(self, fn):
        """run fn(j, sub) for every shard from its own host thread; results in shard order"""
        if len(self.subs) == 1:
            return [fn(0, self.subs[0])]
        return [f.result() for f in [self.pool.submit(fn, j, s) for j, s in enumerate(self.subs)]]

    def set_tree(self, tree):
        self.n_nodes, self.B = tree.n_nodes, tree.n_nodes - 1
        for s in self.subs:
            s.set_tree(tree)

    def set_model(self, model):
        for s in self.subs:
            s.set_model(model)

    def upload_paths(self, fp, capacity=0, global_site_offset=0, n_global=None):
        n, H, Q = fp.n_sites, self.H_INT, self.BLOCK
        k = self.k_req
        while k > 1 and n < k * (2 * H + 2 * Q):      # every shard must own more than its halos
            k -= 1
        for s in self.subs[k:]:
            s.close()
        self.subs = self.subs[:k]
        cuts = [0] + [int(round(j * n / float(k) / Q)) * Q for j in range(1, k)] + [n]
        self.a, self.b = cuts[:-1], cuts[1:]                       # pieces of [0, n)
        self.lo = [a - (H if j > 0 else 0) for j, a in enumerate(self.a)]
        self.hi = [b + (H if j < k - 1 else 0) for j, b in enumerate(self.b)]
        if capacity == 0:
            capacity = int(max(16, 2 * (fp.counts().max() if n else 0) + 8))
        n_global = global_site_offset + n if n_global is None else n_global
        for j, s in enumerate(self.subs):
            s.upload_paths(fp.slice_sites(self.lo[j], self.hi[j]) if k > 1 else fp, capacity,
                           global_site_offset + self.lo[j], n_global)
        self.n_sites = n
        self.halo_mode = False
        self.outer = (0, 0)
        self._drop_blocks()     # the halos are set by set_halo() (ShardedSampler) or by reset()

    def _drop_blocks(self):
        if self._blocks is not None:
            self.subs[0].dev_free(self._blocks)
        self._blocks, self._blocks_batch = None, 0

    # ---- halos
    def set_halo(self, left, right):
        """outer halo blocks of the whole group (multi-GPU); the internal ones are managed here"""
        if (left, right) != self.outer:
            self._drop_blocks()          # ownership of the edge blocks changed
        self.outer, self.halo_mode = (left, right), True
        k, H = len(self.subs), self.H_INT
        for j, s in enumerate(self.subs):
            s.set_halo(left if j == 0 else H, right if j == k - 1 else H)

    def halo_phases_left(self):
        return min(s.halo_phases_left() for s in self.subs)

    def _locate(self, first, count):
        for j in range(len(self.subs)):
            if self.a[j] <= first < self.b[j]:
                if first + count > self.hi[j]:
                    raise ValueError("column range straddles two shards of the group")
                return j, first - self.lo[j]
        raise ValueError("bad column range")

    def column_bytes(self):
        return self.subs[0].column_bytes()

    def get_columns(self, first, count):
        j, f = self._locate(first, count)
        return self.subs[j].get_columns(f, count)

    def put_columns(self, first, count, buf):
        j, f = self._locate(first, count)
        self.subs[j].put_columns(f, count, buf)

    def _refresh_internal(self):
        k, H = len(self.subs), self.H_INT
        cap = max(s.capacity() for s in self.subs)
        for s in self.subs:
            if s.capacity() != cap:
                s.set_capacity(cap)
        for j in range(k - 1):
            L, R = self.subs[j], self.subs[j + 1]
            L.copy_columns_to(self.b[j] - H - self.lo[j], H, R, 0)          # L's edge -> R's left halo
            R.copy_columns_to(H, H, L, self.b[j] - self.lo[j])              # R's edge -> L's right halo
        if k > 1:
            self.set_halo(*self.outer)                                      # marks them fresh

    # ---- the SingleSiteSampler surface
    def reset(self):
        self._refresh_internal()
        self._each(lambda j, s: s.reset())

    def sweep(self, n_sweeps, seed, sweep_base=0):
        if len(self.subs) > 1 and self.halo_phases_left() < 3 * n_sweeps:
            done = 0
            nacc = 0
            while done < n_sweeps:
                kk = min(n_sweeps - done, self.halo_phases_left() // 3)
                if kk == 0:
                    self.reset()
                    continue
                nacc += sum(self._each(lambda j, s: s.sweep(kk, seed, sweep_base + done)))
                done += kk
            return nacc
        return sum(self._each(lambda j, s: s.sweep(n_sweeps, seed, sweep_base)))

    def run_mcmc(self, burn_in, batch, seed, sweep_base=0, average=True):
        if len(self.subs) == 1:
            return self.subs[0].run_mcmc(burn_in, batch, seed, sweep_base, average)
        if self.halo_phases_left() < 3 * (burn_in + batch):
            raise RuntimeError("internal halo of %d columns is too narrow for %d sweeps without a "
                               "reset()" % (self.H_INT, burn_in + batch))
        nb_total = (self.n_sites + self.BLOCK - 1) // self.BLOCK
        V = self.B * 16
        if self._blocks is None or self._blocks_batch < batch:
            self._drop_blocks()
            self._blocks = self.subs[0].dev_alloc(batch * nb_total * V * 8)
            self._blocks_batch = batch
        nacc = sum(self._each(lambda j, s: s.run_mcmc_blocks(burn_in, batch, seed, sweep_base, self._blocks,
                                                             nb_total, self.lo[j] // self.BLOCK)))
        for s in self.subs:
            self.capacity_events += s.capacity_events
            s.capacity_events = []
        J, D = self.subs[0].reduce_blocks(self._blocks, nb_total, batch, average)
        return J, D, nacc

    def scale_jump_times(self, new_branches):
        for s in self.subs:
            s.scale_jump_times(new_branches)

    def _owned_slices(self):
        k = len(self.subs)
        return [(0 if j == 0 else self.a[j] - self.lo[j], self.b[j] - self.lo[j]) for j in range(k)]

    def paths(self):
        if len(self.subs) == 1:
            return self.subs[0].paths()
        ps = self._each(lambda j, s: s.paths())
        return concat_sites([p.slice_sites(lo, hi) for p, (lo, hi) in zip(ps, self._owned_slices())])

    def tri_llh(self):
        ts = [s.tri_llh() for s in self.subs]
        return np.concatenate([t[lo:hi] for t, (lo, hi) in zip(ts, self._owned_slices())])

    def capacity(self):
        return max(s.capacity() for s in self.subs)

    def set_capacity(self, capacity):
        for s in self.subs:
            s.set_capacity(capacity)

    def counters(self):
        out = {}
        for s in self.subs:
            for key, v in s.counters().items():
                out[key] = out.get(key, 0) + v
        out["sweeps"] = self.subs[0].counters()["sweeps"]
        return out

    def set_timing(self, on):
        for s in self.subs:
            s.set_timing(on)

    def kernel_time_ms(self):
        """launch-weighted mean duration of the shards' colour-phase launches, and their number"""
        tot, n = 0.0, 0
        for s in self.subs:
            ms, nl = s.kernel_time_ms()
            tot += ms * nl
            n += nl
        return (tot / n if n else 0.0), n
