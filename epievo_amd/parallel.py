"""Site-sharded multi-GPU driver for launchers that start ONE PROCESS PER GPU (torchrun:
bench.py, the multi-process tests); the C++ EM driver shards inside one process instead
(epievo_amd/csrc/host/epv_sampler.cpp, RCCL linked directly).  Both stand on the same C-ABI
primitives and the same layout rules.

Contiguous shards of the genome with WIDE halos that are updated redundantly, refreshed once
per run_mcmc, and ONE all-gather of the per-branch J/D rows (+ accept count) per run_mcmc.

Why it is correct: one MH update of site i reads the paths of sites
i-2..i+2 and the cached triple log-likelihoods tri[i-1], tri[i+1]; it writes path i and
tri[i-1..i+1].  The RNG and the 3-colouring are keyed by the GLOBAL site index, so a
rank that holds copies of a neighbour's edge columns can update them itself and obtain
exactly what the owner computes.  Each colour phase, the two outermost still-valid halo
columns at every shard-internal edge lose a neighbour and go stale, so a halo of H
columns lasts H/2 phases = H/6 sweeps.  With H >= 6*(burn_in + batch) + 2 a whole
run_mcmc needs NO communication inside it: the halos are refreshed once before
reset(), and the statistics are combined once after it -- instead of 3 exchanges per
sweep.  The redundant work is 2H/n of a shard (0.1 % at n = 1e6, -L 10 -B 50).

Statistics.  Shards are cut on multiples of ROW = 256 * row_blocks sites.  Every rank reduces
its 256-site block partials to rows of ROW sites on its GPU, the rows of all ranks are
all-gathered (device buffers handed to RCCL as they are), and every rank sums all rows of
the genome.  Each stage adds aligned subtrees of ONE balanced binary tree over the global
site index, so a sharded run reproduces the unsharded one bit-for-bit on paths, states, J
AND D, for any number of ranks.

The reference has no parallelism at all (single-threaded); this
module is new capability, not a translation.
"""
import numpy as np

from .host import FlatPaths

BLOCK = 256      # sites per level-0 block of the statistics tree
TAIL = 8         # doubles appended to a rank's rows in the all-gather (accept count)


def halo_width(sweeps_per_refresh):
    """halo columns that last `sweeps_per_refresh` sweeps, in whole 256-site blocks"""
    return max(BLOCK, -(-(6 * int(sweeps_per_refresh) + 2) // BLOCK) * BLOCK)


def shard_cuts(n_global, world, row_blocks=64):
    """cut points of `world` near-equal contiguous shards on whole statistics rows"""
    row = BLOCK * row_blocks
    cuts = [0] + [int(r * n_global / float(world) / row + 0.5) * row for r in range(1, world)] + [n_global]
    if any(b <= a for a, b in zip(cuts[:-1], cuts[1:])):
        raise ValueError("a genome of %d sites is too short for %d shards on %d-site rows" % (n_global, world, row))
    return cuts


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
    """one rank: nothing to exchange, the gathered buffer IS the rank's piece"""
    rank, world = 0, 1

    def exchange(self, dev, send_left, recv_left, send_right, recv_right):
        pass

    def all_gather(self, dev, piece, gathered):
        assert gathered is piece


class TorchComm:
    """torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU
    tests).  The buffers are the device's own (DevBuf: device memory seen by torch through the
    CUDA array interface; the CPU double hands out numpy arrays), so nothing is staged: RCCL
    reads the packed columns and the statistic rows where the kernels wrote them."""

    def __init__(self, dist, device=None):
        import torch
        self.dist, self.torch = dist, torch
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.device = device if device is not None else torch.device("cpu")
        # gloo cannot move GPU tensors: when several ranks SHARE one GPU to rehearse the N > 1
        # path on a 1-GPU box, device buffers are bounced through the host here (rehearsal only;
        # with the nccl backend RCCL reads and writes the device buffers directly)
        self.staged = self.device.type == "cuda" and dist.get_backend() == "gloo"

    def _t(self, buf):
        if hasattr(buf, "np"):                       # host buffer of the CPU device double
            return self.torch.from_numpy(buf.np)
        return self.torch.as_tensor(buf, device=self.device)   # zero-copy view of device memory

    def _sync(self):
        if self.device.type == "cuda":
            self.torch.cuda.synchronize(self.device)

    def exchange(self, dev, send_left, recv_left, send_right, recv_right):
        """swap halo buffers with the left/right neighbour (None = no neighbour on that side)"""
        dist = self.dist
        ops, back = [], []

        def add(send, recv, peer):
            ts, tr = self._t(send), self._t(recv)
            if self.staged:
                ts, dst = ts.cpu(), tr
                tr = self.torch.empty(tr.shape, dtype=tr.dtype)
                back.append((dst, tr))
            ops.append(dist.P2POp(dist.isend, ts, peer))
            ops.append(dist.P2POp(dist.irecv, tr, peer))

        if send_left is not None:
            add(send_left, recv_left, self.rank - 1)
        if send_right is not None:
            add(send_right, recv_right, self.rank + 1)
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        for dst, tmp in back:
            dst.copy_(tmp)
        self._sync()

    def all_gather(self, dev, piece, gathered):
        g, p = self._t(gathered), self._t(piece)
        if self.staged:
            outs = [self.torch.empty(p.shape, dtype=p.dtype) for _ in range(self.world)]
            self.dist.all_gather(outs, p.cpu())
            g.copy_(self.torch.cat(outs))
        elif self.device.type == "cuda":
            self.dist.all_gather_into_tensor(g, p)         # one RCCL all-gather, in place
        else:
            self.dist.all_gather(list(g.chunk(self.world)), p)
        self._sync()


class ShardedSampler:
    """SingleSiteSampler over a site-sharded genome.  `device_factory(device)` builds the
    per-rank engine (the HIP DeviceSampler / LocalGroup in the product; the tests inject an
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
        self._halo_bufs, self._halo_bytes = None, 0
        self._piece = self._gathered = None
        self._piece_batch = 0

    def owned_sites(self):
        return self.n_own - (1 if self.comm.rank == 0 else 0) - \
            (1 if self.comm.rank == self.comm.world - 1 else 0)

    def setup(self, model, tree, fp_own, cuts, capacity=16, sweeps_per_refresh=60, row_blocks=64):
        """fp_own: this rank's owned columns, sites [cuts[rank], cuts[rank+1]) of the genome
        (cuts from shard_cuts: whole statistics rows).  The halo is sized for
        `sweeps_per_refresh` sweeps between refreshes."""
        c = self.comm
        cuts = [int(x) for x in cuts]
        if len(cuts) != c.world + 1 or cuts[0] != 0:
            raise ValueError("cuts must list world + 1 cut points starting at 0")
        row = BLOCK * row_blocks
        if any(x % row for x in cuts[1:-1]):
            raise ValueError("inner cut points must be multiples of %d sites" % row)
        n_own, n_global = cuts[c.rank + 1] - cuts[c.rank], cuts[-1]
        if fp_own.n_sites != n_own:
            raise ValueError("fp_own has %d sites, the cuts give this rank %d" % (fp_own.n_sites, n_own))
        self.cuts, self.row_blocks = cuts, row_blocks
        self.n_own, self.n_global, self.B = n_own, n_global, tree.n_nodes - 1
        H = halo_width(sweeps_per_refresh) if c.world > 1 else 0
        if H > min(b - a for a, b in zip(cuts[:-1], cuts[1:])):
            raise ValueError("a shard is smaller than the %d-column halo" % H)
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
        self.g0 = cuts[c.rank] - left
        # statistics rows of every rank (all ranks compute the same table)
        self.rows_per_rank = [(-(-(b - a) // BLOCK) + row_blocks - 1) // row_blocks for a, b in zip(cuts[:-1], cuts[1:])]
        self.max_rows = max(self.rows_per_rank)
        self.dev.set_tree(tree)
        self.dev.set_model(model)
        self.dev.upload_paths(fp_loc, capacity, self.g0, n_global)
        self.dev.set_halo(left, right)
        self.refresh_halos()

    def _agree_on_capacity(self):
        """packed columns have capacity-dependent size: all ranks move to the widest"""
        if self.comm.world == 1 or not hasattr(self.dev, "capacity"):
            return
        piece, gathered = self.dev.alloc(8), self.dev.alloc(8 * self.comm.world)
        self.dev.write(piece, 0, np.array([float(self.dev.capacity())]))
        self.comm.all_gather(self.dev, piece, gathered)
        cap = int(self.dev.read(gathered, 0, self.comm.world).max())
        piece.free()
        gathered.free()
        if cap != self.dev.capacity():
            self.dev.set_capacity(cap)

    def refresh_halos(self):
        """ship my H outermost owned columns to each neighbour; take theirs as my halos"""
        if self.comm.world == 1:
            return
        self._agree_on_capacity()
        H = self.halo
        nbytes = H * self.dev.column_bytes()
        if self._halo_bufs is None or self._halo_bytes != nbytes:
            for b in self._halo_bufs or []:
                b.free()
            self._halo_bufs, self._halo_bytes = [self.dev.alloc(nbytes) for _ in range(4)], nbytes
        sl, rl, sr, rr = self._halo_bufs
        if self.left:
            self.dev.pack_columns(self.left, H, sl)
        if self.right:
            self.dev.pack_columns(self.n_loc - self.right - H, H, sr)
        self.comm.exchange(self.dev, sl if self.left else None, rl if self.left else None,
                           sr if self.right else None, rr if self.right else None)
        if self.left:
            self.dev.unpack_columns(0, H, rl)
        if self.right:
            self.dev.unpack_columns(self.n_loc - H, H, rr)
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
        """-> (J, D, acc_rate): batch averages over the WHOLE genome, identical on every rank
        and bit-identical to the unsharded run"""
        if self.comm.world > 1 and self.dev.halo_phases_left() < 3 * (burn_in + batch):
            raise RuntimeError("halo too narrow for %d sweeps: call reset() first or set up with "
                               "a larger sweeps_per_refresh" % (burn_in + batch))
        V = self.B * 16
        piece_doubles = self.max_rows * batch * V + TAIL
        if self._piece is None or self._piece_batch != batch:
            for b in {id(x): x for x in (self._piece, self._gathered) if x is not None}.values():
                b.free()
            self._piece = self.dev.alloc(piece_doubles * 8)
            self._gathered = self.dev.alloc(piece_doubles * 8 * self.comm.world) if self.comm.world > 1 \
                else self._piece
            self._piece_batch = batch
        # this rank's rows of the statistics tree stay on the device; the accept count rides in
        # the tail of the same piece, so ONE collective per EM iteration carries everything
        nacc = self.dev.run_mcmc_rows(burn_in, batch, seed, sweep_base, self.row_blocks, self._piece)
        self.dev.write(self._piece, (piece_doubles - TAIL) * 8, np.array([float(nacc)]))
        self.comm.all_gather(self.dev, self._piece, self._gathered)
        J, D = self.dev.reduce_gathered_rows(self._gathered, self.max_rows, piece_doubles, self.rows_per_rank,
                                             batch, True)
        nacc = sum(float(self.dev.read(self._gathered, ((r + 1) * piece_doubles - TAIL) * 8, 1)[0])
                   for r in range(self.comm.world))
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

    BLOCK = 256

    def __init__(self, device=0, shards=2, sweeps_per_refresh=60):
        from concurrent.futures import ThreadPoolExecutor
        from .sampler import DeviceSampler
        self.device, self.k_req = device, max(1, int(shards))
        self.H_INT = halo_width(sweeps_per_refresh)   # internal halo columns (whole blocks)
        self.subs = [DeviceSampler(device) for _ in range(self.k_req)]
        self.pool = ThreadPoolExecutor(max_workers=self.k_req)
        self.n_sites = self.n_nodes = self.B = 0
        self.capacity_events = []
        self._auto_grow = False
        self._blocks, self._blocks_shape = None, None
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
                self._blocks.free()
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
            self._blocks.free()
        self._blocks, self._blocks_shape = None, None

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

    # buffers live on the group's GPU; any context can allocate and fill them
    def alloc(self, nbytes):
        return self.subs[0].alloc(nbytes)

    def write(self, buf, offset, arr):
        self.subs[0].write(buf, offset, arr)

    def read(self, buf, offset, count, dtype=np.float64):
        return self.subs[0].read(buf, offset, count, dtype)

    def pack_columns(self, first, count, buf):
        j, f = self._locate(first, count)
        self.subs[j].pack_columns(f, count, buf)

    def unpack_columns(self, first, count, buf):
        j, f = self._locate(first, count)
        self.subs[j].unpack_columns(f, count, buf)

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

    def _sweep_all(self, k, seed, sweep_base):
        nacc = sum(self._each(lambda j, s: s.sweep(k, seed, sweep_base)))
        # the shards update their shared halo columns redundantly: after an absorbed overflow
        # they must go on proposing under ONE capacity, or one accepts what the other rejects
        if len(self.subs) > 1 and len({s.capacity() for s in self.subs}) > 1:
            self.set_capacity(max(s.capacity() for s in self.subs))
        for s in self.subs:
            self.capacity_events += s.capacity_events
            s.capacity_events = []
        return nacc

    def sweep(self, n_sweeps, seed, sweep_base=0):
        if len(self.subs) > 1 and self.halo_phases_left() < 3 * n_sweeps:
            done = 0
            nacc = 0
            while done < n_sweeps:
                kk = min(n_sweeps - done, self.halo_phases_left() // 3)
                if kk == 0:
                    self.reset()
                    continue
                nacc += self._sweep_all(kk, seed, sweep_base + done)
                done += kk
            return nacc
        return self._sweep_all(n_sweeps, seed, sweep_base)

    def owned_blocks(self):
        """(first local 256-site block with owned columns, number of such blocks)"""
        left, right = self.outer
        b0 = left // self.BLOCK
        return b0, (self.n_sites - right + self.BLOCK - 1) // self.BLOCK - b0

    def _run_blocks(self, burn_in, batch, seed, sweep_base):
        """all shards' run_mcmc with the level-0 partials of the group's owned blocks in ONE
        device buffer [batch][n_blocks][16 B] -> (accepted, n_blocks)"""
        if self.halo_phases_left() < 3 * (burn_in + batch) and len(self.subs) > 1:
            raise RuntimeError("internal halo of %d columns is too narrow for %d sweeps without a "
                               "reset()" % (self.H_INT, burn_in + batch))
        if self.outer[0] % self.BLOCK:
            raise ValueError("the group's left halo must be a whole number of 256-site blocks")
        b0, nb = self.owned_blocks()
        shape = (batch, nb, self.B)
        if self._blocks is None or self._blocks_shape != shape:
            self._drop_blocks()
            self._blocks, self._blocks_shape = self.subs[0].alloc(batch * nb * self.B * 16 * 8), shape
        nacc = sum(self._each(lambda j, s: s.run_mcmc_blocks(burn_in, batch, seed, sweep_base, self._blocks.p,
                                                             nb, self.lo[j] // self.BLOCK - b0)))
        for s in self.subs:
            self.capacity_events += s.capacity_events
            s.capacity_events = []
        return nacc, nb

    def run_mcmc(self, burn_in, batch, seed, sweep_base=0, average=True):
        if len(self.subs) == 1 and not self.halo_mode:
            return self.subs[0].run_mcmc(burn_in, batch, seed, sweep_base, average)
        nacc, nb = self._run_blocks(burn_in, batch, seed, sweep_base)
        J, D = self.subs[0].reduce_blocks(self._blocks.p, nb, batch, average)
        return J, D, nacc

    def run_mcmc_rows(self, burn_in, batch, seed, sweep_base, row_blocks, rows_buf):
        """as DeviceSampler.run_mcmc_rows, the group's shards writing one block buffer"""
        nacc, nb = self._run_blocks(burn_in, batch, seed, sweep_base)
        self.subs[0].blocks_to_rows(self._blocks.p, nb, batch, row_blocks, rows_buf.ptr)
        return nacc

    def reduce_gathered_rows(self, gathered_buf, max_rows, piece_doubles, rows_per_rank, batch, average=True):
        return self.subs[0].reduce_gathered_rows(gathered_buf, max_rows, piece_doubles, rows_per_rank, batch, average)

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

    def set_options(self, **kw):
        for s in self.subs:
            s.set_options(**kw)

    def phase_mode(self):
        """kernels of a colour phase (DeviceSampler.phase_mode) -- of the largest shard"""
        return min(s.phase_mode() for s in self.subs)

    def kernel_time_ms(self):
        """launch-weighted mean duration of the shards' colour-phase launches, and their number"""
        tot, n = 0.0, 0
        for s in self.subs:
            ms, nl = s.kernel_time_ms()
            tot += ms * nl
            n += nl
        return (tot / n if n else 0.0), n
