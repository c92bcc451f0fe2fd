"""An oracle-backed stand-in for epievo_amd.sampler.DeviceSampler, used ONLY by the CPU
tests of the sharding logic (epievo_amd/parallel.py) under gloo.  It mirrors, in Python,
the halo bookkeeping that epv_abi.hip does for the product (phase_range/owned_range)."""
import ctypes as C

import numpy as np

import orc
from epievo_amd.host import FlatPaths


class HostBuf:
    """what OracleDevice.alloc hands out: host memory (the product's DevBuf is device memory)"""

    def __init__(self, nbytes):
        self.np = np.zeros(int(nbytes), np.uint8)
        self.nbytes = int(nbytes)

    def free(self):
        pass


def rows_total(rows):
    """integer sum over axis 0 (the statistics are exact int64 sums: any order gives the same
    bits; numpy restatement of the last stage of the device reduction)"""
    return rows.sum(axis=0, dtype=np.int64)


class OracleDevice:
    def __init__(self, device=0):
        self.o = None
        self.tree = self.model = None

    def set_tree(self, tree):
        self.tree = tree
        self.B = tree.n_nodes - 1

    def set_model(self, model):
        self.model = model
        if self.o is not None:
            self.o.set_model(model)

    def upload_paths(self, fp, capacity=16, global_site_offset=0, n_global=None):
        self.cap, self.n = capacity, fp.n_sites
        self.g0 = global_site_offset
        self.n_global = n_global if n_global is not None else global_site_offset + fp.n_sites
        self.o = orc.Oracle(self.tree, self.model, fp, "B", cap=capacity)
        self.o.L.orc_set_shard(self.o.h, self.g0, self.n_global)
        self.left = self.right = 0
        self.used = 0
        self.halo_mode = False

    def set_halo(self, left, right):
        self.left, self.right, self.used, self.halo_mode = left, right, 0, True

    def halo_phases_left(self):
        hs = [h for h in (self.left, self.right) if h]
        return (min(hs) // 2 - self.used) if hs else (1 << 62)

    def _owned(self):
        lo = self.left if self.left else 1
        hi = self.n - self.right - 1 if self.right else self.n - 2
        return lo, hi

    def _phase_range(self):
        shrink = 2 * (self.used + 1)
        lo = shrink if self.left else 1
        hi = self.n - 1 - shrink if self.right else self.n - 2
        assert (not self.left or lo <= self.left) and (not self.right or hi + self.right >= self.n - 1)
        return lo, hi

    def reset(self):
        self.o.reset()

    def _sweep(self, seed, sweep):
        self.o.seed(seed)
        nacc = 0
        olo, ohi = self._owned()
        for colour in range(3):
            lo, hi = self._phase_range()
            nacc += int(self.o.L.orc_sweep_phase(self.o.h, colour, sweep, lo, hi, olo, ohi))
            self.used += 1
        return nacc

    def sweep(self, n_sweeps, seed, sweep_base=0):
        return sum(self._sweep(seed, sweep_base + w) for w in range(n_sweeps))

    def suffstats(self):
        J, D = np.zeros(self.B * 8), np.zeros(self.B * 8)
        lo, hi = self._owned()
        self.o.L.orc_suffstats_range(self.o.h, lo, hi, orc._p(J, C.c_double), orc._p(D, C.c_double))
        return J, D

    def run_mcmc(self, burn_in, batch, seed, sweep_base=0, average=True):
        w = sweep_base
        for _ in range(burn_in):
            self._sweep(seed, w)
            w += 1
        J, D, nacc = np.zeros(self.B * 8), np.zeros(self.B * 8), 0
        for _ in range(batch):
            nacc += self._sweep(seed, w)
            w += 1
            J1, D1 = self.suffstats()
            J += J1
            D += D1
        if average:
            return J / float(batch), D / float(batch), nacc
        return J, D, nacc

    def scale_jump_times(self, nb):
        self.o.scale_jump_times(nb)

    def paths(self):
        return self.o.paths()

    # ---- device-style buffers
    def alloc(self, nbytes):
        return HostBuf(nbytes)

    def write(self, buf, offset, arr):
        a = np.ascontiguousarray(arr).view(np.uint8).reshape(-1)
        buf.np[offset:offset + a.size] = a

    def read(self, buf, offset, count, dtype=np.float64):
        return buf.np[offset:offset + count * np.dtype(dtype).itemsize].view(dtype).copy()

    def capacity(self):
        return self.cap

    # ---- columns: [count][B] init u8 | [count][B] cnt u32 | [count][B][cap] f64
    def column_bytes(self):
        return self.B * (1 + 4 + 8 * self.cap)

    def pack_columns(self, first, count, buf):
        sub = self.o.paths().slice_sites(first, first + count)
        B, cap = self.B, self.cap
        init = sub.init.reshape(B, count).T.copy()
        cnt = sub.counts().reshape(B, count).T.astype(np.uint32).copy()
        jp = np.zeros((count, B, cap))
        off = sub.offsets[:-1].reshape(B, count)
        for b in range(B):
            for s in range(count):
                k = cnt[s, b]
                jp[s, b, :k] = sub.jumps[int(off[b, s]):int(off[b, s]) + k]
        packed = np.concatenate([init.reshape(-1).view(np.uint8), cnt.reshape(-1).view(np.uint8),
                                 jp.reshape(-1).view(np.uint8)])
        buf.np[:packed.size] = packed

    def unpack_columns(self, first, count, buf):
        B, cap = self.B, self.cap
        raw = buf.np[:count * self.column_bytes()]
        init = raw[:count * B].reshape(count, B)
        cnt = raw[count * B:count * B * 5].view(np.uint32).reshape(count, B)
        jp = raw[count * B * 5:].view(np.float64).reshape(count, B, cap)
        for s in range(count):
            js = np.concatenate([jp[s, b, :cnt[s, b]] for b in range(B)] + [np.zeros(1)])
            i8 = np.ascontiguousarray(init[s])
            c32 = np.ascontiguousarray(cnt[s])
            self.o.L.orc_set_site(self.o.h, first + s, orc._p(i8, C.c_uint8),
                                  orc._p(c32, C.c_uint32), orc._p(js, C.c_double))

    # ---- statistics rows (the product: epv_run_mcmc_blocks + epv_blocks_to_rows)
    def run_mcmc_rows(self, burn_in, batch, seed, sweep_base, row_blocks, rows_buf):
        assert self.left % 256 == 0
        w = sweep_base
        for _ in range(burn_in):
            self._sweep(seed, w)
            w += 1
        lo, hi = self._owned()
        nb = (self.n - self.right + 255) // 256 - self.left // 256
        n_rows = (nb + row_blocks - 1) // row_blocks
        V = self.B * 16
        rows = rows_buf.np[:n_rows * batch * V * 8].view(np.int64).reshape(n_rows, batch, V)
        one = np.zeros((n_rows, V), np.int64)
        nacc = 0
        for i in range(batch):
            nacc += self._sweep(seed, w)
            w += 1
            self.o.L.orc_suffstats_rows(self.o.h, self.left, 256 * row_blocks, n_rows, lo, hi,
                                        orc._p(one, C.c_int64))
            rows[:, i, :] = one
        return nacc

    def reduce_gathered_rows(self, gathered_buf, max_rows, piece_doubles, rows_per_rank, batch, average=True):
        V = self.B * 16
        g = gathered_buf.np.view(np.int64)
        rows = np.concatenate([g[r * piece_doubles:r * piece_doubles + k * batch * V].reshape(k, batch, V)
                               for r, k in enumerate(rows_per_rank)], axis=0)
        tot = rows_total(rows).reshape(batch, self.B, 16)            # [batch][B][16] integers
        scale = np.zeros(self.B + 1)
        self.o.L.orc_stat_scales(self.o.h, orc._p(scale, C.c_double))
        one = tot.astype(np.float64)       # int64 -> double rounds to nearest even, as C does
        one[:, :, 8:] *= (1.0 / scale[1:])[None, :, None]            # 2^-k: exact
        one = one.reshape(batch, V)
        acc = np.zeros(V)
        for i in range(batch):             # the sequential accumulation of run_mcmc
            acc = acc + one[i]
        if average:
            acc = acc / float(batch)
        acc = acc.reshape(self.B, 16)
        return acc[:, :8].reshape(-1).copy(), acc[:, 8:].reshape(-1).copy()

    def set_timing(self, on):
        pass

    def kernel_time_ms(self):
        return 0.0, 0
