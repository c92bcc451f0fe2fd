"""ctypes binding of libepievo_mi355x.so (the C ABI of include/epievo_mi355x.h) and a
Python mirror of the reference's SingleSiteSampler interface on top of it
(/root/reference/src/libepievo/SingleSiteSampler.hpp:35-81: ctor(burn_in, batch),
reset(model, paths), run_mcmc(...) -> J, D, acc_rate).

There is no CPU fallback: if the HIP library is missing or no GPU can be opened,
construction raises."""
import ctypes as C
import os

import numpy as np

from . import _build
from .host import FlatPaths

_lib = None

EPV_OK, EPV_ERR_ARG, EPV_ERR_HIP, EPV_ERR_CAPACITY, EPV_ERR_STATE = 0, 1, 2, 3, 4


class EpvError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("epv error %d: %s" % (code, msg))
        self.code = code


class CapacityError(EpvError):
    pass


class _Counters(C.Structure):
    _fields_ = [("n_overflow", C.c_uint64), ("n_coop_tasks", C.c_uint64),
                ("n_sweeps", C.c_uint64), ("reserved", C.c_uint64)]


MAX_CAPACITY = 2047    # EPV_MAX_CAP: jump slots per (site, branch)

ABI_SYMBOLS = [
    "epv_create", "epv_destroy", "epv_last_error", "epv_set_tree", "epv_set_model",
    "epv_upload_paths", "epv_set_capacity", "epv_get_capacity", "epv_init_paths_indep", "epv_indep_expectation",
    "epv_indep_sufficient_statistics", "epv_indep_update_paths", "epv_set_global_length", "epv_set_update_range", "epv_set_halo",
    "epv_halo_phases_left", "epv_reset", "epv_reset_async", "epv_sweep",
    "epv_sweep_phase", "epv_run_mcmc", "epv_run_mcmc_sums", "epv_get_sufficient_statistics", "epv_scale_jump_times",
    "epv_paths_total_jumps", "epv_download_paths", "epv_get_tri_llh", "epv_column_bytes",
    "epv_get_columns", "epv_put_columns", "epv_copy_columns", "epv_dev_alloc", "epv_dev_free",
    "epv_run_mcmc_blocks", "epv_reduce_blocks", "epv_get_counters", "epv_kernel_time_ms",
    "epv_set_timing", "epv_pack_columns_dev", "epv_unpack_columns_dev", "epv_device_of",
    "epv_blocks_to_rows", "epv_reduce_rows", "epv_reduce_gathered_rows", "epv_dev_write", "epv_dev_read", "epv_set_options", "epv_get_options", "epv_phase_mode",
    "epv_forward_simulate", "epv_forward_last_ms", "epv_copy_columns_async",
]


def lib():
    """Load the HIP library; raise (never fall back) when it has not been built."""
    global _lib
    if _lib is None:
        # EPIEVO_MI355X_LIB: another build of the same library (A/B runs of kernel variants)
        path = os.environ.get("EPIEVO_MI355X_LIB", _build.HIP_SO)
        if not os.path.exists(path):
            raise RuntimeError("%s is missing: run `python -c 'import __graft_entry__ as g; "
                               "g.build()'` (hipcc --offload-arch=gfx950)" % path)
        L = C.CDLL(path)
        dp, u8p, u32p, u64p, vp = (C.POINTER(C.c_double), C.POINTER(C.c_uint8),
                                   C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), C.c_void_p)
        L.epv_create.argtypes = [C.c_int]
        L.epv_create.restype = vp
        L.epv_destroy.argtypes = [vp]
        L.epv_destroy.restype = None
        L.epv_last_error.argtypes = [vp]
        L.epv_last_error.restype = C.c_char_p
        L.epv_set_tree.argtypes = [vp, C.c_int, u32p, u32p, dp]
        L.epv_set_model.argtypes = [vp, dp, dp]
        L.epv_upload_paths.argtypes = [vp, C.c_uint64, u8p, u64p, dp, C.c_uint32, C.c_uint64]
        L.epv_set_capacity.argtypes = [vp, C.c_uint32]
        L.epv_get_capacity.argtypes = [vp, C.POINTER(C.c_uint32)]
        L.epv_init_paths_indep.argtypes = [vp, C.c_uint64, u8p, u8p, C.c_uint64, C.c_uint32]
        L.epv_indep_expectation.argtypes = [vp, dp, dp, dp]
        L.epv_indep_sufficient_statistics.argtypes = [vp, dp, dp]
        L.epv_indep_update_paths.argtypes = [vp, dp, C.c_uint64, C.c_uint32]
        L.epv_set_global_length.argtypes = [vp, C.c_uint64]
        L.epv_set_update_range.argtypes = [vp, C.c_uint64, C.c_uint64]
        L.epv_set_halo.argtypes = [vp, C.c_uint64, C.c_uint64]
        L.epv_halo_phases_left.argtypes = [vp, u64p]
        L.epv_reset.argtypes = [vp]
        L.epv_sweep.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_uint32, u64p]
        L.epv_sweep_phase.argtypes = [vp, C.c_int, C.c_uint64, C.c_uint32, u64p]
        L.epv_run_mcmc.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, dp, dp, u64p]
        L.epv_run_mcmc_sums.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_int, dp, dp, u64p]
        L.epv_get_sufficient_statistics.argtypes = [vp, dp, dp]
        L.epv_scale_jump_times.argtypes = [vp, dp]
        L.epv_paths_total_jumps.argtypes = [vp, u64p]
        L.epv_download_paths.argtypes = [vp, u8p, u64p, dp]
        L.epv_get_tri_llh.argtypes = [vp, dp]
        L.epv_column_bytes.argtypes = [vp]
        L.epv_column_bytes.restype = C.c_uint64
        L.epv_get_columns.argtypes = [vp, C.c_uint64, C.c_uint64, vp]
        L.epv_put_columns.argtypes = [vp, C.c_uint64, C.c_uint64, vp]
        L.epv_copy_columns.argtypes = [vp, C.c_uint64, C.c_uint64, vp, C.c_uint64]
        L.epv_dev_alloc.argtypes = [vp, C.c_uint64, C.POINTER(vp)]
        L.epv_dev_free.argtypes = [vp, vp]
        L.epv_run_mcmc_blocks.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, vp, C.c_uint64,
                                          C.c_int64, u64p]
        L.epv_reduce_blocks.argtypes = [vp, vp, C.c_uint64, C.c_uint64, C.c_int, dp, dp]
        L.epv_pack_columns_dev.argtypes = [vp, C.c_uint64, C.c_uint64, vp]
        L.epv_unpack_columns_dev.argtypes = [vp, C.c_uint64, C.c_uint64, vp]
        L.epv_device_of.argtypes = [vp]
        L.epv_blocks_to_rows.argtypes = [vp, vp, C.c_uint64, C.c_uint64, C.c_uint32, vp]
        L.epv_reduce_rows.argtypes = [vp, vp, C.c_uint64, C.c_uint64, C.c_int, dp, dp]
        L.epv_reduce_gathered_rows.argtypes = [vp, vp, C.c_uint32, C.c_uint64, C.c_uint64, u64p, C.c_uint64, C.c_int,
                                               dp, dp]
        L.epv_set_options.argtypes = [vp, C.c_uint32]
        L.epv_forward_simulate.argtypes = [vp, C.c_uint64, C.POINTER(C.c_uint8), C.c_uint64, C.c_uint32,
                                           C.POINTER(C.c_uint64)]
        L.epv_forward_last_ms.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.epv_get_options.argtypes = [vp, C.POINTER(C.c_uint32)]
        L.epv_phase_mode.argtypes = [vp, C.POINTER(C.c_uint32)]
        L.epv_dev_write.argtypes = [vp, vp, vp, C.c_uint64]
        L.epv_dev_read.argtypes = [vp, vp, vp, C.c_uint64]
        L.epv_get_counters.argtypes = [vp, C.POINTER(_Counters)]
        L.epv_kernel_time_ms.argtypes = [vp, dp, u64p]
        L.epv_set_timing.argtypes = [vp, C.c_int]
        _lib = L
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


class DevBuf:
    """Zero-filled device memory on a context's GPU.  torch (and anything else that speaks the
    CUDA array interface) sees it without a copy, so the halo columns and statistic rows a
    sharded run hands to RCCL never pass through the host."""

    def __init__(self, dev, nbytes):
        self.dev, self.nbytes = dev, int(nbytes)
        self.p = dev.dev_alloc(self.nbytes)
        self.ptr = int(self.p.value)

    @property
    def __cuda_array_interface__(self):
        return {"shape": (self.nbytes,), "typestr": "|u1", "data": (self.ptr, False), "version": 2,
                "strides": None}

    def free(self):
        if self.p is not None and self.dev.h:
            self.dev.dev_free(self.p)
        self.p, self.ptr = None, 0


class DeviceSampler:
    """One context = one GPU.  Thin, explicit face of the C ABI."""

    def __init__(self, device=0):
        self.L = lib()
        self.h = self.L.epv_create(device)
        if not self.h:
            raise RuntimeError("epv_create(%d) failed: no usable HIP device (this build has no "
                               "CPU fallback)" % device)
        self.n_sites = self.n_nodes = self.B = 0
        self.auto_grow = False     # True: widen the jump slots after an overflow and carry on
        self.capacity_events = []  # messages of the overflows that were absorbed
        self.halo = (0, 0)
        self._blocks, self._blocks_shape = None, None

    def close(self):
        if getattr(self, "h", None):
            if getattr(self, "_blocks", None) is not None:
                self._blocks.free()
                self._blocks = None
            self.L.epv_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def _ck(self, rc):
        if rc != EPV_OK:
            msg = self.L.epv_last_error(self.h).decode()
            raise (CapacityError if rc == EPV_ERR_CAPACITY else EpvError)(rc, msg)

    def _ck_mcmc(self, rc):
        """after an MCMC call: a capacity overflow leaves a valid chain (the over-long proposals
        were rejected) and complete outputs, so with auto_grow the slots are doubled for the
        following calls -- what the reference's std::vector paths do on their own -- and the
        call succeeds; otherwise it raises CapacityError like any other failure."""
        if rc == EPV_ERR_CAPACITY and self.auto_grow:
            msg = self.L.epv_last_error(self.h).decode()
            cap = self.capacity()
            if cap < MAX_CAPACITY:
                self.set_capacity(min(MAX_CAPACITY, 2 * cap))
                self.capacity_events.append(msg)
                return
        self._ck(rc)

    def set_options(self, reference_proposal_ratio=False, forward_rejection=False, sample_root=False):
        """see EPV_OPT_* in include/epievo_mi355x.h"""
        self._ck(self.L.epv_set_options(self.h, (1 if reference_proposal_ratio else 0) |
                                        (2 if forward_rejection else 0) | (4 if sample_root else 0)))

    PHASE_KERNELS = {
        0: "epv_mh_propose_kernel + epv_mh_jumps_kernel + epv_mh_accept_kernel",
        1: "epv_mh_propose2_kernel + epv_mh_jumps_all_kernel + epv_mh_accept_kernel",
        2: "epv_mh_propose2_kernel + epv_seg_search_kernel + epv_seg_assemble_kernel + epv_mh_accept_kernel",
        3: "epv_mh_propose2_kernel<fused>: proposal, segment search, assembly and acceptance in one launch",
        4: "epv_mh_propose3_kernel + epv_mh_jumps_all_kernel + epv_mh_accept3_kernel",
    }

    def phase_mode(self):
        """EPV_PHASE_* of include/epievo_mi355x.h: which kernels a colour phase launches"""
        v = C.c_uint32(0)
        self._ck(self.L.epv_phase_mode(self.h, C.byref(v)))
        return int(v.value)

    def capacity(self):
        v = C.c_uint32(0)
        self._ck(self.L.epv_get_capacity(self.h, C.byref(v)))
        return int(v.value)

    def set_capacity(self, capacity):
        self._ck(self.L.epv_set_capacity(self.h, int(capacity)))

    def set_tree(self, tree):
        self.n_nodes, self.B = tree.n_nodes, tree.n_nodes - 1
        self._ck(self.L.epv_set_tree(self.h, tree.n_nodes, _p(tree.parent_ids, C.c_uint32),
                                     _p(tree.subtree_sizes, C.c_uint32),
                                     _p(tree.branches, C.c_double)))

    def set_model(self, model):
        self._ck(self.L.epv_set_model(self.h, _p(model.rates, C.c_double), _p(model.T, C.c_double)))

    def upload_paths(self, fp, capacity=0, global_site_offset=0, n_global=None):
        jumps = fp.jumps if len(fp.jumps) else np.zeros(1)
        self.n_sites = fp.n_sites
        self._ck(self.L.epv_upload_paths(self.h, fp.n_sites, _p(fp.init, C.c_uint8),
                                         _p(fp.offsets, C.c_uint64), _p(jumps, C.c_double),
                                         capacity, global_site_offset))
        if n_global is not None:
            self._ck(self.L.epv_set_global_length(self.h, n_global))
        self.halo = (0, 0)

    def init_paths_indep(self, root, leaf, seed, capacity=0):
        root = np.ascontiguousarray(root, np.uint8)
        leaf = np.ascontiguousarray(leaf, np.uint8)
        self.n_sites = len(root)
        self._ck(self.L.epv_init_paths_indep(self.h, len(root), _p(root, C.c_uint8),
                                             _p(leaf, C.c_uint8), seed, capacity))

    def forward_simulate(self, n_sites, seed, root=None, capacity=0):
        """epievo_sim's forward simulation on the device (set_tree and set_model first); the histories
        are resident afterwards.  Doubles the jump slots until every path fits -> total jumps"""
        self.n_sites = int(n_sites)
        rp = None
        if root is not None:
            root = np.ascontiguousarray(root, np.uint8)
            rp = _p(root, C.c_uint8)
        tot = C.c_uint64(0)
        cap = capacity or 16
        while True:
            rc = self.L.epv_forward_simulate(self.h, self.n_sites, rp, int(seed), cap, C.byref(tot))
            if rc == EPV_ERR_CAPACITY and cap < 2047:
                cap = min(2047, cap * 2)
                continue
            self._ck(rc)
            return int(tot.value)

    def forward_last_ms(self):
        """(device memory management, simulation) wall clock of the last forward_simulate, ms"""
        a, b = C.c_double(0), C.c_double(0)
        self._ck(self.L.epv_forward_last_ms(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def indep_expectation(self, rates):
        r = np.ascontiguousarray(rates, np.float64)
        J, D = np.zeros(self.B * 2), np.zeros(self.B * 2)
        self._ck(self.L.epv_indep_expectation(self.h, _p(r, C.c_double), _p(J, C.c_double), _p(D, C.c_double)))
        return J, D

    def indep_suffstats(self):
        J, D = np.zeros(self.B * 2), np.zeros(self.B * 2)
        self._ck(self.L.epv_indep_sufficient_statistics(self.h, _p(J, C.c_double), _p(D, C.c_double)))
        return J, D

    def indep_update_paths(self, rates, seed, sweep=0):
        r = np.ascontiguousarray(rates, np.float64)
        self._ck_mcmc(self.L.epv_indep_update_paths(self.h, _p(r, C.c_double), seed, sweep))

    def set_update_range(self, first, last):
        self._ck(self.L.epv_set_update_range(self.h, first, last))

    def set_halo(self, left, right):
        self._ck(self.L.epv_set_halo(self.h, left, right))
        self.halo = (int(left), int(right))

    def halo_phases_left(self):
        v = C.c_uint64(0)
        self._ck(self.L.epv_halo_phases_left(self.h, C.byref(v)))
        return int(v.value)

    def reset(self):
        self._ck(self.L.epv_reset(self.h))

    def sweep(self, n_sweeps, seed, sweep_base=0):
        nacc = C.c_uint64(0)
        self._ck_mcmc(self.L.epv_sweep(self.h, n_sweeps, seed, sweep_base, C.byref(nacc)))
        return int(nacc.value)

    def sweep_phase(self, colour, seed, sweep):
        nacc = C.c_uint64(0)
        self._ck(self.L.epv_sweep_phase(self.h, colour, seed, sweep, C.byref(nacc)))
        return int(nacc.value)

    def run_mcmc(self, burn_in, batch, seed, sweep_base=0, average=True):
        J, D = np.zeros(self.B * 8), np.zeros(self.B * 8)
        nacc = C.c_uint64(0)
        self._ck_mcmc(self.L.epv_run_mcmc_sums(self.h, burn_in, batch, seed, sweep_base, int(average),
                                               _p(J, C.c_double), _p(D, C.c_double), C.byref(nacc)))
        return J, D, int(nacc.value)

    def suffstats(self):
        J, D = np.zeros(self.B * 8), np.zeros(self.B * 8)
        self._ck(self.L.epv_get_sufficient_statistics(self.h, _p(J, C.c_double), _p(D, C.c_double)))
        return J, D

    def scale_jump_times(self, new_branches):
        nb = np.ascontiguousarray(new_branches, dtype=np.float64)
        self._ck(self.L.epv_scale_jump_times(self.h, _p(nb, C.c_double)))

    def paths(self):
        tot = C.c_uint64(0)
        self._ck(self.L.epv_paths_total_jumps(self.h, C.byref(tot)))
        init = np.zeros(self.B * self.n_sites, np.uint8)
        off = np.zeros(self.B * self.n_sites + 1, np.uint64)
        jumps = np.zeros(max(tot.value, 1))
        self._ck(self.L.epv_download_paths(self.h, _p(init, C.c_uint8), _p(off, C.c_uint64),
                                           _p(jumps, C.c_double)))
        return FlatPaths(self.n_sites, self.n_nodes, init, off, jumps[:tot.value])

    def tri_llh(self):
        out = np.zeros(self.n_sites)
        self._ck(self.L.epv_get_tri_llh(self.h, _p(out, C.c_double)))
        return out

    def column_bytes(self):
        return int(self.L.epv_column_bytes(self.h))

    def get_columns(self, first, count):
        buf = np.zeros(count * self.column_bytes(), np.uint8)
        self._ck(self.L.epv_get_columns(self.h, first, count, buf.ctypes.data_as(C.c_void_p)))
        return buf

    def put_columns(self, first, count, buf):
        buf = np.ascontiguousarray(buf, dtype=np.uint8)
        self._ck(self.L.epv_put_columns(self.h, first, count, buf.ctypes.data_as(C.c_void_p)))

    # ---- several shards on one GPU (see epievo_amd.parallel.LocalGroup)
    def copy_columns_to(self, first, count, other, other_first):
        self._ck(self.L.epv_copy_columns(self.h, first, count, other.h, other_first))

    def dev_alloc(self, nbytes):
        p = C.c_void_p(0)
        self._ck(self.L.epv_dev_alloc(self.h, nbytes, C.byref(p)))
        return p

    def dev_free(self, p):
        self._ck(self.L.epv_dev_free(self.h, p))

    def run_mcmc_blocks(self, burn_in, batch, seed, sweep_base, d_blocks, n_blocks_total, block_offset):
        nacc = C.c_uint64(0)
        self._ck_mcmc(self.L.epv_run_mcmc_blocks(self.h, burn_in, batch, seed, sweep_base, d_blocks,
                                                 n_blocks_total, block_offset, C.byref(nacc)))
        return int(nacc.value)

    def reduce_blocks(self, d_blocks, n_blocks_total, batch, average=True):
        J, D = np.zeros(self.B * 8), np.zeros(self.B * 8)
        self._ck(self.L.epv_reduce_blocks(self.h, d_blocks, n_blocks_total, batch, int(average),
                                          _p(J, C.c_double), _p(D, C.c_double)))
        return J, D

    # ---- a genome sharded over several GPUs: device-resident halo columns and statistic rows
    # (buffers: DevBuf, or anything with .ptr = a device address on this context's GPU)
    def alloc(self, nbytes):
        return DevBuf(self, nbytes)

    def pack_columns(self, first, count, buf):
        self._ck(self.L.epv_pack_columns_dev(self.h, first, count, C.c_void_p(buf.ptr)))

    def unpack_columns(self, first, count, buf):
        self._ck(self.L.epv_unpack_columns_dev(self.h, first, count, C.c_void_p(buf.ptr)))

    def owned_blocks(self):
        """(first local 256-site block with owned columns, number of such blocks)"""
        left, right = self.halo
        b0 = left // 256
        return b0, (self.n_sites - right + 255) // 256 - b0

    def run_mcmc_rows(self, burn_in, batch, seed, sweep_base, row_blocks, rows_buf):
        """run_mcmc whose statistics stay on the device as rows of the reduction tree:
        rows_buf[row][w][16 B] doubles, row = row_blocks consecutive blocks of the owned
        columns (the shard starts on a whole row of the genome).  -> accepted proposals"""
        if self.halo[0] % 256:
            raise EpvError(EPV_ERR_ARG, "the left halo must be a whole number of 256-site blocks")
        b0, nb = self.owned_blocks()
        shape = (batch, nb, self.B)
        if self._blocks is None or self._blocks_shape != shape:
            if self._blocks is not None:
                self._blocks.free()
            self._blocks, self._blocks_shape = DevBuf(self, batch * nb * self.B * 16 * 8), shape
        nacc = self.run_mcmc_blocks(burn_in, batch, seed, sweep_base, self._blocks.p, nb, -b0)
        self.blocks_to_rows(self._blocks.p, nb, batch, row_blocks, rows_buf.ptr)
        return nacc

    def blocks_to_rows(self, d_blocks, n_blocks_total, batch, row_blocks, d_rows):
        self._ck(self.L.epv_blocks_to_rows(self.h, d_blocks, n_blocks_total, batch, row_blocks, C.c_void_p(d_rows)))

    def reduce_rows(self, d_rows, n_rows, batch, average=True):
        J, D = np.zeros(self.B * 8), np.zeros(self.B * 8)
        self._ck(self.L.epv_reduce_rows(self.h, C.c_void_p(d_rows), n_rows, batch, int(average),
                                        _p(J, C.c_double), _p(D, C.c_double)))
        return J, D

    def write(self, buf, offset, arr):
        a = np.ascontiguousarray(arr)
        self._ck(self.L.epv_dev_write(self.h, C.c_void_p(buf.ptr + offset), a.ctypes.data_as(C.c_void_p), a.nbytes))

    def read(self, buf, offset, count, dtype=np.float64):
        out = np.zeros(count, dtype)
        self._ck(self.L.epv_dev_read(self.h, out.ctypes.data_as(C.c_void_p), C.c_void_p(buf.ptr + offset), out.nbytes))
        return out

    def reduce_gathered_rows(self, gathered_buf, max_rows, piece_doubles, rows_per_rank, batch, average=True):
        J, D = np.zeros(self.B * 8), np.zeros(self.B * 8)
        rpr = np.ascontiguousarray(rows_per_rank, np.uint64)
        self._ck(self.L.epv_reduce_gathered_rows(self.h, C.c_void_p(gathered_buf.ptr), len(rpr), max_rows, piece_doubles,
                                                 _p(rpr, C.c_uint64), batch, int(average),
                                                 _p(J, C.c_double), _p(D, C.c_double)))
        return J, D

    def counters(self):
        c = _Counters()
        self._ck(self.L.epv_get_counters(self.h, C.byref(c)))
        return {"overflow": c.n_overflow, "coop_tasks": c.n_coop_tasks, "sweeps": c.n_sweeps}

    def set_timing(self, on):
        self._ck(self.L.epv_set_timing(self.h, int(on)))

    def kernel_time_ms(self):
        ms, n = C.c_double(0), C.c_uint64(0)
        self._ck(self.L.epv_kernel_time_ms(self.h, C.byref(ms), C.byref(n)))
        return ms.value, int(n.value)


class SingleSiteSampler:
    """Mirror of the reference class (SingleSiteSampler.hpp:35-81).

    mcmc = SingleSiteSampler(burn_in, batch); mcmc.reset(model, tree, paths);
    J, D, acc_rate = mcmc.run_mcmc(seed, em_iter)      # paths stay on the GPU
    paths = mcmc.paths()                               # download when needed
    The reference threads one std::mt19937 through every call; here the random
    stream is the counter-based (seed, sweep) pair, so the caller passes the seed and
    the index of the EM iteration (sweep numbers never repeat across iterations)."""

    def __init__(self, n_burn_in, n_batch, device=0, capacity=0):
        self.burn_in, self.batch = int(n_burn_in), int(n_batch)
        # hard-wired false in the reference (SingleSiteSampler.cpp:441) and set by none of its programs;
        # True = EPV_OPT_SAMPLE_ROOT: root states are proposed too (reference-arithmetic kernels)
        self.SAMPLE_ROOT = False
        self.capacity = capacity
        self.dev = DeviceSampler(device)
        self.dev.auto_grow = True     # paths grow on demand, as the reference's vectors do
        self._uploaded = False

    def _apply_sample_root(self):
        flags = C.c_uint32(0)
        self.dev._ck(self.dev.L.epv_get_options(self.dev.h, C.byref(flags)))
        want = (flags.value | 4) if self.SAMPLE_ROOT else (flags.value & ~4)
        if want != flags.value:
            self.dev._ck(self.dev.L.epv_set_options(self.dev.h, want))

    def reset(self, model, tree, paths=None):
        self.dev.set_tree(tree)
        self.dev.set_model(model)
        if paths is not None:
            self.dev.upload_paths(paths, self.capacity)
            self._uploaded = True
        if not self._uploaded:
            raise EpvError(EPV_ERR_STATE, "reset() needs paths the first time")
        self._apply_sample_root()
        self.dev.reset()

    def run_mcmc(self, seed, em_iter=0):
        self._apply_sample_root()
        base = em_iter * (self.burn_in + self.batch)
        J, D, nacc = self.dev.run_mcmc(self.burn_in, self.batch, seed, base)
        acc_rate = nacc / float(self.batch * (self.dev.n_sites - 2))
        return J, D, acc_rate

    def sweeps(self, n, seed, sweep_base=0):
        self._apply_sample_root()
        return self.dev.sweep(n, seed, sweep_base)

    def scale_jump_times(self, new_branches):
        self.dev.scale_jump_times(new_branches)

    def paths(self):
        return self.dev.paths()
