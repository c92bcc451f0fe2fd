"""ctypes face of libepv_driver.so (include/epievo_mi355x_driver.h): the C++ EM driver
epv::SingleSiteSampler -- the code path of the drop-in CLIs -- for bench.py and the tests.
Every GPU slot in this process (CppSampler(devices=[...])) or one slot per process
(CppSampler(rank=(device, world, rank, id)))."""
import ctypes as C

import numpy as np

from . import _build
from .host import FlatPaths

_lib = None
vp, dp, u8p, u32p, u64p = C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_uint8), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)

DRIVER_SYMBOLS = ["epvd_create", "epvd_unique_id", "epvd_create_rank", "epvd_destroy", "epvd_last_error", "epvd_shard_cuts",
                  "epvd_reset", "epvd_reset_model", "epvd_run_mcmc", "epvd_scale_jump_times", "epvd_download_sizes",
                  "epvd_download", "epvd_layout", "epvd_set_options", "epvd_set_timing", "epvd_kernel_time_ms",
                  "epvd_phase_mode"]


def lib():
    global _lib
    if _lib is None:
        _build.build_hip()
        _build.build_comm()
        L = C.CDLL(_build.build_driver())
        L.epvd_create.restype = vp
        L.epvd_create.argtypes = [C.c_uint64, C.c_uint64, C.c_int, C.POINTER(C.c_int), C.c_uint32]
        L.epvd_unique_id.argtypes = [vp]
        L.epvd_create_rank.restype = vp
        L.epvd_create_rank.argtypes = [C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_int, vp, C.c_uint32]
        L.epvd_destroy.argtypes = [vp]
        L.epvd_last_error.restype = C.c_char_p
        L.epvd_last_error.argtypes = [vp]
        L.epvd_shard_cuts.argtypes = [C.c_uint64, C.c_int, C.c_uint64, C.c_uint64, u64p]
        L.epvd_reset.argtypes = [vp, dp, dp, C.c_int, u32p, u32p, dp, C.c_uint64, u8p, u64p, dp, C.c_uint64]
        L.epvd_reset_model.argtypes = [vp, dp, dp]
        L.epvd_run_mcmc.argtypes = [vp, C.c_uint64, C.c_uint64, dp, dp, dp]
        L.epvd_scale_jump_times.argtypes = [vp, dp, C.c_int]
        L.epvd_download_sizes.argtypes = [vp, u64p, u64p]
        L.epvd_download.argtypes = [vp, u8p, u64p, dp]
        L.epvd_layout.argtypes = [vp, C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), u64p]
        L.epvd_set_options.argtypes = [vp, C.c_uint32]
        L.epvd_set_timing.argtypes = [vp, C.c_int]
        L.epvd_kernel_time_ms.argtypes = [vp, dp, u64p]
        L.epvd_phase_mode.argtypes = [vp, u32p]
        _lib = L
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


class DriverError(RuntimeError):
    pass


def unique_id():
    """RCCL id for a one-slot-per-process run (rank 0 makes it, the launcher passes it around)"""
    buf = (C.c_uint8 * 128)()
    if lib().epvd_unique_id(buf) != 0:
        raise DriverError("epvd_unique_id failed")
    return bytes(buf)


def shard_cuts(n_sites, world, burn_in, batch):
    cuts = np.zeros(world + 1, np.uint64)
    g = lib().epvd_shard_cuts(n_sites, world, burn_in, batch, _p(cuts, C.c_uint64))
    return [int(x) for x in cuts[:g + 1]]


class CppSampler:
    def __init__(self, burn_in, batch, devices=(0,), capacity=0, rank=None):
        self.L = lib()
        self.burn_in, self.batch = int(burn_in), int(batch)
        if rank is None:
            devs = (C.c_int * len(devices))(*devices)
            self.h = self.L.epvd_create(self.burn_in, self.batch, len(devices), devs, capacity)
        else:
            device, world, r, ident = rank
            idb = (C.c_uint8 * 128).from_buffer_copy(ident)
            self.h = self.L.epvd_create_rank(self.burn_in, self.batch, device, world, r, idb, capacity)
        if not self.h:
            raise DriverError(self.L.epvd_last_error(None).decode())
        self.B = 0

    def close(self):
        if getattr(self, "h", None):
            self.L.epvd_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def _ck(self, rc):
        if rc != 0:
            raise DriverError(self.L.epvd_last_error(self.h).decode())

    def reset(self, model, tree=None, fp=None, n_global=0):
        rates = np.ascontiguousarray(model.rates, np.float64)
        T = np.ascontiguousarray(model.T, np.float64)
        if fp is None:
            self._ck(self.L.epvd_reset_model(self.h, _p(rates, C.c_double), _p(T, C.c_double)))
            return
        self.B, self.n_nodes = tree.n_nodes - 1, tree.n_nodes
        jumps = fp.jumps if len(fp.jumps) else np.zeros(1)
        self._ck(self.L.epvd_reset(self.h, _p(rates, C.c_double), _p(T, C.c_double), tree.n_nodes,
                                   _p(tree.parent_ids, C.c_uint32), _p(tree.subtree_sizes, C.c_uint32),
                                   _p(tree.branches, C.c_double), fp.n_sites, _p(fp.init, C.c_uint8),
                                   _p(fp.offsets, C.c_uint64), _p(jumps, C.c_double), n_global))

    def run_mcmc(self, seed, em_iteration=0):
        J, D, acc = np.zeros(self.B * 8), np.zeros(self.B * 8), C.c_double(0)
        self._ck(self.L.epvd_run_mcmc(self.h, seed, em_iteration, _p(J, C.c_double), _p(D, C.c_double), C.byref(acc)))
        return J, D, acc.value

    def scale_jump_times(self, branches):
        nb = np.ascontiguousarray(branches, np.float64)
        self._ck(self.L.epvd_scale_jump_times(self.h, _p(nb, C.c_double), len(nb)))

    def paths(self):
        n, tot = C.c_uint64(0), C.c_uint64(0)
        self._ck(self.L.epvd_download_sizes(self.h, C.byref(n), C.byref(tot)))
        E = self.B * n.value
        init, off, jumps = np.zeros(E, np.uint8), np.zeros(E + 1, np.uint64), np.zeros(max(tot.value, 1))
        self._ck(self.L.epvd_download(self.h, _p(init, C.c_uint8), _p(off, C.c_uint64), _p(jumps, C.c_double)))
        return FlatPaths(n.value, self.n_nodes, init, off, jumps[:tot.value])

    def layout(self):
        buf = C.create_string_buffer(512)
        ns, npart, rccl, halo = C.c_int(0), C.c_int(0), C.c_int(0), C.c_uint64(0)
        self._ck(self.L.epvd_layout(self.h, buf, 512, C.byref(ns), C.byref(npart), C.byref(rccl), C.byref(halo)))
        return {"text": buf.value.decode(), "slots_here": ns.value, "parts_here": npart.value,
                "rccl": bool(rccl.value), "halo": halo.value}

    def set_options(self, reference_proposal_ratio=False, forward_rejection=False):
        self._ck(self.L.epvd_set_options(self.h, (1 if reference_proposal_ratio else 0) | (2 if forward_rejection else 0)))

    def set_timing(self, every):
        self._ck(self.L.epvd_set_timing(self.h, int(every)))

    def kernel_time_ms(self):
        a, k = C.c_double(0), C.c_uint64(0)
        self._ck(self.L.epvd_kernel_time_ms(self.h, C.byref(a), C.byref(k)))
        return a.value, k.value

    def phase_mode(self):
        m = C.c_uint32(0)
        self._ck(self.L.epvd_phase_mode(self.h, C.byref(m)))
        return m.value
