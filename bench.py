#!/usr/bin/env python3
"""bench.py -- site-branch path resamples/s of the MCEM inner loop on MI355X.

Workload (BASELINE.json configs[2], the one the metric is quoted on): the 4-branch
test/tree.nwk tree, n = 1e6 sites per GPU, test/test.param scaled to unit rate, synthetic
histories forward-simulated on the host, true history as the initial MCMC state.
One "step" = the E-step of one EM iteration exactly as epievo_est_params_histories
drives it with -L 10 -B 50: reset() + run_mcmc() = 60 three-colour sweeps, with the
per-branch sufficient statistics J/D reduced after each of the 50 batch sweeps.
    resamples per step = (L + B) * (n_owned_sites) * (n_nodes - 1)
Inputs are resident in HBM before the timed region; file IO, upload/download and the
O(8) host M-step are outside it.

  python bench.py --gpus N --steps K --warmup W
runs the product's C++ EM driver, epv::SingleSiteSampler (epievo_amd/csrc/host/epv_sampler.cpp,
the class the epievo_* CLIs use) through libepv_driver.so.  The genome of N * n sites is cut into
contiguous shards with wide halos that each GPU updates redundantly (the RNG is keyed by the
global site index), so one step needs exactly two exchanges over RCCL (libepv_rccl.so): a halo
refresh before reset() and one all-gather of the integer J/D rows afterwards, both on device
buffers.  Called plainly, one process drives all N GPUs (ncclCommInitAll; EPV_DEVICES=0,0,0,0
rehearses four slots on one GPU through the loopback transport); under
`python -m torch.distributed.run --nproc-per-node N` every rank drives its GPU and joins the
communicator with ncclCommInitRank (torch.distributed/gloo only carries the id, the barriers and
the max over ranks).  N = 1: n = 1e6 (BASELINE config 3, the metric's); N > 1: 1.25e6 per GPU,
i.e. BASELINE config 4's n = 1e7 at N = 8.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec
N_SITES = 1000000
BURN_IN, BATCH = 10, 50
SEED = 42
# HIP events around every 7th colour-phase launch of each context (coprime with the three colours, so
# all of them are sampled): events around EVERY launch cost 4 % of the step they are meant to measure
TIMING_EVERY = 7


def algorithmic_bytes_per_resample(kbar, n_branches):
    """each path read once and written once per sweep in the SoA
    layout {1 B state+count, 4 B offset/count, 8 B per jump} + tri_llh 8 B read + 8 B
    write per site."""
    return 2.0 * (1.0 + 4.0 + 8.0 * kbar) + 16.0 / n_branches


def effective_cpus():
    """CPUs this process may really use: the affinity mask capped by the cgroup CPU quota
    (a GPU box hands each job a share of a much larger host)."""
    n = len(os.sched_getaffinity(0))
    for qf, pf in (("/sys/fs/cgroup/cpu.max", None),
                   ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us")):
        try:
            if pf is None:
                quota, period = open(qf).read().split()[:2]
            else:
                quota, period = open(qf).read().strip(), open(pf).read().strip()
            if quota not in ("max", "-1"):
                n = max(1, min(n, int(float(quota) / float(period) + 0.5)))
            break
        except (OSError, ValueError):
            continue
    return n


def cpu_baseline(model, tree, fp, budget_s=20.0):
    """The reference's own CPU path (oracle/_ref, the unmodified libepievo built in the
    authoring container) or, when that .so is absent, the bit-identical C restatement
    (oracle rung A), on ONE core (the reference is single-threaded), on a bounded sample
    of the same workload: the first n_s sites, a few sequential sweeps."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orc
    n_s = 200000
    sub = fp.slice_sites(0, n_s)
    if orc.have_ref():
        eng, kind = orc.Reference(tree, model, sub, seed=SEED), "reference"
        eng.reset(0, 1)
        sweep = lambda: eng.sweeps(1)
    else:
        eng, kind = orc.Oracle(tree, model, sub, "A", seed=SEED), "port"
        eng.reset()
        sweep = lambda: eng.sweep(0)
    t0 = time.perf_counter()
    k = 0
    while True:
        sweep()
        k += 1
        el = time.perf_counter() - t0
        if el > budget_s or k >= 40:
            break
    rs = k * (n_s - 2) * (tree.n_nodes - 1)
    out = {"value": rs / el, "unit": "site-branch resamples/s", "cores": 1, "kind": kind,
           "sample": "%d sequential sweeps over the first %d sites of the same workload "
                     "(%.1f s, run_mcmc region only)" % (k, n_s, el)}
    # the "fair" CPU number of BASELINE.md section 3 item 2: the same per-site arithmetic under
    # the parallel 3-colour schedule with OpenMP over the sites of a colour, all host cores
    omp = os.path.join(ROOT, "oracle", "liborc_omp.so")
    if os.path.exists(omp):
        try:
            cores = effective_cpus()
            os.environ["OMP_NUM_THREADS"] = str(cores)     # read when libgomp is loaded ...
            orc._orc, orc.ORC_SO = None, omp
            try:                                            # ... or set on an already loaded one
                import ctypes
                ctypes.CDLL("libgomp.so.1").omp_set_num_threads(cores)
            except OSError:
                pass
            o = orc.Oracle(tree, model, sub, "B", cap=16, seed=SEED)
            o.reset()
            t0, k2 = time.perf_counter(), 0
            while time.perf_counter() - t0 < 6.0 and k2 < 200:
                o.sweep(k2)
                k2 += 1
            el2 = time.perf_counter() - t0
            out["all_cores"] = {"value": k2 * (n_s - 2) * (tree.n_nodes - 1) / el2, "cores": cores,
                                "kind": "port (oracle rung B: 3-colour schedule, Philox, OpenMP)",
                                "sample": "%d sweeps over the same %d sites (%.1f s)" % (k2, n_s, el2)}
        except Exception as e:  # the extra number is best-effort
            out["all_cores"] = {"error": str(e)}
    return out


class CppEngine:
    """the product's C++ EM driver (epv::SingleSiteSampler through libepv_driver.so): every GPU of
    the run in this process (ncclCommInitAll), or -- under torch.distributed.run -- this rank's GPU
    with the communicator made by ncclCommInitRank from an id rank 0 broadcasts"""

    def __init__(self, args, model, tree, world, rank, local_rank, dist):
        from epievo_amd import driver, host
        self.world, self.rank = world, rank
        cap = 16 if args.config != "pair" else 0
        if dist is None:
            # plain launch: --gpus N devices, or the list of EPV_DEVICES (repeats rehearse N GPUs on fewer)
            env = os.environ.get("EPV_DEVICES")
            devices = [int(x) for x in env.split(",")] if env else list(range(args.gpus))
            self.n_gpus = len(devices)
            self.n_global = args.sites * self.n_gpus
            self.s = driver.CppSampler(BURN_IN, BATCH, devices=devices, capacity=cap)
            self.fp = host.simulate(model, tree, self.n_global, SEED)
            self.s.reset(model, tree, self.fp)
            self.fp_sample = self.fp
        else:
            self.n_gpus = world
            self.n_global = args.sites * world
            ident = [driver.unique_id() if rank == 0 else None]
            dist.broadcast_object_list(ident, src=0)
            cuts = driver.shard_cuts(self.n_global, world, BURN_IN, BATCH)
            if len(cuts) - 1 != world:
                sys.exit("a genome of %d sites cannot feed %d GPUs" % (self.n_global, world))
            self.s = driver.CppSampler(BURN_IN, BATCH, capacity=cap or 32, rank=(local_rank, world, rank, ident[0]))
            # every rank simulates its own columns; the halos arrive from the neighbours before the first reset
            self.fp_sample = host.simulate(model, tree, cuts[rank + 1] - cuts[rank], SEED + rank)
            self.s.reset(model, tree, self.fp_sample, n_global=self.n_global)
        self.model = model
        self.lay = self.s.layout()
        self.k_eff = max(1, self.lay["parts_here"] // max(1, self.lay["slots_here"]))
        self.transport = ("rccl" if self.lay["rccl"] else "loopback") if self.n_gpus > 1 else "none (one GPU)"
        self.halo = self.lay["halo"]
        self.owned_per_gpu = self.n_global / float(self.n_gpus)

    def step(self, i):
        self.s.reset(self.model)
        return self.s.run_mcmc(SEED, i)

    def set_timing(self, every):
        self.s.set_timing(every)

    def kernel_time_ms(self):
        return self.s.kernel_time_ms()

    def set_options(self, **kw):
        self.s.set_options(**kw)

    def phase_mode(self):
        return self.s.phase_mode()

    def describe(self):
        return "C++ epv::SingleSiteSampler (epv_sampler.cpp + libepv_rccl.so); " + self.lay["text"]


class TorchEngine:
    """the Python one-process-per-GPU driver (epievo_amd/parallel.py over torch.distributed): the same
    C-ABI primitives, kept for A/B runs against the C++ driver (--driver torch)"""

    def __init__(self, args, model, tree, world, rank, local_rank, dist, torch):
        from epievo_amd import host
        from epievo_amd.parallel import ShardedSampler, TorchComm, NullComm, LocalGroup, shard_cuts
        comm = TorchComm(dist, torch.device("cuda", local_rank)) if dist is not None else NullComm()
        self.n_gpus = world
        self.n_global = args.sites * world
        cuts = shard_cuts(self.n_global, world)
        self.fp_sample = host.simulate(model, tree, cuts[rank + 1] - cuts[rank], SEED + rank)
        k_local = args.shards_per_gpu if args.shards_per_gpu > 0 else (3 if tree.n_nodes - 1 <= 8 else 2)
        self.ss = ShardedSampler(comm, device=local_rank,
                                 device_factory=(lambda dev: LocalGroup(dev, k_local, BURN_IN + BATCH)) if k_local > 1 else None)
        self.ss.setup(model, tree, self.fp_sample, cuts, capacity=16 if args.config != "pair" else 0,
                      sweeps_per_refresh=BURN_IN + BATCH)
        self.ss.dev.set_timing(False)
        self.k_eff = len(self.ss.dev.subs) if hasattr(self.ss.dev, "subs") else 1
        self.transport = ("rccl" if args.backend == "nccl" else "gloo (host bounce)") if world > 1 else "none (one GPU)"
        self.halo = self.ss.halo
        self.owned_per_gpu = float(self.ss.owned_sites())

    def step(self, i):
        self.ss.reset()
        return self.ss.run_mcmc(BURN_IN, BATCH, SEED, sweep_base=i * (BURN_IN + BATCH))

    def set_timing(self, every):
        self.ss.dev.set_timing(every)

    def kernel_time_ms(self):
        return self.ss.dev.kernel_time_ms()

    def set_options(self, **kw):
        self.ss.dev.set_options(**kw)

    def phase_mode(self):
        return self.ss.dev.phase_mode()

    def describe(self):
        return "Python driver (epievo_amd/parallel.py over torch.distributed)"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--sites", type=int, default=0,
                    help="sites per GPU; 0 = 1e6 on one GPU (BASELINE config 3, the metric's), 1.25e6 per GPU on "
                         "several (8 GPUs = BASELINE config 4's n = 1e7)")
    ap.add_argument("--config", default="tree", choices=["tree", "pair", "bal16", "bal8", "bal32", "bal64", "cat20", "cat6", "star4"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-reference-leg", action="store_true",
                    help="skip the extra steps in reference-arithmetic mode (profiling runs)")
    ap.add_argument("--driver", default="cpp", choices=["cpp", "torch"],
                    help="cpp: the product's C++ EM driver epv::SingleSiteSampler (what the epievo_* CLIs run), every "
                         "GPU in this process when launched plainly, one rank per GPU under torch.distributed.run; "
                         "torch: the Python driver over torch.distributed (A/B)")
    ap.add_argument("--shards-per-gpu", type=int, default=0,
                    help="contexts per GPU: their launches fill each other's tails; results are bit-identical "
                         "to 1.  0 = by tree size: 3 on small trees (fused colour phase), 2 on large ones")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="--driver torch only: nccl (= RCCL) or gloo (lets several ranks share one GPU to "
                         "rehearse the N>1 code path on a 1-GPU box)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    launched = world > 1 or (os.environ.get("EPV_BENCH_FORCE_DIST") == "1" and "RANK" in os.environ)
    if launched:
        args.gpus = world
    elif args.gpus > 1 and args.driver == "torch":
        sys.exit("--driver torch needs: python -m torch.distributed.run --nproc-per-node %d bench.py --gpus %d ..."
                 % (args.gpus, args.gpus))
    if args.shards_per_gpu > 0:
        os.environ["EPV_CONTEXTS_PER_GPU"] = str(args.shards_per_gpu)    # the C++ driver reads it
    n_gpus_nominal = args.gpus
    if not launched and os.environ.get("EPV_DEVICES"):
        n_gpus_nominal = len(os.environ["EPV_DEVICES"].split(","))
    if args.sites <= 0:
        args.sites = N_SITES if n_gpus_nominal == 1 else 1250000

    # stdout carries exactly one JSON line (rank 0).  Libraries print there too (RCCL's version
    # banner, gloo's connection notes): send everything else written to fd 1 to stderr and
    # restore the descriptor only for the result line.
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch
    from epievo_amd.workloads import ref_test_model, config

    dist = None
    if launched:
        import torch.distributed as dist
        if args.driver == "cpp":
            # control plane only (the id broadcast, barriers, the max over ranks): the data plane is
            # RCCL inside libepv_rccl.so, through the communicator the C++ driver makes
            local_rank = local_rank % max(torch.cuda.device_count(), 1)   # (fewer GPUs than ranks: a rehearsal)
            torch.cuda.set_device(local_rank)
            dist.init_process_group("gloo")
        elif args.backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            local_rank = local_rank % max(torch.cuda.device_count(), 1)
            torch.cuda.set_device(local_rank)
            dist.init_process_group("gloo")

    model = ref_test_model()
    tree = config(args.config)
    fallback = None
    if args.driver == "cpp":
        try:
            eng, err = CppEngine(args, model, tree, world, rank, local_rank, dist), None
        except Exception as e:      # e.g. RCCL cannot form the communicator (two ranks on one GPU)
            eng, err = None, "%s: %s" % (type(e).__name__, e)
        if dist is not None:
            # the ranks agree: either all run the C++ driver or all switch to the Python one, whose
            # exchanges then go over the control-plane group (gloo: device buffers bounced through the host)
            bad = torch.tensor([0 if err is None else 1], dtype=torch.int32)
            dist.all_reduce(bad, op=dist.ReduceOp.MAX)
            if int(bad.item()):
                errs = [None] * world
                dist.all_gather_object(errs, err)
                fallback = "C++ driver unavailable on this launch (%s); ran the Python driver over gloo" % \
                           next(e for e in errs if e)
                eng = None
        elif err is not None:
            raise RuntimeError(err)
        if eng is None:
            args.backend = "gloo"
            eng = TorchEngine(args, model, tree, world, rank, local_rank, dist, torch)
    else:
        eng = TorchEngine(args, model, tree, world, rank, local_rank, dist, torch)
    n_gpus, n_global, n_local = eng.n_gpus, eng.n_global, args.sites
    fp_own = eng.fp_sample
    kbar = len(fp_own.jumps) / float(fp_own.n_sites * (tree.n_nodes - 1))

    def barrier():
        for d in range(torch.cuda.device_count() if dist is None else 0):
            torch.cuda.synchronize(d)            # plain launch: this process drives every GPU
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    step = eng.step
    for i in range(args.warmup):
        step(i)
    barrier()
    eng.kernel_time_ms()          # clear the timing accumulators
    eng.set_timing(TIMING_EVERY)  # HIP events around every TIMING_EVERY-th colour-phase launch of each context
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    barrier()
    el = time.perf_counter() - t0
    eng.set_timing(0)
    avg_ms, n_launch = eng.kernel_time_ms()

    def rank_max(x):
        if dist is None:
            return x
        on_gpu = args.driver == "torch" and args.backend == "nccl"
        t = torch.tensor([x], dtype=torch.float64, device="cuda" if on_gpu else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    el = rank_max(el)

    # the same steps with the proposal ratio q(old)/q(new) evaluated by the reference's sums
    # (EPV_OPT_REFERENCE_PROPOSAL_RATIO) instead of the exact 0 they amount to: reported beside
    # the headline so that the cost of that arithmetic -- which changes no path -- is on record
    k_ref, el_ref = 0, 0.0
    if not args.no_reference_leg:
        eng.set_options(reference_proposal_ratio=True)
        k_ref = max(1, min(2, args.steps))
        step(args.warmup + args.steps)
        barrier()
        t0 = time.perf_counter()
        for i in range(k_ref):
            step(args.warmup + args.steps + 1 + i)
        barrier()
        el_ref = rank_max(time.perf_counter() - t0)
        eng.set_options()

    B = tree.n_nodes - 1
    owned_total = n_global - 2
    resamples = float(args.steps) * (BURN_IN + BATCH) * owned_total * B
    value = resamples / el

    if rank == 0:
        bytes_per = algorithmic_bytes_per_resample(kbar, B)
        k_eff = eng.k_eff
        # a timed launch covers one colour phase of ONE of the k_eff contexts of a GPU
        per_launch_units = eng.owned_per_gpu / 3.0 * B / k_eff
        achieved = per_launch_units * bytes_per / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        traffic, traffic_source = None, None
        tj = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tj):
            try:
                tq = json.load(open(tj)).get(args.config, {})
                # FETCH_SIZE reads half the bytes of the 128-B lines a kernel touches on gfx950, for every
                # access width and stride of these kernels (profiles/r03_fetch_calibration.txt): doubled
                traffic = tq.get("hbm_bytes_per_launch_fetch_x2")
                if traffic is not None:
                    # a citation, not this run's counters: PMC passes cannot run next to the timed region
                    traffic_source = ("profiles/%s_pmc_hbm_%s.csv (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate "
                                      "passes, one context per GPU; 2 x FETCH_SIZE + WRITE_SIZE, the gfx950 correction "
                                      "calibrated on this kernel's access patterns in profiles/r03_fetch_calibration.txt; "
                                      "committed file, not measured in this run)" % (tq.get("round", "?"), args.config))
            except Exception:
                traffic = None
        # what really bounds these kernels: VALU issue.  Instruction counts per launch come from
        # the committed PMC profile of this workload (profiles/issue.json, rocprofv3 --pmc
        # SQ_INSTS_VALU ...; they are properties of the code and the data, not of the box)
        issue = None
        ij = os.path.join(ROOT, "profiles", "issue.json")
        if os.path.exists(ij) and avg_ms > 0:
            try:
                q = json.load(open(ij)).get(args.config)
                if q:
                    # the profile was taken with one context per GPU: scale to this launch's sites
                    insts = q["valu_wave_insts_per_launch"] * per_launch_units / q["resamples_per_launch"]
                    bound_ms = insts * q["cycles_per_inst"] / (q["simds"] * q["clock_ghz"] * 1e9) * 1e3
                    issue = {"bound": "valu-issue", "valu_wave_insts_per_launch": insts,
                             "cycles_per_inst": q["cycles_per_inst"], "simds": q["simds"],
                             "clock_ghz": q["clock_ghz"], "bound_ms_per_launch": bound_ms,
                             "measured_ms_per_launch": avg_ms, "concurrent_launches": k_eff,
                             # k_eff launches share the SIMDs: the device issues k_eff * insts in avg_ms
                             "frac": k_eff * bound_ms / avg_ms, "lane_utilisation": q.get("lane_utilisation"),
                             "source": "profiles/%s_pmc_valu_%s.csv" % (q["round"], args.config),
                             "profiled_phase_mode": q.get("phase_mode")}
            except Exception:
                issue = None
        # how many colour-phase launches were really in flight together: the time the timed launches
        # took, scaled to all phase launches of the timed region, over its wall clock (the k_eff
        # contexts of a GPU never overlap perfectly; the statistics kernels share the device too)
        phase_launches_per_gpu = float(args.steps) * (BURN_IN + BATCH) * 3 * k_eff
        overlap = min(float(k_eff), avg_ms * 1e-3 * phase_launches_per_gpu / el) if avg_ms > 0 else float(k_eff)
        achieved_device = achieved * overlap
        if issue is not None:
            issue["measured_overlap"] = overlap
            issue["frac"] = overlap * issue["bound_ms_per_launch"] / avg_ms
        from epievo_amd.sampler import DeviceSampler
        phase_mode = eng.phase_mode()
        phase_kernels = DeviceSampler.PHASE_KERNELS[phase_mode]
        if args.config == "tree":
            if n_gpus == 1:
                workload = ("tree.nwk (4 branches), n=%d sites per GPU, one step = reset + run_mcmc(-L %d -B %d) as "
                            "epievo_est_params_histories drives it" % (n_local, BURN_IN, BATCH))
            else:
                workload = ("tree.nwk (4 branches), n=%d sites over %d GPUs (%d per GPU; 8 GPUs = BASELINE config 4's "
                            "n = 1e7), one step = reset + run_mcmc(-L %d -B %d) as epievo_est_params_histories -g all "
                            "drives it" % (n_global, n_gpus, n_local, BURN_IN, BATCH))
        else:
            workload = "%s, n=%d per GPU" % (args.config, n_local)
        out = {
            "metric": "site-branch path resamples/sec at n=1e6, 4-leaf tree",
            "value": value, "unit": "site-branch resamples/s", "n_gpus": n_gpus,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": el / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": workload, "sites_per_gpu": n_local, "sites_total": n_global, "branches": B,
                       "burn_in": BURN_IN, "batch": BATCH, "mean_jumps_per_path": kbar, "shards_per_gpu": k_eff,
                       "driver": eng.describe() + ("; " + fallback if fallback else ""), "transport": eng.transport,
                       "launch": ("torch.distributed.run, one rank per GPU" if dist is not None
                                  else "one process drives every GPU"),
                       "sharding": "contiguous site shards cut on 16384-site rows, %d-column redundant halos "
                                   "refreshed once per step, one all-gather of integer J/D rows per step; "
                                   "%d GPU shard(s) x %d concurrent context(s) per GPU" % (eng.halo, n_gpus, k_eff)},
            # frac = what the DEVICE sustains: up to k_eff launches (one per context of this GPU) run
            # concurrently, each timed with its own HIP events on its own stream; achieved = the
            # per-launch rate x the MEASURED overlap of those launches
            "roofline": {"bound": "hbm", "kernel": phase_kernels + " (one colour phase of one context = one timed launch group)",
                         "achieved": achieved_device,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved_device / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_source, "bytes_per_resample": bytes_per,
                         "resamples_per_launch": per_launch_units, "avg_launch_ms": avg_ms,
                         "launches_timed": n_launch, "concurrent_launches": k_eff, "measured_overlap": overlap,
                         "achieved_per_launch": achieved, "phase_mode": phase_mode, "issue": issue},
        }
        out["config"]["proposal_ratio"] = (
            "exact: with the root state kept, log q(old) - log q(new) telescopes to 0 (DESIGN.md section 2), so the "
            "sums are skipped; the chain equals the one with the reference's sums unless a rounding difference of "
            "~1e-12 flips an accept decision (probability ~1e-3 per full run)")
        if k_ref:
            out["reference_proposal_arithmetic"] = {
                "value": float(k_ref) * (BURN_IN + BATCH) * owned_total * B / el_ref, "unit": "site-branch resamples/s",
                "ms_per_step": el_ref / k_ref * 1e3, "steps": k_ref,
                "note": "EPV_OPT_REFERENCE_PROPOSAL_RATIO: the two log-probability sums of "
                        "SingleSiteSampler.cpp:180-339 evaluated as the reference does"}
        if not args.no_cpu_baseline and n_gpus == 1:   # reported at N=1 only
            out["cpu_baseline"] = cpu_baseline(model, tree, fp_own)
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
