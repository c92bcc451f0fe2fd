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
For N > 1 it is launched by torch.distributed.run (one rank per GPU, RCCL): the genome
of N * n sites is cut into contiguous shards with wide halos that each rank updates
redundantly (the RNG is keyed by the global site index), so one step needs exactly two
exchanges: a halo refresh before reset() and one all-gather of the J/D rows afterwards,
both on device buffers handed to RCCL as they are (epievo_amd/parallel.py).
Rank 0 prints ONE JSON line.
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--sites", type=int, default=N_SITES, help="sites per GPU")
    ap.add_argument("--config", default="tree", choices=["tree", "pair", "bal16"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-reference-leg", action="store_true",
                    help="skip the extra steps in reference-arithmetic mode (profiling runs)")
    ap.add_argument("--shards-per-gpu", type=int, default=0,
                    help="contexts per GPU (epievo_amd.parallel.LocalGroup): their launches fill each "
                         "other's tails; results are bit-identical to 1.  0 = by tree size: 3 on small "
                         "trees (fused colour phase), 2 on large ones")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl (= RCCL) is the product path; gloo lets several ranks share one "
                         "GPU to rehearse the N>1 code path on a 1-GPU box")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit("launch with: python -m torch.distributed.run --nproc-per-node %d bench.py "
                     "--gpus %d ..." % (args.gpus, args.gpus))
        args.gpus = world

    # stdout carries exactly one JSON line (rank 0).  Libraries print there too (RCCL's version
    # banner, gloo's connection notes): send everything else written to fd 1 to stderr and
    # restore the descriptor only for the result line.
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch
    from epievo_amd import host
    from epievo_amd.parallel import ShardedSampler, TorchComm, NullComm, LocalGroup, shard_cuts
    from epievo_amd.workloads import ref_test_model, config

    dist = None
    # EPV_BENCH_FORCE_DIST=1 under torchrun with one rank exercises the RCCL set-up and the
    # barrier / all-reduce legs on a 1-GPU box (the shard exchange itself needs >= 2 GPUs)
    if world > 1 or (os.environ.get("EPV_BENCH_FORCE_DIST") == "1" and "RANK" in os.environ):
        import torch.distributed as dist
        if args.backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            comm = TorchComm(dist, torch.device("cuda", local_rank))
        else:
            local_rank = local_rank % max(torch.cuda.device_count(), 1)
            torch.cuda.set_device(local_rank)
            dist.init_process_group("gloo")
            # the buffers stay on the GPU; gloo cannot move them, TorchComm bounces them through the host
            comm = TorchComm(dist, torch.device("cuda", local_rank))
    else:
        comm = NullComm()

    model = ref_test_model()
    tree = config(args.config)
    n_local = args.sites
    n_global = n_local * world
    # contiguous shards cut on whole rows of the statistics tree (16384 sites): n_local sites per
    # GPU on average, at most one row more or less on any one
    cuts = shard_cuts(n_global, world)
    n_own = cuts[rank + 1] - cuts[rank]
    # every rank simulates its own shard (+ halos come from the neighbours' edges)
    fp_own = host.simulate(model, tree, n_own, SEED + rank)
    kbar = len(fp_own.jumps) / float(n_own * (tree.n_nodes - 1))

    k_local = args.shards_per_gpu if args.shards_per_gpu > 0 else (3 if tree.n_nodes - 1 <= 8 else 2)
    ss = ShardedSampler(comm, device=local_rank,
                        device_factory=(lambda dev: LocalGroup(dev, k_local, BURN_IN + BATCH)) if k_local > 1 else None)
    # 16 jump slots per (site, branch) on the short trees; the T = 1 branch picks its own
    ss.setup(model, tree, fp_own, cuts, capacity=16 if args.config != "pair" else 0,
             sweeps_per_refresh=BURN_IN + BATCH)
    ss.dev.set_timing(False)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def step(i):
        ss.reset()
        return ss.run_mcmc(BURN_IN, BATCH, SEED, sweep_base=i * (BURN_IN + BATCH))

    for i in range(args.warmup):
        step(i)
    barrier()
    ss.dev.kernel_time_ms()          # clear the timing accumulators
    ss.dev.set_timing(TIMING_EVERY)  # HIP events around every TIMING_EVERY-th colour-phase launch of each context
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    barrier()
    el = time.perf_counter() - t0
    ss.dev.set_timing(False)
    avg_ms, n_launch = ss.dev.kernel_time_ms()

    def rank_max(x):
        if dist is None:
            return x
        t = torch.tensor([x], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    el = rank_max(el)

    # the same steps with the proposal ratio q(old)/q(new) evaluated by the reference's sums
    # (EPV_OPT_REFERENCE_PROPOSAL_RATIO) instead of the exact 0 they amount to: reported beside
    # the headline so that the cost of that arithmetic -- which changes no path -- is on record
    k_ref, el_ref = 0, 0.0
    if not args.no_reference_leg:
        ss.dev.set_options(reference_proposal_ratio=True)
        k_ref = max(1, min(2, args.steps))
        step(args.warmup + args.steps)
        barrier()
        t0 = time.perf_counter()
        for i in range(k_ref):
            step(args.warmup + args.steps + 1 + i)
        barrier()
        el_ref = rank_max(time.perf_counter() - t0)
        ss.dev.set_options()

    B = tree.n_nodes - 1
    owned_total = n_global - 2
    resamples = float(args.steps) * (BURN_IN + BATCH) * owned_total * B
    value = resamples / el

    if rank == 0:
        bytes_per = algorithmic_bytes_per_resample(kbar, B)
        k_eff = len(ss.dev.subs) if hasattr(ss.dev, "subs") else 1
        # a timed launch covers one colour phase of ONE of the k_eff shards of this GPU
        per_launch_units = ss.owned_sites() / 3.0 * B / k_eff
        achieved = per_launch_units * bytes_per / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        traffic = None
        tj = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tj):
            try:
                traffic = json.load(open(tj)).get(args.config, {}).get("hbm_bytes_per_launch")
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
                             "source": "profiles/%s_pmc_valu_%s.csv" % (q["round"], args.config)}
            except Exception:
                issue = None
        achieved_device = achieved * k_eff
        from epievo_amd.sampler import DeviceSampler
        phase_mode = ss.dev.phase_mode()
        phase_kernels = DeviceSampler.PHASE_KERNELS[phase_mode]
        if issue is not None:
            issue["profiled_phase_mode"] = q.get("phase_mode")
        out = {
            "metric": "site-branch path resamples/sec at n=1e6, 4-leaf tree",
            "value": value, "unit": "site-branch resamples/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": el / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "tree.nwk (4 branches), n=%d sites per GPU, one step = "
                                   "reset + run_mcmc(-L %d -B %d) as epievo_est_params_histories "
                                   "drives it" % (n_local, BURN_IN, BATCH) if args.config == "tree"
                       else "%s, n=%d per GPU" % (args.config, n_local),
                       "sites_per_gpu": n_local, "branches": B, "burn_in": BURN_IN, "batch": BATCH,
                       "mean_jumps_per_path": kbar, "shards_per_gpu": k_eff,
                       "sharding": "contiguous site shards cut on 16384-site rows of the statistics tree, "
                                   "%d-column redundant halos refreshed once per step, "
                                   "%d GPU shard(s) x %d concurrent context(s) per GPU" % (ss.halo, world, k_eff)},
            # frac = what the DEVICE sustains: k_eff launches (one per context of this GPU) run
            # concurrently, each timed with its own HIP events on its own stream
            "roofline": {"bound": "hbm", "kernel": phase_kernels + " (one colour phase of one context = one timed launch group)",
                         "achieved": achieved_device,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved_device / HBM_PEAK_GBS,
                         "traffic": traffic, "bytes_per_resample": bytes_per,
                         "resamples_per_launch": per_launch_units, "avg_launch_ms": avg_ms,
                         "launches_timed": n_launch, "concurrent_launches": k_eff,
                         "achieved_per_launch": achieved, "phase_mode": phase_mode, "issue": issue},
        }
        out["config"]["proposal_ratio"] = ("exact (q(old)/q(new) = 1 when the root state is kept: "
                                           "DESIGN.md section 4.1); same paths as the reference's sums")
        if k_ref:
          out["reference_proposal_arithmetic"] = {
              "value": float(k_ref) * (BURN_IN + BATCH) * owned_total * B / el_ref, "unit": "site-branch resamples/s",
              "ms_per_step": el_ref / k_ref * 1e3, "steps": k_ref,
              "note": "EPV_OPT_REFERENCE_PROPOSAL_RATIO: the two log-probability sums of "
                      "SingleSiteSampler.cpp:180-339 evaluated as the reference does"}
        if not args.no_cpu_baseline and world == 1:   # reported at N=1 only
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
