#!/usr/bin/env python3
"""A/B of compile-time variants of the HIP library through bench.py itself (two contexts per
GPU, the statistics passes, everything the headline number contains), all inside ONE gpurun
call (boxes differ by several per cent).  Build the variants first with tools/ab_defs.py build.
  python tools/ab_bench.py [--config tree] [--sites N] [--reps 2] NAME ...
EPIEVO_MI355X_LIB selects the library a bench.py process loads."""
import argparse, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--config", default="tree")
ap.add_argument("--sites", type=int, default=1000000)
ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--shards-per-gpu", type=int, default=2)
ap.add_argument("names", nargs="+")
a = ap.parse_args()
for rep in range(a.reps):
    for name in a.names:
        lib = os.path.join(ROOT, "build_ab", "libepv_%s.so" % name)
        env = dict(os.environ, EPIEVO_MI355X_LIB=lib)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "1",
                            "--no-cpu-baseline", "--no-reference-leg", "--config", a.config, "--sites", str(a.sites),
                            "--shards-per-gpu", str(a.shards_per_gpu), "--driver", "torch"], env=env, capture_output=True, text=True)
        try:
            j = json.loads(r.stdout.strip().split("\n")[-1])
            print("%-10s %s n=%d k=%d  %.4e resamples/s  %.2f ms/step  launch %.4f ms" %
                  (name, a.config, a.sites, a.shards_per_gpu, j["value"], j["ms_per_step"],
                   j["roofline"]["avg_launch_ms"]), flush=True)
        except Exception as e:
            print(name, "FAILED", e, r.stderr[-400:], flush=True)
