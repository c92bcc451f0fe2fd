#!/usr/bin/env python3
"""A/B of run-time knobs (environment variables) through bench.py inside ONE gpurun call.
  python tools/ab_envbench.py [--config tree] [--sites N] [--shards-per-gpu K] [--reps 2] "NAME:VAR=val,VAR2=val" ...
An empty setting ("base:") is the default configuration."""
import argparse, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--config", default="tree")
ap.add_argument("--sites", type=int, default=1000000)
ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--shards-per-gpu", type=int, default=2)
ap.add_argument("--extra", default="", help="extra bench.py arguments")
ap.add_argument("specs", nargs="+")
a = ap.parse_args()
for rep in range(a.reps):
    for spec in a.specs:
        name, _, kv = spec.partition(":")
        env = dict(os.environ)
        for item in filter(None, kv.split(",")):
            k, _, v = item.partition("=")
            env[k] = v
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "1",
                            "--no-cpu-baseline", "--no-reference-leg", "--config", a.config, "--sites", str(a.sites),
                            "--shards-per-gpu", str(a.shards_per_gpu)] + a.extra.split(), env=env, capture_output=True, text=True)
        try:
            j = json.loads(r.stdout.strip().split("\n")[-1])
            print("%-12s %s n=%d k=%d  %.4e resamples/s  %.2f ms/step  launch %.4f ms" %
                  (name, a.config, a.sites, a.shards_per_gpu, j["value"], j["ms_per_step"],
                   j["roofline"]["avg_launch_ms"]), flush=True)
        except Exception as e:
            print(name, "FAILED", e, r.stderr[-400:], flush=True)
