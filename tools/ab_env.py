#!/usr/bin/env python3
"""A/B of run-time knobs (environment variables read by the library) within ONE gpurun call.
  python tools/ab_env.py <cfg> <n> "NAME=VAL" "" ...     ("" = no variable set)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cfg, n, envs = sys.argv[1], int(sys.argv[2]), sys.argv[3:]
code = r'''
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
from epievo_amd.workloads import simulate
from epievo_amd.sampler import DeviceSampler
model, tree, fp = simulate(%r, %d, seed=42)
d = DeviceSampler(0); d.set_tree(tree); d.set_model(model); d.upload_paths(fp, 0 if %r == "pair" else 16); d.reset()
d.sweep(3, 1, 0)
d.set_timing(True)
d.sweep(10, 1, 3)
ms, nl = d.kernel_time_ms()
print("%%-28s %%s n=%%d  phase %%.4f ms" %% (sys.argv[1], %r, %d, ms), flush=True)
''' % (ROOT, ROOT + "/tests", cfg, n, cfg, cfg, n)
for rep in range(2):
    for e in envs:
        env = dict(os.environ)
        if e:
            k, _, v = e.partition("=")
            env[k] = v
        subprocess.call([sys.executable, "-c", code, e or "(default)"], env=env, stderr=subprocess.DEVNULL)
