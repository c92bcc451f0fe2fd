#!/usr/bin/env python3
"""A/B of compile-time variants of the HIP library within ONE gpurun call (boxes differ by
>10 %, so numbers from different calls are not comparable).
  python tools/ab_defs.py build "NAME=-DX=1 -DY=2" ...     (here: cross-compile into build_ab/)
  python tools/ab_defs.py run <cfg> <n> NAME ...            (on the GPU box)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "build_ab")


def so(name):
    return os.path.join(OUT, "libepv_%s.so" % name)


if sys.argv[1] == "build":
    os.makedirs(OUT, exist_ok=True)
    for spec in sys.argv[2:]:
        name, _, defs = spec.partition("=")
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                               "-ffp-contract=off", "-fno-fast-math"] + defs.split() +
                              ["-I", ROOT + "/include", "-I", ROOT + "/epievo_amd/csrc", "-o", so(name),
                               ROOT + "/epievo_amd/csrc/epv_abi.hip"])
        print("built", so(name))
else:
    cfg, n, names = sys.argv[2], int(sys.argv[3]), sys.argv[4:]
    for rep in range(2):
        for name in names:
            code = r'''
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
from epievo_amd import _build
_build.HIP_SO = %r
from epievo_amd.workloads import simulate
from epievo_amd.sampler import DeviceSampler
model, tree, fp = simulate(%r, %d, seed=42)
d = DeviceSampler(0); d.set_tree(tree); d.set_model(model); d.upload_paths(fp, 0 if %r == "pair" else 16); d.reset()
d.sweep(3, 1, 0)
d.set_timing(True)
d.sweep(20, 1, 3)
ms, nl = d.kernel_time_ms()
print("%%-12s %%s n=%%d  phase %%.4f ms" %% (%r, %r, %d, ms), flush=True)
''' % (ROOT, ROOT + "/tests", so(name), cfg, n, cfg, name, cfg, n)
            subprocess.call([sys.executable, "-c", code], stderr=subprocess.DEVNULL)
