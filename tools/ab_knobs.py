#!/usr/bin/env python3
"""A/B of compile-time knobs of the jumps kernel within ONE process sequence on ONE box
(devices differ by >10 %, so numbers from different gpurun calls are not comparable).
  python tools/ab_knobs.py <cfg> <n> "R,W;R,W;..."   """
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cfg, n, combos = sys.argv[1], int(sys.argv[2]), sys.argv[3]
for rep in range(2):
    for combo in combos.split(";"):
        R, W = combo.split(",")
        so = "/tmp/libepv_R%s_W%s.so" % (R, W)
        if not os.path.exists(so):
            subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                                   "-ffp-contract=off", "-fno-fast-math", "-DEPV_INLINE_TRIALS=%su" % R,
                                   "-DEPV_COOP_WINDOW=%su" % W, "-I", ROOT + "/include", "-I",
                                   ROOT + "/epievo_amd/csrc", "-o", so, ROOT + "/epievo_amd/csrc/epv_abi.hip"])
        code = r'''
import sys, time
sys.path.insert(0, %r); sys.path.insert(0, %r)
from epievo_amd import _build
_build.HIP_SO = %r
from common import simulate
from epievo_amd.sampler import DeviceSampler
model, tree, fp = simulate(%r, %d, seed=42)
d = DeviceSampler(0); d.set_tree(tree); d.set_model(model); d.upload_paths(fp, 0 if %r == "pair" else 16); d.reset()
d.sweep(3, 1, 0)
d.set_timing(True)
d.sweep(10, 1, 3)
ms, nl = d.kernel_time_ms()
print("R=%%-3s W=%%-3s phase %%.3f ms" %% (%r, %r, ms))
''' % (ROOT, ROOT + "/tests", so, cfg, n, cfg, R, W)
        subprocess.call([sys.executable, "-c", code], stderr=subprocess.DEVNULL)
