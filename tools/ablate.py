#!/usr/bin/env python3
"""Timing-only ablation of epv_mh_phase_kernel (results of ablated builds are WRONG by
construction; only their launch time is read).  Builds variants of the HIP library with
-DEPV_ABLATE_* into /tmp and times 10 sweeps of BASELINE config 3 through the C ABI.
Run on the GPU box:  python tools/ablate.py"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

VARIANTS = [[], ["COOP"], ["COOP", "TRIAL"], ["LLH"], ["CURPATH"],
            ["COOP", "TRIAL", "LLH", "CURPATH"]]
EXTRA = [x for x in sys.argv[3:]]


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "tree"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
    for v in (VARIANTS if not EXTRA else [[]]):
        so = "/tmp/libepv_%s.so" % ("_".join(v) or "base")
        cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
               "-ffp-contract=off", "-fno-fast-math", "-I", ROOT + "/include", "-I",
               ROOT + "/epievo_amd/csrc", "-o", so, ROOT + "/epievo_amd/csrc/epv_abi.hip"]
        cmd += ["-DEPV_ABLATE_" + x for x in v] + EXTRA
        subprocess.check_call(cmd)
        code = r'''
import sys, time
sys.path.insert(0, %r); sys.path.insert(0, %r)
from epievo_amd import _build
_build.HIP_SO = %r
from epievo_amd.workloads import simulate
from epievo_amd.sampler import DeviceSampler
model, tree, fp = simulate(%r, %d, seed=42)
d = DeviceSampler(0); d.set_tree(tree); d.set_model(model); d.upload_paths(fp, 16); d.reset()
try:
    d.sweep(2, 1, 0)
except Exception as e:
    pass
d.set_timing(True)
try:
    d.sweep(10, 1, 2)
except Exception as e:
    pass
ms, nl = d.kernel_time_ms()
print("%%-36s kernel %%.3f ms" %% (%r, ms))
''' % (ROOT, ROOT + "/tests", so, cfg, n, "+".join(v) or "base")
        subprocess.call([sys.executable, "-c", code])


if __name__ == "__main__":
    main()
