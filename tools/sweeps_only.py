"""tiny driver for rocprofv3: 3 warm-up + 10 sweeps of a configuration
  python tools/sweeps_only.py <cfg> <n>"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
from epievo_amd.workloads import simulate
from epievo_amd.sampler import DeviceSampler
cfg, n = sys.argv[1], int(sys.argv[2])
model, tree, fp = simulate(cfg, n, seed=42)
d = DeviceSampler(0); d.set_tree(tree); d.set_model(model); d.upload_paths(fp, 0 if cfg == "pair" else 16); d.reset()
d.sweep(3, 1, 0)
d.sweep(10, 1, 3)
