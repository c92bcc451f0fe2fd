import sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from epievo_amd.workloads import simulate
from epievo_amd.sampler import DeviceSampler
cfg = sys.argv[1] if len(sys.argv) > 1 else "tree"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
model, tree, fp = simulate(cfg, n, seed=42)
d = DeviceSampler(0); d.set_tree(tree); d.set_model(model); d.upload_paths(fp, 0 if cfg == "pair" else 16); d.reset()
d.sweep(2, 1, 0)
d.set_timing(True)
t0 = time.perf_counter(); nacc = d.sweep(10, 1, 2); el = time.perf_counter() - t0
ms, nl = d.kernel_time_ms()
print(cfg, n, "resamples/s %.3e" % (10 * (n - 2) * (tree.n_nodes - 1) / el), "phase ms %.3f" % ms)
