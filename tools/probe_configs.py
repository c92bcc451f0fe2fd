import sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from epievo_amd.workloads import simulate
from epievo_amd.sampler import DeviceSampler
for cfg, n in (("tree", 1000000), ("pair", 100000), ("pair", 1000000), ("bal16", 200000)):
    model, tree, fp = simulate(cfg, n, seed=42)
    d = DeviceSampler(0); d.set_tree(tree); d.set_model(model); d.upload_paths(fp, 0 if cfg=="pair" else 16); d.reset()
    d.sweep(2, 1, 0)
    c0 = d.counters()
    d.set_timing(True)
    t0 = time.perf_counter(); nacc = d.sweep(10, 1, 2); el = time.perf_counter() - t0
    ms, nl = d.kernel_time_ms()
    c1 = d.counters()
    B = tree.n_nodes - 1
    print(cfg, n, "resamples/s %.3e" % (10 * (n - 2) * B / el), "kernel ms %.3f" % ms, "acc %.3f" % (nacc / (10.0 * (n - 2))),
          "coop tasks per site-branch %.4f" % ((c1["coop_tasks"] - c0["coop_tasks"]) / (10.0 * (n - 2) * B)))
