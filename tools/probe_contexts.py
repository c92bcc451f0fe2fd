import sys, time, os
sys.path.insert(0, '/root/repo')
from epievo_amd.workloads import simulate
from epievo_amd.parallel import LocalGroup
model, tree, fp = simulate("tree", 1000000, seed=42)
for k in (3, 4, 6):
    for timing in (False, True):
        g = LocalGroup(0, k, 60)
        g.set_tree(tree); g.set_model(model); g.upload_paths(fp, 16)
        g.set_timing(timing)
        g.reset(); g.run_mcmc(10, 50, 1)
        t0 = time.time()
        for s in range(3):
            g.reset(); g.run_mcmc(10, 50, 2 + s)
        el = (time.time() - t0) / 3
        print("k=%d timing=%s  %.2f ms/step  %.3e" % (k, timing, el * 1e3, 60 * 999998 * 4 / el), flush=True)
        g.close()
