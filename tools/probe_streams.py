"""Timing prototype: k independent contexts on ONE GPU driven from k Python threads
(ctypes releases the GIL), each running run_mcmc on n/k sites -- does overlapping their
kernels fill the idle issue slots?"""
import sys, time, threading
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from epievo_amd.workloads import simulate
from epievo_amd.sampler import DeviceSampler
n = 1000000
for k in (1, 2, 3, 4):
    devs = []
    model, tree, fp = simulate("tree", n // k, seed=42)
    for i in range(k):
        d = DeviceSampler(0); d.set_tree(tree); d.set_model(model); d.upload_paths(fp, 16); d.reset()
        devs.append(d)
    def work(d, base):
        d.reset(); d.run_mcmc(10, 50, 42, base)
    for rep in range(2):
        ths = [threading.Thread(target=work, args=(d, 60 * rep)) for d in devs]
        t0 = time.perf_counter()
        for t in ths: t.start()
        for t in ths: t.join()
        el = time.perf_counter() - t0
    print("k=%d  step %.1f ms  %.3e resamples/s" % (k, el * 1e3, 60 * 4 * (n - 2 * k) / el))
