import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from epievo_amd import driver, host
from epievo_amd.workloads import ref_test_model, config
model, tree = ref_test_model(), config("tree")
fp = host.simulate(model, tree, 1000000, 42)
s = driver.CppSampler(10, 50, devices=[0], capacity=16)
s.reset(model, tree, fp)
for i in range(3):
    s.reset(model); s.run_mcmc(42, i)
tr = tm = 0.0
for i in range(10):
    t0 = time.perf_counter(); s.reset(model); t1 = time.perf_counter(); s.run_mcmc(42, 3 + i); t2 = time.perf_counter()
    tr += t1 - t0; tm += t2 - t1
print("per step: reset %.3f ms, run_mcmc %.3f ms" % (tr * 100, tm * 100))
