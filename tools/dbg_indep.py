import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, orc
from common import simulate
from epievo_amd.sampler import DeviceSampler
RATES = np.array([0.7, 1.9])
model, tree, fp = simulate("tree", 5000, seed=7)
d = DeviceSampler(0); d.set_tree(tree); d.set_model(model); d.upload_paths(fp, 32)
o = orc.Oracle(tree, model, fp, "B", cap=32, seed=123)
d.indep_update_paths(RATES, 123, sweep=0xF0000000); o.indep_update_paths(RATES, 0xF0000000)
g, e = d.paths(), o.paths()
B, n = 4, 5000
print("init eq", np.array_equal(g.init, e.init), "counts eq", np.array_equal(g.counts(), e.counts()), len(g.jumps), len(e.jumps))
bi = np.nonzero(g.init != e.init)[0]; print("init diffs", len(bi), bi[:10] // n, bi[:10] % n)
bc = np.nonzero(g.counts() != e.counts())[0]; print("count diffs", len(bc), bc[:10] // n, bc[:10] % n, g.counts()[bc[:10]], e.counts()[bc[:10]])
if len(bc) == 0 and len(bi) == 0:
    dj = np.nonzero(g.jumps != e.jumps)[0]; print("jump diffs", len(dj), g.jumps[dj[:5]], e.jumps[dj[:5]])
def consistent(p):
    init = p.init.reshape(B, n); es = init ^ (p.counts().reshape(B, n) & 1).astype(np.uint8)
    bad = []
    for b in range(B):
        par = tree.parent_ids[b + 1]
        if par: bad.append(int((init[b] != es[par - 1]).sum()))
    return bad
print("GPU inconsistent child inits:", consistent(g), " oracle:", consistent(e))
s = 628
for nm, p in (("gpu", g), ("orc", e)):
    c = p.counts().reshape(B, n)[:, s]; i = p.init.reshape(B, n)[:, s]
    print(nm, "site", s, "init", i, "counts", c)
print("orig", fp.init.reshape(B, n)[:, s], fp.counts().reshape(B, n)[:, s])
