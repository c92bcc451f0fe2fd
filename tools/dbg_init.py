import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, orc
from common import simulate
from epievo_amd import host
from epievo_amd.sampler import DeviceSampler
model, tree, fp = simulate("pair", 5000, seed=3)
root = fp.init; leaf = fp.init ^ (fp.counts() & 1).astype(np.uint8)
d = DeviceSampler(0); d.set_tree(host.Tree.single_branch(1.0)); d.set_model(model)
d.init_paths_indep(root, leaf, seed=77, capacity=32)
g = d.paths(); e = orc.init_paths_indep("orc", 77, model.rates, root, leaf, 1.0, "B")
print("init eq", np.array_equal(g.init, e.init), "counts eq", np.array_equal(g.counts(), e.counts()), len(g.jumps), len(e.jumps))
bad = np.nonzero(g.counts() != e.counts())[0]
print("n bad", len(bad), bad[:10], g.counts()[bad[:10]], e.counts()[bad[:10]], "colours", bad[:10] % 3)
if len(bad) == 0:
    diff = np.nonzero(g.jumps != e.jumps)[0]
    print("jump diffs", len(diff), diff[:5], g.jumps[diff[:5]], e.jumps[diff[:5]])
    site = np.searchsorted(g.offsets, diff[:5], side="right") - 1
    print("sites", site)
