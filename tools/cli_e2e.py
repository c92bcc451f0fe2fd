"""End-to-end wall time of the drop-in CLIs on the GPU box (BASELINE configs 2 and 3),
file IO included."""
import os, subprocess, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
import numpy as np
from epievo_amd.workloads import simulate, TEST_PARAM_TEXT, TREE_NWK_TEXT
from epievo_amd import host, _build
d = tempfile.mkdtemp()
open(d + "/p.param", "w").write(TEST_PARAM_TEXT); open(d + "/t.nwk", "w").write(TREE_NWK_TEXT)
# config 3: tree.nwk, n = 1e6, -B 50 -L 10 (3 EM iterations instead of 20 to keep this short)
model, tree, fp = simulate("tree", 1000000, seed=42)
host.write_paths(d + "/in.paths", tree.node_names, tree.branches, fp)
t0 = time.time()
r = subprocess.run([_build.BIN_DIR + "/epievo_est_params_histories", "-i", "3", "-B", "50", "-L", "10", "-s", "42",
                    "-o", d + "/out.paths", "-p", d + "/out.param", "-v", d + "/p.param", d + "/t.nwk", d + "/in.paths"],
                   capture_output=True, text=True)
el = time.time() - t0
print("config 3 (n=1e6, 3 EM iterations x 60 sweeps): %.2f s wall, rc=%d" % (el, r.returncode))
print(r.stderr[-400:])
# config 2: single branch T = 1.0, n = 1e5, -L 100
model, tree, fp = simulate("pair", 100000, seed=42)
root = fp.init; leaf = fp.init ^ (fp.counts() & 1).astype(np.uint8)
with open(d + "/obs.states", "w") as f:
    f.write("#root\tleaf\n")
    f.write("".join("%d\t%d\t%d\n" % (i, root[i], leaf[i]) for i in range(fp.n_sites)))
t0 = time.time()
r = subprocess.run([_build.BIN_DIR + "/epievo_sim_pairwise", "-L", "100", "-T", "1.0", "-s", "42", "-o", d + "/pw.paths",
                    "-v", d + "/p.param", d + "/obs.states"], capture_output=True, text=True)
el = time.time() - t0
print("config 2 (T=1.0, n=1e5, -L 100): %.2f s wall, rc=%d  -> %.3e resamples/s end to end" % (el, r.returncode, 100 * 99998 / el))
print(r.stderr[-200:])
