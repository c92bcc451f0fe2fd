"""BASELINE config 3 end to end through the drop-in CLI: tree.nwk, n = 1e6,
epievo_est_params_histories -i 20 -B 50 (-L 10), file IO included."""
import os, subprocess, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
from epievo_amd.workloads import simulate, TEST_PARAM_TEXT, TREE_NWK_TEXT
from epievo_amd import host, _build
d = tempfile.mkdtemp()
open(d + "/p.param", "w").write(TEST_PARAM_TEXT); open(d + "/t.nwk", "w").write(TREE_NWK_TEXT)
model, tree, fp = simulate("tree", 1000000, seed=42)
host.write_paths(d + "/in.paths", tree.node_names, tree.branches, fp)
t0 = time.time()
r = subprocess.run([_build.BIN_DIR + "/epievo_est_params_histories", "-i", "20", "-B", "50", "-L", "10", "-s", "42",
                    "-o", d + "/out.paths", "-p", d + "/out.param", "-v", d + "/p.param", d + "/t.nwk", d + "/in.paths"],
                   capture_output=True, text=True)
el = time.time() - t0
print("config 3 (n=1e6, -i 20 -B 50 -L 10): %.2f s wall, rc=%d -> %.3e site-branch resamples/s end to end"
      % (el, r.returncode, 20 * 60 * 999998 * 4 / el))
print(r.stderr[-900:])
