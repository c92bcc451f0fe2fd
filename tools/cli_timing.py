#!/usr/bin/env python3
"""config 3 through epievo_est_params_histories with EPV_CLI_TIMING=1: where the iterations' wall
clock goes, for 1 and 3 contexts per GPU.  python tools/cli_timing.py"""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from epievo_amd.workloads import simulate, TEST_PARAM_TEXT, TREE_NWK_TEXT   # noqa: E402
from epievo_amd import host, _build                                        # noqa: E402
d = tempfile.mkdtemp()
open(d + "/p.param", "w").write(TEST_PARAM_TEXT)
open(d + "/t.nwk", "w").write(TREE_NWK_TEXT)
model, tree, fp = simulate("tree", 1000000, seed=42)
host.write_paths(d + "/in.paths", tree.node_names, tree.branches, fp)
import hashlib
digest = {}
for k, extra in (("1", []), ("3", []), ("3", ["-e", "20"]), ("1", []), ("3", []), ("3", ["-e", "20"])):
    t0 = time.time()
    r = subprocess.run([_build.BIN_DIR + "/epievo_est_params_histories", "-i", "20", "-B", "50", "-L", "10", "-s", "42",
                        "-o", d + "/out.paths", "-p", d + "/out.param", "-v"] + extra + [d + "/p.param", d + "/t.nwk",
                        d + "/in.paths"], capture_output=True, text=True,
                       env=dict(os.environ, EPV_CONTEXTS_PER_GPU=k, EPV_CLI_TIMING="1"))
    el = time.time() - t0
    digest[(k, tuple(extra))] = hashlib.sha256(open(d + "/out.paths", "rb").read()).hexdigest()[:16]
    print("contexts %s %s: %.2f s wall (rc %d), final paths file %s;" % (k, " ".join(extra) or "(default: every iteration)", el, r.returncode, digest[(k, tuple(extra))]),
          [l for l in r.stderr.split("\n") if l.startswith("[TIMING")], flush=True)
assert len(set(digest.values())) == 1, digest     # the same final file whatever the layout and the write cadence
