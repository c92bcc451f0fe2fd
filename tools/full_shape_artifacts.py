#!/usr/bin/env python3
"""Timings of BASELINE.json's configurations at their stated shapes on ONE MI355X, kept under
profiles/ (run on the GPU box through gpurun; results land in gpurun_out/<tag>/ and are copied):
  python tools/full_shape_artifacts.py <tag>
    <tag>_bench_config2_pair_n1e5.json      bench.py --config pair --sites 100000
    <tag>_bench_config4_genome_n1e7.json    bench.py --sites 10000000  (config 4's genome on one GPU)
    <tag>_bench_config5_shard_bal16.json    bench.py --config bal16 --sites 1250000 (one GPU's share)
    <tag>_bench_config5_full_n1e7.json      bench.py --config bal16 --sites 10000000 (the stated size on ONE GPU, 77 GB)
    <tag>_bench_rehearsal_4slots.json       EPV_DEVICES=0,0,0,0 bench.py --gpus 4 (the N > 1 entry, loopback transport)
    <tag>_config3_cli_e2e.txt               config 3 through the drop-in CLI, file IO included;
                                            one context, the default three, and four rehearsal slots
"""
import json, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
tag = sys.argv[1]
out = os.path.join(ROOT, "gpurun_out", tag)
os.makedirs(out, exist_ok=True)


def bench(name, *args, env=None):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1",
                        "--no-cpu-baseline"] + list(args), capture_output=True, text=True,
                       env=dict(os.environ, **(env or {})))
    line = r.stdout.strip().split("\n")[-1] if r.stdout.strip() else ""
    open(os.path.join(out, "%s_%s.json" % (tag, name)), "w").write(line + "\n")
    try:
        j = json.loads(line)
        print("%-34s %.4e resamples/s  %.2f ms/step" % (name, j["value"], j["ms_per_step"]), flush=True)
    except Exception:
        print(name, "FAILED", r.stderr[-500:], flush=True)


bench("bench_config2_pair_n1e5", "--config", "pair", "--sites", "100000")
bench("bench_config4_genome_n1e7", "--sites", "10000000")
bench("bench_config5_shard_bal16", "--config", "bal16", "--sites", "1250000")
bench("bench_config5_full_n1e7", "--config", "bal16", "--sites", "10000000", "--no-reference-leg")
bench("bench_rehearsal_4slots", "--gpus", "4", "--sites", "500000", env={"EPV_DEVICES": "0,0,0,0"})

from epievo_amd.workloads import simulate, TEST_PARAM_TEXT, TREE_NWK_TEXT   # noqa: E402
from epievo_amd import host, _build                                        # noqa: E402
d = tempfile.mkdtemp()
open(d + "/p.param", "w").write(TEST_PARAM_TEXT)
open(d + "/t.nwk", "w").write(TREE_NWK_TEXT)
model, tree, fp = simulate("tree", 1000000, seed=42)
host.write_paths(d + "/in.paths", tree.node_names, tree.branches, fp)
lines = []
for name, env, extra in (("one context", {"EPV_CONTEXTS_PER_GPU": "1"}, []), ("three contexts (default)", {}, []),
                         ("three contexts, -e 20 (paths file written once)", {}, ["-e", "20"]),
                         ("four rehearsal slots on one GPU", {"EPV_DEVICES": "0,0,0,0", "EPV_CONTEXTS_PER_GPU": "1"}, [])):
    t0 = time.time()
    r = subprocess.run([_build.BIN_DIR + "/epievo_est_params_histories", "-i", "20", "-B", "50", "-L", "10", "-s", "42",
                        "-o", d + "/out.paths", "-p", d + "/out.param", "-v"] + extra + [d + "/p.param", d + "/t.nwk",
                        d + "/in.paths"], capture_output=True, text=True, env=dict(os.environ, **env))
    el = time.time() - t0
    last = [l for l in r.stderr.split("\n") if l and l[0].isdigit()][-1:]
    lines.append("config 3 (tree.nwk, n=1e6, -i 20 -B 50 -L 10) through epievo_est_params_histories, %s: %.2f s wall "
                 "(rc %d), %.3e site-branch resamples/s end to end incl. reading and writing the 111 MB paths file "
                 "every iteration; last -v line: %s" % (name, el, r.returncode, 20 * 60 * 999998 * 4 / el, last))
    print(lines[-1], flush=True)
open(os.path.join(out, "%s_config3_cli_e2e.txt" % tag), "w").write("\n".join(lines) + "\n")
