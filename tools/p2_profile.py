"""wave-time per section of epv_mh_propose2_kernel (build_ab/libepv_prof.so: tools/ab_defs.py build
"prof=-DEPV_P2_PROFILE").  python tools/p2_profile.py [sites] [config]"""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + '/tests')
os.environ.setdefault("EPIEVO_MI355X_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'build_ab', 'libepv_prof.so'))
from epievo_amd.workloads import simulate
from epievo_amd.sampler import DeviceSampler, lib
N_SITES = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
CFG = sys.argv[2] if len(sys.argv) > 2 else "tree"
model, tree, fp = simulate(CFG, N_SITES, seed=42)
d = DeviceSampler(0); d.set_tree(tree); d.set_model(model); d.upload_paths(fp, 0 if CFG == "pair" else 16); d.reset()
d.sweep(2, 1, 0)
out = (C.c_ulonglong * 16)()
lib().epv_debug_p2_profile(out)
base = list(out)
d.sweep(10, 1, 2)
lib().epv_debug_p2_profile(out)
tot = [out[i] - base[i] for i in range(16)]
waves = tot[15]
names = ["start..staged+meta", "scan/plan", "descriptor pass", "dense eval", "pruning", "downward", "hand-over",
         "fused: search", "fused: assemble", "fused: accept"]
s = sum(tot[:10])
for n_, t in zip(names, tot):
    print("%-22s %8.0f ticks/wave  %5.1f %%" % (n_, t / waves, 100.0 * t / s))
print("total %.0f ticks per wave (%d waves)" % (s / waves, waves))
