#!/bin/bash
# instruction mix of the kernels (wave-instructions per launch by unit), one context per GPU:
#   bash tools/prof_mix.sh <tag> [extra bench args]
tag=$1; shift
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
export EPV_FUSED_PHASE=${EPV_FUSED_PHASE:-1}
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_WAVES -d $out/pmc_mix -o m -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-reference-leg --shards-per-gpu 1 "$@" > /dev/null 2> $out/pmc_mix.err || { tail -5 $out/pmc_mix.err; exit 1; }
python - <<PY
import sys, collections
sys.path.insert(0, "profiles")
import pmc_load, glob
path = glob.glob("$out/pmc_mix/**/*results.db", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in pmc_load.rows(path):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
    if k.startswith("epv_"):
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_BRANCH", "SQ_WAVES"]
print("kernel,launches," + ",".join(n.lower() for n in names))
for k in sorted(acc):
    v = acc[k]
    print(k + ",%d," % max(len(x) for x in v.values()) + ",".join("%.0f" % (sum(v[n]) / len(v[n])) if v.get(n) else "nan" for n in names))
PY
