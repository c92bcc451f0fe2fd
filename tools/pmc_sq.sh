#!/bin/bash
# SQ counters (one context) of a bench configuration:  bash tools/pmc_sq.sh <tag> --config bal16 --sites 1250000
tag=$1; shift
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
cp profiles/issue.json /tmp/issue.json.keep
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVES -d $out/pmc_sq -o s -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-reference-leg --shards-per-gpu 1 "$@" > /dev/null 2> $out/pmc_sq.err || exit 1
python profiles/summarize_valu.py $(find $out/pmc_sq -name "*results.db" | head -1) scratch_$tag $tag 1 | cut -d, -f1-3,5-8,11-12
mv profiles/${tag}_pmc_valu_scratch_$tag.csv $out/ 2>/dev/null
cp /tmp/issue.json.keep profiles/issue.json
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_LDS -d $out/pmc_mix -o s -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-reference-leg --shards-per-gpu 1 "$@" > /dev/null 2> $out/pmc_mix.err || exit 1
python - <<PY
import sqlite3, collections, glob
db = glob.glob("$out/pmc_mix/**/*results.db", recursive=True)[0]
cur = sqlite3.connect(db).cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
pmc = [t for t in tabs if "pmc_event" in t][0]; info = [t for t in tabs if "info_pmc" in t][0]
kd = [t for t in tabs if "kernel_dispatch" in t][0]; ks = [t for t in tabs if "kernel_symbol" in t][0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
q = "select s.kernel_name, i.name, e.value, d.id from %s e join %s i on e.pmc_id = i.id join %s d on e.event_id = d.event_id join %s s on d.kernel_id = s.id" % (pmc, info, kd, ks)
try:
    rows = list(cur.execute(q))
except Exception as ex:
    print("query failed", ex, tabs); rows = []
for name, c, v, did in rows:
    k = name.split("(")[0].split("<")[0].replace("void ", "")
    acc[k][c] += v; cnt[k].add(did)
for k in sorted(acc):
    if "epv_mh" in k:
        n = len(cnt[k])
        print(k, n, {c: round(v / n / 1e6, 3) for c, v in sorted(acc[k].items())})
PY
