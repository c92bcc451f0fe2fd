#!/bin/bash
# quick kernel-trace + SQ counters of the current build, one context per GPU:
#   bash tools/prof_quick.sh <tag> [extra bench args]
tag=$1; shift
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $out/trace -o t -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-reference-leg --shards-per-gpu 1 "$@" > $out/b.json 2> $out/trace.err || exit 1
python tools/kstats.py $(find $out/trace -name "*results.db" | head -1) > $out/kstats.txt
cat $out/kstats.txt
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVES -d $out/pmc_sq -o s -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-reference-leg --shards-per-gpu 1 "$@" > /dev/null 2> $out/pmc_sq.err || exit 1
python profiles/summarize_valu.py $(find $out/pmc_sq -name "*results.db" | head -1) scratch_$tag $tag 1 | cut -d, -f1-3,5-8,11-12
rm -f profiles/${tag}_pmc_valu_scratch_$tag.csv
