#!/bin/bash
# kernel trace (default contexts) + SQ counters and instruction mix (one context) of a bench configuration:
#   bash tools/prof_cfg.sh <tag> --config bal16 --sites 1250000
tag=$1; shift
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
python bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $out/bench.json 2> $out/bench.err || { tail -5 $out/bench.err; exit 1; }
python -c "import json; d=json.load(open('$out/bench.json')); print('bench value %.4g  ms/step %.3f  launch %.1f us  mode %d  ref-arith %.4g' % (d['value'], d['ms_per_step'], 1e3*d['roofline']['avg_launch_ms'], d['roofline']['phase_mode'], d['reference_proposal_arithmetic']['value']))"
rocprofv3 --kernel-trace --stats -d $out/trace -o t -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-reference-leg "$@" > $out/bench_traced.json 2> $out/trace.err || exit 1
python tools/kstats.py $(find $out/trace -name "*results.db" | head -1) > $out/kstats.txt
cat $out/kstats.txt
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVES -d $out/pmc_sq -o s -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-reference-leg --shards-per-gpu 1 "$@" > /dev/null 2> $out/pmc_sq.err || exit 1
python profiles/summarize_valu.py $(find $out/pmc_sq -name "*results.db" | head -1) scratch_$tag $tag 1 | cut -d, -f1-3,5-8,11-12
mv profiles/${tag}_pmc_valu_scratch_$tag.csv $out/ 2>/dev/null
git checkout profiles/issue.json 2>/dev/null
