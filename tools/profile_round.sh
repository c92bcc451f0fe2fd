#!/bin/bash
# Measurement pass of a round on the GPU box (run from the repo root through gpurun):
#   bash tools/profile_round.sh <tag>
# writes under gpurun_out/<tag>/: bench.json (the driver's command line), kernel-trace stats of the
# same command, the SQ (VALU issue) counters and the two HBM traffic passes, each in its own
# rocprofv3 run (counters never together with a trace domain), and the summaries under profiles/.
set -o pipefail
tag=$1
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
python bench.py > $out/bench.json 2> $out/bench.err || exit 1
cat $out/bench.json
rocprofv3 --kernel-trace --stats -d $out/trace -o t -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-reference-leg > $out/bench_traced.json 2> $out/trace.err || exit 1
python tools/kstats.py $(find $out/trace -name "*kernel_trace.csv" -o -name "*results.db" | head -1) > $out/kstats.txt
cat $out/kstats.txt
# counters: one context per GPU so that every launch covers a whole colour phase, in the phase mode
# the bench line above ran in (three contexts of 1736 waves each take the fused kernel; one context
# of 5209 waves would not by itself)
export EPV_FUSED_PHASE=$(python -c "import json; print(1 if json.load(open('$out/bench.json'))['roofline'].get('phase_mode') == 3 else 0)")
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVES -d $out/pmc_sq -o s -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-reference-leg --shards-per-gpu 1 > $out/pmc_sq.json 2> $out/pmc_sq.err || exit 1
rocprofv3 --pmc FETCH_SIZE -d $out/pmc_fetch -o f -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-reference-leg --shards-per-gpu 1 > /dev/null 2> $out/pmc_fetch.err || exit 1
rocprofv3 --pmc WRITE_SIZE -d $out/pmc_write -o w -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-reference-leg --shards-per-gpu 1 > /dev/null 2> $out/pmc_write.err || exit 1
units=$(python -c "print((1000000 - 2) / 3.0 * 4)")
python profiles/summarize_valu.py $(find $out/pmc_sq -name "*results.db" | head -1) tree $tag $units
python profiles/summarize_pmc.py $(find $out/pmc_fetch -name "*results.db" | head -1) $(find $out/pmc_write -name "*results.db" | head -1) tree $tag
cp profiles/${tag}_pmc_valu_tree.csv profiles/${tag}_pmc_hbm_tree.csv profiles/issue.json profiles/traffic.json $out/
cp $out/bench.json $out/${tag}_bench.json
