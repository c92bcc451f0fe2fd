#!/bin/bash
# A/B of the large-tree proposal kernel on a bench configuration: bench line + kernel trace for EPV_PROPOSE_V3=1 / 0
#   bash tools/ab_p3.sh <tag> --config bal16 --sites 1250000
tag=$1; shift
export TMPDIR=/tmp
for v in 1 0; do
  out=gpurun_out/$tag/v3_$v
  mkdir -p $out
  export EPV_PROPOSE_V3=$v
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-reference-leg "$@" > $out/bench.json 2> $out/bench.err || { tail -5 $out/bench.err; exit 1; }
  python -c "import json; d=json.load(open('$out/bench.json')); print('V3=$v bench value %.4g  ms/step %.3f  launch %.1f us  mode %d' % (d['value'], d['ms_per_step'], 1e3*d['roofline']['avg_launch_ms'], d['roofline']['phase_mode']))"
  rocprofv3 --kernel-trace --stats -d $out/trace -o t -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-reference-leg --shards-per-gpu 1 "$@" > $out/bench_traced.json 2> $out/trace.err || exit 1
  python tools/kstats.py $(find $out/trace -name "*results.db" | head -1) > $out/kstats.txt
  cat $out/kstats.txt
done
