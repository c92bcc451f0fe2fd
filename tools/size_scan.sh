#!/bin/bash
# per-kernel duration against the number of sites (one context per GPU): how much of a launch
# is ramp/tail and how the waves of one SIMD share it.  bash tools/size_scan.sh <tag> n1 n2 ...
tag=$1; shift
export TMPDIR=/tmp
for n in "$@"; do
  out=gpurun_out/$tag/n$n
  mkdir -p $out
  rocprofv3 --kernel-trace --stats -d $out/trace -o t -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-reference-leg --shards-per-gpu 1 --sites $n > $out/b.json 2> $out/trace.err || exit 1
  echo "== n=$n  waves/phase=$(( (n / 3 + 63) / 64 ))"
  python tools/kstats.py $(find $out/trace -name "*results.db" | head -1) | grep -E "propose2|jumps|accept"
done
