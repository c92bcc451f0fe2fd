#!/bin/bash
# fused colour phase vs separate kernels against the launch size:  bash tools/fused_scan.sh <config> n1 n2 ...
cfg=$1; shift
for n in "$@"; do for k in 1 2; do for f in 0 1; do
  EPV_FUSED_PHASE=$f python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-reference-leg --config $cfg --sites $n --shards-per-gpu $k 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('$cfg n=$n k=$k fused=$f  %.3e  %.2f ms/step  launch %.4f ms' % (j['value'], j['ms_per_step'], j['roofline']['avg_launch_ms']))"
done; done; done
