#!/bin/bash
# fused phase: contexts per GPU x sites per wave at one size:  bash tools/fused_scan2.sh <config> <n>
cfg=$1; n=$2
for k in 1 2 3 4; do for l in 64 32; do
  EPV_FUSED_PHASE=1 EPV_FUSED_LANES=$l python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-reference-leg --config $cfg --sites $n --shards-per-gpu $k 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('$cfg n=$n k=$k lanes=$l  %.3e  %.2f ms/step  launch %.4f ms' % (j['value'], j['ms_per_step'], j['roofline']['avg_launch_ms']))"
done; done
