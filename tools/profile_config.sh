#!/bin/bash
# PMC evidence of a non-default configuration (one context per GPU, each counter set in its own rocprofv3
# pass) -> profiles/<tag>_pmc_{valu,hbm}_<config>.csv + the config's entries of issue.json / traffic.json,
# and the bench line of the configuration with those entries in effect:
#   bash tools/profile_config.sh <tag> <config> <sites> <bench-json-name>
set -o pipefail
tag=$1; cfg=$2; sites=$3; name=$4
out=gpurun_out/${tag}_$cfg
mkdir -p $out
export TMPDIR=/tmp
args="--config $cfg --sites $sites --steps 1 --warmup 0 --no-cpu-baseline --no-reference-leg --shards-per-gpu 1"
# pair at n = 1e5 runs the fused phase with several contexts; profile that kernel path
if [ "$cfg" = "pair" ]; then export EPV_FUSED_PHASE=1; fi
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVES -d $out/pmc_sq -o s -- python3 bench.py $args > /dev/null 2> $out/pmc_sq.err || exit 1
rocprofv3 --pmc FETCH_SIZE -d $out/pmc_fetch -o f -- python3 bench.py $args > /dev/null 2> $out/pmc_fetch.err || exit 1
rocprofv3 --pmc WRITE_SIZE -d $out/pmc_write -o w -- python3 bench.py $args > /dev/null 2> $out/pmc_write.err || exit 1
units=$(python -c "
import sys
sys.path.insert(0, '.')
from epievo_amd.workloads import config
print(($sites - 2) / 3.0 * (config('$cfg').n_nodes - 1))")
python profiles/summarize_valu.py $(find $out/pmc_sq -name "*results.db" | head -1) $cfg $tag $units | cut -d, -f1-3,5-8,11-12
python profiles/summarize_pmc.py $(find $out/pmc_fetch -name "*results.db" | head -1) $(find $out/pmc_write -name "*results.db" | head -1) $cfg $tag
unset EPV_FUSED_PHASE
python bench.py --config $cfg --sites $sites --steps 3 --warmup 1 --no-cpu-baseline > $out/${tag}_${name}.json 2> $out/bench.err || exit 1
python -c "import json; d=json.load(open('$out/${tag}_${name}.json')); print('$cfg value %.4g  ms/step %.3f  frac %.4f issue %s' % (d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['issue'] and d['roofline']['issue']['frac']))"
cp profiles/${tag}_pmc_valu_$cfg.csv profiles/${tag}_pmc_hbm_$cfg.csv profiles/issue.json profiles/traffic.json $out/
