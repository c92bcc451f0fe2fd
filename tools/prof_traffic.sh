#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE in separate passes) of the current build, one context:
#   bash tools/prof_traffic.sh <tag> [extra bench args]
tag=$1; shift
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE -d $out/pmc_fetch -o f -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-reference-leg --shards-per-gpu 1 "$@" > /dev/null 2> $out/pmc_fetch.err || exit 1
rocprofv3 --pmc WRITE_SIZE -d $out/pmc_write -o w -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-reference-leg --shards-per-gpu 1 "$@" > /dev/null 2> $out/pmc_write.err || exit 1
cp profiles/traffic.json $out/traffic_before.json
python profiles/summarize_pmc.py $(find $out/pmc_fetch -name "*results.db" | head -1) $(find $out/pmc_write -name "*results.db" | head -1) scratch $tag
grep "mh_\|suffstat" profiles/${tag}_pmc_hbm_scratch.csv
mv profiles/${tag}_pmc_hbm_scratch.csv $out/
cp $out/traffic_before.json profiles/traffic.json
