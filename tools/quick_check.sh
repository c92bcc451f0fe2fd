#!/bin/bash
# fast loop on the GPU box: the statistics / sharding parity tests, then the default bench shape
# under a kernel trace (three contexts, as the driver runs it):  bash tools/quick_check.sh <tag> [pytest -k expr]
tag=$1; kexpr=${2:-"run_mcmc or local_group or sharded or full_size or columns"}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q -k "$kexpr" > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest.log
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-reference-leg > $out/bench.json 2> $out/bench.err || exit 1
python -c "import json; d=json.load(open('$out/bench.json')); print('bench value %.4g  ms/step %.3f  launch %.1f us' % (d['value'], d['ms_per_step'], 1e3*d['roofline']['avg_launch_ms']))"
rocprofv3 --kernel-trace --stats -d $out/trace -o t -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-reference-leg > $out/bench_traced.json 2> $out/trace.err || exit 1
python tools/kstats.py $(find $out/trace -name "*results.db" | head -1) > $out/kstats.txt
cat $out/kstats.txt
