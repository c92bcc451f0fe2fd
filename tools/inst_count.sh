#!/bin/bash
# VALU wave-instructions per launch of the MH kernels for library variants (tools/ab_defs.py build):
#   bash tools/inst_count.sh <tag> NAME ...     (one context, fused phase forced)
# per stage of the fused phase: python tools/ab_defs.py build "s0=" "s1=-DEPV_DBG_SKIP=1" "s2=-DEPV_DBG_SKIP=2" "s3=-DEPV_DBG_SKIP=3"
tag=$1; shift
export TMPDIR=/tmp
export EPV_FUSED_PHASE=1
cp profiles/issue.json /tmp/issue.json.keep      # summarize_valu.py adds an entry per call
for name in "$@"; do
  out=gpurun_out/$tag/$name
  mkdir -p $out
  export EPIEVO_MI355X_LIB=$PWD/build_ab/libepv_$name.so
  rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_THREAD_CYCLES_VALU -d $out/pmc -o s -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-reference-leg --shards-per-gpu 1 --driver torch > /dev/null 2> $out/err.txt || exit 1
  echo "== $name"
  python profiles/summarize_valu.py $(find $out/pmc -name "*results.db" | head -1) scratch_$name $tag 1 2>/dev/null | grep -E "propose2" | cut -d, -f1-3,5,6,10,11
  rm -f profiles/${tag}_pmc_valu_scratch_$name.csv
done
cp /tmp/issue.json.keep profiles/issue.json
