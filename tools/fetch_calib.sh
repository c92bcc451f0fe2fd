#!/bin/bash
# FETCH_SIZE against known byte counts in the sampler's access patterns -> profiles/<tag>_fetch_calibration.txt
tag=${1:-r03}
out=gpurun_out/${tag}_fetch_calib
mkdir -p $out
export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O3 -o $out/fetch_calib tools/fetch_calib.hip || exit 1
rocprofv3 --pmc FETCH_SIZE -d $out/pmc -o c -- $out/fetch_calib > $out/expected.csv 2> $out/pmc.err || { tail $out/pmc.err; exit 1; }
python - <<PY
import sys, glob, csv
sys.path.insert(0, "profiles")
import pmc_load
db = glob.glob("$out/pmc/**/*results.db", recursive=True)[0]
exp = list(csv.DictReader(open("$out/expected.csv")))
got = [float(r["Counter_Value"]) * 1024.0 for r in pmc_load.rows(db) if r["Counter_Name"] == "FETCH_SIZE" and "read_kernel" in r["Kernel_Name"]]
lines = ["pattern,bytes_requested,bytes_of_touched_lines,FETCH_SIZE_bytes,FETCH_SIZE/touched_lines,FETCH_SIZE/requested"]
for e, g in zip(exp, got):
    lines.append("%s,%s,%s,%.0f,%.3f,%.3f" % (e["kernel"], e["bytes_requested"], e["bytes_of_touched_128B_lines"], g,
                                            g / float(e["bytes_of_touched_128B_lines"]), g / float(e["bytes_requested"])))
open("$out/${tag}_fetch_calibration.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
