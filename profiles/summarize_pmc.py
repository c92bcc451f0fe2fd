#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc counter_collection CSVs (FETCH_SIZE and WRITE_SIZE collected
in SEPARATE passes, as MI355X_MICROARCH.md prescribes) into per-kernel HBM bytes per
launch, and write profiles/traffic.json for bench.py's roofline.traffic field.

  python profiles/summarize_pmc.py <fetch_counter_collection.csv> <write_counter_collection.csv> <config> <round-tag>

Units: FETCH_SIZE / WRITE_SIZE are KiB.  On gfx950 FETCH_SIZE reads HALF the bytes of the
128-byte lines a kernel touches -- calibrated for this sampler's own access patterns (2- and
8-byte loads per lane, unit stride, every third element, one lane in sixteen) by
tools/fetch_calib.hip: profiles/r03_fetch_calibration.txt, ratio 0.500 in every case -- so the
read side is doubled (hbm_bytes_per_launch_fetch_x2 is the figure bench.py cites; the raw sum is
kept beside it).  WRITE_SIZE was checked on epv_reset_kernel, which writes exactly n*8 bytes of
tri: 7812.5 KiB for n = 1e6, i.e. exact.
"""
import collections
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import pmc_load  # noqa: E402


def per_kernel(path, counter):
    acc = collections.defaultdict(list)
    for r in pmc_load.rows(path):
        if r["Counter_Name"] == counter:
            name = r["Kernel_Name"].split("(")[0]
            name = name[5:] if name.startswith("void ") else name   # templates: "void k<false>"
            acc[name.split("<")[0]].append(float(r["Counter_Value"]))
    return {k: (len(v), sum(v) / len(v) * 1024.0) for k, v in acc.items()}


def main():
    fetch_csv, write_csv, config, tag = sys.argv[1:5]
    f, w = per_kernel(fetch_csv, "FETCH_SIZE"), per_kernel(write_csv, "WRITE_SIZE")
    here = os.path.dirname(os.path.abspath(__file__))
    rows = []
    for k in sorted(set(f) | set(w)):
        nf, bf = f.get(k, (0, 0.0))
        nw, bw = w.get(k, (0, 0.0))
        rows.append((k, nf, bf, bw))
    with open(os.path.join(here, "%s_pmc_hbm_%s.csv" % (tag, config)), "w") as out:
        out.write("kernel,launches,fetch_bytes_per_launch,write_bytes_per_launch,hbm_bytes_per_launch\n")
        for k, n, bf, bw in rows:
            out.write("%s,%d,%.0f,%.0f,%.0f\n" % (k, n, bf, bw, bf + bw))
    tj = os.path.join(here, "traffic.json")
    data = json.load(open(tj)) if os.path.exists(tj) else {}
    parts = [r for r in rows if r[0] in ("epv_mh_phase_kernel", "epv_mh_propose_kernel", "epv_mh_propose2_kernel",
                                         "epv_mh_propose3_kernel", "epv_mh_jumps_kernel", "epv_mh_jumps_all_kernel", "epv_mh_accept_kernel",
                                         "epv_mh_accept3_kernel")]
    mh = ("+".join(r[0] for r in parts), parts[0][1], sum(r[2] for r in parts), sum(r[3] for r in parts))
    data[config] = {"kernel": mh[0], "round": tag,
                    "fetch_bytes_per_launch": mh[2], "write_bytes_per_launch": mh[3],
                    "hbm_bytes_per_launch": mh[2] + mh[3],
                    "hbm_bytes_per_launch_fetch_x2": 2 * mh[2] + mh[3]}
    json.dump(data, open(tj, "w"), indent=1, sort_keys=True)
    print(json.dumps(data[config]))


if __name__ == "__main__":
    main()
