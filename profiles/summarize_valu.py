#!/usr/bin/env python3
"""Aggregate a rocprofv3 --pmc run over the SQ counters into per-kernel instruction-issue
figures, write profiles/<tag>_pmc_valu_<config>.csv and the entry of profiles/issue.json that
bench.py's roofline.issue block reads.

  python profiles/summarize_valu.py <counter_collection.csv> <config> <round-tag> <resamples per launch>

Counters (one pass, 8 SQ slots): SQ_INSTS_VALU (VALU wave-instructions), SQ_ACTIVE_INST_VALU and
SQ_WAVE_CYCLES / SQ_WAIT_ANY / SQ_WAIT_INST_ANY / SQ_BUSY_CYCLES (quad-cycles, see
MI355X_MICROARCH.md "rocprofv3 PMC slots"), SQ_THREAD_CYCLES_VALU (active lanes summed over the
VALU instructions: / SQ_INSTS_VALU / 64 = lane utilisation), SQ_WAVES.  The csv also carries
each kernel's VGPR count, LDS and scratch size per work-item.
Issue bound of a launch: a wave64 VALU instruction (fp64 or 32-bit) occupies its SIMD for 4
cycles, the chip has 256 CUs x 4 SIMDs, so t_issue = SQ_INSTS_VALU * 4 / (1024 * f_clk).
"""
import collections
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import pmc_load  # noqa: E402

CLOCK_GHZ = 2.4      # MI355X peak engine clock
SIMDS = 1024
MH = ("epv_mh_propose_kernel", "epv_mh_propose2_kernel", "epv_mh_propose3_kernel", "epv_mh_jumps_kernel", "epv_mh_jumps_all_kernel",
      "epv_mh_accept_kernel", "epv_mh_accept3_kernel")


def short(name):
    name = name.split("(")[0]
    name = name[5:] if name.startswith("void ") else name
    return name.split("<")[0]


def main():
    path, config, tag = sys.argv[1:4]
    units = float(sys.argv[4])     # site-branch resamples one profiled colour-phase launch covers
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    res = {}
    for r in pmc_load.rows(path):
        k = short(r["Kernel_Name"])
        if not k.startswith("epv_"):
            continue
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        res[k] = (int(r["VGPR_Count"]), int(r["Accum_VGPR_Count"]), int(r["SGPR_Count"]), int(r["LDS_Block_Size"]),
                  int(r["Scratch_Size"]), int(r["Workgroup_Size"]))
    names = ["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY",
             "SQ_WAIT_INST_ANY", "SQ_BUSY_CYCLES", "SQ_WAVES"]
    here = os.path.dirname(os.path.abspath(__file__))
    out = os.path.join(here, "%s_pmc_valu_%s.csv" % (tag, config))
    total = tc = 0.0
    with open(out, "w") as f:
        f.write("kernel,launches," + ",".join(n.lower() + "_per_launch" for n in names) +
                ",lane_utilisation,issue_bound_us,vgpr,agpr,sgpr,lds_bytes_per_block,scratch_bytes_per_lane,block\n")
        for k in sorted(acc):
            v = acc[k]
            mean = {n: (sum(v[n]) / len(v[n]) if v.get(n) else float("nan")) for n in names}
            n_l = max(len(x) for x in v.values())
            insts = mean["SQ_INSTS_VALU"]
            util = mean["SQ_THREAD_CYCLES_VALU"] / insts / 64.0 if insts else float("nan")
            bound = insts * 4.0 / (SIMDS * CLOCK_GHZ * 1e3)
            f.write("%s,%d,%s,%.3f,%.1f,%s\n" % (k, n_l, ",".join("%.0f" % mean[n] for n in names), util, bound,
                                                 ",".join(str(x) for x in res[k])))
            if k in MH:
                total += insts
                tc += mean["SQ_THREAD_CYCLES_VALU"]
    ij = os.path.join(here, "issue.json")
    data = json.load(open(ij)) if os.path.exists(ij) else {}
    mh = [k for k in sorted(acc) if k in MH]
    # EPV_PHASE_* of include/epievo_mi355x.h, from the kernels that ran
    mode = 3 if mh == ["epv_mh_propose2_kernel"] else 4 if "epv_mh_propose3_kernel" in mh else \
        0 if "epv_mh_propose_kernel" in mh else \
        2 if "epv_seg_search_kernel" in acc else 1
    data[config] = {"round": tag, "kernels": "+".join(mh), "phase_mode": mode,
                    "valu_wave_insts_per_launch": total, "cycles_per_inst": 4, "simds": SIMDS,
                    "clock_ghz": CLOCK_GHZ, "resamples_per_launch": units,
                    "lane_utilisation": tc / total / 64.0 if total else None}
    json.dump(data, open(ij, "w"), indent=1, sort_keys=True)
    print(open(out).read())


if __name__ == "__main__":
    main()
