"""Per-kernel duration statistics from a rocprofv3 --kernel-trace run: pass either the
*_kernel_trace.csv or the rocpd *_results.db it wrote."""
import collections
import csv
import sqlite3
import sys


def rows_from(path):
    if path.endswith(".db"):
        cur = sqlite3.connect(path).cursor()
        tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
        kd = [t for t in tabs if "kernel_dispatch" in t][0]
        ks = [t for t in tabs if "kernel_symbol" in t][0]
        for name, a, b in cur.execute("select s.kernel_name, d.start, d.end from %s d join %s s "
                                      "on d.kernel_id = s.id" % (kd, ks)):
            yield name, (b - a) / 1e3
    else:
        for r in csv.DictReader(open(path)):
            yield r["Kernel_Name"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3


acc = collections.defaultdict(list)
for name, us in rows_from(sys.argv[1]):
    name = name.split("(")[0]
    name = (name[5:] if name.startswith("void ") else name).split("<")[0]
    if name.startswith("_Z"):
        import re
        m = re.match(r"_Z(\d+)", name)
        name = name[2 + len(m.group(1)):][:int(m.group(1))]
    if name.startswith("epv_"):
        acc[name].append(us)
for k, v in sorted(acc.items()):
    v2 = sorted(v)
    print("%-26s n=%5d avg %8.1f us  med %8.1f  min %8.1f  max %8.1f" % (k, len(v), sum(v) / len(v), v2[len(v2) // 2], v2[0], v2[-1]))
