"""Rows of a rocprofv3 --pmc run, from either output format: the counter_collection.csv of
--output-format csv or the rocpd results.db (the default of ROCm 7.2), whose counters_collection
view holds one row per counter INSTANCE (shader engine / XCC) -- summed per dispatch here, as the
csv writer does."""
import collections
import csv
import sqlite3


def rows(path):
    if not path.endswith(".db"):
        for r in csv.DictReader(open(path)):
            yield r
        return
    cur = sqlite3.connect(path).cursor()
    acc = collections.OrderedDict()
    q = ("select dispatch_id, kernel_name, counter_name, value, vgpr_count, accum_vgpr_count, sgpr_count, "
         "lds_block_size, scratch_size, workgroup_size from counters_collection")
    for d, k, c, v, vg, ag, sg, lds, scr, wg in cur.execute(q):
        key = (d, c)
        if key not in acc:
            acc[key] = {"Kernel_Name": k, "Counter_Name": c, "Counter_Value": 0.0, "VGPR_Count": vg,
                        "Accum_VGPR_Count": ag, "SGPR_Count": sg, "LDS_Block_Size": lds, "Scratch_Size": scr,
                        "Workgroup_Size": wg, "Dispatch_Id": d}
        acc[key]["Counter_Value"] += float(v)
    for r in acc.values():
        yield r
