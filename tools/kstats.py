import csv, collections, sys
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(list)
for r in rows:
    name = r["Kernel_Name"].split("(")[0]
    if name.startswith("epv_"):
        acc[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(acc.items()):
    v2 = sorted(v)
    print("%-26s n=%4d avg %8.1f us  med %8.1f  max %8.1f" % (k, len(v), sum(v)/len(v), v2[len(v2)//2], max(v)))
