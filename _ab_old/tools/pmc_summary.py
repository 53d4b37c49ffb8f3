"""Summarise a rocprofv3 --pmc counter_collection.csv: per kernel-name, mean of each counter per dispatch."""
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
pat = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    if pat and not re.search(pat, r["Kernel_Name"]):
        continue
    name = re.sub(r"\(.*", "", r["Kernel_Name"])[-70:]
    agg[(name, r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print(k)
    for c, vals in sorted(v.items()):
        print("   %-28s %14.0f  (n=%d)" % (c, sum(vals) / len(vals), len(vals)))
