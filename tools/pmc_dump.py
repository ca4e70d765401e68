"""Per-kernel averages of whatever counters a rocprofv3 --pmc run collected.  Usage: pmc_dump.py <dir> [name filter]"""
import collections, csv, glob, os, sys
f = glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True)[0]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
per = collections.defaultdict(dict)
for r in csv.DictReader(open(f)):
    d = per[r["Dispatch_Id"]]
    d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    d["name"] = r["Kernel_Name"]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for d in per.values():
    k = d["name"].replace("(anonymous namespace)::", "").replace("void ", "")[:60]
    if flt not in k:
        continue
    cnt[k] += 1
    for c, v in d.items():
        if c != "name":
            agg[k][c] += v
for k in agg:
    print(k, "calls", cnt[k])
    for c, v in sorted(agg[k].items()):
        print(f"   {c:32s} {v / cnt[k]:16.1f}")
