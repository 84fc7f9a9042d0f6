"""Aggregate a rocprofv3 --pmc counter_collection.csv per kernel name (sum per counter / dispatches)."""
import csv, sys, collections
rows = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if len(sys.argv) > 2 and sys.argv[2] not in k:
        continue
    key = (k, r["Grid_Size"]) if len(sys.argv) > 3 else k
    rows[key][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[key].add(r["Dispatch_Id"])
for k, c in sorted(rows.items(), key=lambda kv: -sum(kv[1].values()))[:40]:
    n = len(cnt[k])
    print(str(k)[:110], "dispatches", n)
    print("   ", "  ".join("%s=%.4g" % (name, v / n) for name, v in sorted(c.items())))
