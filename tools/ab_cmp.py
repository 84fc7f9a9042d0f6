"""tools/ab_cmp.py <dir> <row-regex> [name ...]: per-layer microseconds of several bench.py detail dumps (tools/ab_rows.sh) side by side."""
import json, re, sys, os, glob
d = sys.argv[1]; pat = re.compile(sys.argv[2])
names = sys.argv[3:] or sorted(os.path.basename(f)[7:-5] for f in glob.glob(os.path.join(d, "detail_*.json")))
tabs = {n: json.load(open(os.path.join(d, "detail_%s.json" % n)))["per_tag"] for n in names}
rows = [k for k in tabs[names[0]] if pat.search(k)]
print("%-58s" % "layer (us per launch)" + "".join("%9s" % n[:8] for n in names))
tot = {n: 0.0 for n in names}
for k in rows:
    line = "%-58s" % k[:58]
    for n in names:
        v = tabs[n].get(k)
        us = v["total_ms"] / v["launches"] * 1e3 if v else float("nan")
        tot[n] += us if v else 0
        line += "%9.1f" % us
    print(line)
print("%-58s" % "sum" + "".join("%9.1f" % tot[n] for n in names))
