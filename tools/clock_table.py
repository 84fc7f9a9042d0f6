"""Per-kernel effective shader clock from a rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace run (tools/measure_clock.sh)."""
import collections, csv, glob, os, re, sys
d = sys.argv[1]
cc = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
kt = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
dur = {}
for r in csv.DictReader(open(kt[0])):
    dur[r["Dispatch_Id"]] = (float(r["End_Timestamp"]) - float(r["Start_Timestamp"]), r["Kernel_Name"], r.get("Grid_Size", ""))
acc = collections.defaultdict(lambda: [0.0, 0.0, 0])
for r in csv.DictReader(open(cc[0])):
    if r["Counter_Name"] != "GRBM_GUI_ACTIVE" or r["Dispatch_Id"] not in dur:
        continue
    ns, name, grid = dur[r["Dispatch_Id"]]
    k = re.sub(r"\(.*", "", name).replace("void ", "")[:70]
    a = acc[k]
    a[0] += float(r["Counter_Value"]); a[1] += ns; a[2] += 1
XCDS = 8    # rocprofv3 reports GRBM_GUI_ACTIVE summed over the eight XCDs (each counts its own busy cycles)
print("# effective shader clock per kernel = GRBM_GUI_ACTIVE / 8 XCDs / duration (sum over the dispatches of one bench pass + warm-up; PMC run: kernels serialised; dispatches of a few microseconds over-count: the counter also runs between them)")
for k, (cyc, ns, n) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:40]:
    print("%-72s n=%4d  %9.1f us  %.3f GHz" % (k, n, ns / 1e3, cyc / XCDS / ns if ns else 0))
