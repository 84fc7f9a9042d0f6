"""Timeline of the LAST bench pass from a rocprofv3 --kernel-trace csv: per kernel start / end (us from the pass's first kernel), queue, and
which kernels of the other queue overlap it."""
import csv, glob, os, re, sys
d = sys.argv[1]
kt = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = []
for r in csv.DictReader(open(kt)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")[:60], r.get("Queue_Id", "?")))
rows.sort()
# last pass: starts at the last mapping_fused kernel
starts = [i for i, r in enumerate(rows) if r[2].startswith("mapping_fused")]
i0 = starts[-1]
t0 = rows[i0][0]
last = rows[i0:]
qs = sorted(set(r[3] for r in last))
print("# queues:", qs, " pass length %.1f us" % ((max(r[1] for r in last) - t0) / 1e3))
busy = {q: 0 for q in qs}
for s, e, n, q in last:
    busy[q] += e - s
print("# sum of kernel durations per queue (us):", {q: round(v / 1e3, 1) for q, v in busy.items()})
main_q = max(busy, key=busy.get)
for s, e, n, q in last:
    if q == main_q and (e - s) > 100000:
        ov = [(max(s, s2), min(e, e2), n2) for s2, e2, n2, q2 in last if q2 != main_q and s2 < e and e2 > s]
        ovt = sum(b - a for a, b, _ in ov)
        print("%9.1f %9.1f %8.1f us  %-58s other-queue overlap %7.1f us in %d kernels" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, n, ovt / 1e3, len(ov)))
oth = [(s, e, n) for s, e, n, q in last if q != main_q]
if oth:
    print("# other queues: first start %.1f, last end %.1f, n=%d" % ((min(s for s, _, _ in oth) - t0) / 1e3, (max(e for _, e, _ in oth) - t0) / 1e3, len(oth)))
