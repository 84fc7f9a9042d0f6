"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE).

FETCH_SIZE / WRITE_SIZE are in KiB (rocprofv3 derived counters: TCC_EA0_RDREQ*64B etc. / 1024).
gfx950 correction (MI355X_MICROARCH.md §HBM): FETCH_SIZE reports HALF the bytes of a wide
coalesced streaming read -> doubled here (`fetch_corrected`); WRITE_SIZE is uncalibrated and
reported as is.  Output: JSON {kernel: {launches, fetch_kib, write_kib, bytes_per_launch}}.
"""
import collections
import csv
import json
import sys


def agg(path, counter):
    tot = collections.defaultdict(float)
    n = collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"]
        tot[k] += float(r["Counter_Value"])
        n[k].add(r["Dispatch_Id"])
    return tot, {k: len(v) for k, v in n.items()}


fetch, nf = agg(sys.argv[1], "FETCH_SIZE")
write, nw = agg(sys.argv[2], "WRITE_SIZE")
out = {}
for k in fetch:
    n = nf[k]
    f, w = fetch[k] / n, write.get(k, 0.0) / max(nw.get(k, 1), 1)
    out[k] = dict(launches=n, fetch_kib_per_launch=f, write_kib_per_launch=w,
                  bytes_per_launch=(2.0 * f + w) * 1024.0, note="fetch doubled (gfx950 FETCH_SIZE undercount)")
json.dump(dict(sorted(out.items(), key=lambda kv: -kv[1]["bytes_per_launch"] * kv[1]["launches"])), sys.stdout, indent=1)
