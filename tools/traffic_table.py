"""Per-kernel HBM-side traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of ONE bench run.

FETCH_SIZE / WRITE_SIZE are in KiB (rocprofv3 derived counters: TCC_EA0_RDREQ * 64 B etc. / 1024); Infinity-Cache hits are
counted, not excluded.  gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE reports HALF the bytes of a wide coalesced
streaming read -> doubled here; WRITE_SIZE is uncalibrated and reported as is.

The bench run behind it (tools/measure_traffic.sh) uses GLASS_BENCH_UNIFORM_POP=1: EVERY pass carries the full population, so
a kernel's dispatches are its P = 64 launches only (round 2 averaged 24 launches at P = 64 with 6 at P = 4 and understated
the up-conv's traffic by 23 %).  Rows: per kernel symbol (average over its launches = over the layers it runs) and per
(kernel, grid size) = per layer.

  python tools/traffic_table.py fetch_counter_collection.csv write_counter_collection.csv > traffic.json
"""
import collections
import csv
import json
import re
import sys


def short(name):
    name = re.sub(r"\s+", "", name)
    name = re.sub(r"^void", "", name)
    return re.sub(r"\(.*$", "", name)


def load(path, counter):
    rows = collections.OrderedDict()          # dispatch id -> (kernel, grid, value)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        d = int(r["Dispatch_Id"])
        k, g, v = rows.get(d, (short(r["Kernel_Name"]), int(r["Grid_Size"]), 0.0))
        rows[d] = (k, g, v + float(r["Counter_Value"]))
    return rows


def main():
    fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    per_kernel, per_layer = {}, {}

    def add(table, key, f_kib, w_kib):
        a = table.setdefault(key, dict(launches_fetch=0, launches_write=0, fetch_kib=0.0, write_kib=0.0))
        if f_kib is not None:
            a["launches_fetch"] += 1
            a["fetch_kib"] += f_kib
        if w_kib is not None:
            a["launches_write"] += 1
            a["write_kib"] += w_kib
    for d, (k, g, v) in fetch.items():
        add(per_kernel, k, v, None)
        add(per_layer, "%s grid=%d" % (k, g), v, None)
    for d, (k, g, v) in write.items():
        add(per_kernel, k, None, v)
        add(per_layer, "%s grid=%d" % (k, g), None, v)

    def finish(table):
        out = {}
        for key, a in table.items():
            f = 2.0 * 1024.0 * a["fetch_kib"] / max(a["launches_fetch"], 1)
            w = 1024.0 * a["write_kib"] / max(a["launches_write"], 1)
            out[key] = dict(launches=a["launches_fetch"], fetch_bytes_per_launch=f, write_bytes_per_launch=w, bytes_per_launch=f + w)
        return dict(sorted(out.items(), key=lambda kv: -kv[1]["bytes_per_launch"] * kv[1]["launches"]))
    json.dump(dict(note="bytes per launch, every launch at the full population (GLASS_BENCH_UNIFORM_POP=1); fetch = 2 x FETCH_SIZE (gfx950 "
                        "undercount of wide reads, MI355X_MICROARCH.md), write = WRITE_SIZE as reported (uncalibrated); Infinity-Cache hits "
                        "are counted as traffic",
                   per_kernel=finish(per_kernel), per_layer=finish(per_layer)), sys.stdout, indent=1)


if __name__ == "__main__":
    main()
