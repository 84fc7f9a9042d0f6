#!/usr/bin/env python3
"""Timeline of ONE single-token GPT-2 decode step from a rocprofv3 --kernel-trace database (rocpd sqlite):
    rocprofv3 --kernel-trace --stats -d <dir> -o gpt2 -- python bench.py --config gpt2 --steps 2 --warmup 1 --no-cpu-baseline
    python tools/gpt2_timeline.py <dir>/gpt2_results.db
Prints every launch of the last complete step (start relative to the step, start-to-end duration, grid) and the per-kernel sums: inside the
captured graph a launch starts when its predecessor ends, so a duration includes the dispatch cost (the 64-workgroup helpers' ~5 us are
almost all dispatch: gpt2_advance_kernel, ONE thread, measured 4.1-4.2 us in the unfused forms)."""
import collections
import sqlite3
import sys


def main(path):
    con = sqlite3.connect(path)
    rows = list(con.execute("select name, start, end, grid_x, workgroup_x from kernels order by start"))
    # a step ends with the fused pick / embed / advance kernel (round 4) or, in the unfused forms, with gpt2_advance_kernel
    adv = [i for i, r in enumerate(rows) if "gpt2_pick_embed" in r[0]]
    if len(adv) < 3:
        adv = [i for i, r in enumerate(rows) if "gpt2_advance" in r[0]]
    if len(adv) < 3:
        sys.exit("no complete decode step in the trace")
    i0, i1 = adv[-2], adv[-1]
    step = rows[i0 + 1:i1 + 1]
    t0 = step[0][1]
    print("# one decode step: %d launches, %.1f us from the previous step's last launch to this one's" % (len(step), (step[-1][2] - rows[i0][2]) / 1e3))
    for r in step:
        print("%-72s start %8.2f us  dur %6.2f us  workgroups %5d x %4d threads" % (r[0][:72], (r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, r[3] // r[4], r[4]))
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in step:
        agg[r[0][:72]][0] += 1
        agg[r[0][:72]][1] += (r[2] - r[1]) / 1e3
    print("# per kernel")
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("%-72s n=%3d  total %7.1f us  avg %6.2f us" % (k, v[0], v[1], v[1] / v[0]))


if __name__ == "__main__":
    main(sys.argv[1])
