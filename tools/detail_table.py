"""Per-layer table from a bench.py detail dump (GLASS_BENCH_DETAIL=<file>): time share, algorithmic TFLOP/s, GB/s."""
import json
import sys

d = json.load(open(sys.argv[1]))
key = sys.argv[2] if len(sys.argv) > 2 else "per_tag"
rows = [dict(name=k, **v) for k, v in d[key].items()]
tot = sum(r["total_ms"] for r in rows)
rows.sort(key=lambda r: -r["total_ms"])
print(f"total {tot:.2f} ms over the profiled passes")
for r in rows[: int(sys.argv[3]) if len(sys.argv) > 3 else 50]:
    ms = r["total_ms"] / r["launches"]
    print(f"{r['name'][:62]:62s} n={r['launches']:4d} avg={ms * 1e3:8.1f}us {100 * r['total_ms'] / tot:5.1f}% "
          f"{r['flops'] / r['launches'] / ms * 1e-9:8.1f} TF/s {r['bytes'] / r['launches'] / ms * 1e-6:8.1f} GB/s")
