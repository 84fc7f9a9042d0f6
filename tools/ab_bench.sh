#!/bin/bash
# A/B of environment knobs on ONE box: tools/ab_bench.sh <out_dir> "<name>:<ENV=1 ENV2=..>" ...   (name "base": no knobs)
# Each variant: python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-legs, JSON line + per-layer table under <out_dir>/.
# Knobs only exist in the developer build (`make -C clip_glass_amd/csrc AB=1` -> tools/lib/libglass_ab.so): every variant that names a
# knob runs on it (GLASS_LIB), "base" runs on the release library — so a knob run also A/Bs the two builds against each other.
out=$1; shift
mkdir -p "$out"
root=$(cd "$(dirname "$0")/.." && pwd)
ablib="$root/tools/lib/libglass_ab.so"
for spec in "$@"; do
  name=${spec%%:*}; envs=${spec#*:}
  [ "$envs" = "$spec" ] && envs=""
  echo "== $name [$envs]" | tee -a "$out/summary.txt"
  if [ -n "$envs" ] && [ -z "${envs##*GLASS_*}" ] && [ "${envs#*GLASS_LIB=}" = "$envs" ]; then
    [ -f "$ablib" ] || { echo "  $ablib missing: make -C clip_glass_amd/csrc AB=1" | tee -a "$out/summary.txt"; continue; }
    envs="GLASS_LIB=$ablib $envs"
  fi
  env $envs GLASS_BENCH_DETAIL="$out/detail_$name.json" timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-legs > "$out/bench_$name.json" 2> "$out/bench_$name.err"
  python - "$out/bench_$name.json" <<'PY' | tee -a "$out/summary.txt"
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d.get("roofline", {})
    print("  value %.1f %s  ms/step %.3f  dominant %s avg_ms %.3f" % (d["value"], d["unit"], d["ms_per_step"], r.get("kernel"), r.get("avg_ms", 0)))
except Exception as e:
    print("  FAILED", e)
PY
  [ -f "$out/detail_$name.json" ] && python tools/detail_table.py "$out/detail_$name.json" per_tag 30 > "$out/table_$name.txt" 2>&1
done
