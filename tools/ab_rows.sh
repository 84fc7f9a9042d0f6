#!/bin/bash
# tools/ab_rows.sh <out_dir> <row-regex> name[:lib] ...  — bench.py per-layer rows of several library builds on ONE box (GLASS_LIB), rows matching the regex
out=$1; pat=$2; shift 2
mkdir -p "$out"
root=$(cd "$(dirname "$0")/.." && pwd)
for spec in "$@"; do
  name=${spec%%:*}; lib=${spec#*:}
  [ "$lib" = "$spec" ] && lib=""
  envs=""
  [ -n "$lib" ] && envs="GLASS_LIB=$root/$lib"
  env $envs GLASS_BENCH_DETAIL="$out/detail_$name.json" timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-legs > "$out/bench_$name.json" 2> "$out/bench_$name.err"
  echo "== $name $(python -c "import json,sys; d=json.loads(open('$out/bench_$name.json').read().strip().splitlines()[-1]); print('%.1f cand/s %.3f ms' % (d['value'], d['ms_per_step']))" 2>/dev/null)"
  python tools/detail_table.py "$out/detail_$name.json" per_tag 80 | grep -E "$pat" | awk '{printf "   %-64s %s %s\n", $1, $3, $4}'
done
