#!/bin/bash
# HBM-side traffic of every kernel of one bench pass from the PMC counters (separate passes, as MI355X_MICROARCH.md prescribes:
# FETCH_SIZE and WRITE_SIZE do not fit one pass; no tracing flags besides the counters).  On the GPU box:
#   bash tools/measure_traffic.sh <tag> [ffhq|biggan512|gpt2]     -> gpurun_out/pmc_<tag>[_<config>]/traffic.json
# ffhq (default) is the headline; the other two are the legs of the default bench line (their `roofline.traffic`).
set -e
TAG=${1:-r05}
CFG=${2:-ffhq}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_$TAG
[ "$CFG" != "ffhq" ] && OUT=${OUT}_$CFG
mkdir -p $OUT
# single stream, no event instrumentation, and EVERY pass at the full population (the set-up pass included): a kernel's
# dispatches are then its P = 64 launches only
# (GLASS_NO_CLIP_OVERLAP is read by bench.py, which calls glass_engine_set_overlap(0): the release library itself reads no environment)
export GLASS_NO_CLIP_OVERLAP=1 GLASS_BENCH_NOPROF=1 GLASS_BENCH_UNIFORM_POP=1
ARGS="--config $CFG --steps 1 --warmup 1 --no-cpu-baseline --no-legs"
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT -o fetch -- python bench.py $ARGS > /dev/null 2> $OUT/fetch.err
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT -o write -- python bench.py $ARGS > /dev/null 2> $OUT/write.err
python tools/traffic_table.py $OUT/fetch_counter_collection.csv $OUT/write_counter_collection.csv > $OUT/traffic.json
if [ "$CFG" = "gpt2" ]; then
  # one `_evaluate` = prefill + 30 single-token steps + text tower; the run made 2 of them (1 warm-up + 1 timed)
  python - $OUT/traffic.json <<'PY' > $OUT/traffic_gpt2.json
import json, sys
t = json.load(open(sys.argv[1]))["per_kernel"]
tot = sum(r["bytes_per_launch"] * r["launches"] for k, r in t.items() if not k.startswith("__amd_rocclr"))
dec = sum(r["bytes_per_launch"] * r["launches"] for k, r in t.items() if k.startswith(("gemm_f32", "gpt2_", "argmax", "splitk")) or "gpt2" in k)
print(json.dumps(dict(bytes_per_evaluate=tot / 2, bytes_per_decode=dec / 2, evaluates=2,
                      source="rocprofv3 --pmc FETCH_SIZE x 2 + WRITE_SIZE over one `bench.py --config gpt2 --steps 1 --warmup 1` run "
                             "(tools/measure_traffic.sh <tag> gpt2): sum over the decode kernels' dispatches / 2 evaluates; stored, not "
                             "collected by this run", per_kernel=t), indent=1))
PY
  head -c 600 $OUT/traffic_gpt2.json
else
  head -c 1500 $OUT/traffic.json
fi
