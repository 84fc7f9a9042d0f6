#!/bin/bash
# HBM-side traffic of every kernel of one bench pass from the PMC counters (separate passes, as MI355X_MICROARCH.md prescribes:
# FETCH_SIZE and WRITE_SIZE do not fit one pass; no tracing flags besides the counters).  On the GPU box:
#   bash tools/measure_traffic.sh <tag>     -> gpurun_out/pmc_<tag>/traffic.json
set -e
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
# single stream, no event instrumentation, and EVERY pass at the full population (the set-up pass included): a kernel's
# dispatches are then its P = 64 launches only
export GLASS_NO_CLIP_OVERLAP=1 GLASS_BENCH_NOPROF=1 GLASS_BENCH_UNIFORM_POP=1
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT -o fetch -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-legs > /dev/null 2> $OUT/fetch.err
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT -o write -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-legs > /dev/null 2> $OUT/write.err
python tools/traffic_table.py $OUT/fetch_counter_collection.csv $OUT/write_counter_collection.csv > $OUT/traffic.json
head -c 1500 $OUT/traffic.json
