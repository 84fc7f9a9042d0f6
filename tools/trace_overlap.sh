#!/bin/bash
# kernel trace of a few bench steps in the default two-stream mode:  bash tools/trace_overlap.sh <tag> [GLASS_LIB]
TAG=${1:-cur}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/ktrace_$TAG
mkdir -p $OUT
[ -n "$2" ] && export GLASS_LIB=$2
export GLASS_BENCH_NOPROF=1
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $OUT -o kt -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-legs > $OUT/bench.json 2> $OUT/kt.err
python tools/trace_overlap.py $OUT > $OUT/timeline.txt
tail -70 $OUT/timeline.txt
