#!/bin/bash
# Effective shader clock of every kernel of one bench step: GRBM_GUI_ACTIVE (cycles) / dispatch duration (one PMC pass + kernel trace).
# Run on the GPU box:  bash tools/measure_clock.sh <tag> [GLASS_LIB path]
TAG=${1:-r06}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_clk_$TAG
mkdir -p $OUT
[ -n "$2" ] && export GLASS_LIB=$2
# (GLASS_NO_CLIP_OVERLAP is read by bench.py, which calls glass_engine_set_overlap(0): the release library itself reads no environment)
export GLASS_NO_CLIP_OVERLAP=1 GLASS_BENCH_NOPROF=1 GLASS_BENCH_UNIFORM_POP=1
timeout 400 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT -o clk -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-legs > /dev/null 2> $OUT/clk.err
ls $OUT
python tools/clock_table.py $OUT > $OUT/clock_table.txt
head -40 $OUT/clock_table.txt
