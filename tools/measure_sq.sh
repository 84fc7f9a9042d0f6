#!/bin/bash
# SQ counters of every kernel of one bench step (two PMC passes, no tracing flags).  Run on the GPU box:  bash tools/measure_sq.sh <tag>
set -e
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_sq_$TAG
mkdir -p $OUT
# (GLASS_NO_CLIP_OVERLAP is read by bench.py, which calls glass_engine_set_overlap(0): the release library itself reads no environment)
export GLASS_NO_CLIP_OVERLAP=1 GLASS_BENCH_NOPROF=1 GLASS_BENCH_UNIFORM_POP=1
timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAVES --output-format csv -d $OUT -o sq1 -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-legs > /dev/null 2> $OUT/sq1.err
timeout 400 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT -o sq2 -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-legs > /dev/null 2> $OUT/sq2.err
python tools/sq_table.py $OUT/sq1_counter_collection.csv $OUT/sq2_counter_collection.csv > $OUT/sq_counters.txt
head -20 $OUT/sq_counters.txt
