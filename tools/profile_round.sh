#!/bin/bash
# One round's measurement set on ONE GPU box (run through gpurun):  bash tools/profile_round.sh r03
#   1. python bench.py                                  -> gpurun_out/<tag>/bench.json (+ per-layer table of an instrumented pass)
#   2. rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline   -> kernel stats of the SAME command + its JSON line
#   3. tools/measure_traffic.sh (PMC FETCH_SIZE / WRITE_SIZE, two passes), tools/measure_sq.sh (SQ counters, two passes)
#   4. device probes (tools/_bin/*), the biggan512 / gpt2 legs, the timeline of one GPT-2 decode step, the GPU test suite
# tools/publish_profiles.sh <tag> then copies the summaries into profiles/ (tracked).
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG
mkdir -p $OUT
GLASS_BENCH_DETAIL=$OUT/bench_detail.json python bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -c 600 $OUT/bench.json; echo
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o stats -- python bench.py --no-cpu-baseline --no-legs > $OUT/bench_under_rocprof.json 2> $OUT/rocprof.err
ls $OUT/prof | head
bash tools/measure_traffic.sh $TAG > $OUT/traffic.log 2>&1
bash tools/measure_sq.sh $TAG > $OUT/sq.log 2>&1
bash tools/measure_traffic.sh $TAG biggan512 > $OUT/traffic_biggan512.log 2>&1      # the legs' `roofline.traffic` (was null)
bash tools/measure_traffic.sh $TAG gpt2 > $OUT/traffic_gpt2.log 2>&1
for b in mfma_peak hbm_peak inflight_probe launch_boundary grid_barrier; do   # (tools/_bin does not travel: built on the box)
  hipcc --offload-arch=gfx950 -O3 tools/$b.hip -o /tmp/$b 2>/dev/null && timeout 180 /tmp/$b > $OUT/$b.txt 2>&1
done
python bench.py --config biggan512 --steps 20 --warmup 3 > $OUT/bench_biggan512.json 2>/dev/null
python bench.py --config gpt2 --steps 5 --warmup 1 > $OUT/bench_gpt2.json 2>/dev/null
# timeline of one GPT-2 decode step (rocpd database -> text; the database itself stays on the box)
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_gpt2 -o gpt2 -- python bench.py --config gpt2 --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python tools/gpt2_timeline.py /tmp/prof_gpt2/gpt2_results.db > $OUT/gpt2_step_timeline.txt 2>&1
(timeout 600 python -m pytest tests -q -m gpu 2>&1 | tail -5) > $OUT/pytest_gpu.log
cat $OUT/pytest_gpu.log
