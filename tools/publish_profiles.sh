#!/bin/bash
# Copy the summaries of tools/profile_round.sh's call (gpurun_out/<tag> + gpurun_out/pmc_<tag>, scratch) into profiles/ (tracked).
set -e
R=${1:-r03}
cd "$(dirname "$0")/.."
cp gpurun_out/$R/bench.json profiles/${R}_bench.json
cp gpurun_out/$R/bench_detail.json profiles/${R}_bench_detail.json
cp gpurun_out/$R/bench_under_rocprof.json profiles/${R}_bench_under_rocprof.json
cp gpurun_out/$R/prof/stats_kernel_stats.csv profiles/${R}_rocprofv3_kernel_stats.csv
cp gpurun_out/pmc_$R/traffic.json profiles/${R}_traffic.json
cp gpurun_out/pmc_$R/traffic.json profiles/traffic_latest.json
if [ -f gpurun_out/pmc_${R}_biggan512/traffic.json ]; then cp gpurun_out/pmc_${R}_biggan512/traffic.json profiles/${R}_traffic_biggan512.json; cp gpurun_out/pmc_${R}_biggan512/traffic.json profiles/traffic_latest_biggan512.json; fi
if [ -f gpurun_out/pmc_${R}_gpt2/traffic_gpt2.json ]; then cp gpurun_out/pmc_${R}_gpt2/traffic_gpt2.json profiles/${R}_traffic_gpt2.json; cp gpurun_out/pmc_${R}_gpt2/traffic_gpt2.json profiles/traffic_latest_gpt2.json; fi
cp gpurun_out/pmc_sq_$R/sq_counters.txt profiles/${R}_sq_counters.txt
cp gpurun_out/$R/pytest_gpu.log profiles/${R}_pytest_gpu.log
cp gpurun_out/$R/bench_biggan512.json profiles/${R}_bench_biggan512.json
cp gpurun_out/$R/bench_gpt2.json profiles/${R}_bench_gpt2.json
[ -f gpurun_out/$R/gpt2_step_timeline.txt ] && cp gpurun_out/$R/gpt2_step_timeline.txt profiles/${R}_gpt2_step_timeline.txt
{ echo "# tools/mfma_peak (v_mfma_f32_32x32x16_f16; >= 50 ms timed after a 100 ms warm-up; zero vs non-zero operands)"; cat gpurun_out/$R/mfma_peak.txt;
  echo; echo "# tools/hbm_peak (16-byte accesses, 1 GiB arrays, best of 5)"; cat gpurun_out/$R/hbm_peak.txt;
  echo; echo "# tools/inflight_probe (HBM read rate vs waves per CU x 16-byte loads in flight per thread)"; cat gpurun_out/$R/inflight_probe.txt;
  echo; echo "# tools/launch_boundary (dependent kernel boundary, same stream: eager and hipGraph)"; cat gpurun_out/$R/launch_boundary.txt 2>/dev/null;
  echo; echo "# tools/grid_barrier (one counter vs XCD-hierarchical, relaxed sc1 polling)"; cat gpurun_out/$R/grid_barrier.txt 2>/dev/null; } > profiles/${R}_device_peaks.txt
ls -la profiles | grep $R
