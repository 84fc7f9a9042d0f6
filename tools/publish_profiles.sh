#!/bin/bash
# Copy the summaries of the last profiling call (gpurun_out/r02 + gpurun_out/pmc_r02, scratch) into profiles/ (tracked).
#   on the GPU box (one gpurun call):  see the command block at the top of DESIGN.md section 5
set -e
R=${1:-r02}
cd "$(dirname "$0")/.."
cp gpurun_out/$R/bench.json profiles/${R}_bench.json
cp gpurun_out/$R/bench_detail.json profiles/${R}_bench_detail.json
cp gpurun_out/$R/bench_under_rocprof.json profiles/${R}_bench_under_rocprof.json
cp gpurun_out/$R/prof/stats_kernel_stats.csv profiles/${R}_rocprofv3_kernel_stats.csv
cp gpurun_out/pmc_$R/traffic.json profiles/${R}_traffic.json
cp gpurun_out/pmc_$R/traffic.json profiles/traffic_latest.json
{ echo "# tools/mfma_peak (v_mfma_f32_32x32x16_f16; >= 50 ms timed after a 100 ms warm-up; zero vs non-zero operands)"; cat gpurun_out/$R/mfma_peak.txt;
  echo; echo "# tools/inflight_probe (HBM read rate vs waves per CU x 16-byte loads in flight per thread)"; cat gpurun_out/$R/inflight_probe.txt; } > profiles/${R}_device_peaks.txt
for f in down_trace stream_trace; do [ -f gpurun_out/$f.txt ] && cp gpurun_out/$f.txt profiles/${R}_phase_trace_${f%_trace}.txt; done
ls -la profiles | grep $R
