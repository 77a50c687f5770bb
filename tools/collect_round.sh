#!/bin/bash
# Build container: copy the judged summaries of a measurement round from gpurun_out/<round> into profiles/<round>.
set -e
R=${1:-r02}; O=gpurun_out/$R; mkdir -p profiles/$R
cp $O/bench_n1_default.json $O/bench_n1_under_rocprof.json $O/pmc_traffic.json $O/bench_n1_trace_summary.txt profiles/$R/
for f in pmc_sq.json bench_n1_train.json knn_query.json lidar_bench.json lidar_bench_overlap.json lidar_trace_summary.txt; do
  [ -f $O/$f ] && cp $O/$f profiles/$R/
done
cp "$(ls -t $O/stats/*/*_kernel_stats.csv | head -1)" profiles/$R/bench_n1_default_kernel_stats.csv
[ -d $O/lidar_stats ] && cp "$(ls -t $O/lidar_stats/*/*_kernel_stats.csv | head -1)" profiles/$R/lidar_kernel_stats.csv
ls -la profiles/$R
