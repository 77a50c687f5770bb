#!/bin/bash
# Build container: copy the judged summaries of a measurement round from gpurun_out/<round> into profiles/<round>.
set -e
R=${1:-r03}; O=gpurun_out/$R; mkdir -p profiles/$R
for f in bench_n1_default.json bench_n1_under_rocprof.json pmc_traffic.json pmc_sq.json bench_n1_trace_summary.txt \
         bench_n1_train.json train_kernel_summary.txt knn_query.json lidar_bench.json lidar_trace_summary.txt \
         pmc_traffic_lidar.json pmc_sq_lidar.json pmc_wide.json wide_ablate.log conv_ablate.log conv_tile.log conv_buf.log \
         lds_dma_oob.txt wide4.log swin_attn_bench.log swin_attn_bench_gather.log swin1m_kernel_stats.csv swin1m.log; do
  [ -f $O/$f ] && cp $O/$f profiles/$R/
done
cp "$(ls -t $O/stats/*/*_kernel_stats.csv | head -1)" profiles/$R/bench_n1_default_kernel_stats.csv
[ -d $O/lidar_stats ] && cp "$(ls -t $O/lidar_stats/*/*_kernel_stats.csv | head -1)" profiles/$R/lidar_kernel_stats.csv
[ -d $O/train_stats ] && cp "$(ls -t $O/train_stats/*/*_kernel_stats.csv | head -1)" profiles/$R/train_kernel_stats.csv
ls -la profiles/$R
