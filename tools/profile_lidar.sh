#!/bin/bash
# BASELINE configs[2] (PTv3 semseg, 120k-point LiDAR-like scan): bench line with the per-family HIP-event times and a
# rocprofv3 kernel trace of the same command.  usage (under gpurun): bash tools/profile_lidar.sh r02
R=${1:-r02}; O=gpurun_out/$R; mkdir -p $O
export TMPDIR=/tmp
A="--model semseg --kind lidar --points 120000 --cpu-sample 0 --no-extra"
python bench.py $A --no-overlap > $O/lidar_bench.json 2> $O/lidar_bench.err
python bench.py $A --no-kernel-events > $O/lidar_bench_overlap.json 2>> $O/lidar_bench.err
rm -rf $O/lidar_stats
rocprofv3 --kernel-trace --stats --output-format csv -d $O/lidar_stats -- python bench.py $A --no-overlap --steps 5 --warmup 2 --no-kernel-events > $O/lidar_stats.log 2>&1
python tools/summarize_trace.py "$(ls -t $O/lidar_stats/*/*_kernel_trace.csv | head -1)" 5 > $O/lidar_trace_summary.txt
head -12 $O/lidar_trace_summary.txt
