#!/bin/bash
# Round measurement on the GPU box: default bench line, rocprofv3 kernel stats of the same command, PMC traffic + SQ
# passes (counters in their own runs, --kernel-trace only), the configs[2] LiDAR workload, train step, kNN.
# usage (from repo root, under gpurun): bash tools/measure_round.sh r02
set -e
R=${1:-r02}; O=gpurun_out/$R; mkdir -p $O
export TMPDIR=/tmp
# NOTE: only gpurun_out/ travels back from the GPU box; copy the summaries into profiles/$R afterwards with
#       tools/collect_round.sh $R (runs in the build container).
python bench.py > $O/bench_default.log 2>&1; grep '^{"metric"' $O/bench_default.log | tail -1 > $O/bench_n1_default.json
echo "bench done: $(cut -c1-160 $O/bench_n1_default.json)"
rm -rf $O/stats $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py --cpu-sample 0 --no-extra > $O/bench_stats.log 2>&1
grep '^{"metric"' $O/bench_stats.log | tail -1 > $O/bench_n1_under_rocprof.json
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -- python bench.py --steps 5 --warmup 2 --cpu-sample 0 --no-extra --no-kernel-events > $O/pmc_$c.log 2>&1
done
rm -f $O/pmc_traffic.json
python tools/pmc_traffic.py "$(ls -t $O/pmc_FETCH_SIZE/*/*_counter_collection.csv | head -1)" "$(ls -t $O/pmc_WRITE_SIZE/*/*_counter_collection.csv | head -1)" 5 \
  $O/pmc_traffic.json '{"points": 100000, "scenes": 1, "dtype": "bf16", "kind": "surface"}'
# the trace holds warm-up + timed + event-bracketed + latency forwards; summarise the last 15 (one forward in flight)
python tools/summarize_trace.py "$(ls -t $O/stats/*/*_kernel_trace.csv | head -1)" 15 > $O/bench_n1_trace_summary.txt
head -14 $O/bench_n1_trace_summary.txt
# issue / stall / matrix-core shares per kernel (SQ counters; separate pass, no other trace domains)
rm -rf $O/pmc_sq
rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU \
  --kernel-trace --output-format csv -d $O/pmc_sq -- python bench.py --steps 5 --warmup 2 --cpu-sample 0 --no-extra --no-kernel-events > $O/pmc_sq.log 2>&1
python tools/pmc_sq.py "$(ls -t $O/pmc_sq/*/*_counter_collection.csv | head -1)" 5 $O/pmc_sq.json
# BASELINE configs[2]: PTv3 semseg on a 120k-point LiDAR-like scan (bench line + kernel trace)
bash tools/profile_lidar.sh $R > $O/lidar_profile.log 2>&1 || true
head -8 $O/lidar_trace_summary.txt
# the training step and kNN (A18)
python bench.py --mode train --steps 10 --warmup 3 2>/dev/null | grep '^{"metric"' > $O/bench_n1_train.json
cut -c1-220 $O/bench_n1_train.json
python tools/bench_pointops.py > $O/knn_query.json 2>/dev/null; python tools/bench_pointops.py 100000 100000 16 >> $O/knn_query.json 2>/dev/null
cat $O/knn_query.json | cut -c1-160
