#!/bin/bash
# Round measurement on the GPU box: default bench line, rocprofv3 kernel stats of the same command, PMC traffic + SQ
# passes (counters in their own runs, --kernel-trace only) for the headline AND for BASELINE configs[2] (120k LiDAR),
# the train step with its kernel summary, kNN.
# usage (from repo root, under gpurun): bash tools/measure_round.sh r03
set -e
R=${1:-r03}; O=gpurun_out/$R; mkdir -p $O
PART=${2:-ABC}     # a gpurun call is capped at 20 minutes: run the parts (A headline, B LiDAR, C train / kNN / Swin3D) in separate calls
export TMPDIR=/tmp
# NOTE: only gpurun_out/ travels back from the GPU box; copy the summaries into profiles/$R afterwards with
#       tools/collect_round.sh $R (runs in the build container).
if [[ $PART == *A* ]]; then
python bench.py > $O/bench_default.log 2>&1; grep '^{"metric"' $O/bench_default.log | tail -1 > $O/bench_n1_default.json
echo "bench done: $(cut -c1-160 $O/bench_n1_default.json)"
fi
pmc_pair() {  # $1 = tag, rest = bench args: FETCH / WRITE passes + SQ pass of one workload
  local tag=$1; shift
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $O/pmc_${tag}_$c
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_${tag}_$c -- python bench.py "$@" --steps 5 --warmup 2 --cpu-sample 0 --no-extra --no-kernel-events > $O/pmc_${tag}_$c.log 2>&1
  done
  rm -rf $O/pmc_${tag}_sq
  rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU \
    --kernel-trace --output-format csv -d $O/pmc_${tag}_sq -- python bench.py "$@" --steps 5 --warmup 2 --cpu-sample 0 --no-extra --no-kernel-events > $O/pmc_${tag}_sq.log 2>&1
}
# ---- headline (100k surface scene)
if [[ $PART == *A* ]]; then
rm -rf $O/stats
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py --cpu-sample 0 --no-extra > $O/bench_stats.log 2>&1
grep '^{"metric"' $O/bench_stats.log | tail -1 > $O/bench_n1_under_rocprof.json
python tools/summarize_trace.py "$(ls -t $O/stats/*/*_kernel_trace.csv | head -1)" 15 > $O/bench_n1_trace_summary.txt
head -14 $O/bench_n1_trace_summary.txt
pmc_pair default
python tools/pmc_traffic.py "$(ls -t $O/pmc_default_FETCH_SIZE/*/*_counter_collection.csv | head -1)" "$(ls -t $O/pmc_default_WRITE_SIZE/*/*_counter_collection.csv | head -1)" 5 \
  $O/pmc_traffic.json '{"points": 100000, "scenes": 1, "dtype": "bf16", "kind": "surface"}'
python tools/pmc_sq.py "$(ls -t $O/pmc_default_sq/*/*_counter_collection.csv | head -1)" 5 $O/pmc_sq.json
fi
# ---- BASELINE configs[2]: PTv3 semseg on a 120k-point LiDAR-like scan
if [[ $PART == *B* ]]; then
A="--model semseg --kind lidar --points 120000"
python bench.py $A --cpu-sample 0 --no-extra > $O/lidar_bench.json 2> $O/lidar_bench.err
rm -rf $O/lidar_stats
rocprofv3 --kernel-trace --stats --output-format csv -d $O/lidar_stats -- python bench.py $A --cpu-sample 0 --no-extra --steps 5 --warmup 2 --no-kernel-events > $O/lidar_stats.log 2>&1
python tools/summarize_trace.py "$(ls -t $O/lidar_stats/*/*_kernel_trace.csv | head -1)" 5 > $O/lidar_trace_summary.txt
head -10 $O/lidar_trace_summary.txt
pmc_pair lidar $A
python tools/pmc_traffic.py "$(ls -t $O/pmc_lidar_FETCH_SIZE/*/*_counter_collection.csv | head -1)" "$(ls -t $O/pmc_lidar_WRITE_SIZE/*/*_counter_collection.csv | head -1)" 5 \
  $O/pmc_traffic_lidar.json '{"points": 120000, "scenes": 1, "dtype": "bf16", "kind": "lidar", "model": "semseg"}'
python tools/pmc_sq.py "$(ls -t $O/pmc_lidar_sq/*/*_counter_collection.csv | head -1)" 5 $O/pmc_sq_lidar.json
fi
if [[ $PART == *C* ]]; then
# ---- the training step: bench line with its roofline + kernel stats of the same command
python bench.py --mode train --steps 10 --warmup 3 2>/dev/null | grep '^{"metric"' > $O/bench_n1_train.json
cut -c1-220 $O/bench_n1_train.json
rm -rf $O/train_stats
rocprofv3 --kernel-trace --stats --output-format csv -d $O/train_stats -- python bench.py --mode train --steps 5 --warmup 2 --no-kernel-events > $O/train_stats.log 2>&1
python tools/train_kernel_stats.py "$(ls -t $O/train_stats/*/*_kernel_stats.csv | head -1)" 7 > $O/train_kernel_summary.txt 2>/dev/null || true
head -12 $O/train_kernel_summary.txt || true
python tools/bench_pointops.py > $O/knn_query.json 2>/dev/null; python tools/bench_pointops.py 100000 100000 16 >> $O/knn_query.json 2>/dev/null
# ---- Swin3D (BASELINE configs[4]): attention micro-benchmark (both kernels), kernel stats of the 1M-point forward
python tools/bench_swin.py 300000 2>/dev/null > $O/swin_attn_bench.log
PTV3_SWIN_ATTN_MFMA=0 python tools/bench_swin.py 300000 2>/dev/null > $O/swin_attn_bench_gather.log
rm -rf $O/swin_stats
rocprofv3 --kernel-trace --stats --output-format csv -d $O/swin_stats -- python tools/bench_swin.py 1000000 model > $O/swin1m.log 2>&1
cp "$(ls -t $O/swin_stats/*/*_kernel_stats.csv | head -1)" $O/swin1m_kernel_stats.csv; rm -rf $O/swin_stats
tail -1 $O/swin1m.log
fi
