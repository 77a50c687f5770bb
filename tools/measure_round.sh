#!/bin/bash
# Round measurement on the GPU box: default bench line, rocprofv3 kernel stats of the same command, PMC traffic passes.
# usage (from repo root, under gpurun): bash tools/measure_round.sh r01
set -e
R=${1:-r01}; O=gpurun_out/$R; mkdir -p $O profiles/$R
export TMPDIR=/tmp
# NOTE: only gpurun_out/ travels back from the GPU box; copy the summaries into profiles/$R afterwards with
#       tools/collect_round.sh $R (runs in the build container).
python bench.py > $O/bench_default.log 2>&1; grep '^{"metric"' $O/bench_default.log | tail -1 > $O/bench_n1_default.json
echo "bench done: $(cut -c1-160 $O/bench_n1_default.json)"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py --cpu-sample 0 > $O/bench_stats.log 2>&1
grep '^{"metric"' $O/bench_stats.log | tail -1 > $O/bench_n1_under_rocprof.json
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -- python bench.py --steps 5 --warmup 2 --cpu-sample 0 --no-kernel-events > $O/pmc_$c.log 2>&1
done
python tools/pmc_traffic.py $O/pmc_FETCH_SIZE/*/*_counter_collection.csv $O/pmc_WRITE_SIZE/*/*_counter_collection.csv 5 \
  $O/pmc_traffic.json '{"points": 100000, "scenes": 1, "dtype": "bf16", "kind": "surface"}'
python tools/summarize_trace.py $O/stats/*/*_kernel_trace.csv 15 > $O/bench_n1_trace_summary.txt
head -12 $O/bench_n1_trace_summary.txt
