# other workloads of DESIGN section 4 (one JSON value per line): sizes, LiDAR-like scene, fp32, batched scenes, 1M points
run() { name=$1; shift; timeout -k 10 280 python bench.py --cpu-sample 0 --no-kernel-events "$@" 2> gpurun_out/sweep_$name.err | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('$name', d['value'], d['ms_per_step'])" || echo "$name FAILED"; }
run p80k --points 80000
run lidar120k --points 120000 --kind lidar
run fp32 --dtype fp32
run scenes2 --scenes 2
run scenes4 --scenes 4
run p1M --points 1000000 --steps 5 --warmup 2
run noov --no-overlap
