#!/bin/bash
# bench under different ROCm runtime env settings (which one explains the rocprofv3-vs-plain gap?)
run() { echo "== $*"; env "$@" python bench.py --steps 20 --warmup 5 --cpu-sample 0 --no-kernel-events 2>&1 | grep -o '"value": [0-9.]*, "unit": "Mpoints/s", "n_gpus": 1, "steps": [0-9]*, "warmup": [0-9]*, "ms_per_step": [0-9.]*'; }
run A=1
run HSA_ENABLE_SDMA=0
run HIP_FORCE_DEV_KERNARG=1
run HSA_ENABLE_INTERRUPT=0
run GPU_MAX_HW_QUEUES=8
run HSA_ENABLE_SDMA=0 HIP_FORCE_DEV_KERNARG=1 HSA_ENABLE_INTERRUPT=0
