#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 600 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-north-star --dump-launches gpurun_out/launches_train_final.json > gpurun_out/launch_train.log 2>&1; grep -o '"ms_per_step": [0-9.]*' gpurun_out/launch_train.log | head -1
timeout 600 python bench.py --workload infer --encoder 101 --steps 30 --warmup 3 --no-cpu-baseline --dump-launches gpurun_out/launches_infer_r101_final.json > gpurun_out/launch_infer.log 2>&1; grep -o '"ms_per_step": [0-9.]*' gpurun_out/launch_infer.log | head -1
