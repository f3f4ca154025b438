#!/bin/bash
# experiment: fused decoder ReLU backward (stats_kind 2) + new final_bwd -- kernel tests, bench A/B
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
export MSC_TUNE_CACHE="$PWD/gpurun_out/tune_cache.json"
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_unet.py -m gpu -q -rf --tb=short -p no:cacheprovider > gpurun_out/pytest_k.log 2>&1; echo "pytest rc=$?"; tail -25 gpurun_out/pytest_k.log
timeout 600 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --dump-launches gpurun_out/launches_train.json > gpurun_out/bench_train_fused.log 2>&1; echo "bench fused rc=$?"
grep '^{' gpurun_out/bench_train_fused.log | tail -1 > gpurun_out/bench_train_fused.json; grep -o '"ms_per_step": [0-9.]*' gpurun_out/bench_train_fused.json; grep -o '"family_ms_per_step": {[^}]*}' gpurun_out/bench_train_fused.json
MSC_FUSE_RELU_BWD=0 timeout 600 python bench.py --steps 100 --warmup 5 --no-cpu-baseline > gpurun_out/bench_train_unfused.log 2>&1; echo "bench unfused rc=$?"
grep '^{' gpurun_out/bench_train_unfused.log | tail -1 > gpurun_out/bench_train_unfused.json; grep -o '"ms_per_step": [0-9.]*' gpurun_out/bench_train_unfused.json; grep -o '"family_ms_per_step": {[^}]*}' gpurun_out/bench_train_unfused.json
tail -3 gpurun_out/bench_train_fused.log | cut -c1-300
