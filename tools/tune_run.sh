cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
export MSC_TUNE_DB=0 MSC_TUNE_CACHE=$PWD/gpurun_out/tune_new.json
rm -f gpurun_out/tune_new.json
timeout 600 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --dump-launches gpurun_out/launches_train_new.json > gpurun_out/tune_train.log 2>&1; echo "train rc=$?"; grep -o '"ms_per_step": [0-9.]*' gpurun_out/tune_train.log; grep -o '"family_ms_per_step": {[^}]*}' gpurun_out/tune_train.log
timeout 600 python bench.py --workload infer --encoder 101 --steps 30 --warmup 3 --no-cpu-baseline --dump-launches gpurun_out/launches_infer_r101_new.json > gpurun_out/tune_infer101.log 2>&1; echo "infer101 rc=$?"; grep -o '"ms_per_step": [0-9.]*' gpurun_out/tune_infer101.log
timeout 600 python bench.py --workload infer --steps 30 --warmup 3 --no-cpu-baseline > gpurun_out/tune_infer34.log 2>&1; echo "infer34 rc=$?"; grep -o '"ms_per_step": [0-9.]*' gpurun_out/tune_infer34.log
timeout 600 python bench.py --workload infer --encoder 101 --size 320 --steps 30 --warmup 3 --no-cpu-baseline > gpurun_out/tune_infer320.log 2>&1; echo "infer320 rc=$?"; grep -o '"ms_per_step": [0-9.]*' gpurun_out/tune_infer320.log
timeout 600 python bench.py --size 320 --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/tune_train320.log 2>&1; echo "train320 rc=$?"; grep -o '"ms_per_step": [0-9.]*' gpurun_out/tune_train320.log
timeout 600 python bench.py --workload tta --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/tune_tta.log 2>&1; echo "tta rc=$?"; grep -o '"ms_per_step": [0-9.]*' gpurun_out/tune_tta.log
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x --tb=short -p no:cacheprovider 2>&1 | tail -3
