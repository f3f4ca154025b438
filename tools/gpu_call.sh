cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
export MSC_TUNE_CACHE="$PWD/gpurun_out/tune_cache.json"
timeout 120 python probes/thr_debug.py > gpurun_out/thr_debug.txt 2>&1; cat gpurun_out/thr_debug.txt | tail -30
timeout 600 python -m pytest tests/test_gpu_unet.py tests/test_gpu_annot.py tests/test_gpu_configs.py tests/test_gpu_post.py tests/test_gpu_kernels.py -m gpu -q -rf --tb=short -p no:cacheprovider -k "bn_on_load or golden or category_layers or resize or loss_kernels or chain" > gpurun_out/pytest_g2.log 2>&1; echo "pytest rc=$?"; tail -30 gpurun_out/pytest_g2.log
MSC_BN_ON_LOAD=1 ./run_gpu_round.sh b32 2>&1 | head -20
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/bench_ns.log 2>&1; echo "bench rc=$?"; grep '^{' gpurun_out/bench_ns.log | tail -1 > gpurun_out/bench_ns.json; python -c "
import json; d=json.load(open('gpurun_out/bench_ns.json')); print(d['value'], d['ms_per_step']); print(json.dumps(d.get('north_star'), indent=1))"
AB="MSC_FORCE_COLLECTIVES=0 MSC_FORCE_COLLECTIVES=1 MSC_FORCE_COLLECTIVES=0 MSC_FORCE_COLLECTIVES=1" ./run_gpu_round.sh ab
