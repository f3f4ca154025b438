#!/bin/bash
# scratch runner for one-off GPU calls of a round (gpurun -- ./tools/gpu_call.sh); the stages that matter live in run_gpu_round.sh
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
export MSC_TUNE_CACHE="$PWD/gpurun_out/tune_cache.json"
timeout 300 ./probes/coop_bn_probe > gpurun_out/coop_bn_probe.txt 2>&1; echo "probe rc=$?"; cat gpurun_out/coop_bn_probe.txt
timeout 900 python -m pytest tests/test_gpu_replay_hazard.py tests/test_gpu_post.py tests/test_gpu_prep.py -m gpu -q -rf --tb=short -p no:cacheprovider -s > gpurun_out/pytest_hazard.log 2>&1; echo "pytest rc=$?"; tail -30 gpurun_out/pytest_hazard.log
