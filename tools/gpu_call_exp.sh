cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 150 python -m pytest tests/test_gpu_unet.py -m gpu -q --tb=line -p no:cacheprovider -x 2>&1 | grep -E "passed|failed|error" | tail -2
