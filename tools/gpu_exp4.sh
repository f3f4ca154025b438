#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_post.py tests/test_gpu_pipeline.py tests/test_gpu_annot.py -m gpu -q -x --tb=short -p no:cacheprovider > gpurun_out/pytest_post.log 2>&1; echo "pytest rc=$?"; tail -8 gpurun_out/pytest_post.log | cut -c1-300
timeout 300 python bench.py --workload post --no-cpu-baseline 2>&1 | grep '^{' | cut -c1-900
