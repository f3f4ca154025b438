#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "$K" --tb=short -p no:cacheprovider > gpurun_out/pytest_k.log 2>&1; echo "pytest rc=$?"; tail -30 gpurun_out/pytest_k.log | cut -c1-400
