#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -rf --tb=short -p no:cacheprovider -k "relu_backward_and_bias or stem" 2>&1 | tail -5
