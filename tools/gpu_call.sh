#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 240 python -m pytest tests/test_gpu_two_ranks.py -m gpu -q -rf --tb=short -p no:cacheprovider -k "refused" 2>&1 | tail -30
