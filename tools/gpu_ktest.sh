#!/bin/bash
# a subset of the kernel tests on the GPU box:  K="<pytest -k expression>" bash tools/gpu_ktest.sh [test files...]
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
FILES="${@:-tests/test_gpu_kernels.py}"
timeout 900 python -m pytest $FILES -m gpu -q -x ${K:+-k "$K"} --tb=short -p no:cacheprovider > gpurun_out/pytest_k.log 2>&1; echo "pytest rc=$?"; tail -30 gpurun_out/pytest_k.log | cut -c1-400
