#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
MSC_WGRAD_PARTS=3 timeout 900 python -m pytest tests/test_gpu_unet.py tests/test_abi.py -m gpu -q -rf --tb=short -p no:cacheprovider -k "train or deterministic or abi or rccl" 2>&1 | tail -4
AB="MSC_WGRAD_PARTS=1 MSC_WGRAD_PARTS=2 MSC_WGRAD_PARTS=3 MSC_WGRAD_PARTS=7 MSC_WGRAD_PARTS=1 MSC_WGRAD_PARTS=3" ./run_gpu_round.sh ab
