#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
MSC_WGRAD_PERSIST=40 timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_unet.py -m gpu -q -rf --tb=short -p no:cacheprovider -k "wgrad or group or train or grad" 2>&1 | tail -5
O="MSC_OVERLAP_WGRAD=1,MSC_WGRAD_FLUSH=dec,l3"
AB="MSC_X=0 MSC_CORUN=1 MSC_CORUN=1,$O,MSC_WGRAD_PERSIST=256 MSC_CORUN=1,$O,MSC_WGRAD_PERSIST=512 $O,MSC_WGRAD_PERSIST=256 MSC_CORUN=1,$O,MSC_WGRAD_PERSIST=256,MSC_WGRAD_GROUP_TILE=64 MSC_WGRAD_PERSIST=512 MSC_WGRAD_PERSIST=768 MSC_X=0" ./run_gpu_round.sh ab
