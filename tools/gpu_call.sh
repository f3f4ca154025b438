#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
for b in 256 512 1024; do echo "MSC_STEM_BLOCKS=$b"; MSC_STEM_BLOCKS=$b timeout 300 python tools/stem_probe.py 2>&1 | grep "cfg" ; done
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_unet.py -m gpu -q -rf --tb=short -p no:cacheprovider -k "stem or every_kernel or eval_logits" 2>&1 | tail -3
