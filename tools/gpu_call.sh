#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
export MSC_TUNE_CACHE="$PWD/gpurun_out/tune_cache.json"
timeout 300 python probes/wreg_diag.py 2>&1 | grep -v "^C w=\|amdgpu.ids" | head -40
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -rf --tb=line -p no:cacheprovider -k "halo_tile_kernel_for_3x3" 2>&1 | tail -5
timeout 900 python -m pytest tests/test_gpu_post.py tests/test_gpu_pipeline.py tests/test_gpu_prep.py -m gpu -q -rf --tb=short -p no:cacheprovider 2>&1 | tail -5
timeout 600 python tools/conv_cfg_table.py --shapes dec --cfgs 42,46,53,59,60 2>&1 | grep -v amdgpu.ids
for cfg in "MSC_RECT_TILED=0" "MSC_RECT_TILED=1"; do env $cfg timeout 600 python tools/post_chain_ab.py 2>&1 | grep "per 128" | sed "s/^/[$cfg] /"; done
