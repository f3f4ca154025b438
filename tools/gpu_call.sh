cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
export MSC_TUNE_CACHE="$PWD/gpurun_out/tune_cache.json"
timeout 300 python tools/deconv_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/deconv_probe.txt
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -rf --tb=short -p no:cacheprovider -k "transposed_conv" 2>&1 | tail -6
