#!/bin/bash
# Round-3 GPU sessions (stages picked by argument); logs -> gpurun_out/
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
export MSC_TUNE_CACHE="$PWD/gpurun_out/tune_cache.json"
R="$PWD"
pt() {   # pt <tag> <timeout> <pytest args...>
  local tag=$1 to=$2; shift 2
  timeout $to python -m pytest "$@" -m gpu -q -rf --tb=short -p no:cacheprovider --durations=6 > gpurun_out/pytest_$tag.log 2>&1; echo "pytest $tag rc=$?"; tail -25 gpurun_out/pytest_$tag.log
}
bench_line() {
  local tag=$1 to=$2; shift 2
  timeout $to python bench.py "$@" > gpurun_out/bench_$tag.log 2>&1; echo "bench $tag rc=$?"
  grep '^{' gpurun_out/bench_$tag.log | tail -1 > gpurun_out/bench_$tag.json; cut -c1-600 gpurun_out/bench_$tag.json
}
for s in "$@"; do
case $s in
bneck)  pt bneck 600 tests/test_gpu_kernels.py -k "fused_bottleneck";;
k2gib)  pt k2gib 600 tests/test_gpu_kernels.py -k "beyond_2gib";;
kernels) pt kernels 900 tests/test_gpu_kernels.py;;
configs) pt configs 1200 tests/test_gpu_configs.py;;
unet)   pt unet 900 tests/test_gpu_unet.py tests/test_gpu_fullsize.py tests/test_gpu_annot.py;;
all)    pt all 2400 tests;;
infer)  bench_line infer_r101 600 --workload infer --encoder 101 --no-cpu-baseline --dump-launches gpurun_out/launches_infer_r101.json
        MSC_FUSE_BNECK=0 bench_line infer_r101_unfused 600 --workload infer --encoder 101 --no-cpu-baseline;;
trainq) bench_line train 900 --no-cpu-baseline --steps 50;;
stamps) MSC_BNECK_ABL=8 timeout 120 python tools/bneck_probe.py 2>&1 | tail -16;;
probe)  for a in 0 7; do MSC_BNECK_ABL=$a timeout 120 python tools/bneck_probe.py 2>&1 | tail -1; done
        CMID=128 HW=32 timeout 120 python tools/bneck_probe.py 2>&1 | tail -1; CMID=64 HW=64 timeout 120 python tools/bneck_probe.py 2>&1 | tail -1;;
smoke)  timeout 600 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/smoke.log;;
esac
done
