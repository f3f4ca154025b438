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
post)   pt post 900 tests/test_gpu_post.py tests/test_gpu_pipeline.py tests/test_gpu_annot.py;;
postq)  pt postq 600 tests/test_gpu_post.py tests/test_gpu_annot.py;;
e2e)    bench_line e2e 900 --workload e2e --no-cpu-baseline; python -c "import json; d=json.load(open('gpurun_out/bench_e2e.json')); print(json.dumps(d['config']['variants'], indent=1)); print(d['roofline']['dense_crf'])";;
e2e_ws2) MSC_WS_PASSES=2 bench_line e2e_ws2 900 --workload e2e --no-cpu-baseline; python -c "import json; d=json.load(open('gpurun_out/bench_e2e_ws2.json')); print({k:(round(v['post_only_img_s']),round(v['end_to_end_img_s'])) for k,v in d['config']['variants'].items() if isinstance(v,dict)})";;
postb)  bench_line post 600 --workload post --no-cpu-baseline;;
prof_e2e) ( cd /tmp; export TMPDIR=/tmp; rm -rf "$R/gpurun_out/prof_e2e"; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_e2e" -- python "$R/bench.py" --workload e2e --steps 5 --warmup 2 --no-cpu-baseline > "$R/gpurun_out/prof_e2e.log" 2>&1; echo "prof e2e rc=$?" )
        python tools/kernel_stats_summary.py gpurun_out/prof_e2e > gpurun_out/kernel_stats_e2e.txt 2>&1; head -60 gpurun_out/kernel_stats_e2e.txt | cut -c1-180;;
tailprof) timeout 300 python tools/tail_profile.py 0 2>&1 | grep -v '^$' | head -50 | cut -c1-150; timeout 300 python tools/tail_profile.py 5 2>&1 | grep -v '^$' | head -45 | cut -c1-150;;
hiptrace) ( cd /tmp; export TMPDIR=/tmp; rm -rf "$R/gpurun_out/hip_tail"; timeout 600 rocprofv3 --hip-trace --kernel-trace --stats --output-format csv -d "$R/gpurun_out/hip_tail" -- python "$R/tools/tail_profile.py" 0 > "$R/gpurun_out/hip_tail.log" 2>&1; echo rc=$? )
        f=$(ls gpurun_out/hip_tail/*/*hip_api_stats.csv 2>/dev/null | head -1); echo $f; head -25 "$f" | cut -c1-160; f2=$(ls gpurun_out/hip_tail/*/*kernel_stats.csv | head -1); head -30 "$f2" | cut -c1-200;;
fin)    pt fin 900 tests/test_gpu_kernels.py -k 'fused_final or epilogues'; pt fin2 900 tests/test_gpu_unet.py tests/test_gpu_fullsize.py;;
splitk) pt splitk 600 tests/test_gpu_kernels.py -k 'split_k or epilogues or every_kernel'; pt splitk2 900 tests/test_gpu_unet.py tests/test_gpu_parity_timed.py;;
deconv) timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k 'transposed or every_kernel or transpose' 2>&1 | tail -3; for nb in 0 256 512; do MSC_DECONV_BLOCKS=$nb timeout 300 python bench.py --workload infer --encoder 101 --steps 30 --no-cpu-baseline --dump-launches gpurun_out/l_$nb.json > /dev/null 2>&1; python -c "import json; d=json.load(open('gpurun_out/l_$nb.json')); print('deconv blocks $nb:', [round(x['us'],1) for x in d if x['k']=='conv' and x['mode']==1 and x['Cout']==32], [round(x['us'],1) for x in d if x['k']=='conv' and x['Cin']==32 and x['Cout']==32], round(sum(x['us'] for x in d),1))"; done;;
join) timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_unet.py -q -x -k 'residual_join or batchnorm_backward or train or backward or gradient or graph or every_kernel' > gpurun_out/join.log 2>&1; grep -E 'passed|failed' gpurun_out/join.log | tail -1; grep -E '^FAILED|Error' gpurun_out/join.log | head; for v in 1 0 1 0; do MSC_FUSE_JOIN_BWD=$v timeout 600 python bench.py --steps 100 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); f=d['roofline']['family_ms_per_step']; print('join fuse $v:', round(d['ms_per_step'],3), 'conv', f['msc_conv_igemm'], 'reduce', f['msc_bn_bwd_reduce'])"; done;;
stem) timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k 'stem_halo' 2>&1 | tail -3; for nb in 0 256 512 1024; do MSC_STEM_BLOCKS=$nb MSC_TUNE_DB=0 MSC_TUNE_CACHE=/tmp/t_$nb.json timeout 300 python bench.py --workload infer --encoder 101 --steps 30 --no-cpu-baseline --dump-launches gpurun_out/l_$nb.json > /dev/null 2>&1; python -c "import json; d=json.load(open('gpurun_out/l_$nb.json')); print('stem blocks $nb:', [round(x['us'],1) for x in d if x['k']=='conv' and x['KH']==7], round(sum(x['us'] for x in d),1))"; done;;
c32) timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k '32_channel or fused_final or relu_backward or every_kernel' 2>&1 | tail -3; for nb in 0 512 1024 2048; do MSC_C32_BLOCKS=$nb timeout 300 python bench.py --workload infer --encoder 101 --steps 30 --no-cpu-baseline --dump-launches gpurun_out/l_$nb.json > /dev/null 2>&1; python -c "import json; d=json.load(open('gpurun_out/l_$nb.json')); print('c32 blocks $nb:', [round(x['us'],1) for x in d if x['k']=='conv' and x['Cin']==32 and x['Cout']==32], round(sum(x['us'] for x in d),1))"; done;;
stream) timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k 'streaming' 2>&1 | tail -3; for m in fwd eval dgrad; do MODE=$m timeout 300 python tools/conv1x1_probe.py 2>&1 | grep -v amdgpu.ids | tail -12; done;;
smoke)  timeout 600 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/smoke.log;;
esac
done
