#!/bin/bash
# One GPU-box session: probes, parity tests, smoke, bench.  Logs go to gpurun_out/.
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
export MSC_TUNE_CACHE="${GRAFT_REPO_ROOT:-.}/gpurun_out/tune_cache.json"
R="$PWD"
STAGES="${1:-tests smoke bench}"
bench_line() {   # bench_line <tag> <timeout> <bench.py args...>: one JSON line -> gpurun_out/bench_<tag>.json
  local tag=$1 to=$2; shift 2
  timeout $to python bench.py "$@" > gpurun_out/bench_$tag.log 2>&1; echo "bench $tag rc=$?"
  grep '^{' gpurun_out/bench_$tag.log | tail -1 > gpurun_out/bench_$tag.json; cut -c1-400 gpurun_out/bench_$tag.json
}
prof_stats() {   # prof_stats <tag> <bench.py args...>: rocprofv3 kernel-trace statistics of the same command -> gpurun_out/prof_<tag>/
  local tag=$1; shift
  ( cd /tmp; export TMPDIR=/tmp; rm -rf "$R/gpurun_out/prof_$tag"
    timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_$tag" -- python "$R/bench.py" "$@" --no-cpu-baseline --no-north-star > "$R/gpurun_out/prof_$tag.log" 2>&1; echo "prof $tag rc=$?" )
  python tools/kernel_stats_summary.py gpurun_out/prof_$tag > gpurun_out/kernel_stats_$tag.txt 2>&1; head -12 gpurun_out/kernel_stats_$tag.txt | cut -c1-200
}
for s in $STAGES; do
case $s in
probe) timeout 60 ./probes/tr_probe > gpurun_out/tr_probe.txt 2>&1; echo "probe rc=$?";;
tests) timeout 1800 python -m pytest tests -m gpu -q -rf --tb=short -p no:cacheprovider --durations=8 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -40 gpurun_out/pytest_gpu.log;;
ptests) timeout 1200 python -m pytest tests/test_gpu_parity_timed.py -m gpu -q -rf --tb=short -p no:cacheprovider > gpurun_out/pytest_parity.log 2>&1; echo "pytest rc=$?"; tail -30 gpurun_out/pytest_parity.log; cat gpurun_out/parity_timed.json;;
smoke) timeout 600 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -5 gpurun_out/smoke.log;;
bench) # every line DESIGN.md quotes; the default command (BASELINE.json's metric) last
       bench_line infer_r34 600 --workload infer
       bench_line infer_r101 600 --workload infer --encoder 101
       bench_line infer_r101_320 600 --workload infer --encoder 101 --size 320 --no-cpu-baseline
       bench_line train_r101_320 900 --size 320 --steps 100 --no-cpu-baseline
       bench_line train_r101_fp32 900 --dtype fp32 --steps 20 --no-cpu-baseline
       bench_line infer_r101_fp32 600 --workload infer --encoder 101 --dtype fp32 --steps 20 --no-cpu-baseline
       bench_line tta_r152_fp16 900 --workload tta
       bench_line post 600 --workload post
       bench_line annot 600 --workload annot
       bench_line e2e 900 --workload e2e
       bench_line train 1200;;
benchq) bench_line train 1200 --no-cpu-baseline;;
prof)  python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-breakdown --no-north-star > gpurun_out/tune_warm.log 2>&1   # fills the tune cache so the profile holds no tuning launches
       prof_stats train --steps 20 --warmup 2
       python bench.py --workload infer --encoder 101 --steps 2 --warmup 1 --no-cpu-baseline --no-breakdown --no-north-star > gpurun_out/tune_warm.log 2>&1
       prof_stats infer_r101 --workload infer --encoder 101 --steps 20 --warmup 2
       prof_stats post --workload post --steps 20 --warmup 2
       prof_stats e2e --workload e2e --steps 5 --warmup 2;;
proft) python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-breakdown --no-north-star > gpurun_out/tune_warm.log 2>&1
       prof_stats train --steps 20 --warmup 2;;
pmc)   python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-breakdown --no-north-star > gpurun_out/tune_warm.log 2>&1
       cd /tmp; export TMPDIR=/tmp
       for c in FETCH_SIZE WRITE_SIZE; do
         timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/pmc_$c" -- python "$GRAFT_REPO_ROOT/bench.py" --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-breakdown --no-north-star > "$GRAFT_REPO_ROOT/gpurun_out/pmc_$c.log" 2>&1; echo "pmc $c rc=$?"
       done
       cd "$GRAFT_REPO_ROOT"; python tools/pmc_summary.py gpurun_out gpurun_out/pmc_traffic.json > gpurun_out/pmc_summary.txt 2>&1; head -12 gpurun_out/pmc_summary.txt;;
pmcsq) # SQ / LDS / L2 counters of the train step, one rocprofv3 pass per counter group (counter runs carry kernel-trace only)
       python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-breakdown --no-north-star > gpurun_out/tune_warm.log 2>&1
       cd /tmp; export TMPDIR=/tmp
       i=0
       for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" "TCC_HIT_sum TCC_MISS_sum"; do
         i=$((i+1)); rm -rf "$GRAFT_REPO_ROOT/gpurun_out/pmcsq_$i"
         timeout 900 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/pmcsq_$i" -- python "$GRAFT_REPO_ROOT/bench.py" --steps 1 --warmup 1 --no-graph --no-cpu-baseline --no-breakdown --no-north-star > "$GRAFT_REPO_ROOT/gpurun_out/pmcsq_$i.log" 2>&1; echo "pmcsq $i rc=$?"
       done
       cd "$GRAFT_REPO_ROOT"; python tools/pmc_sq_summary.py gpurun_out/pmcsq_1 gpurun_out/pmcsq_2 gpurun_out/pmcsq_3 > gpurun_out/pmcsq_summary.txt 2>&1; head -30 gpurun_out/pmcsq_summary.txt | cut -c1-260;;
b32)   timeout 900 python -m pytest tests/test_gpu_parity_timed.py tests/test_gpu_fullsize.py -m gpu -q -rf --tb=short -p no:cacheprovider -k "batch32" --durations=5 > gpurun_out/pytest_b32.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/pytest_b32.log; cat gpurun_out/parity_timed.json | head -60;;
ktests) timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_unet.py -m gpu -q -rf --tb=short -p no:cacheprovider > gpurun_out/pytest_k.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/pytest_k.log;;
bnl)   # BatchNorm + ReLU on load (msc_conv_desc.in_bn, ABI v9): the kernel test, the end-to-end comparison with the separate apply launches, then the step A/B
       timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -rf --tb=short -p no:cacheprovider -k "on_load" > gpurun_out/pytest_bnl.log 2>&1; echo "pytest rc=$?"; tail -25 gpurun_out/pytest_bnl.log
       timeout 600 python -m pytest tests/test_gpu_unet.py -m gpu -q -rf --tb=short -p no:cacheprovider -k "bn_on_load" > gpurun_out/pytest_bnl2.log 2>&1; echo "pytest rc=$?"; tail -25 gpurun_out/pytest_bnl2.log
       AB="MSC_BN_ON_LOAD=0 MSC_BN_ON_LOAD=1 MSC_BN_ON_LOAD=1,MSC_BN_ON_LOAD_3X3=0 MSC_BN_ON_LOAD=0" "$0" ab;;
ab)    # A/B of environment switches on the train step: AB="NAME=VAL,NAME2=VAL2 NAME=VAL ..." (one run per word)
       python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-breakdown --no-north-star > gpurun_out/tune_warm.log 2>&1
       for cfg in $AB; do
         tag=$(echo "$cfg" | tr ',=/' '___')
         ( IFS=,; for kv in $cfg; do export "$kv"; done; unset IFS; timeout 600 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-north-star $ABFLAGS > "gpurun_out/ab_$tag.log" 2>&1 )
         echo "$cfg: $(grep -o '"ms_per_step": [0-9.]*' "gpurun_out/ab_$tag.log" | head -1) $(grep -o '"wgrad": {[^}]*}' "gpurun_out/ab_$tag.log" | head -1) $(grep -o '"msc_adam_pack": [0-9.]*' "gpurun_out/ab_$tag.log" | head -1)"
       done;;
esac
done
