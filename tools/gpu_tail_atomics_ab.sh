#!/bin/bash
# loss_sums / final_bwd: blocks per launch against the same-address atomics at their tails (rocprofv3 kernel durations of the train step)
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"; mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-breakdown --no-north-star > gpurun_out/tune_warm.log 2>&1
for cfg in "2048 1024" "1024 512" "512 512" "512 256" "256 256" "128 128"; do
  set -- $cfg
  ( cd /tmp; export TMPDIR=/tmp; rm -rf "$R/gpurun_out/prof_tail"
    MSC_LOSS_BLOCKS=$1 MSC_FINAL_BWD_BLOCKS=$2 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_tail" -- python "$R/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --no-breakdown --no-north-star > "$R/gpurun_out/prof_tail.log" 2>&1 )
  python tools/kernel_stats_summary.py gpurun_out/prof_tail > gpurun_out/kernel_stats_tail.txt 2>&1
  echo "MSC_LOSS_BLOCKS=$1 MSC_FINAL_BWD_BLOCKS=$2: $(grep -E 'loss_sums_kernel|final_bwd_kernel|loss_grad_kernel' gpurun_out/kernel_stats_tail.txt | awk '{printf "%s avg %s us; ", $NF, $4}' | cut -c1-300)"
done
