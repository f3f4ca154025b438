#!/bin/bash
# kernel trace of the train step (graph replay): per (kernel, grid) durations -> gpurun_out/trace_train.txt
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"; mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0 MSC_TUNE_CACHE="$R/gpurun_out/tune_cache.json"
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-breakdown > gpurun_out/tune_warm.log 2>&1
( cd /tmp; export TMPDIR=/tmp; rm -rf "$R/gpurun_out/trace_train"
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$R/gpurun_out/trace_train" -- python "$R/bench.py" --steps 20 --warmup 2 --no-cpu-baseline --no-breakdown $TRACEFLAGS > "$R/gpurun_out/trace_train.log" 2>&1; echo "trace rc=$?" )
python tools/trace_by_grid.py gpurun_out/trace_train > gpurun_out/trace_train.txt 2>&1; head -5 gpurun_out/trace_train.txt
rm -rf gpurun_out/trace_train
