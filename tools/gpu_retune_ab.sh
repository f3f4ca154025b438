#!/bin/bash
# same-box comparison: the shipped tuning db vs a fresh measurement of every conv shape (kept in gpurun_out/tune_new.json)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
run() { timeout 600 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-breakdown "$@" 2>&1 | grep -o '"ms_per_step": [0-9.]*' | head -1; }
echo "shipped: $(run) $(run)"
TUNE_EXTRA="320 tta" bash tools/tune_run.sh > gpurun_out/retune.log 2>&1
export MSC_TUNE_DB=0 MSC_TUNE_CACHE=$PWD/gpurun_out/tune_new.json
echo "fresh:   $(run) $(run)"
echo "fresh infer101: $(run --workload infer --encoder 101)"
unset MSC_TUNE_DB MSC_TUNE_CACHE
echo "shipped infer101: $(run --workload infer --encoder 101)"
