#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
./run_gpu_round.sh "tests smoke"
python bench.py > gpurun_out/bench_train.log 2>&1; grep '^{' gpurun_out/bench_train.log | tail -1 > gpurun_out/bench_train.json; cut -c1-3000 gpurun_out/bench_train.json
