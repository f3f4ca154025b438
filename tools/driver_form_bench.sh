#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_train_driver_form.log 2>&1; grep '^{' gpurun_out/bench_train_driver_form.log | tail -1 > gpurun_out/bench_train_driver_form.json; python - <<'P'
import json; d=json.load(open('gpurun_out/bench_train_driver_form.json')); r=d['roofline']; print(d['value'], d['ms_per_step'], r['frac'], r['traffic'], d['north_star']['forward']['img_s'])
P
