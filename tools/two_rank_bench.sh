#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
MSC_DIST_BACKEND=gloo MSC_DIST_ONE_DEVICE=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 2 > gpurun_out/bench_two_ranks_one_gpu.log 2>&1; echo "rc=$?"
grep '^{' gpurun_out/bench_two_ranks_one_gpu.log | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print(d['n_gpus'], d['config']['global_batch'], d['config']['parallelism'], d['config']['hipgraph'], round(d['ms_per_step'],1), d['config']['final_loss'])"
