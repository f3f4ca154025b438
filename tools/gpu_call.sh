#!/bin/bash
# scratch runner for one-off GPU calls of a round (gpurun -- ./tools/gpu_call.sh); the stages that matter live in run_gpu_round.sh
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
./run_gpu_round.sh "ktests benchq"
