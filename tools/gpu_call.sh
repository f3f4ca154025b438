#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
AB="MSC_DOWN4=1 MSC_DOWN4=0 MSC_DOWN4=1 MSC_DOWN4=0" ./run_gpu_round.sh ab
