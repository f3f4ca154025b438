#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 60 ./probes/kernarg_latency_probe 2>&1 | tee gpurun_out/kernarg_latency.txt
