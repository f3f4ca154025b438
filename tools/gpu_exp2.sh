#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
for cfg in "256 2048" "64 2048" "256 1024" "256 4096" "128 2048" "64 4096"; do
  set -- $cfg
  echo "== MSC_BN_CT=$1 MSC_BN_BLOCKS=$2"
  MSC_BN_CT=$1 MSC_BN_BLOCKS=$2 timeout 300 python tools/bn_probe.py 2>&1 | grep "^M="
done | tee gpurun_out/bn_probe.txt
