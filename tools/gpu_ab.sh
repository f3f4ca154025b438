#!/bin/bash
# A/B of environment switches on the train step: AB="NAME=VAL,NAME2=VAL2 NAME=VAL ..." (one run per word; "base" = no switch)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0 MSC_TUNE_CACHE="$PWD/gpurun_out/tune_cache.json"
for cfg in $AB; do
  tag=$(echo "$cfg" | tr ',=' '__')
  ( if [ "$cfg" != base ]; then IFS=,; for kv in $cfg; do export "$kv"; done; unset IFS; fi
    timeout 600 python bench.py --steps ${ABSTEPS:-100} --warmup 5 --no-cpu-baseline $ABFLAGS > "gpurun_out/ab_$tag.log" 2>&1 )
  echo "$cfg: $(grep -o '"ms_per_step": [0-9.]*' "gpurun_out/ab_$tag.log" | head -1) $(grep -o '"family_ms_per_step": {[^}]*}' "gpurun_out/ab_$tag.log" | head -1)"
done
