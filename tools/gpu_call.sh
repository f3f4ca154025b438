#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
L=open-solution-mapping-challenge_amd/lib
for env in "MSC_X=1" "MSC_DOWN4=0" "MSC_HIP_LIB=$L/libmsc_hip_prev.so" "MSC_X=1"; do
  echo "== $env"; env $env timeout 600 python -m pytest tests/test_gpu_configs.py -m gpu -q --tb=line -p no:cacheprovider -k "trajectory" 2>&1 | grep -E "passed|failed|AssertionError" | cut -c1-200
  python - <<'P'
import json
try:
    d=json.load(open('gpurun_out/parity_configs.json')); t=d.get('bf16_r101_256_trajectory',{}); print('rel_max %.4f rel_mean %.4f iou %.4f'%(t.get('rel_max',-1),t.get('rel_mean',-1),t.get('final_mask_iou',-1)))
except Exception as e: print(e)
P
done
