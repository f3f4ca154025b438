#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
L=open-solution-mapping-challenge_amd/lib
AB="MSC_HIP_LIB=$L/libmsc_hip_prev.so MSC_X=0 MSC_HIP_LIB=$L/libmsc_hip_prev.so MSC_X=0 MSC_HIP_LIB=$L/libmsc_hip_prev.so MSC_X=0" ./run_gpu_round.sh ab
for f in gpurun_out/ab_MSC_HIP_LIB_*prev.so.log gpurun_out/ab_MSC_X_0.log; do grep -o '"msc_conv_igemm": [0-9.]*' $f | head -1; done
for v in "MSC_HIP_LIB=$L/libmsc_hip_prev.so" "MSC_X=0" "MSC_HIP_LIB=$L/libmsc_hip_prev.so" "MSC_X=0"; do echo "infer101 $v: $(env $v timeout 300 python bench.py --workload infer --encoder 101 --steps 200 --warmup 10 --no-cpu-baseline --no-breakdown 2>&1 | grep -o '"ms_per_step": [0-9.]*' | head -1)"; done
