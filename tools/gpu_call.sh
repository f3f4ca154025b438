cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
R="$PWD"
export MSC_TUNE_CACHE="$PWD/gpurun_out/tune_cache.json"
rm -f gpurun_out/tune_cache.json
for pk in 1 0; do MSC_CRF_PK=$pk timeout 300 python tools/crf_probe.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/crf_probe.txt; done
./run_gpu_round.sh tests 2>&1 | tail -25
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/bench_ns.log 2>&1; echo "bench rc=$?"; grep '^{' gpurun_out/bench_ns.log | tail -1 > gpurun_out/bench_ns.json; python -c "
import json; d=json.load(open('gpurun_out/bench_ns.json')); print(d['value'], d['ms_per_step'], d['roofline']['family_ms_per_step']); print(json.dumps(d.get('north_star'))[:1500])"
AB="MSC_BN_ON_LOAD=0 MSC_BN_ON_LOAD=1" ./run_gpu_round.sh ab
