cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
export MSC_TUNE_CACHE="$PWD/gpurun_out/tune_cache.json"
AB="MSC_BN_ON_LOAD=0 MSC_BN_ON_LOAD=1 MSC_BN_ON_LOAD=1,MSC_BN_ON_LOAD_MIN_PIXELS=0 MSC_BN_ON_LOAD=0 MSC_BN_ON_LOAD=1 MSC_BN_ON_LOAD=1,MSC_BN_ON_LOAD_MIN_PIXELS=0" ./run_gpu_round.sh ab
for t in MSC_BN_ON_LOAD_0 MSC_BN_ON_LOAD_1 MSC_BN_ON_LOAD_1_MSC_BN_ON_LOAD_MIN_PIXELS_0; do python - $t <<'PY'
import json, sys
t = sys.argv[1]
d = json.loads([l for l in open('gpurun_out/ab_%s.log' % t) if l.startswith('{')][-1]); f = d['roofline']['family_ms_per_step']
print(t, 'step %.3f conv %.3f (frac %.3f) bn_apply %.3f bn_bwd %.3f' % (d['ms_per_step'], f['msc_conv_igemm'], d['roofline']['frac'], f['msc_bn_apply'], f['msc_bn_bwd_apply']))
PY
done
./run_gpu_round.sh b32 2>&1 | head -8
