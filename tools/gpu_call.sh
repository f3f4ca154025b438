cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
R="$PWD"
export MSC_TUNE_CACHE="$PWD/gpurun_out/tune_cache.json"
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-breakdown --no-north-star > gpurun_out/tune_warm.log 2>&1
for cfg in 0 1 10000 0 1 10000; do
  MSC_PREFETCH_W=$cfg timeout 600 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-north-star > gpurun_out/ab_prefetch_$cfg.log 2>&1
  python - $cfg <<'PY'
import json, sys
cfg = sys.argv[1]
line = [l for l in open('gpurun_out/ab_prefetch_%s.log' % cfg) if l.startswith('{')][-1]
d = json.loads(line); f = d['roofline']['family_ms_per_step']
print('MSC_PREFETCH_W=%s: step %.3f ms  conv family %.3f (b2b %.3f)  prefetch %.3f  bn_apply %.3f  bn_bwd_apply %.3f  wgrad %.3f  sum %.3f' % (
    cfg, d['ms_per_step'], f.get('msc_conv_igemm', 0), d['roofline']['back_to_back']['ms_per_step'], f.get('msc_l2_prefetch', 0), f.get('msc_bn_apply', 0), f.get('msc_bn_bwd_apply', 0),
    d['roofline']['wgrad']['ms_per_step'], d['roofline']['sum_kernel_ms_per_step']))
PY
done
