# Re-measure the per-layer kernel configuration of every conv shape of the benchmarked workloads (MSC_TUNE_DB=0 ignores the
# shipped db); the result (gpurun_out/tune_new.json) is merged into open-solution-mapping-challenge_amd/tune/gfx950.json.
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
export MSC_TUNE_DB=0 MSC_TUNE_CACHE=$PWD/gpurun_out/tune_new.json
rm -f gpurun_out/tune_new.json
run() { tag=$1; shift; timeout 900 python bench.py "$@" --no-cpu-baseline > gpurun_out/tune_$tag.log 2>&1; echo "$tag rc=$? $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/tune_$tag.log | head -1)"; }
run train --steps 30 --warmup 3 --dump-launches gpurun_out/launches_train_new.json
grep -o '"family_ms_per_step": {[^}]*}' gpurun_out/tune_train.log
run infer101 --workload infer --encoder 101 --steps 30 --warmup 3 --dump-launches gpurun_out/launches_infer_r101_new.json
run infer34 --workload infer --steps 30 --warmup 3
for extra in $TUNE_EXTRA; do
case $extra in
  320) run infer320 --workload infer --encoder 101 --size 320 --steps 30 --warmup 3; run train320 --size 320 --steps 20 --warmup 3;;
  tta) run tta --workload tta --steps 5 --warmup 2;;
esac
done
