#!/bin/bash
# does the driver's short run (--steps 20 --warmup 5) measure the same step as a long one?  same box, alternating
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
run() { timeout 600 python bench.py --gpus 1 "$@" --no-cpu-baseline --no-breakdown --no-north-star 2>&1 | grep -o '"ms_per_step": [0-9.]*' | head -1; }
for i in 1 2 3; do
  echo "steps 20 warmup 5:   $(run --steps 20 --warmup 5)"
  echo "steps 200 warmup 5:  $(run --steps 200 --warmup 5)"
  echo "steps 20 warmup 100: $(run --steps 20 --warmup 100)"
done
echo "default line (as the driver runs it):"; timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2>&1 | grep '^{' | cut -c1-1200
