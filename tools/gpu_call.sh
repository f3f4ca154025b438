cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
# the product's own mitigation without the conftest fixture: the failing order through tools/dirty_probe5.py (plain = no manual synchronise)
MODE=plain timeout 900 python tools/dirty_probe5.py 2>&1 | grep "^MODE"
MODE=plain timeout 900 python tools/dirty_probe5.py 2>&1 | grep "^MODE"
timeout 900 python -m pytest tests/test_gpu_configs.py -m gpu -q --tb=line -p no:cacheprovider 2>&1 | tail -3
