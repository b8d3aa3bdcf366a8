cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_layerwise_gpu.py -m gpu -q -k "cpp or example or host" 2>&1 < /dev/null | grep -E "passed|failed|error" | tail -3
