cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python tools/run_layerwise_gpu.py 2>&1 < /dev/null | tail -25 | cut -c1-3000
