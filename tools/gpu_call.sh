cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "node2vec or metric_graph" 2>&1 < /dev/null | grep -E "passed|failed|rror|assert" | tail -8
for cfg in "2 0 100000" "2 0 1000"; do timeout 120 python tools/n2v_one.py $cfg 2>&1 < /dev/null | grep stats; done
