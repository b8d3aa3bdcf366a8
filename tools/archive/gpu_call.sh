cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "node2vec or walk or metric_graph" 2>&1 < /dev/null | grep -E "passed|failed|rror" | tail -3
