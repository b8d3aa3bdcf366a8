cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "inline_line or sample_neighbor_vs_oracle or small_fanout" 2>&1 < /dev/null | tail -5
