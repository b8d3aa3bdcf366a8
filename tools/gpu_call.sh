cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "node2vec or walk" 2>&1 < /dev/null | tail -12
timeout 300 python tools/prof_n2v.py 2>&1 < /dev/null | tail -16 | cut -c1-1500 | tee gpurun_out/r2g_prof_n2v.txt
