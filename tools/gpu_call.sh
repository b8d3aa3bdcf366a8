cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/c3_pytest.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/c3_pytest.log
timeout 400 python tools/ab_round2.py > gpurun_out/c3_ab.log 2>&1; echo "ab rc=$?"; tail -12 gpurun_out/c3_ab.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/c3_bench.json 2> gpurun_out/c3_bench.err; echo "bench rc=$?"; tail -c 1500 gpurun_out/c3_bench.json; tail -5 gpurun_out/c3_bench.err
