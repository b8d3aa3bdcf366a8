cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 300 examples/cpp/sharded_fanout 1 300000 2048 > gpurun_out/c17_cpp.log 2>&1; echo "cpp rc=$?"; tail -3 gpurun_out/c17_cpp.log
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/c17_pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/c17_pytest.log | cut -c1-600
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/c17_smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/c17_smoke.log
timeout 200 python bench.py --force-sharded --steps 10 --warmup 3 --repeats 3 --no-cpu-baseline --no-check --no-small-batch > gpurun_out/c17_sharded1.json 2> gpurun_out/c17_sharded1.err; echo "sharded1 rc=$?"; tail -1 gpurun_out/c17_sharded1.json | cut -c1-200
