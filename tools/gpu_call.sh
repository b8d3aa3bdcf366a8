cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q 2>&1 < /dev/null | grep -E "passed|failed|error" | tail -3 | tee gpurun_out/r2_gpu_pytest_summary.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 < /dev/null | grep -i smoke
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r2_final_bench.json 2>/dev/null < /dev/null; tail -1 gpurun_out/r2_final_bench.json | cut -c1-260
