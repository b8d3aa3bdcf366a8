cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 < /dev/null | tail -4
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 < /dev/null | tail -2
timeout 900 bash tools/round2_profile.sh r2g 2>&1 < /dev/null | grep -E "rc=" 
timeout 900 python bench.py > gpurun_out/r2g_bench.json 2> gpurun_out/r2g_bench.err < /dev/null; echo "bench rc=$?"; tail -1 gpurun_out/r2g_bench.json | cut -c1-300
for w in products hetero; do timeout 600 python bench.py --workload $w > gpurun_out/r2g_$w.json 2>/dev/null < /dev/null; tail -1 gpurun_out/r2g_$w.json | cut -c1-200; done
timeout 600 python bench.py --workload deepwalk --n2v --steps 5 --warmup 1 --repeats 3 > gpurun_out/r2g_deepwalk.json 2>/dev/null < /dev/null; tail -1 gpurun_out/r2g_deepwalk.json | cut -c1-200
