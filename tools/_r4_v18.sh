export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -x -q -k "sets or config5 or python_example or hetero" > gpurun_out/r4_v18_pytest.txt 2>&1; tail -3 gpurun_out/r4_v18_pytest.txt
timeout 300 python bench.py --workload hetero --no-cpu-baseline > gpurun_out/r4_v18_hetero.json 2> gpurun_out/r4_v18_hetero.err; tail -c 1500 gpurun_out/r4_v18_hetero.json
timeout 300 python bench.py --workload hetero --no-cpu-baseline --tuning 47=0 > gpurun_out/r4_v18_hetero_nolds.json 2>/dev/null; tail -c 600 gpurun_out/r4_v18_hetero_nolds.json
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_sage -o sage -- python $GRAFT_REPO_ROOT/tools/sage_one.py > $GRAFT_REPO_ROOT/gpurun_out/r4_v18_sage_one.txt 2>&1
cd $GRAFT_REPO_ROOT; f=$(find /tmp/prof_sage -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r4_v18_sage_blocks_kernel_stats.csv; head -12 gpurun_out/r4_v18_sage_blocks_kernel_stats.csv | cut -c1-160
tail -5 gpurun_out/r4_v18_sage_one.txt
