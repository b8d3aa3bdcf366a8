export TMPDIR=/tmp
export PYTHONPATH=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -x -q -k "sage or flow or blocks or example" > gpurun_out/r4_v19_pytest.txt 2>&1; tail -3 gpurun_out/r4_v19_pytest.txt
python tools/sage_one.py 2>&1 | tail -2
python tools/sage_one.py 2>&1 | tail -1
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_sage -o sage -- python $GRAFT_REPO_ROOT/tools/sage_one.py > $GRAFT_REPO_ROOT/gpurun_out/r4_v19_sage_one.txt 2>&1
cd $GRAFT_REPO_ROOT; f=$(find /tmp/prof_sage -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r4_v19_sage_blocks_kernel_stats.csv; head -14 gpurun_out/r4_v19_sage_blocks_kernel_stats.csv | cut -c1-160
