export TMPDIR=/tmp
export PYTHONPATH=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -x -q -k "sage or flow or blocks or example or unique" > gpurun_out/r4_v20_pytest.txt 2>&1; tail -3 gpurun_out/r4_v20_pytest.txt
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_sage -o sage -- python $GRAFT_REPO_ROOT/tools/sage_one.py --steps 50 > $GRAFT_REPO_ROOT/gpurun_out/r4_v20_sage_one.txt 2>&1
cd $GRAFT_REPO_ROOT; f=$(find /tmp/prof_sage -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r4_v20_sage_blocks_kernel_stats.csv; grep "Flow\|SampleNeighborPivot" gpurun_out/r4_v20_sage_blocks_kernel_stats.csv | cut -d, -f1-4,6,7 | cut -c40-200
python tools/ab_count_read.py 2>&1 | grep "no read\|pageable"
