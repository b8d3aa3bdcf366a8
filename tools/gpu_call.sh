cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 300 python tools/prof_row.py > gpurun_out/c6_row.log 2>&1; echo "row rc=$?"; tail -2 gpurun_out/c6_row.log
R=$GRAFT_REPO_ROOT; cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/c6_rowstats -o row -- python $R/tools/prof_row.py > $R/gpurun_out/c6_rowstats.log 2>&1; echo "rocprof rc=$?"; cd $R
grep -E "SampleNeighbor" gpurun_out/c6_rowstats/row_kernel_stats.csv | cut -c1-200
timeout 400 python tools/ab_round2.py > gpurun_out/c6_ab.log 2>&1; echo "ab rc=$?"; grep -E "^row=|^B1024" gpurun_out/c6_ab.log
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/c6_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/c6_pytest.log | cut -c1-200
