cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/c4_pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/c4_pytest.log
timeout 300 python tools/prof_row.py > gpurun_out/c4_row.log 2>&1; echo "row rc=$?"; tail -3 gpurun_out/c4_row.log
cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/c4_rowstats -o row -- python $GRAFT_REPO_ROOT/tools/prof_row.py > $GRAFT_REPO_ROOT/gpurun_out/c4_rowstats.log 2>&1; echo "rocprof rc=$?"; cd $GRAFT_REPO_ROOT
find gpurun_out/c4_rowstats -name "*kernel_stats.csv" | head -1 | xargs -I{} head -12 {}
