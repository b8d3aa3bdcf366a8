cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out; R=$PWD
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/dw_stats -o dw -- python $R/bench.py --workload deepwalk --n2v --steps 3 --warmup 1 --repeats 1 > /dev/null 2>&1 < /dev/null; echo "deepwalk rc=$?"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/he_stats -o he -- python $R/bench.py --workload hetero --steps 6 --warmup 2 --repeats 1 > /dev/null 2>&1 < /dev/null; echo "hetero rc=$?"
cd $R
for d in dw he; do f=$(find gpurun_out/${d}_stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/r2_${d}_kernel_stats.csv && head -8 gpurun_out/r2_${d}_kernel_stats.csv | cut -c1-150; done
rm -rf gpurun_out/dw_stats gpurun_out/he_stats
