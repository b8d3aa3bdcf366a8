#!/bin/bash
# usage: tools/round2_profile.sh <tag>     (on the GPU box through gpurun)
# bench line, rocprofv3 kernel stats and PMC passes (one counter set per pass,
# --kernel-trace only) of the same single-stream command, plus the FETCH_SIZE /
# WRITE_SIZE calibration on known request shapes -> gpurun_out/<tag>_*
tag=${1:-r2}
R="${GRAFT_REPO_ROOT:-$(pwd)}"
cd "$R"; export TMPDIR=/tmp; mkdir -p gpurun_out
CMD="python $R/bench.py --steps 4 --warmup 1 --streams 1 --repeats 1 --no-cpu-baseline --no-check --no-small-batch"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_stats -o bench -- $CMD > $R/gpurun_out/${tag}_stats.log 2>&1; echo "stats rc=$?"
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_32B_sum"; do
  name=$(echo $c | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/${tag}_pmc_$name -o pmc -- $CMD > $R/gpurun_out/${tag}_pmc_$name.log 2>&1; echo "pmc $name rc=$?"
  timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/${tag}_cal_$name -o pmc -- $R/tools/ubench_fetch > $R/gpurun_out/${tag}_cal_$name.log 2>&1; echo "calib $name rc=$?"
done
cd "$R"
$R/tools/ubench_fetch > gpurun_out/${tag}_cal_asked.txt 2>&1
python tools/pmc_round2.py gpurun_out $tag > gpurun_out/${tag}_pmc_summary.json 2> gpurun_out/${tag}_pmc_summary.err; echo "summary rc=$?"; head -c 3000 gpurun_out/${tag}_pmc_summary.json
