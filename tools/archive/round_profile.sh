#!/bin/bash
# usage: tools/round_profile.sh <tag>     (run on the GPU box through gpurun)
# bench line + rocprofv3 kernel stats + PMC passes (one counter set per pass,
# --kernel-trace only) of the same command -> gpurun_out/<tag>_*
tag=${1:-r1}
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
echo "bench rc=$?"; tail -c 400 gpurun_out/${tag}_bench.json
CMD="python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-check"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_stats -o bench -- $CMD > gpurun_out/${tag}_stats.log 2>&1
echo "stats rc=$?"
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
  name=$(echo $c | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/${tag}_pmc_$name -o pmc -- $CMD > gpurun_out/${tag}_pmc_$name.log 2>&1
  echo "pmc $name rc=$?"
done
# calibration of FETCH_SIZE on a known pattern: tools/ubench_tcp, 4 GiB working set
if [ -x tools/ubench_tcp ]; then
  timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/${tag}_pmc_calib -o pmc -- tools/ubench_tcp > gpurun_out/${tag}_pmc_calib.log 2>&1
  echo "calib rc=$?"
fi
