#!/bin/bash
# usage: tools/round3_profile.sh <tag>     (on the GPU box through gpurun)
# The round's profile set of the one-kernel step: rocprofv3 kernel trace + stats over a few
# metric steps, the counter passes (tools/pmc_fl.sh), and the summary bench.py reads for
# roofline.traffic (profiles/pmc_latest.json).  Everything lands in gpurun_out/<tag>_*;
# tools/round3_collect.py copies the summaries into profiles/.
tag=$1
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_trace -o trace \
   -- python tools/fl_one.py --steps 12 > gpurun_out/${tag}_trace.log 2>&1
echo "trace rc=$?"
bash tools/pmc_fl.sh ${tag} "" > gpurun_out/${tag}_pmc.log 2>&1
echo "pmc rc=$?"
for wl in deepwalk hetero; do
  timeout 280 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_${wl}_trace -o trace \
     -- python bench.py --workload ${wl} --steps 3 --warmup 1 > gpurun_out/${tag}_${wl}_trace.log 2>&1
  echo "${wl} trace rc=$?"
done
timeout 600 python bench.py ${BENCH_ARGS} > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
echo "bench rc=$?"
tail -c 600 gpurun_out/${tag}_bench.json
