#!/bin/bash
# usage: tools/round4_profile.sh <tag>     (on the GPU box through gpurun; PMC=1 adds the counter passes)
# Round 4's profile set: (1) kernel trace + stats of the two-stream headline loop (do two launches of
# the step's kernel overlap?), (2) kernel stats of one-stream steps, (3) the bench line, optionally
# (4) the counter passes of tools/pmc_fl.sh.  tools/round4_collect.py copies the summaries into profiles/.
tag=$1
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_2s -o trace \
   -- python tools/two_stream_trace.py --steps 24 --streams ${STREAMS:-3} > gpurun_out/${tag}_2s.log 2>&1
echo "two-stream trace rc=$?"; tail -2 gpurun_out/${tag}_2s.log
python tools/trace_overlap.py gpurun_out/${tag}_2s gpurun_out/${tag}_two_stream_trace.csv > gpurun_out/${tag}_two_stream_summary.json
cat gpurun_out/${tag}_two_stream_summary.json
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_trace -o trace \
   -- python tools/fl_one.py --steps 12 > gpurun_out/${tag}_trace.log 2>&1
echo "trace rc=$?"
if [ "$PMC" = "1" ]; then
  bash tools/pmc_fl.sh ${tag} "" > gpurun_out/${tag}_pmc.log 2>&1
  echo "pmc rc=$?"
fi
timeout 900 python bench.py ${BENCH_ARGS} > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
echo "bench rc=$?"
tail -c 1500 gpurun_out/${tag}_bench.json
