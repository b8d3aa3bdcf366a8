#!/bin/bash
# usage: tools/pmc_fl_quick.sh <tag> "<tuning>"  - one counter pass (L2 <-> fabric requests) over
# tools/fl_one.py; prints the per-dispatch means of the sampling kernels
tag=$1; tuning=$2
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv \
    -d gpurun_out/pmcq_${tag}_1 -o pmc -- python tools/fl_one.py --tuning "$tuning" > gpurun_out/pmcq_${tag}.log 2>&1
python tools/pmc_fl_summary.py gpurun_out "pmcq_${tag}_" | tr -d '\n '; echo
