#!/bin/bash
# usage (on the GPU box through gpurun): tools/r5_profile.sh <tag> <what...>
#   stats:<name>   rocprofv3 --kernel-trace --stats of `tools/r5_one.py <name...>`  -> gpurun_out/<tag>_<name>_trace/
#   pmc:<name>     two separate --pmc passes (TCC read / write) of the same command   -> gpurun_out/<tag>_<name>_pmc{1,2}/
# <name> with '+' for spaces, e.g. stats:hashed+--unweighted
tag=$1; shift
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
for job in "$@"; do
  kind=${job%%:*}; name=${job#*:}
  cmd=${name//+/ }
  safe=${name//+/_}; safe=${safe//-/}
  if [ "$kind" = "stats" ]; then
    timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_${safe}_trace -o trace \
      -- python tools/r5_one.py $cmd > gpurun_out/${tag}_${safe}_trace.log 2>&1
    echo "stats $name rc=$?"; tail -2 gpurun_out/${tag}_${safe}_trace.log
  else
    i=0
    for ctrs in "TCC_EA0_RDREQ_sum FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" \
                "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAVES"; do
      i=$((i+1))
      timeout 420 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv \
        -d gpurun_out/${tag}_${safe}_pmc_$i -o pmc -- python tools/r5_one.py $cmd > gpurun_out/${tag}_${safe}_pmc_$i.log 2>&1
      echo "pmc $name pass $i rc=$?"
    done
    python tools/pmc_fl_summary.py gpurun_out "${tag}_${safe}_pmc_" > gpurun_out/${tag}_${safe}_pmc.json
    head -c 1500 gpurun_out/${tag}_${safe}_pmc.json
  fi
done
