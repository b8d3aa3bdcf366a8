#!/bin/bash
# usage: tools/pmc_fl.sh <tag> "<tuning>"   -> gpurun_out/pmc_<tag>_*/ + gpurun_out/pmc_<tag>.json
# Separate rocprofv3 --pmc passes (SQ instruction mix, SQ cycles, TCC read, TCC write) over
# tools/fl_one.py, then the per-dispatch means of the sampling kernels.
tag=$1; tuning=$2
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
i=0
for ctrs in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_FLAT" \
            "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
            "TCC_EA0_RDREQ_sum FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_WAVES"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv \
    -d gpurun_out/pmc_${tag}_$i -o pmc -- python tools/fl_one.py --tuning "$tuning" > gpurun_out/pmc_${tag}_$i.log 2>&1
  echo "pass $i rc=$?"
done
python tools/pmc_fl_summary.py gpurun_out "pmc_${tag}_" > gpurun_out/pmc_${tag}.json
cat gpurun_out/pmc_${tag}.json
