#!/bin/bash
# Kernels of the library that spill registers to scratch (hipcc resource remarks), with their
# VGPR count and occupancy: a shared device function that grows can push every kernel that
# inlines it over its register budget - run after touching k1_search.h / device_fns.h.
cd "$(dirname "$0")/../euler_amd/csrc"
for f in *.hip; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result -I. -I../../include \
        -c $f -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 |
    grep "remark:" | grep "Function Name\|VGPRs:\|ScratchSize\|Occupancy \[waves" | paste - - - - |
    awk -v f=$f '{ if ($16 + 0 > 0) print f, $5, "VGPRs", $10, "scratch", $16, "occupancy", $22 }' | grep -v rocprim
done
