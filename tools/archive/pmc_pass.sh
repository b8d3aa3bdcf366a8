#!/bin/bash
# usage: tools/pmc_pass.sh <tag> "<space separated counters>" -- <command...>
# One rocprofv3 counter pass (own run, --kernel-trace only) -> gpurun_out/pmc_<tag>/
tag=$1; ctrs=$2; shift 3
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv \
  -d gpurun_out/pmc_$tag -o pmc -- "$@" > gpurun_out/pmc_$tag.log 2>&1
echo "pmc $tag rc=$?"
