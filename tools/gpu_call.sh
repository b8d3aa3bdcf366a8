cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 300 python tools/prof_sharded.py > gpurun_out/c19_sharded.log 2>&1; echo "rc=$?"; grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Lib" gpurun_out/c19_sharded.log | tail -30
