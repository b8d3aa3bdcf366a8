cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 400 python tools/ab_round2.py > gpurun_out/c8_ab.log 2>&1; echo "ab rc=$?"; grep -E "^row=|^B1024|Error|error" gpurun_out/c8_ab.log | cut -c1-400
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/c8_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/c8_pytest.log | cut -c1-200
