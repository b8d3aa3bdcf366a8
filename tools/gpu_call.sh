cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out; R=$PWD
for cfg in "2 0 100000" "2 0 1000"; do timeout 120 python tools/n2v_one.py $cfg 2>&1 < /dev/null | grep stats; done
