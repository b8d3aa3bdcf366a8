cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -x -q -k "walk or goldens or uniform or world" 2>&1 < /dev/null | tail -4
for v in "6 0" "6 8" "3 0"; do timeout 300 python tools/prof_walk.py $v 2>&1 < /dev/null | grep variant; done
