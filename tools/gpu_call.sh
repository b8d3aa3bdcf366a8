cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
bash tools/round2_profile.sh r2a 2>&1 | tail -60
timeout 600 python -m pytest tests -m gpu -x -q -k "mp_ or config5 or scatter" > gpurun_out/c12_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/c12_pytest.log | cut -c1-300
timeout 300 python bench.py --workload hetero > gpurun_out/c12_hetero.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/c12_hetero.json')); print(d['value'], d['ms_per_step'], d['config']['phases_ms'])"
