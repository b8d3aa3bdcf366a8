cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
T=r5_v3
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/${T}_gpu_pytest.txt 2>&1
echo "pytest rc=$?"; tail -8 gpurun_out/${T}_gpu_pytest.txt
R5_ITERS=20 timeout 300 python tools/r5_one.py sharded_step 2>&1 | grep RESULT | tee gpurun_out/${T}_sharded_step_one_in_flight.txt
timeout 300 python tools/r5_one.py sharded_walk --cohorts 1 2>&1 | grep RESULT | tee gpurun_out/${T}_sharded_walk.txt
timeout 600 python bench.py --force-sharded --no-cpu-baseline > gpurun_out/${T}_sharded_metric.json 2> gpurun_out/${T}_sharded_metric.err; echo "sharded metric rc=$?"
R5_ITERS=8 bash tools/r5_profile.sh ${T} stats:hashed stats:hashed+--unweighted stats:hetero stats:sharded_step stats:sharded_walk stats:sage stats:sample_node
R5_ITERS=6 bash tools/r5_profile.sh ${T} pmc:hashed pmc:hashed+--unweighted pmc:hetero pmc:sharded_step pmc:sample_node
python - <<'P'
import json
l=[x for x in open('gpurun_out/r5_v3_sharded_metric.json').read().splitlines() if x.startswith('{')]
if l:
    d=json.loads(l[-1]); print('sharded_metric', d['value'], d['ms_per_step'], d['config']['repeat_ms_per_step'])
P
