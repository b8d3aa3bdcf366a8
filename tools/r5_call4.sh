cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
T=r5_v4
timeout 2700 python -m pytest tests -m gpu --maxfail=8 -q > gpurun_out/${T}_gpu_pytest.txt 2>&1
echo "pytest rc=$?"; tail -8 gpurun_out/${T}_gpu_pytest.txt
for i in 1 2 3; do timeout 600 python -m pytest "tests/test_gpu_parity.py::test_gpu_sharded_sampler_single_rank" tests/test_gpu_parity.py::test_dedup_split_pack_expand tests/test_gpu_parity.py::test_dedup_split_dense_id_table -q 2>&1 | tail -1; done | tee gpurun_out/${T}_front_end_repeats.txt
timeout 300 python tools/r5_one.py sharded_walk --cohorts 1 2>&1 | grep RESULT | tee gpurun_out/${T}_sharded_walk.txt
R5_ITERS=20 timeout 300 python tools/r5_one.py sharded_step 2>&1 | grep RESULT | tee gpurun_out/${T}_sharded_step_one_in_flight.txt
timeout 600 python bench.py --force-sharded > gpurun_out/${T}_sharded_metric.json 2> gpurun_out/${T}_sharded_metric.err; echo "sharded metric rc=$?"
timeout 600 python bench.py --force-sharded --workload deepwalk > gpurun_out/${T}_sharded_deepwalk.json 2> gpurun_out/${T}_sharded_deepwalk.err; echo "sharded deepwalk rc=$?"
timeout 600 python bench.py --force-sharded --workload hetero > gpurun_out/${T}_sharded_hetero.json 2> gpurun_out/${T}_sharded_hetero.err; echo "sharded hetero rc=$?"
R5_ITERS=8 bash tools/r5_profile.sh ${T} stats:sharded_step stats:sharded_walk pmc:sharded_step
python - <<'P'
import json
for f in ('sharded_metric','sharded_deepwalk','sharded_hetero'):
    l=[x for x in open('gpurun_out/r5_v4_%s.json'%f).read().splitlines() if x.startswith('{')]
    if l:
        d=json.loads(l[-1]); print(f, d['value'], d['ms_per_step'], d['config']['repeat_ms_per_step'], (d.get('roofline') or {}).get('frac'), (d.get('cpu_baseline') or {}).get('value'))
P
