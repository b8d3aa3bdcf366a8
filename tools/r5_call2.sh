cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
T=r5_v2
timeout 1800 python -m pytest tests/test_bench_gpu.py tests/test_sharded_gpu_world2.py "tests/test_layerwise_gpu.py::test_python_deepwalk_example" "tests/test_gpu_parity.py::test_sage_blocks_two_host_threads_one_stream" "tests/test_gpu_parity.py::test_synthetic_hashed_ids_two_types" "tests/test_gpu_parity.py::test_dedup_split_pack_expand" "tests/test_gpu_parity.py::test_dedup_split_dense_id_table" tests/test_gpu_parity.py::test_gpu_sharded_sampler_single_rank -q > gpurun_out/${T}_new_tests.txt 2>&1
echo "new tests rc=$?"; tail -25 gpurun_out/${T}_new_tests.txt
for k in 1 2 4; do timeout 300 python tools/r5_one.py sharded_walk --cohorts $k 2>&1 | grep RESULT; done | tee gpurun_out/${T}_sharded_walk.txt
R5_ITERS=20 timeout 300 python tools/r5_one.py sharded_step 2>&1 | grep RESULT | tee gpurun_out/${T}_sharded_step_one_in_flight.txt
for tn in "49=1" "49=0"; do
  timeout 300 python tools/r5_one.py hashed --tuning $tn 2>&1 | grep RESULT
  timeout 300 python tools/r5_one.py hashed --unweighted --tuning $tn 2>&1 | grep RESULT
done | tee gpurun_out/${T}_fat_slots_ab.txt
# kernel traces: the sharded walk (2 cohorts), the sharded step
R5_ITERS=4 bash tools/r5_profile.sh ${T} stats:sharded_walk stats:sharded_step
timeout 600 python bench.py --force-sharded --no-cpu-baseline > gpurun_out/${T}_sharded_metric.json 2> gpurun_out/${T}_sharded_metric.err; echo "sharded metric rc=$?"
timeout 600 python bench.py --force-sharded --workload deepwalk --no-cpu-baseline > gpurun_out/${T}_sharded_deepwalk.json 2> gpurun_out/${T}_sharded_deepwalk.err; echo "sharded deepwalk rc=$?"
python - <<'P'
import json
for f in ('sharded_metric','sharded_deepwalk'):
    l=[x for x in open('gpurun_out/r5_v2_%s.json'%f).read().splitlines() if x.startswith('{')]
    if l:
        d=json.loads(l[-1]); print(f, d['value'], d['ms_per_step'], d['config']['repeat_ms_per_step'])
P
