cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
T=r5_v1
# 1. new / changed code first, fail fast
timeout 1500 python -m pytest tests/test_bench_gpu.py tests/test_sharded_gpu_world2.py "tests/test_layerwise_gpu.py::test_python_deepwalk_example" "tests/test_gpu_parity.py::test_sage_blocks_two_host_threads_one_stream" -x -q > gpurun_out/${T}_new_tests.txt 2>&1
echo "new tests rc=$?"; tail -15 gpurun_out/${T}_new_tests.txt
# 2. sharded walk, cohorts 1 / 2 / 4, against the unsharded walk
for k in 1 2 4; do timeout 300 python tools/r5_one.py sharded_walk --cohorts $k 2>&1 | tail -1; done | tee gpurun_out/${T}_sharded_walk.txt
R5_ITERS=20 timeout 300 python tools/r5_one.py sharded_step 2>&1 | tail -1 | tee gpurun_out/${T}_sharded_step_one_in_flight.txt
# 3. the sharded lines on one rank
timeout 600 python bench.py --force-sharded > gpurun_out/${T}_sharded_metric.json 2> gpurun_out/${T}_sharded_metric.err; echo "sharded metric rc=$?"; tail -c 600 gpurun_out/${T}_sharded_metric.json
timeout 600 python bench.py --force-sharded --workload deepwalk > gpurun_out/${T}_sharded_deepwalk.json 2> gpurun_out/${T}_sharded_deepwalk.err; echo "sharded deepwalk rc=$?"; tail -c 1500 gpurun_out/${T}_sharded_deepwalk.json
timeout 600 python bench.py --force-sharded --workload hetero > gpurun_out/${T}_sharded_hetero.json 2> gpurun_out/${T}_sharded_hetero.err; echo "sharded hetero rc=$?"; tail -c 1500 gpurun_out/${T}_sharded_hetero.json
# 4. the default line
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err; echo "bench rc=$?"; tail -c 300 gpurun_out/${T}_bench.err
python - <<'P'
import json
d=json.loads(open('gpurun_out/r5_v1_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'])
for k,v in d['config']['secondary'].items():
    print(k, json.dumps(v)[:400])
c=d['cpu_baseline']
print(json.dumps(c.get('sample_node'))[:600]); print(json.dumps(c.get('deepwalk'))[:600])
P
