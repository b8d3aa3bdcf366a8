cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/c14_pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/c14_pytest.log | cut -c1-400
timeout 600 python bench.py --workload deepwalk --n2v --steps 5 --warmup 1 --repeats 3 > gpurun_out/c14_deepwalk.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/c14_deepwalk.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['config']['node2vec'])"
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/c14_metric.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/c14_metric.json')); print(d['ms_per_step'], d['config']['repeat_ms_per_step'], d['config']['one_stream_ms_per_step'], d['config']['small_batch'], d['roofline']['frac'], d['roofline']['launch_ms'])"
