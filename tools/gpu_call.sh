cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -x -q -k "mp_ or config5 or scatter or world" > gpurun_out/c13_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/c13_pytest.log | cut -c1-300
timeout 300 python bench.py --workload hetero > gpurun_out/c13_hetero.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/c13_hetero.json')); print(d['value'], d['ms_per_step'], d['config']['phases_ms'], d['roofline']['scatter_mean'])"
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-check --no-small-batch --repeats 7"
for rep in 1 2; do
for cfg in "--streams 2" "--streams 2 --tuning 19=1" "--streams 2 --tuning 3=4096" "--streams 2 --tuning 19=1,3=4096" "--streams 3 --tuning 19=1,3=4096" "--streams 2 --tuning 19=1,3=2048" "--streams 2 --tuning 19=1,3=8192"; do
  echo "== $cfg"; timeout 200 $B $cfg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['repeat_ms_per_step'], d['config']['one_stream_ms_per_step'])"
done; done
