cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/c11_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/c11_pytest.log | cut -c1-300
for w in products hetero deepwalk; do
  extra=""; [ $w = deepwalk ] && extra="--n2v --steps 5 --warmup 1 --repeats 3"
  timeout 600 python bench.py --workload $w $extra > gpurun_out/c11_$w.json 2> gpurun_out/c11_$w.err; echo "$w rc=$?"; cut -c1-1200 gpurun_out/c11_$w.json; tail -2 gpurun_out/c11_$w.err
done
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/c11_metric.json 2> gpurun_out/c11_metric.err; echo "metric rc=$?"; python -c "
import json; d=json.load(open('gpurun_out/c11_metric.json')); print(d['ms_per_step'], d['config']['small_batch'], d['roofline']['frac'])"
