cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
T=r5_v5
timeout 900 python -m pytest tests -m gpu --maxfail=8 -q > gpurun_out/${T}_gpu_pytest.txt 2>&1
echo "pytest rc=$?"; tail -8 gpurun_out/${T}_gpu_pytest.txt
timeout 600 python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err; echo "bench rc=$?"
timeout 600 python bench.py --force-sharded > gpurun_out/${T}_sharded_metric.json 2> gpurun_out/${T}_sharded_metric.err; echo "sharded metric rc=$?"
timeout 600 python bench.py --force-sharded --workload deepwalk > gpurun_out/${T}_sharded_deepwalk.json 2> gpurun_out/${T}_sharded_deepwalk.err; echo "sharded deepwalk rc=$?"
timeout 600 python bench.py --force-sharded --workload hetero > gpurun_out/${T}_sharded_hetero.json 2> gpurun_out/${T}_sharded_hetero.err; echo "sharded hetero rc=$?"
python - <<'P'
import json
for f in ('bench','sharded_metric','sharded_deepwalk','sharded_hetero'):
    l=[x for x in open('gpurun_out/r5_v5_%s.json'%f).read().splitlines() if x.startswith('{')]
    if l:
        d=json.loads(l[-1]); print(f, d['value'], d['ms_per_step'], d['config'].get('repeat_ms_per_step'), (d.get('roofline') or {}).get('frac'), (d.get('roofline') or {}).get('avg_launch_ms'), (d.get('cpu_baseline') or {}).get('value'))
        if f=='bench':
            for k,v in (d.get('secondary') or {}).items():
                if isinstance(v,dict): print('  ',k, v.get('value'), v.get('ms_per_step'), v.get('one_stream_ms_per_step'), v.get('roofline_frac'))
P
