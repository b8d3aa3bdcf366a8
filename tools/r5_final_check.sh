# the round's last check of a tree on the GPU box: the whole -m gpu suite, then the driver's bench command
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
T=${1:-r5_last}
timeout 1500 python -m pytest tests -m gpu --maxfail=5 -q > gpurun_out/${T}_gpu_pytest.txt 2>&1
rc=$?
echo "pytest rc=$rc"; tail -6 gpurun_out/${T}_gpu_pytest.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
echo "bench rc=$?"
python - "$T" <<'P'
import json,sys
l=[x for x in open('gpurun_out/%s_bench.json'%sys.argv[1]).read().splitlines() if x.startswith('{')]
d=json.loads(l[-1]); print('bench', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['config']['graph_bytes_per_gpu'])
for k,v in (d['config'].get('secondary') or {}).items():
    if isinstance(v,dict): print('  ',k, v.get('value'), v.get('ms_per_step'), v.get('one_stream_ms_per_step'), v.get('roofline_frac'))
print(json.dumps(d['config']['secondary']['sage_blocks'].get('small_batch')))
P
