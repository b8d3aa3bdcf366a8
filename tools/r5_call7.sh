cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
T=r5_final
timeout 900 python -m pytest tests -m gpu --maxfail=5 -q > gpurun_out/${T}_gpu_pytest.txt 2>&1
rc=$?
echo "pytest rc=$rc"; tail -8 gpurun_out/${T}_gpu_pytest.txt
if [ $rc -ne 0 ]; then exit 1; fi
PMC=1 BENCH_ARGS="--gpus 1 --steps 20 --warmup 5" bash tools/round4_profile.sh ${T}
python - <<'P'
import json
l=[x for x in open('gpurun_out/r5_final_bench.json').read().splitlines() if x.startswith('{')]
d=json.loads(l[-1]); print('bench', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['config']['graph_bytes_per_gpu'])
for k,v in (d['config'].get('secondary') or {}).items():
    if isinstance(v,dict): print('  ',k, v.get('value'), v.get('ms_per_step'), v.get('one_stream_ms_per_step'), v.get('roofline_frac'), v.get('graph_bytes'))
P
