cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/c15_pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/c15_pytest.log | cut -c1-400
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/c15_bench.json 2> gpurun_out/c15_bench.err; echo "bench rc=$?"; python -c "
import json; d=json.load(open('gpurun_out/c15_bench.json')); print(d['value'], d['ms_per_step'], d['config']['repeat_ms_per_step'], d['config']['one_stream_ms_per_step'], d['config']['small_batch'], d['roofline']['frac'], d['roofline']['launch_ms']); print(d['cpu_baseline']['sample'])"
bash tools/round2_profile.sh r2b 2>&1 | grep -E "rc=" 
python - <<'PY'
import json
d=json.load(open('gpurun_out/r2b_pmc_summary.json'))
for k,e in d['kernels'].items():
    print(k[:60], {n:round(v['mean']) for n,v in e.items() if isinstance(v,dict)}, round(e.get('hbm_bytes_per_launch',0)/1e6,1))
print(d.get('pmc_latest'))
PY
