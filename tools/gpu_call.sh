cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "gather_scatter or mp_ops or mp_grad" 2>&1 < /dev/null | tail -5
for f in "" "--unfused-aggregation"; do timeout 300 python bench.py --workload hetero $f 2>/dev/null < /dev/null | tail -1 > gpurun_out/r2h_hetero$f.json; python -c "
import json; d=json.loads(open('gpurun_out/r2h_hetero$f.json').read()); print(round(d['value']/1e9,2), round(d['ms_per_step'],3), d['config']['phases_ms'], d['roofline']['frac'])"; done
