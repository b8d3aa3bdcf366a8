cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 300 python bench.py --workload hetero 2>gpurun_out/r2h.err < /dev/null | tail -1 > gpurun_out/r2h_hetero.json; tail -3 gpurun_out/r2h.err; python -c "
import json; d=json.loads(open('gpurun_out/r2h_hetero.json').read()); print(round(d['value']/1e9,2), round(d['ms_per_step'],3), d['config']['phases_ms']); print(json.dumps(d['roofline'])[:1500])"
