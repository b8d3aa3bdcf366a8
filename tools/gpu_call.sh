cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
for st in 2 3 1; do timeout 300 python bench.py --workload hetero --streams $st 2>gpurun_out/r2h.err < /dev/null | tail -1 > gpurun_out/r2h_hetero_s$st.json; python -c "
import json; d=json.loads(open('gpurun_out/r2h_hetero_s$st.json').read()); print($st, round(d['value']/1e9,2), round(d['ms_per_step'],3), d['config'].get('one_stream_ms_per_step'))"; done; tail -2 gpurun_out/r2h.err
