export PYTHONPATH=$GRAFT_REPO_ROOT
for t in 43=7 43=9 43=10 43=11; do timeout 300 python bench.py --workload deepwalk --no-cpu-baseline --tuning $t 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$t', d['value']/1e9, d['ms_per_step'], d['config'].get('repeat_ms_per_step'))"; done
