export TMPDIR=/tmp
export PYTHONPATH=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q -k "walk" > gpurun_out/r4_v21_pytest.txt 2>&1; tail -3 gpurun_out/r4_v21_pytest.txt
cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_walk -o walk -- python $GRAFT_REPO_ROOT/tools/walk_one.py --calls 6 > /tmp/walk_one.txt 2>&1
cd $GRAFT_REPO_ROOT; python tools/walk_gaps.py /tmp/prof_walk gpurun_out/r4_v21_walk_gaps.txt | tail -8
timeout 300 python bench.py --workload deepwalk --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r4_v21_deepwalk.json; python -c "
import json; d=json.loads(open('gpurun_out/r4_v21_deepwalk.json').read()); print(d['value'], d['ms_per_step'], d['config'].get('repeat_ms_per_step'))"
