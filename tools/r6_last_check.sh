# round 6, last sessions: the whole -m gpu suite, the driver's bench command, then the one-rank sharded DeepWalk + node2vec line
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
T=${1:-r6_last}
timeout 1500 python -m pytest tests -m gpu --maxfail=5 -q > gpurun_out/${T}_gpu_pytest.txt 2>&1
echo "pytest rc=$?"; grep -a "passed\|failed" gpurun_out/${T}_gpu_pytest.txt | tail -2
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
echo "bench rc=$?"
timeout 600 python bench.py --force-sharded --workload deepwalk --n2v > gpurun_out/${T}_sharded_1rank_deepwalk.json 2> gpurun_out/${T}_sharded_1rank_deepwalk.err
echo "sharded deepwalk rc=$?"
python - "$T" <<'P'
import json,sys
T=sys.argv[1]
l=[x for x in open('gpurun_out/%s_bench.json'%T).read().splitlines() if x.startswith('{')]
d=json.loads(l[-1]); print('bench', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_ms'])
l=[x for x in open('gpurun_out/%s_sharded_1rank_deepwalk.json'%T).read().splitlines() if x.startswith('{')]
d=json.loads(l[-1]); print('sharded deepwalk', d['value'], d['ms_per_step']); print(json.dumps(d['config'].get('node2vec'))[:1500])
P
