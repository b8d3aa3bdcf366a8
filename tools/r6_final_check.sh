# round 6, final tree: the whole -m gpu suite, the driver's bench command, the one-rank sharded lines (metric, hetero, DeepWalk + node2vec)
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
T=${1:-r6_final6}
timeout 1500 python -m pytest tests -m gpu --maxfail=5 -q > gpurun_out/${T}_gpu_pytest.txt 2>&1
echo "pytest rc=$?"; grep -a "passed\|failed" gpurun_out/${T}_gpu_pytest.txt | tail -2
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
echo "bench rc=$?"
for W in metric hetero; do
  timeout 600 python bench.py --force-sharded --workload $W > gpurun_out/${T}_sharded_1rank_$W.json 2> gpurun_out/${T}_sharded_1rank_$W.err
  echo "sharded $W rc=$?"
done
timeout 600 python bench.py --force-sharded --workload deepwalk --n2v > gpurun_out/${T}_sharded_1rank_deepwalk.json 2> gpurun_out/${T}_sharded_1rank_deepwalk.err
echo "sharded deepwalk rc=$?"
python - "$T" <<'P'
import json,sys
T=sys.argv[1]
def last(p):
    l=[x for x in open(p).read().splitlines() if x.startswith('{')]
    return json.loads(l[-1])
d=last('gpurun_out/%s_bench.json'%T); print('bench', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_ms'])
for W in ('metric','hetero','deepwalk'):
    d=last('gpurun_out/%s_sharded_1rank_%s.json'%(T,W)); print('sharded', W, d['value'], d['ms_per_step'], d.get('roofline',{}).get('frac'))
print(json.dumps(d['config'].get('node2vec'))[:600])
P
