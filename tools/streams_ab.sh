cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "inflate_idx or sparse_gather" 2>&1 | tail -3
for S in 3 4 2 5 6 3; do
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --streams $S --no-cpu-baseline --no-secondary --no-small-batch --no-check 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('streams', $S, 'ms/step', round(d['ms_per_step'],4), 'sustained', d.get('sustained',{}).get('ms_per_step'), 'rep', d['repeat_ms_per_step'])
"
done 2>&1 | tee gpurun_out/r6_streams_ab.txt
