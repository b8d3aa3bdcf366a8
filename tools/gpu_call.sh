cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
bash tools/round2_profile.sh r2f 2>&1 | grep -E "rc="
timeout 1500 python bench.py --steps 20 --warmup 5 --cpu-protocol full > gpurun_out/r2f_bench_fullcpu.json 2> gpurun_out/r2f_bench_fullcpu.err; echo "bench full rc=$?"; tail -1 gpurun_out/r2f_bench_fullcpu.json | python -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d['cpu_baseline']; print(d['value'], d['ms_per_step']); print(c['sample']); print({B:(c[B]['cpu']['as_shipped']['edges_per_s'], c[B]['cpu']['as_shipped']['rounds'], c[B]['cpu']['best']['edges_per_s'], c[B]['cpu']['best']['what'], c[B]['gpu_same_graph']['edges_per_s']) for B in ('B1024','B131072')})"
for w in products hetero; do timeout 600 python bench.py --workload $w > gpurun_out/r2f_$w.json 2>/dev/null; tail -1 gpurun_out/r2f_$w.json | cut -c1-400; done
timeout 600 python bench.py --workload deepwalk --n2v --steps 5 --warmup 1 --repeats 3 > gpurun_out/r2f_deepwalk.json 2>/dev/null; tail -1 gpurun_out/r2f_deepwalk.json | cut -c1-300
