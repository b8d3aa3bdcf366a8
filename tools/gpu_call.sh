cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-check"
for cfg in "--streams 2 --tuning 19=1" "--streams 3 --tuning 19=1" "--streams 4 --tuning 19=1" "--streams 2 --tuning 19=1,3=4096,12=4096" "--streams 3 --tuning 19=1,3=4096,12=4096" "--streams 2 --tuning 19=1,12=4096" "--streams 2 --tuning 19=1,3=4096" "--streams 2 --tuning 19=1,12=8192" "--streams 3 --tuning 19=1,12=4096" "--streams 2 --tuning 19=1,3=6144,12=6144" "--streams 2 --tuning 19=1,14=1" ; do
  echo "== $cfg"; timeout 200 $B $cfg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['repeat_ms_per_step'], d['config']['one_stream_ms_per_step'])"
done
