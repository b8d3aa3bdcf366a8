cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/c16_pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/c16_pytest.log | cut -c1-600
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/c16_bench.json 2>gpurun_out/c16_bench.err; python -c "
import json; d=json.load(open('gpurun_out/c16_bench.json')); print(d['ms_per_step'], d['config']['repeat_ms_per_step'], d['config']['one_stream_ms_per_step'], d['config']['small_batch'], d['roofline']['frac'], d['roofline']['launch_ms'])"
timeout 300 python bench.py --gpus 2 --oversubscribe --steps 4 --warmup 1 --repeats 2 --batch 16384 --nodes 2000000 --edges 20000000 > gpurun_out/c16_gpus2.json 2> gpurun_out/c16_gpus2.err; echo "gpus2 rc=$?"; cut -c1-1500 gpurun_out/c16_gpus2.json; tail -5 gpurun_out/c16_gpus2.err | cut -c1-300
timeout 200 python bench.py --force-sharded --steps 10 --warmup 3 --repeats 3 --no-cpu-baseline --no-check --no-small-batch > gpurun_out/c16_sharded1.json 2> gpurun_out/c16_sharded1.err; echo "sharded1 rc=$?"; python -c "
import json; d=json.load(open('gpurun_out/c16_sharded1.json')); print(d['ms_per_step'], d['config']['transport'], d['config']['exchanged_bytes_per_step'])"
