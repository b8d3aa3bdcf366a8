export PYTHONPATH=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q -k "fanout_local_dedup" > gpurun_out/r4_v26_pytest.txt 2>&1; tail -15 gpurun_out/r4_v26_pytest.txt | cut -c1-250
timeout 300 python tools/all_types_one.py 2>&1 | grep -v amdgpu.ids
