"""bench.py as the driver runs it on a node, dry: `--gpus 8` with the ranks sharing the one
GPU of the box (--oversubscribe: gloo, host-staged exchange - RCCL refuses two ranks per
device), a small graph.  What must hold the first time the bench meets an 8-GPU node: ONE
JSON line from rank 0 with the whole-job value, the sharded K1 roofline reduced over the
ranks, the replicated-graph leg beside the sharded headline, the communicator size."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


def _run(args, timeout=900):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True,
                         text=True, timeout=timeout, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-3000:]
    return json.loads(lines[0])


def test_bench_dry_run_eight_ranks_share_one_gpu(torch_cuda):
    small = ["--nodes", "2000000", "--edges", "20000000", "--batch", "4096", "--steps", "2",
             "--warmup", "1", "--repeats", "1", "--no-cpu-baseline", "--pipeline", "2"]
    line = _run(["--gpus", "8", "--oversubscribe"] + small)
    assert line["steps"] == 2 and line["warmup"] == 1 and line["scaling"] == "weak"
    cfg = line["config"]
    assert cfg["ranks"] == 8 and "8 ranks" in cfg["transport"]
    assert line["value"] > 0 and abs(line["value"] - 8 * 4096 * 275 * 2 / (line["ms_per_step"] * 2e-3)) \
        < 1e-6 * line["value"]
    roof = line["roofline"]
    assert roof["bound"] == "hbm" and roof["frac"] > 0 and len(roof["per_rank"]) == 8
    assert cfg["replicas"]["value"] > 0 and cfg["exchanged_bytes_per_step"] > 0
    # the replicated-graph mode as the headline
    line = _run(["--gpus", "2", "--oversubscribe", "--replicas"] + small)
    assert line["config"]["ranks"] == 2 and "replicas" in line["config"]["partitioning"]
    # (the name of what the library really launched: 4 096 roots are below the one-kernel builds' floor)
    assert line["roofline"]["kernel"] in ("SampleFanoutPlainKernel", "SampleFanoutLeanKernel", "SampleFanoutLocalKernel",
                                          "hop by hop") and line["config"]["parity_checked_edges"]


def test_bench_sharded_workloads_on_one_rank(torch_cuda):
    """--force-sharded on one rank, the three workloads on a small graph: every line carries its own
    roofline (the sharded step's kernels) and, with the CPU cell on, a cpu_baseline."""
    small = ["--nodes", "2000000", "--edges", "20000000", "--steps", "2", "--warmup", "1", "--repeats", "1",
             "--force-sharded", "--no-cpu-baseline"]
    line = _run(small + ["--batch", "8192"])
    assert line["roofline"]["kernel"].startswith("SampleNeighborPivotKernel") and line["roofline"]["traffic"] is None
    line = _run(small + ["--batch", "8192", "--workload", "hetero"])
    assert line["roofline"]["frac"] > 0 and len(line["roofline"]["separate_launches"]) == 3
    assert line["roofline"]["kernel"].startswith("SampleNeighborSets")
    line = _run(small + ["--workload", "deepwalk"])
    assert line["roofline"]["frac"] > 0 and line["config"]["walk_stats"]["host_waits"] == 0   # (enqueued: key 63)
    assert line["config"]["parity_checked_steps"] == 64 * 40
