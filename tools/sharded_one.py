"""The sharded fanout step on ONE rank (self exchanges), 3 minibatches in flight, for a
kernel trace: python tools/sharded_one.py [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, torch.distributed as dist
import euler_amd
from euler_amd.distributed import gpu_sharded_sampler, run_interleaved
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
N = 100_000_000
G = euler_amd.Graph.synthetic(euler_amd.synth_params(20240521, N, 10 * N, weighted=True), device=0,
                              partitions=1, shard_index=0, shards=1)
G.set_seed(20240521)
B = 131072
K = int(sys.argv[2]) if len(sys.argv) > 2 else 3
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
gen = torch.Generator(device=dev); gen.manual_seed(1234)
roots = torch.randint(1, N + 1, (steps, B), generator=gen, device=dev, dtype=torch.int64)
samplers = [gpu_sharded_sampler(G, partitions=1) for _ in range(K)]
streams = [torch.cuda.Stream(device=dev) for _ in range(K)]
et, FANOUT = [[0], [0]], [25, 10]


def run(first, last):
    def make(j):
        return samplers[j % K].sample_fanout_steps(roots[first + j], et, FANOUT, N + 1,
                                                   call_id=2 * (first + j))
    run_interleaved(make, last - first, K, enter=lambda k: torch.cuda.stream(streams[k]),
                    on_result=lambda job, value: None)
    for s in streams:
        s.synchronize()


torch.cuda.synchronize()
for _ in range(3):
    run(0, steps)
torch.cuda.synchronize()
print("MARK", flush=True)
t0 = time.perf_counter()
run(0, steps)
torch.cuda.synchronize()
print("ms_per_step", (time.perf_counter() - t0) / steps * 1e3)
dist.destroy_process_group()
