"""Repeatability of the one-rank sharded step and what the caching allocator does
meanwhile (device allocations during the timed region).

  python tools/diag_sharded.py [in_flight] [repeats]"""
import os, sys, time, json
sys.path.insert(0, '.')
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29544")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
import euler_amd
from euler_amd.distributed import gpu_sharded_sampler, run_interleaved
K = int(sys.argv[1]) if len(sys.argv) > 1 else 2
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
N = 100_000_000
G = euler_amd.Graph.synthetic(euler_amd.synth_params(20240521, N, 1_000_000_000, weighted=True))
G.set_seed(20240521)
B, steps = 131072, 20
fresh = bool(os.environ.get('DIAG_FRESH'))      # new roots every repetition (like bench.py)
gen = torch.Generator(device=dev); gen.manual_seed(1234)
roots = torch.randint(1, N + 1, (steps * (reps if fresh else 1), B), generator=gen, device=dev)
samplers = [gpu_sharded_sampler(G, partitions=1) for _ in range(K)]
streams = [torch.cuda.Stream(device=dev) for _ in range(K)]
torch.cuda.synchronize()
import gc
if os.environ.get('DIAG_NOGC'):
    gc.collect(); gc.freeze(); gc.disable()
out = []
for rep in range(reps):
    if os.environ.get('DIAG_BARRIER'):
        dist.barrier(); torch.cuda.synchronize()
    s0 = torch.cuda.memory_stats()
    t0 = time.perf_counter()
    base = rep * steps if fresh else 0
    run_interleaved(lambda j: samplers[j % K].sample_fanout_steps(roots[base + j], [[0], [0]], [25, 10],
                                                                  N + 1, call_id=2 * j),
                    steps, K, enter=lambda k: torch.cuda.stream(streams[k]))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps * 1e3
    s1 = torch.cuda.memory_stats()
    out.append({"ms_per_step": round(dt, 3),
                "device_allocs": s1["num_device_alloc"] - s0["num_device_alloc"],
                "device_frees": s1["num_device_free"] - s0["num_device_free"],
                "reserved_GB": round(s1["reserved_bytes.all.current"] / 1e9, 2)})
print(json.dumps(out))
dist.destroy_process_group()
