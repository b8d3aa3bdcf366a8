"""Round 6: the one-rank sharded node2vec walk step by step: walks of length 1 .. 10 (time and the kernels'
counters, euler_gpu_random_walk_stats) - the difference of two lengths is one step.
  python tools/sharded_n2v_steps.py [key=value,...]"""
import os, sys, time, ctypes as C
sys.path.insert(0, '.')
import torch, euler_amd
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
from euler_amd.distributed import gpu_sharded_sampler, c_sharded_node2vec_walk
from euler_amd import _lib
N, SEED = 100_000_000, 20240521
G = euler_amd.Graph.synthetic(euler_amd.synth_params(SEED, N, 10 * N, weighted=True), device=0, partitions=1,
                              shard_index=0, shards=1)
G.set_seed(SEED)
S = gpu_sharded_sampler(G, partitions=1)
gen = torch.Generator(device=dev); gen.manual_seed(1234)
starts = torch.randint(1, N + 1, (100_000,), generator=gen, device=dev, dtype=torch.int64)
for kv in filter(None, (sys.argv[1] if len(sys.argv) > 1 else "").split(",")):
    k, v = kv.split("="); _lib.check(_lib.lib().euler_gpu_set_tuning(int(k), int(v)))
prev_t, prev_s = 0.0, [0] * 8
for L in range(1, 11):
    et = [[0]] * L
    c_sharded_node2vec_walk(G, S.c_transport, starts, et, 0.25, 4.0, N + 1, 3, 1, S.dense_table)
    ts = []
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        c_sharded_node2vec_walk(G, S.c_transport, starts, et, 0.25, 4.0, N + 1, 3, 1, S.dense_table)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    _lib.check(_lib.lib().euler_gpu_random_walk_stats(None, 2))
    c_sharded_node2vec_walk(G, S.c_transport, starts, et, 0.25, 4.0, N + 1, 3, 1, S.dense_table)
    buf = (C.c_uint64 * 8)()
    _lib.check(_lib.lib().euler_gpu_random_walk_stats(buf, 1))
    st = list(buf)
    t = sorted(ts)[1]
    print("RESULT L=%2d  %.2f ms (+%.2f)  step stats [par steps, par entries, seq steps, seq entries, cursor moves, "
          "chain chunks, big steps, big entries]: %s" % (L, t, t - prev_t, [a - b for a, b in zip(st, prev_s)]), flush=True)
    prev_t, prev_s = t, st
