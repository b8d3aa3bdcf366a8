"""Round 6: the one-rank sharded node2vec walk (euler_gpu_sharded_node2vec_walk, 100 000 walkers x 10, p = 0.25,
q = 4 on the metric graph) by wall clock for tuning settings, one process:
  python tools/sharded_n2v_ab.py [key=value,...]..."""
import os, sys, time
sys.path.insert(0, '.')
import torch, euler_amd
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29543")
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
et = [[0]] * 10
ref = G.random_walk(starts, et, 0.25, 4.0, N + 1, call_id=3)
for cfg in (sys.argv[1:] or [""]):
    for kv in filter(None, cfg.split(",")):
        k, v = kv.split("="); _lib.check(_lib.lib().euler_gpu_set_tuning(int(k), int(v)))
    out = c_sharded_node2vec_walk(G, S.c_transport, starts, et, 0.25, 4.0, N + 1, 3, 1, S.dense_table)
    res = []
    for rep in range(int(os.environ.get("AB_REPS", 3))):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = c_sharded_node2vec_walk(G, S.c_transport, starts, et, 0.25, 4.0, N + 1, 3, 1, S.dense_table)
        torch.cuda.synchronize()
        res.append(round((time.perf_counter() - t0) * 1e3, 2))
        if os.environ.get("AB_MEM"):
            print("  free GB after rep %d: %.2f (%.1f ms)" % (rep, torch.cuda.mem_get_info()[0] / 2**30, res[-1]), flush=True)
    print("RESULT cfg '%s': %s ms per walk, same as unsharded: %s" % (cfg, res, torch.equal(out, ref)), flush=True)
