"""Round 6: the one-rank sharded step / sharded DeepWalk by wall clock, one process, for A/B runs of
two library builds on ONE box (EULER_GPU_LIB_PATH):  python tools/sharded_ab.py [step] [walk]"""
import os, sys, time
sys.path.insert(0, '.')
import torch, euler_amd
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
from euler_amd.distributed import gpu_sharded_sampler, c_sharded_random_walk
N = int(os.environ.get("AB_N", 100_000_000))
SEED = 20240521
what = sys.argv[1:] or ["step", "walk"]
G = euler_amd.Graph.synthetic(euler_amd.synth_params(SEED, N, 10 * N, weighted=True), device=0, partitions=1,
                              shard_index=0, shards=1)
G.set_seed(SEED)
S = gpu_sharded_sampler(G, partitions=1)
gen = torch.Generator(device=dev); gen.manual_seed(1234)
if "step" in what:
    roots = torch.randint(1, N + 1, (12, 131072), generator=gen, device=dev, dtype=torch.int64)
    for i in range(3):
        S.sample_fanout(roots[i], [[0], [0]], [25, 10], N + 1, call_id=2 * i)
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(12):
            S.sample_fanout(roots[i], [[0], [0]], [25, 10], N + 1, call_id=2 * i)
        torch.cuda.synchronize()
        print("RESULT step one in flight: %.4f ms" % ((time.perf_counter() - t0) / 12 * 1e3), flush=True)
if "walk" in what:
    W, L = 1_000_000, 40
    starts = torch.randint(1, N + 1, (4, W), generator=gen, device=dev, dtype=torch.int64)
    et = [[0]] * L
    from euler_amd import _lib
    for enq, K, self_x, ch, tail, split in ((1, 1, 0, 16, 16, 10), (0, 1, 0, 16, 16, 10), (0, 2, 0, 16, 16, 10), (1, 1, 0, 16, 0, 10),
                                            (1, 1, 0, 16, 16, 0), (1, 1, 0, 64, 16, 0), (1, 1, 1, 16, 16, 10), (0, 1, 1, 16, 16, 10)):
        # (tuning key 63: the walk enqueued without host waits / a wait per step; key 52: a lone
        # rank makes the exchanges with itself)
        _lib.check(_lib.lib().euler_gpu_set_tuning(63, enq))
        _lib.check(_lib.lib().euler_gpu_set_tuning(52, self_x))
        _lib.check(_lib.lib().euler_gpu_set_tuning(64, ch))      # columns per LDS tile of the path kernel
        _lib.check(_lib.lib().euler_gpu_set_tuning(66, tail))    # first step that sends its level as it is
        _lib.check(_lib.lib().euler_gpu_set_tuning(67, split))   # level at which the path writer splits into two passes
        for i in range(2):
            out = c_sharded_random_walk(G, S.c_transport, starts[i], et, N + 1, 40 * i, 1, K, S.dense_table)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(8):
            out = c_sharded_random_walk(G, S.c_transport, starts[i % 4], et, N + 1, 40 * i, 1, K, S.dense_table)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 8 * 1e3
        ref = G.random_walk(starts[3], et, 1.0, 1.0, N + 1, call_id=40 * 7)
        print("RESULT walk 1M x 40, enqueued %d, %d cohorts, self-exchange %d, path tile %d, tail %d, split %d: %.3f ms  same as unsharded: %s"
              % (enq, K, self_x, ch, tail, split, ms, torch.equal(out, ref)), flush=True)
