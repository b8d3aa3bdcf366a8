"""One-process runners of the kernels round 5 profiles - the command `rocprofv3` wraps.
  python tools/r5_one.py hashed [--unweighted]     the general-form one-kernel fanout (WB = 2 / 4, or 6 / 5)
  python tools/r5_one.py hetero                    SampleNeighborSetsKernel + its aggregation (one enqueue)
  python tools/r5_one.py sample_node               SampleNodeKernel, 32M draws over 100M-node tables
  python tools/r5_one.py sharded_walk [--cohorts K]  euler_gpu_sharded_random_walk, one rank, 1M x 40
  python tools/r5_one.py sharded_step              the sharded fanout step, one rank, ONE minibatch in flight
  python tools/r5_one.py sage                      euler_gpu_sage_blocks, 16 384 roots
  python tools/r5_one.py sage_multi [--M 64 --B 1024]   euler_gpu_sage_blocks_multi
Everything on ONE stream, a few iterations, so that kernel durations are not stretched by overlap."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import euler_amd
import bench

what = sys.argv[1]
N = int(os.environ.get("R5_NODES", 100_000_000))
SEED = bench.GRAPH_SEED
B = 131072
it = int(os.environ.get("R5_ITERS", 12))


def arg(name, default):
    return type(default)(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else default


from euler_amd import _lib
for kv in filter(None, arg("--tuning", "").split(",")):
    k_, v_ = kv.split("=")
    _lib.check(_lib.lib().euler_gpu_set_tuning(int(k_), int(v_)))


if what == "hashed":
    weighted = "--unweighted" not in sys.argv
    p = euler_amd.synth_params(SEED, N, 10 * N, n_types=2, weighted=weighted, hashed_ids=True)
    G = euler_amd.Graph.synthetic(p); G.set_seed(SEED)
    gen = torch.Generator(device="cuda"); gen.manual_seed(2468)
    roots = bench._mix64_t(torch.randint(1, N + 1, (4, B), generator=gen, device="cuda", dtype=torch.int64))
    for et in ([[0], [0]], [[0, 1], [0, 1]]):
        for i in range(3):
            G.sample_fanout(roots[i % 4], et, [25, 10], -1, call_id=2 * i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(it):
            G.sample_fanout(roots[i % 4], et, [25, 10], -1, call_id=2 * i)
        torch.cuda.synchronize()
        print("RESULT hashed T2 %s %s tuning '%s': %.4f ms per step on one stream, graph %.1f GB"
              % ("weighted" if weighted else "unweighted", et, arg("--tuning", ""),
                 (time.perf_counter() - t0) / it * 1e3, G.device_bytes / 1e9), flush=True)
elif what == "hetero":
    T, D, CNT = 8, 128, 10
    p = euler_amd.synth_params(SEED, N, 10 * N, n_types=T, weighted=True)
    G = euler_amd.Graph.synthetic(p); G.set_seed(SEED)
    feat = torch.randn(N + 2, D, device="cuda", generator=torch.Generator("cuda").manual_seed(7))
    gen = torch.Generator(device="cuda"); gen.manual_seed(1234)
    roots = torch.randint(1, N + 1, (4, B), generator=gen, device="cuda", dtype=torch.int64)
    sets = ([3], [1, 4, 6], list(range(T)))
    for i in range(it):
        G.sample_neighbor_sets(roots[i % 4], sets, CNT, N + 1, call_id=3 * i, feat=feat)
    torch.cuda.synchronize()
elif what == "sample_node":
    p = euler_amd.synth_params(SEED, N, 10 * N, weighted=True)
    G = euler_amd.Graph.synthetic(p); G.set_seed(SEED)
    ids = np.arange(1, N + 1, dtype=np.uint64)
    types = (((ids * np.uint64(0x9E3779B97F4A7C15)) >> np.uint64(61)) & np.uint64(3)).astype(np.int32)
    weights = (0.5 + 4.0 * np.random.default_rng(11).random(N, dtype=np.float32)).astype(np.float32)
    G.set_node_sampler(None, types, weights, 4)
    for i in range(it):
        G.sample_node(32 * 1024 * 1024, -1 if i % 2 == 0 else 1, call_id=i)
    torch.cuda.synchronize()
elif what in ("sharded_walk", "sharded_step"):
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    from euler_amd.distributed import gpu_sharded_sampler, c_sharded_random_walk
    p = euler_amd.synth_params(SEED, N, 10 * N, weighted=True)
    G = euler_amd.Graph.synthetic(p, device=0, partitions=1, shard_index=0, shards=1); G.set_seed(SEED)
    S = gpu_sharded_sampler(G, partitions=1)
    gen = torch.Generator(device=dev); gen.manual_seed(1234)
    if what == "sharded_walk":
        W, L = (1_000_000 if N >= 100_000_000 else max(1000, N // 100)), 40
        K = arg("--cohorts", 2)
        starts = torch.randint(1, N + 1, (4, W), generator=gen, device=dev, dtype=torch.int64)
        et = [[0]] * L
        for i in range(2):
            c_sharded_random_walk(G, S.c_transport, starts[i], et, N + 1, 40 * i, 1, K, S.dense_table)
        torch.cuda.synchronize()
        print("MARK", flush=True)
        t0 = time.perf_counter()
        for i in range(it):
            out, stats = c_sharded_random_walk(G, S.c_transport, starts[i % 4], et, N + 1, 40 * i, 1, K,
                                               S.dense_table, return_stats=True)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / it * 1e3
        G.set_seed(SEED)
        t0 = time.perf_counter()
        for i in range(it):
            ref = G.random_walk(starts[i % 4], et, 1.0, 1.0, N + 1, call_id=40 * i)
        torch.cuda.synchronize()
        ms_u = (time.perf_counter() - t0) / it * 1e3
        assert torch.equal(out, ref)
        print("RESULT sharded walk, one rank, %d walkers x %d, %d cohorts: %.3f ms per walk (unsharded %.3f ms) %s"
              % (W, L, K, ms, ms_u, stats))
    else:
        roots = torch.randint(1, N + 1, (it, B), generator=gen, device=dev, dtype=torch.int64)
        for i in range(3):
            S.sample_fanout(roots[i], [[0], [0]], [25, 10], N + 1, call_id=2 * i)
        torch.cuda.synchronize()
        print("MARK", flush=True)
        t0 = time.perf_counter()
        for i in range(it):
            S.sample_fanout(roots[i], [[0], [0]], [25, 10], N + 1, call_id=2 * i)
        torch.cuda.synchronize()
        print("RESULT sharded step, one rank, one minibatch in flight: %.4f ms per step"
              % ((time.perf_counter() - t0) / it * 1e3))
    dist.destroy_process_group()
elif what == "sage":
    from euler_amd.dataflow import SageDataFlow
    p = euler_amd.synth_params(SEED, N, 10 * N, weighted=True)
    G = euler_amd.Graph.synthetic(p); G.set_seed(SEED)
    gen = torch.Generator(device="cuda"); gen.manual_seed(77)
    r = torch.randint(1, N + 1, (16384,), generator=gen, device="cuda", dtype=torch.int64)
    for i in range(it):
        G.sage_blocks(r, [[0], [0]], [25, 10], default_node=N + 1, sync=False)
    torch.cuda.synchronize()
elif what == "sage_multi":
    # euler_gpu_sage_blocks_multi: M = 64 minibatches of 1 024 roots per enqueue
    p = euler_amd.synth_params(SEED, N, 10 * N, weighted=True)
    G = euler_amd.Graph.synthetic(p); G.set_seed(SEED)
    gen = torch.Generator(device="cuda"); gen.manual_seed(77)
    M_, B_ = arg("--M", 64), arg("--B", 1024)
    r = torch.randint(1, N + 1, (M_, B_), generator=gen, device="cuda", dtype=torch.int64)
    for i in range(3):
        G.sage_blocks_multi(r, [[0], [0]], [25, 10], default_node=N + 1, sync=False)
    torch.cuda.synchronize()
    ms = bench._events(lambda: G.sage_blocks_multi(r, [[0], [0]], [25, 10], default_node=N + 1, sync=False), it)
    print("RESULT sage_blocks_multi M=%d B=%d: %.4f ms per launch, %.2f us per minibatch" % (M_, B_, ms, ms * 1e3 / M_), flush=True)
else:
    raise SystemExit("unknown: " + what)
