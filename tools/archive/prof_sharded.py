"""Where a sharded fanout step spends its time on ONE rank (world 1, RCCL
all-to-all with itself), phase by phase with a device sync after each phase.

  python tools/prof_sharded.py"""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
import euler_amd
from euler_amd.distributed import gpu_sharded_sampler
N = 100_000_000
p = euler_amd.synth_params(20240521, N, 1_000_000_000, weighted=True)
G = euler_amd.Graph.synthetic(p)
G.set_seed(20240521)
S = gpu_sharded_sampler(G, partitions=1, group=None)
B = 131072
gen = torch.Generator(device='cuda'); gen.manual_seed(1234)
roots = torch.randint(1, N + 1, (B,), generator=gen, device='cuda')
acc = {}


def tick(name, t0):
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    acc[name] = acc.get(name, 0.0) + (t1 - t0) * 1e3
    return t1


def hop(tag, r, count, call_id, mask, group):
    n = r.numel()
    t = time.perf_counter()
    shard_off, shard_ids, pos = S.dedup_split_fn(r, S.partitions, S.world, mask, group)
    t = tick(tag + ' dedup_split', t)
    send_counts = [int(shard_off[s + 1] - shard_off[s]) for s in range(S.world)]
    sc = torch.tensor(send_counts, dtype=torch.int64, device=r.device)
    rc = torch.empty_like(sc)
    dist.all_to_all_single(rc, sc)
    recv_counts = [int(x) for x in rc.tolist()]
    t = tick(tag + ' counts exchange', t)
    owned = S._exchange(shard_ids, send_counts, recv_counts)
    t = tick(tag + ' ids exchange', t)
    ids, w, ty, m = S.local_sample(owned, [0], count, N + 1, call_id)
    t = tick(tag + ' local sample', t)
    packed = S.pack_fn(ids, w, ty, m, count, 0)
    t = tick(tag + ' pack', t)
    back = S._exchange(packed, recv_counts, send_counts)
    t = tick(tag + ' rows exchange', t)
    out = S.expand_fn(pos, back, count, 0)
    t = tick(tag + ' expand_packed', t)
    return out


iters = 10
for it in range(iters + 2):
    if it == 2:
        acc.clear()
    torch.cuda.synchronize()
    a = hop('hop1', roots, 25, 2 * it, None, 1)
    b = hop('hop2', a[0].reshape(-1), 10, 2 * it + 1, a[3], 25)
res = {k: round(v / iters, 4) for k, v in acc.items()}
res['sum'] = round(sum(res.values()), 4)
# the same step without the per-phase syncs, and the unsharded fanout
torch.cuda.synchronize(); t0 = time.perf_counter()
for it in range(iters):
    S.sample_fanout(roots, [[0], [0]], [25, 10], N + 1, call_id=2 * it)
torch.cuda.synchronize()
res['sharded step (no phase syncs)'] = round((time.perf_counter() - t0) / iters * 1e3, 4)
t0 = time.perf_counter()
for it in range(iters):
    G.sample_fanout(roots, [[0], [0]], [25, 10], N + 1, call_id=2 * it)
torch.cuda.synchronize()
res['local fanout'] = round((time.perf_counter() - t0) / iters * 1e3, 4)
print(json.dumps(res, indent=1))
dist.destroy_process_group()
