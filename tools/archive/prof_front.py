"""Wall time of the sharded front end (euler_gpu_dedup_split) on hop-1 and hop-2
shaped inputs of the metric workload; no graph needed.

  python tools/prof_front.py [shards]
  rocprofv3 --hip-trace --kernel-trace --stats -d out -- python tools/prof_front.py"""
import sys, time, json
sys.path.insert(0, '.')
import torch
from euler_amd import ops
shards = int(sys.argv[1]) if len(sys.argv) > 1 else 8
gen = torch.Generator(device='cuda'); gen.manual_seed(7)
hop1 = torch.randint(1, 100_000_001, (131072,), generator=gen, device='cuda')
pool = torch.randint(1, 100_000_001, (286_000,), generator=gen, device='cuda')
hop2 = pool[torch.randint(0, pool.numel(), (3_276_800,), generator=gen, device='cuda')]
res = {}
for name, ids in (('hop1', hop1), ('hop2', hop2)):
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(20):
            off, sid, pos = ops.dedup_split(ids, shards, shards)
        torch.cuda.synchronize()
        res.setdefault(name, []).append(round((time.perf_counter() - t0) / 20 * 1e3, 4))
    assert torch.equal(sid[pos.long()], ids)
    res[name + ' distinct'] = int(off[-1])
print(json.dumps(res))
