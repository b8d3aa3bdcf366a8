"""Does it matter that the `count` samples of one root sit in adjacent lanes?
Sample 1 neighbour for 32.8 M roots: (a) every hop-2 root repeated 10x in a row
(the access pattern of the count=10 launch), (b) the same multiset shuffled."""
import sys, json, ctypes as C
sys.path.insert(0, '.')
import torch, euler_amd
from euler_amd import _lib
L = _lib.lib()
p = euler_amd.synth_params(20240521, 100_000_000, 1_000_000_000, weighted=True)
G = euler_amd.Graph.synthetic(p)
G.set_seed(20240521)
B = 131072
gen = torch.Generator(device='cuda'); gen.manual_seed(1234)
roots = torch.randint(1, 100_000_001, (B,), generator=gen, device='cuda')
out = G.sample_fanout(roots, [[0],[0]], [25,10], 100_000_001, call_id=0)
hop2 = out[0][1].contiguous()
rep = hop2.repeat_interleave(10).contiguous()
shuf = rep[torch.randperm(rep.numel(), device='cuda')].contiguous()
srt = torch.sort(rep)[0].contiguous()
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
et1 = (C.c_int32*1)(0)
res = {}
for variant in (1, 3):
    L.euler_gpu_set_tuning(0, variant)
    for name, r, cnt in (('hop2_c10', hop2, 10), ('rep_c1', rep, 1), ('shuf_c1', shuf, 1), ('sorted_c1', srt, 1)):
        n = r.numel()
        oid = torch.empty(n*cnt, dtype=torch.int64, device='cuda'); ow = torch.empty(n*cnt, dtype=torch.float32, device='cuda'); ot = torch.empty(n*cnt, dtype=torch.int32, device='cuda')
        ms = C.c_float(0)
        _lib.check(L.euler_gpu_time_sample_neighbor(G._h, st, 20240521, C.c_void_p(r.data_ptr()), n, et1, 1, cnt, 1, C.c_void_p(oid.data_ptr()), C.c_void_p(ow.data_ptr()), C.c_void_p(ot.data_ptr()), 10, C.byref(ms)))
        res['v%d %s' % (variant, name)] = round(ms.value, 4)
print(json.dumps(res))
