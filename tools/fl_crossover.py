"""Where does the one-kernel fanout (wave per 4 roots) overtake the workgroup-per-root kernel
(key 23) / the hop-by-hop path?  The 2-hop fanout [25, 10] on the metric graph, timed from C
(euler_gpu_time_sample_fanout) for a range of batch sizes with key 33 = 0 (one-kernel always)
and key 33 = 2^30 (never).  -> gpurun_out/fl_crossover.json"""
import json, sys, ctypes as C
sys.path.insert(0, '.')
import torch, euler_amd
from euler_amd import _lib
L = _lib.lib()
N = 100_000_000
G = euler_amd.Graph.synthetic(euler_amd.synth_params(20240521, N, 10 * N, weighted=True))
G.set_seed(20240521)
gen = torch.Generator(device='cuda'); gen.manual_seed(5)
FAN = [25, 10]
cnt = (C.c_int32 * 2)(*FAN); et = (C.c_int32 * 2)(0, 0)
rows = []
for n in (1024, 2048, 4096, 8192, 16384, 32768, 65536):
    r = torch.randint(1, N + 1, (n,), generator=gen, device='cuda', dtype=torch.int64)
    o_n = [torch.empty(n * 25, dtype=torch.int64, device='cuda'), torch.empty(n * 250, dtype=torch.int64, device='cuda')]
    o_w = [torch.empty(n * 25, dtype=torch.float32, device='cuda'), torch.empty(n * 250, dtype=torch.float32, device='cuda')]
    o_t = [torch.empty(n * 25, dtype=torch.int32, device='cuda'), torch.empty(n * 250, dtype=torch.int32, device='cuda')]
    ws = torch.empty(max(int(L.euler_gpu_sample_fanout_workspace(n, cnt, 2)), 16), dtype=torch.uint8, device='cuda')
    pn = (C.c_void_p * 2)(*[t.data_ptr() for t in o_n]); pw = (C.c_void_p * 2)(*[t.data_ptr() for t in o_w])
    pt = (C.c_void_p * 2)(*[t.data_ptr() for t in o_t])
    row = {'roots': n}
    for name, v in (('one_kernel', 0), ('other', 1 << 30)):
        _lib.check(L.euler_gpu_set_tuning(33, v))
        ms = C.c_float(0)
        for it in (5, 50):
            _lib.check(L.euler_gpu_time_sample_fanout(G._h, C.c_void_p(torch.cuda.current_stream().cuda_stream), 20240521,
                                                      C.c_void_p(r.data_ptr()), n, et, 1, cnt, 2, N + 1, pn, pw, pt,
                                                      C.c_void_p(ws.data_ptr()), it, C.byref(ms)))
        row[name + '_us'] = round(ms.value * 1e3, 1)
    rows.append(row); print(json.dumps(row), flush=True)
_lib.check(L.euler_gpu_set_tuning(33, 4096))
json.dump(rows, open('gpurun_out/fl_crossover.json', 'w'), indent=1)
