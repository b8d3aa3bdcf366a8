"""Phases of the hop-by-hop fanout listing all edge types (hashed ids, 2 type groups): per hop
[duplicate detection, sampling, expansion] in ms (euler_gpu_time_sample_fanout_phases)."""
import sys, ctypes as C
sys.path.insert(0, '.')
import torch, euler_amd, bench
from euler_amd import _lib
L = _lib.lib()
N = 100_000_000
p = euler_amd.synth_params(bench.GRAPH_SEED, N, 10 * N, n_types=2, weighted=True, hashed_ids=True)
G = euler_amd.Graph.synthetic(p)
gen = torch.Generator(device="cuda"); gen.manual_seed(2468)
B = 131072
r = bench._mix64_t(torch.randint(1, N + 1, (B,), generator=gen, device="cuda", dtype=torch.int64)).contiguous()
FANOUT = [25, 10]
dev = r.device
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for et, k in (([0, 0], 1), ([0, 1, 0, 1], 2)):
    for tun in ((), ((27, 0),)):
        if k == 2 and tun: continue
        for key, v in tun: L.euler_gpu_set_tuning(key, v)
        layers = 2
        cnt_a = (C.c_int32 * layers)(*FANOUT)
        et_a = (C.c_int32 * len(et))(*et)
        o_n, o_w, o_t, m = [], [], [], B
        for c in FANOUT:
            m *= c
            o_n.append(torch.empty(m, dtype=torch.int64, device=dev))
            o_w.append(torch.empty(m, dtype=torch.float32, device=dev))
            o_t.append(torch.empty(m, dtype=torch.int32, device=dev))
        wsz = int(L.euler_gpu_sample_fanout_workspace(B, cnt_a, layers))
        fws = torch.empty(max(wsz, 16), dtype=torch.uint8, device=dev)
        pn = (C.c_void_p * layers)(*[t.data_ptr() for t in o_n])
        pw_ = (C.c_void_p * layers)(*[t.data_ptr() for t in o_w])
        pt = (C.c_void_p * layers)(*[t.data_ptr() for t in o_t])
        ms_all = (C.c_float * (3 * layers))()
        nu_all = (C.c_int64 * layers)()
        for it in (3, 10):
            _lib.check(L.euler_gpu_time_sample_fanout_phases(
                G._h, st, bench.GRAPH_SEED, C.c_void_p(r.data_ptr()), B, et_a, k, cnt_a,
                layers, -1, pn, pw_, pt, C.c_void_p(fws.data_ptr()), it, ms_all, nu_all))
        print("k", k, "tuning", tun, "phases ms", [round(x, 4) for x in ms_all], "unique", list(nu_all))
        for key, v in tun: L.euler_gpu_set_tuning(key, 1)
