"""Round 6: A/B of launch geometries / builds of the metric step's kernel on ONE box.
Per configuration (comma-separated `key=value` tuning pairs): the kernel alone (HIP events
around 20 launches through euler_gpu_time_sample_fanout), the (unique rows, index) form
(no 524 MB expansion: the read side by itself) by HIP events, and the three-stream loop;
outputs compared bit for bit with the first configuration's.
  python tools/r6_sweep.py --configs "" 35=6,29=32 ... [--nodes N --edges E] [--lib PATH]"""
import argparse, ctypes as C, json, os, sys, time
sys.path.insert(0, '.')
ap = argparse.ArgumentParser()
ap.add_argument('--configs', nargs='*', default=[''])
ap.add_argument('--nodes', type=int, default=100_000_000)
ap.add_argument('--edges', type=int, default=1_000_000_000)
ap.add_argument('--batch', type=int, default=131072)
ap.add_argument('--reps', type=int, default=3)
ap.add_argument('--no-unique', action='store_true')
ap.add_argument('--no-streams', action='store_true')
a = ap.parse_args()
import torch, euler_amd
from euler_amd import _lib
L = _lib.lib()
N, B = a.nodes, a.batch
G = euler_amd.Graph.synthetic(euler_amd.synth_params(20240521, N, a.edges, weighted=True))
G.set_seed(20240521)
gen = torch.Generator(device='cuda'); gen.manual_seed(1234)
roots = torch.randint(1, N + 1, (24, B), generator=gen, device='cuda')
FAN = [25, 10]
cnt = (C.c_int32 * 2)(*FAN); et = (C.c_int32 * 2)(0, 0)
o_n, o_w, o_t, m = [], [], [], B
for c in FAN:
    m *= c
    o_n.append(torch.empty(m, dtype=torch.int64, device='cuda'))
    o_w.append(torch.empty(m, dtype=torch.float32, device='cuda'))
    o_t.append(torch.empty(m, dtype=torch.int32, device='cuda'))
ws = torch.empty(max(int(L.euler_gpu_sample_fanout_workspace(B, cnt, 2)), 16), dtype=torch.uint8, device='cuda')
pn = (C.c_void_p * 2)(*[t.data_ptr() for t in o_n]); pw = (C.c_void_p * 2)(*[t.data_ptr() for t in o_w])
pt = (C.c_void_p * 2)(*[t.data_ptr() for t in o_t])
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
side = [torch.cuda.Stream() for _ in range(3)]


def events(fn, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for i in range(n):
        fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


ref = None
out = {}
for cfg in a.configs:
    for kv in filter(None, cfg.split(',')):
        k, v = kv.split('=')
        _lib.check(L.euler_gpu_set_tuning(int(k), int(v)))
    ms = C.c_float(0)
    alone = []
    for rep in range(a.reps):
        _lib.check(L.euler_gpu_time_sample_fanout(G._h, st, 20240521, C.c_void_p(roots[0].data_ptr()), B, et, 1, cnt, 2,
                                                  N + 1, pn, pw, pt, C.c_void_p(ws.data_ptr()), 20, C.byref(ms)))
        alone.append(round(ms.value, 4))
    got = (o_n[0].clone(), o_n[1].clone(), o_w[1].clone(), o_w[0].clone())
    if ref is None:
        ref = got
    same = all(torch.equal(x, y) for x, y in zip(ref, got))
    res = {'alone_ms': alone, 'same_as_first': same}
    if not a.no_unique:
        try:
            G.sample_fanout_unique(roots[0], [[0], [0]], FAN, N + 1, call_id=0)
            res['unique_rows_ms'] = [round(events(lambda i: G.sample_fanout_unique(roots[i % 24], [[0], [0]], FAN, N + 1,
                                                                                      call_id=2 * i), 12), 4)
                                     for _ in range(a.reps)]
        except Exception as e:      # (a geometry the form does not take)
            res['unique_rows_ms'] = str(e)[:80]
    if not a.no_streams:
        three = []
        for rep in range(a.reps):
            torch.cuda.synchronize(); t1 = time.perf_counter()
            for i in range(48):
                with torch.cuda.stream(side[i % 3]):
                    G.sample_fanout(roots[i % 24], [[0], [0]], FAN, N + 1, call_id=2 * i)
            torch.cuda.synchronize(); three.append(round((time.perf_counter() - t1) / 48 * 1e3, 4))
        res['three_stream_ms_per_step'] = three
    out[cfg] = res
    print('CFG', repr(cfg), res, flush=True)
    # back to the defaults the next configuration starts from
    for kv in filter(None, cfg.split(',')):
        k, v = kv.split('=')
        dflt = {'28': 0, '29': 0, '30': 0, '31': 1, '32': -1, '35': 5, '45': 1, '53': 1, '54': 0, '55': 5, '57': 0}.get(k)
        if dflt is not None:
            L.euler_gpu_set_tuning(int(k), dflt)
print(json.dumps({'graph_bytes': G.device_bytes, 'results': out}))
