"""Where a wave of the one-kernel fanout (lean build) spends its time: s_memtime stamps at
the phase boundaries of every tile of one metric step (euler_gpu_set_debug_buffer).
  python tools/fl_phases.py [--tuning 28=4,29=32,30=64]"""
import argparse, json, sys, ctypes as C
sys.path.insert(0, '.')
import numpy as np, torch, euler_amd
from euler_amd import _lib
ap = argparse.ArgumentParser()
ap.add_argument('--tuning', default='')
ap.add_argument('--nodes', type=int, default=100_000_000)
ap.add_argument('--edges', type=int, default=1_000_000_000)
ap.add_argument('--batch', type=int, default=131072)
a = ap.parse_args()
L = _lib.lib()
gr = 4
for kv in filter(None, a.tuning.split(',')):
    k, v = kv.split('=')
    _lib.check(L.euler_gpu_set_tuning(int(k), int(v)))
    if int(k) == 28:
        gr = int(v)
p = euler_amd.synth_params(20240521, a.nodes, a.edges, weighted=True)
G = euler_amd.Graph.synthetic(p)
G.set_seed(20240521)
gen = torch.Generator(device='cuda'); gen.manual_seed(1234)
roots = torch.randint(1, a.nodes + 1, (4, a.batch), generator=gen, device='cuda')
tiles = (a.batch + gr - 1) // gr
dbg = torch.zeros(tiles * 8, dtype=torch.int64, device='cuda')
for i in range(3):
    G.sample_fanout(roots[i], [[0], [0]], [25, 10], a.nodes + 1, call_id=2 * i)
torch.cuda.synchronize()
_lib.check(L.euler_gpu_set_debug_buffer(C.c_void_p(dbg.data_ptr())))
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
G.sample_fanout(roots[3], [[0], [0]], [25, 10], a.nodes + 1, call_id=6)
e1.record()
torch.cuda.synchronize()
_lib.check(L.euler_gpu_set_debug_buffer(None))
d = dbg.cpu().numpy().reshape(tiles, 8).astype(np.int64)
t = d[:, :6]
t0 = t[:, 0].min()
names = ['P1 hop-1 sampling', 'P2 slots', 'P3 first chunk sampling', 'P4 + later chunks', 'hop-1 stores + drain']
dur = np.diff(t, axis=1)
slots = d[:, 6]
out = {'kernel_ms_with_stamps': round(e0.elapsed_time(e1), 4), 'tiles': int(tiles),
       'span_ticks': int(t[:, 5].max() - t0),
       'mean_ticks_per_phase': {n: round(float(dur[:, i].mean()), 1) for i, n in enumerate(names)},
       'p99_ticks_per_phase': {n: int(np.percentile(dur[:, i], 99)) for i, n in enumerate(names)},
       'wave_total_ticks': {'mean': round(float((t[:, 5] - t[:, 0]).mean()), 1),
                            'p50': int(np.percentile(t[:, 5] - t[:, 0], 50)),
                            'p99': int(np.percentile(t[:, 5] - t[:, 0], 99)),
                            'max': int((t[:, 5] - t[:, 0]).max())},
       'slots_per_tile': {'mean': round(float(slots.mean()), 2), 'p99': int(np.percentile(slots, 99)),
                          'max': int(slots.max())},
       # waves resident over time: starts / ends per decile of the kernel's span
       'start_decile_hist': np.histogram((t[:, 0] - t0) / max(1, (t[:, 5].max() - t0)), bins=10, range=(0, 1))[0].tolist(),
       'end_decile_hist': np.histogram((t[:, 5] - t0) / max(1, (t[:, 5].max() - t0)), bins=10, range=(0, 1))[0].tolist()}
# total by slot-count class
for lo_, hi_ in ((0, 8), (8, 16), (16, 32), (32, 64), (64, 1000)):
    sel = (slots >= lo_) & (slots < hi_)
    if sel.any():
        out['tiles_with_%d_to_%d_slots' % (lo_, hi_)] = {
            'n': int(sel.sum()), 'mean_total_ticks': round(float((t[sel, 5] - t[sel, 0]).mean()), 1)}
print(json.dumps(out, indent=1))
