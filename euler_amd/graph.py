"""Device-resident Euler graph + the sampling / message-passing operators on it.

Host-side mirror of what the reference reaches through
`QueryProxy::RunAsyncGremlin` for the hot path (tf_euler/kernels/*.cc): every
method enqueues hand-written HIP kernels of libeuler_gpu.so on the current
torch stream and returns torch tensors living in HBM.  torch is plumbing here
(device memory, streams); the computation is in euler_amd/csrc.
"""
import contextlib
import ctypes as C
import math
import threading

import numpy as np
import torch

from . import _lib
from ._lib import check, lib

_I32 = C.POINTER(C.c_int32)


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    """The current HIP stream of the current device as a pointer (torch.cuda.current_stream()
    builds a Stream object for it: 8 us a call, a quarter of a small op's enqueue cost)."""
    if _RAW_STREAM is not None:
        return C.c_void_p(_RAW_STREAM(torch.cuda.current_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


_NULL_CTX = contextlib.nullcontext()


_I32_CACHE = {}
_FANOUT_PLANS = {}
_WS_BYTES = {}


def _fanout_plan(edge_types, counts):
    """(edge types [layers, k] as int32 array, its pointer, k, counts array, its pointer) of a
    fanout call; cached for list arguments (a training loop passes the same ones every step)."""
    key = None
    try:
        key = (tuple(tuple(int(t) for t in row) for row in edge_types), tuple(int(c) for c in counts))
        hit = _FANOUT_PLANS.get(key)
        if hit is not None:
            return hit
    except TypeError:
        key = None
    layers = len(counts)
    et = np.ascontiguousarray(np.asarray(edge_types, dtype=np.int32).reshape(layers, -1)) if layers \
        else np.zeros((0, 0), np.int32)
    cnt = np.ascontiguousarray(np.asarray(counts, dtype=np.int32).reshape(-1))
    res = (et, et.ctypes.data_as(_I32), int(et.size // layers) if layers else 0, cnt, cnt.ctypes.data_as(_I32))
    if key is not None and len(_FANOUT_PLANS) < 1024:
        et.flags.writeable = False      # shared by every later call with these arguments
        cnt.flags.writeable = False
        _FANOUT_PLANS[key] = res
    return res


def _i32_array(values):
    """(array, int32*, size) of a list of small ints (edge types, counts); flat lists are
    cached - the same few lists come back every minibatch."""
    key = None
    if type(values) in (list, tuple) and len(values) <= 32 and all(type(v) is int for v in values):
        key = tuple(values)
        hit = _I32_CACHE.get(key)
        if hit is not None:
            return hit
    a = np.ascontiguousarray(np.asarray(values, dtype=np.int32).reshape(-1))
    res = (a, a.ctypes.data_as(_I32), int(a.size))
    if key is not None and len(_I32_CACHE) < 1024:
        a.flags.writeable = False       # shared by every later call with this list
        _I32_CACHE[key] = res
    return res


def _read_counts(counts):
    """A few device-side counts as Python ints.  Tensor.cpu() is the cheapest way measured
    (profiles/r4_ab_count_read.txt: a pinned buffer + stream / event synchronize costs the
    same or more; what the read costs is the drain of the stream, ~55 us, not the copy)."""
    return [int(c) for c in counts.cpu().tolist()]


def _as_i64_cuda(x, device):
    if not torch.is_tensor(x):
        x = torch.as_tensor(np.asarray(x).astype(np.int64, copy=False))
    if x.dtype != torch.int64:
        x = x.to(torch.int64)
    return x.to(device).contiguous()


def _np(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def synth_params(seed, n_nodes, n_edges, n_types=1, weighted=True, scale=None, hashed_ids=False):
    """Parameters of the deterministic synthetic power-law graph (RMAT
    marginals a,b,c,d = .57,.19,.19,.05, minimum degree 1).
    deg_table[z] = expected extra out-degree of a node whose (id-1) has z one
    bits, normalised so the edge total is n_edges."""
    if scale is None:
        scale = max(1, int(math.ceil(math.log2(max(n_nodes, 2)))))
    # number of x in [0, n_nodes) with popcount(x & mask) == z, by digit DP
    cnt = [0] * 65
    mask_bits = scale
    hi = n_nodes >> mask_bits           # full cycles of the low `scale` bits
    lo = n_nodes & ((1 << mask_bits) - 1)
    if hi:
        for z in range(mask_bits + 1):
            cnt[z] += hi * math.comb(mask_bits, z)
    ones = 0
    for b in range(mask_bits - 1, -1, -1):
        if (lo >> b) & 1:
            for z in range(b + 1):
                cnt[ones + z] += math.comb(b, z)
            ones += 1
    p = _lib.SynthParams()
    p.seed = seed
    p.n_nodes = n_nodes
    p.n_edges_target = n_edges
    p.scale = scale
    p.n_types = n_types
    p.weighted = 1 if weighted else 0
    # hashed_ids: node x is known outside by mix64(x) (arbitrary u64 ids, hash id map) - what a
    # dataset converted by euler/tools looks like - instead of x itself (identity id map)
    p.hashed_ids = 1 if hashed_ids else 0
    wz = [(0.76 ** (scale - z)) * (0.24 ** z) for z in range(scale + 1)]
    norm = sum(c * w for c, w in zip(cnt, wz))
    extra = max(0.0, float(n_edges - n_nodes))
    for z in range(64):
        p.deg_table[z] = extra * wz[z] / norm if (z <= scale and norm > 0) else 0.0
    return p


class Graph:
    """Immutable graph in HBM (CSR + row metadata + alias tables)."""

    def __init__(self, handle, device, meta=None):
        self._h = handle
        self.device = torch.device("cuda", device)
        self._dev_index = self.device.index
        self.device_index = device
        self.seed = 0
        self._call_id = 0
        self._lock = threading.Lock()
        self.node_type_names = (meta or {}).get("node_types", {})
        self.edge_type_names = (meta or {}).get("edge_types", {})

    # ------------------------------------------------------------ creation
    @classmethod
    def from_csr(cls, row_id, row_ptr, type_end, nbr, prefix_w, type_prefix,
                 n_edge_types, node_type=None, node_weight=None,
                 sampler_order=None, device=0, partitions=1, shard_index=0,
                 shards=1, features=None, sparse_features=None):
        """features = (n_float, feat_ptr [n+1], feat_idx [n*F], feat_val): the
        reference's per-node float_features_idx_ / float_features_
        (core/graph/node.h) concatenated over rows; sparse_features = the same
        four for uint64_features_idx_ / uint64_features_."""
        row_id = _np(row_id, np.uint64)
        row_ptr = _np(row_ptr, np.int64)
        type_end = _np(type_end, np.int32).reshape(-1)
        nbr = _np(nbr, np.uint64)
        prefix_w = _np(prefix_w, np.float32)
        type_prefix = _np(type_prefix, np.float32).reshape(-1)
        c = _lib.HostCSR()
        c.n_rows = len(row_id)
        c.n_edge_types = int(n_edge_types)
        keep = [row_id, row_ptr, type_end, nbr, prefix_w, type_prefix]
        c.row_id = row_id.ctypes.data_as(_lib.u64p)
        c.row_ptr = row_ptr.ctypes.data_as(_lib.i64p)
        c.type_end = type_end.ctypes.data_as(_lib.i32p)
        c.nbr = nbr.ctypes.data_as(_lib.u64p)
        c.prefix_w = prefix_w.ctypes.data_as(_lib.f32p)
        c.type_prefix = type_prefix.ctypes.data_as(_lib.f32p)
        n_node_types = 1
        if node_type is not None:
            node_type = _np(node_type, np.int32)
            keep.append(node_type)
            c.node_type = node_type.ctypes.data_as(_lib.i32p)
            n_node_types = int(node_type.max()) + 1 if len(node_type) else 1
        if node_weight is not None:
            node_weight = _np(node_weight, np.float32)
            keep.append(node_weight)
            c.node_weight = node_weight.ctypes.data_as(_lib.f32p)
        if sampler_order is not None:
            sampler_order = _np(sampler_order, np.uint64)
            keep.append(sampler_order)
            c.sampler_order = sampler_order.ctypes.data_as(_lib.u64p)
        c.n_node_types = n_node_types
        if features is not None:
            nf, fptr, fidx, fval = features
            fptr = _np(fptr, np.int64)
            fidx = _np(fidx, np.int32).reshape(-1)
            fval = _np(fval, np.float32)
            keep += [fptr, fidx, fval]
            c.n_float_features = int(nf)
            c.feat_ptr = fptr.ctypes.data_as(_lib.i64p)
            c.feat_idx = fidx.ctypes.data_as(_lib.i32p)
            c.feat_val = fval.ctypes.data_as(_lib.f32p)
        if sparse_features is not None:
            nu, uptr, uidx, uval = sparse_features
            uptr = _np(uptr, np.int64)
            uidx = _np(uidx, np.int32).reshape(-1)
            uval = _np(uval, np.uint64)
            keep += [uptr, uidx, uval]
            c.n_u64_features = int(nu)
            c.ufeat_ptr = uptr.ctypes.data_as(_lib.i64p)
            c.ufeat_idx = uidx.ctypes.data_as(_lib.i32p)
            c.ufeat_val = uval.ctypes.data_as(_lib.u64p)
        h = C.c_void_p()
        check(lib().euler_gpu_graph_create_shard(C.byref(c), device, partitions,
                                                 shard_index, shards, C.byref(h)))
        return cls(h, device)

    @classmethod
    def synthetic(cls, params, device=0, partitions=1, shard_index=0, shards=1):
        h = C.c_void_p()
        check(lib().euler_gpu_graph_create_synthetic(
            C.byref(params), device, partitions, shard_index, shards, C.byref(h)))
        return cls(h, device)

    @classmethod
    def load(cls, data_path, device=0, shard_index=0, shards=1):
        h = C.c_void_p()
        check(lib().euler_gpu_graph_load(str(data_path).encode(), device,
                                         shard_index, shards, C.byref(h)))
        return cls(h, device)

    def close(self):
        if self._h is not None and self._h.value:
            lib().euler_gpu_graph_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -------------------------------------------------------------- facts
    @property
    def num_nodes(self):
        return lib().euler_gpu_graph_num_nodes(self._h)

    @property
    def num_edges(self):
        return lib().euler_gpu_graph_num_edges(self._h)

    @property
    def num_edge_types(self):
        return lib().euler_gpu_graph_num_edge_types(self._h)

    @property
    def num_node_types(self):
        return lib().euler_gpu_graph_num_node_types(self._h)

    @property
    def partitions(self):
        """euler.meta's partitions_num for a graph loaded from a data directory (0
        otherwise): the `partitions` a sharded sampler must route ids with."""
        return int(lib().euler_gpu_graph_partitions(self._h))

    @property
    def device_bytes(self):
        return lib().euler_gpu_graph_bytes(self._h)

    def node_weight_sums(self):
        out = np.zeros(max(self.num_node_types, 1), np.float32)
        check(lib().euler_gpu_graph_node_weight_sums(
            self._h, out.ctypes.data_as(_lib.f32p)))
        return out

    def export_rows(self, ids):
        """Rows of the device CSR for `ids` as host numpy arrays."""
        ids = _np(ids, np.uint64)
        n = len(ids)
        T = self.num_edge_types
        row_ptr = np.zeros(n + 1, np.int64)
        check(lib().euler_gpu_graph_export_rows(
            self._h, ids.ctypes.data_as(_lib.u64p), n,
            row_ptr.ctypes.data_as(_lib.i64p), None, None, None, None))
        tot = int(row_ptr[-1])
        type_end = np.zeros(n * T, np.int32)
        nbr = np.zeros(max(tot, 1), np.uint64)
        pw = np.zeros(max(tot, 1), np.float32)
        tp = np.zeros(n * T, np.float32)
        check(lib().euler_gpu_graph_export_rows(
            self._h, ids.ctypes.data_as(_lib.u64p), n,
            row_ptr.ctypes.data_as(_lib.i64p), type_end.ctypes.data_as(_lib.i32p),
            nbr.ctypes.data_as(_lib.u64p), pw.ctypes.data_as(_lib.f32p),
            tp.ctypes.data_as(_lib.f32p)))
        return row_ptr, type_end, nbr[:tot], pw[:tot], tp

    def index_overflow_rows(self, cap=4096):
        """Node ids (identity id maps only) of rows in which a bucket of the weight-bucket index
        overflows its block, as found while the index was built (at most 4 096): draws that land
        there take the fallback search.  Tests draw roots from them on purpose."""
        out = np.zeros(max(int(cap), 1), np.uint64)
        n = C.c_int64(0)
        check(lib().euler_gpu_graph_index_overflow_rows(self._h, out.ctypes.data_as(_lib.u64p), int(cap),
                                                        C.byref(n)))
        return out[:n.value]

    # ---------------------------------------------------------------- RNG

    def _on_device(self):
        """Context of the graph's device; nothing to switch (and 5 us less per op) when the
        caller already runs on it."""
        if torch.cuda.current_device() == self._dev_index:
            return _NULL_CTX
        return torch.cuda.device(self.device)

    def set_seed(self, seed, call_id=0):
        """Fix the sampling stream: ids are a pure function of (seed, call_id,
        node id / sample index, draw index)."""
        with self._lock:
            self.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
            self._call_id = int(call_id)

    def _take_call_ids(self, n, call_id=None):
        if call_id is not None:
            return int(call_id) & 0xFFFFFFFF
        with self._lock:
            c = self._call_id
            self._call_id = (self._call_id + n) & 0xFFFFFFFF
        return c

    # ------------------------------------------------------------ sampling
    def sample_neighbor(self, nodes, edge_types, count, default_node=-1,
                        layout="tf", call_id=None, return_mask=False, dedup=True):
        """tf_euler SampleNeighbor (tf_euler/kernels/sample_neighbor_op.cc):
        nodes [n] int64 -> (neighbors [n,count] int64, weights f32, types
        int32).  layout='core' gives the API_SAMPLE_NB fill (0, 0.0, 0)."""
        nodes = _as_i64_cuda(nodes, self.device)
        shape = tuple(nodes.shape) + (int(count),)
        flat = nodes.reshape(-1)
        n = flat.numel()
        out_n = torch.empty(shape, dtype=torch.int64, device=self.device)
        out_w = torch.empty(shape, dtype=torch.float32, device=self.device)
        out_t = torch.empty(shape, dtype=torch.int32, device=self.device)
        mask = (torch.empty(n, dtype=torch.uint8, device=self.device)
                if return_mask else None)
        et, et_p, k = _i32_array(edge_types)
        lay = _lib.LAYOUT_TF if layout == "tf" else _lib.LAYOUT_CORE
        with self._on_device():
            if dedup:
                check(lib().euler_gpu_sample_neighbor(
                    self._h, _stream(), self.seed, self._take_call_ids(1, call_id),
                    _ptr(flat), n, None, 1, et_p, k, int(count), lay,
                    int(default_node), _ptr(out_n), _ptr(out_w), _ptr(out_t),
                    _ptr(mask)))
            else:       # roots known to be distinct (sharded sampler)
                check(lib().euler_gpu_sample_neighbor_distinct(
                    self._h, _stream(), self.seed, self._take_call_ids(1, call_id),
                    _ptr(flat), n, et_p, k, int(count), lay, int(default_node),
                    _ptr(out_n), _ptr(out_w), _ptr(out_t), _ptr(mask)))
        res = (out_n, out_w, out_t)
        return res + (mask,) if return_mask else res

    def sample_neighbor_sets(self, nodes, type_sets, count, default_node=-1, call_id=None,
                             feat=None, aggr="mean"):
        """S SampleNeighbor ops over the same nodes - one per edge-type set, as a heterogeneous
        (RGCN-style) model issues them - in ONE launch (euler_gpu_sample_neighbor_sets): returns
        (neighbors [S, n, count] int64, weights f32, types int32); set s draws with the call id
        of the s-th of S consecutive sample_neighbor calls, and the result equals theirs bit
        for bit.  With `feat` ([rows, d] f32 indexed by node id) the sampled neighbours'
        rows are aggregated per (set, root) in the same enqueue
        (euler_gpu_sample_aggregate_sets): a fourth result [S, n, d] = add / max / mean over
        the `count` rows, the bits of scatter_(aggr, gather(feat, neighbors))."""
        flat = _as_i64_cuda(nodes, self.device).reshape(-1)
        n, S = flat.numel(), len(type_sets)
        ks = [len(ts) for ts in type_sets]
        ks_a, ks_p, _ = _i32_array(ks)
        et_a, et_p, _ = _i32_array([int(t) for ts in type_sets for t in ts])
        out_n = torch.empty((S, n, int(count)), dtype=torch.int64, device=self.device)
        out_w = torch.empty((S, n, int(count)), dtype=torch.float32, device=self.device)
        out_t = torch.empty((S, n, int(count)), dtype=torch.int32, device=self.device)
        cid = self._take_call_ids(S, call_id)
        with self._on_device():
            if feat is None:
                check(lib().euler_gpu_sample_neighbor_sets(
                    self._h, _stream(), self.seed, cid, _ptr(flat), n, et_p, ks_p, S, int(count),
                    int(default_node), _ptr(out_n), _ptr(out_w), _ptr(out_t)))
                return out_n, out_w, out_t
            assert feat.dtype == torch.float32 and feat.dim() == 2 and feat.is_contiguous()
            if not 0 <= int(default_node) < feat.shape[0]:
                raise ValueError("sample_neighbor_sets(feat=...): default_node must name a row of the "
                                 "feature table (the row the default fill aggregates, e.g. max_id + 1); "
                                 "got %d for %d rows" % (int(default_node), feat.shape[0]))
            mode = {"add": 0, "max": 1, "mean": 2}[aggr]
            agg = torch.empty((S, n, feat.shape[1]), dtype=torch.float32, device=self.device)
            check(lib().euler_gpu_sample_aggregate_sets(
                self._h, _stream(), self.seed, cid, _ptr(flat), n, et_p, ks_p, S, int(count),
                int(default_node), mode, _ptr(feat), int(feat.shape[0]), int(feat.shape[1]),
                _ptr(out_n), _ptr(out_w), _ptr(out_t), _ptr(agg)))
        return out_n, out_w, out_t, agg

    def sample_neighbor_packed(self, nodes, edge_types, count, default_node=-1,
                               call_id=None):
        """The shard side of a multi-GPU hop: TF-layout SampleNeighbor of the
        (distinct) ids this shard owns, returned as wire rows for ops.expand_packed:
        [n, 4 * count + 2] int32 (ids | weights | types | mask, pad), or
        [n, 3 * count + 2 (padded to even)] without the types when one edge type is listed."""
        flat = _as_i64_cuda(nodes, self.device).reshape(-1)
        n = flat.numel()
        et, et_p, k = _i32_array(edge_types)
        # one listed type: the type column stays off the wire
        words = ((3 if k == 1 else 4) * int(count) + 2 + 1) & ~1
        rows = torch.empty((n, words), dtype=torch.int32, device=self.device)
        with self._on_device():
            check(lib().euler_gpu_sample_neighbor_packed(
                self._h, _stream(), self.seed, self._take_call_ids(1, call_id),
                _ptr(flat), n, et_p, k, int(count), int(default_node), _ptr(rows)))
        return rows

    def sample_neighbor_sets_packed(self, nodes, type_sets, count, default_node=-1, call_id=None):
        """sample_neighbor_packed for several edge-type sets over the same (distinct) ids in ONE
        launch (euler_gpu_sample_neighbor_sets_packed): a list of wire-row tensors, one per set -
        the rows of len(type_sets) sample_neighbor_packed calls with consecutive call ids (views
        of one allocation)."""
        flat = _as_i64_cuda(nodes, self.device).reshape(-1)
        n, S = flat.numel(), len(type_sets)
        ks = [len(ts) for ts in type_sets]
        ks_a, ks_p, _ = _i32_array(ks)
        et_a, et_p, _ = _i32_array([int(t) for ts in type_sets for t in ts])
        words = [((3 if k == 1 else 4) * int(count) + 2 + 1) & ~1 for k in ks]
        buf = torch.empty((n * sum(words),), dtype=torch.int32, device=self.device)
        cid = self._take_call_ids(S, call_id)
        with self._on_device():
            check(lib().euler_gpu_sample_neighbor_sets_packed(
                self._h, _stream(), self.seed, cid, _ptr(flat), n, et_p, ks_p, S, int(count),
                int(default_node), _ptr(buf)))
        out, off = [], 0
        for w in words:
            out.append(buf[off:off + n * w].view(n, w))
            off += n * w
        return out

    def sample_fanout(self, nodes, edge_types, counts, default_node=-1,
                      call_id=None):
        """tf_euler sample_fanout (euler_ops/neighbor_ops.py:122-158 over
        tf_euler/kernels/sample_fanout_op.cc): returns (neighbors_list,
        weights_list, types_list) with flattened per-hop tensors."""
        nodes = _as_i64_cuda(nodes, self.device).reshape(-1)
        layers = len(counts)
        et, et_p, k, cnt, cnt_p = _fanout_plan(edge_types, counts)
        n = nodes.numel()
        outs_n, outs_w, outs_t = [], [], []
        m = n
        for c in counts:
            m *= int(c)
            outs_n.append(torch.empty(m, dtype=torch.int64, device=self.device))
            outs_w.append(torch.empty(m, dtype=torch.float32, device=self.device))
            outs_t.append(torch.empty(m, dtype=torch.int32, device=self.device))
        ws_key = (n,) + tuple(int(c) for c in counts)
        ws_bytes = _WS_BYTES.get(ws_key)
        if ws_bytes is None:
            ws_bytes = int(lib().euler_gpu_sample_fanout_workspace(n, cnt_p, layers))
            if len(_WS_BYTES) < 4096:
                _WS_BYTES[ws_key] = ws_bytes
        ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=self.device)
        pn = (C.c_void_p * layers)(*[t.data_ptr() for t in outs_n])
        pw = (C.c_void_p * layers)(*[t.data_ptr() for t in outs_w])
        pt = (C.c_void_p * layers)(*[t.data_ptr() for t in outs_t])
        with self._on_device():
            check(lib().euler_gpu_sample_fanout(
                self._h, _stream(), self.seed,
                self._take_call_ids(layers, call_id), _ptr(nodes), n, et_p, k,
                cnt_p, layers, int(default_node), pn, pw, pt, _ptr(ws)))
        return [nodes] + outs_n, outs_w, outs_t

    def sample_fanout_multi(self, batches, edge_types, counts, default_node=-1, call_id=None,
                            call_ids=None):
        """M minibatches of sample_fanout in ONE enqueue (euler_gpu_sample_fanout_multi):
        `batches` is an [M, B] tensor or a list of M equally long root tensors; minibatch b
        draws with call id call_id + b * len(counts) - what M consecutive sample_fanout calls
        take from the graph's counter - or call_ids[b] (a device uint32 / int32 tensor).
        Returns a list of M (neighbors_list, weights_list, types_list), each what
        sample_fanout returns for that minibatch (views of one allocation per hop), bit for
        bit."""
        if isinstance(batches, (list, tuple)):
            batches = torch.stack([_as_i64_cuda(b, self.device).reshape(-1) for b in batches], 0)
        roots = _as_i64_cuda(batches, self.device)
        assert roots.dim() == 2, "batches: [M, B]"
        roots = roots.contiguous()
        m, n = int(roots.shape[0]), int(roots.shape[1])
        layers = len(counts)
        et, et_p, k, cnt, cnt_p = _fanout_plan(edge_types, counts)
        outs_n, outs_w, outs_t, rows = [], [], [], n
        for c in counts:
            rows *= int(c)
            outs_n.append(torch.empty((m, rows), dtype=torch.int64, device=self.device))
            outs_w.append(torch.empty((m, rows), dtype=torch.float32, device=self.device))
            outs_t.append(torch.empty((m, rows), dtype=torch.int32, device=self.device))
        ws_key = (m * n,) + tuple(int(c) for c in counts)
        ws_bytes = _WS_BYTES.get(ws_key)
        if ws_bytes is None:
            ws_bytes = int(lib().euler_gpu_sample_fanout_workspace(m * n, cnt_p, layers))
            if len(_WS_BYTES) < 4096:
                _WS_BYTES[ws_key] = ws_bytes
        ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=self.device)
        pn = (C.c_void_p * layers)(*[t.data_ptr() for t in outs_n])
        pw = (C.c_void_p * layers)(*[t.data_ptr() for t in outs_w])
        pt = (C.c_void_p * layers)(*[t.data_ptr() for t in outs_t])
        ids_p = None
        if call_ids is not None:
            call_ids = call_ids.to(device=self.device, dtype=torch.int32).contiguous()
            assert call_ids.numel() == m
            ids_p = _ptr(call_ids)
        with self._on_device():
            check(lib().euler_gpu_sample_fanout_multi(
                self._h, _stream(), self.seed, self._take_call_ids(layers * m, call_id), layers,
                ids_p, m, _ptr(roots), n, et_p, k, cnt_p, layers, int(default_node), pn, pw, pt,
                _ptr(ws)))
        return [([roots[b]] + [t[b] for t in outs_n], [t[b] for t in outs_w], [t[b] for t in outs_t])
                for b in range(m)]

    def sample_fanout_unique(self, nodes, edge_types, counts, default_node=-1, call_id=None):
        """The 2-hop fanout in the (unique rows, index) form (euler_gpu_sample_fanout_unique):
        returns (hop-1 ids [n, c1], weights, types, row_index [n * c1] int64, rows_id
        [n * c1, c2], rows_w, rows_t) with hop-2 tensor = rows[row_index]; only the rows
        row_index names are written.  Raises EulerGpuError(EINVAL) for graphs / fanouts the
        one-kernel path does not serve."""
        nodes = _as_i64_cuda(nodes, self.device).reshape(-1)
        assert len(counts) == 2
        et = np.asarray(edge_types, dtype=np.int32).reshape(2, -1)
        assert et.shape[1] == 1, "one listed edge type per hop"
        et, et_p, _ = _i32_array(et)
        cnt, cnt_p, _ = _i32_array(counts)
        n, c1, c2 = nodes.numel(), int(counts[0]), int(counts[1])
        dev = self.device
        id1 = torch.empty((n, c1), dtype=torch.int64, device=dev)
        w1 = torch.empty((n, c1), dtype=torch.float32, device=dev)
        t1 = torch.empty((n, c1), dtype=torch.int32, device=dev)
        idx = torch.empty(n * c1, dtype=torch.int32, device=dev)
        rid = torch.empty((n * c1, c2), dtype=torch.int64, device=dev)
        rw = torch.empty((n * c1, c2), dtype=torch.float32, device=dev)
        rt = torch.empty((n * c1, c2), dtype=torch.int32, device=dev)
        ws_bytes = lib().euler_gpu_sample_fanout_workspace(n, cnt_p, 2)
        ws = torch.empty(max(int(ws_bytes), 16), dtype=torch.uint8, device=dev)
        with self._on_device():
            check(lib().euler_gpu_sample_fanout_unique(
                self._h, _stream(), self.seed, self._take_call_ids(2, call_id), _ptr(nodes), n,
                et_p, cnt_p, int(default_node), _ptr(id1), _ptr(w1), _ptr(t1), _ptr(idx),
                _ptr(rid), _ptr(rw), _ptr(rt), _ptr(ws)))
        # uint32 row numbers read as int64 for indexing
        return id1, w1, t1, idx.to(torch.int64) & 0xFFFFFFFF, rid, rw, rt

    def sample_fanout_with_feature(self, nodes, edge_types, counts, default_node,
                                   feature_ids, dimensions, call_id=None):
        """tf_euler sample_fanout_with_feature, dense part (euler_ops/neighbor_ops.py:49-70
        over tf_euler/kernels/sample_fanout_with_feature_op.cc:135-233) as ONE enqueue
        (euler_gpu_sample_fanout_with_feature): returns (neighbors_list, weights_list,
        types_list, dense_features) with dense_features[layer * len(feature_ids) + j] =
        [m_layer, dimensions[j]] float32, layer 0 = the roots."""
        nodes = _as_i64_cuda(nodes, self.device).reshape(-1)
        layers = len(counts)
        et = np.asarray(edge_types, dtype=np.int32).reshape(layers, -1)
        et, et_p, _ = _i32_array(et)
        k = et.size // layers if layers else 0
        cnt, cnt_p, _ = _i32_array(counts)
        fid, fid_p, _ = _i32_array(feature_ids)
        dim, dim_p, _ = _i32_array(dimensions)
        nd = len(feature_ids)
        n = nodes.numel()
        outs_n, outs_w, outs_t, dense = [], [], [], []
        m = n
        sizes = [n]
        for c in counts:
            m *= int(c)
            sizes.append(m)
            outs_n.append(torch.empty(m, dtype=torch.int64, device=self.device))
            outs_w.append(torch.empty(m, dtype=torch.float32, device=self.device))
            outs_t.append(torch.empty(m, dtype=torch.int32, device=self.device))
        for ml in sizes:
            for d in dimensions:
                dense.append(torch.empty((ml, int(d)), dtype=torch.float32, device=self.device))
        ws_bytes = lib().euler_gpu_sample_fanout_workspace(n, cnt_p, layers)
        ws = torch.empty(max(int(ws_bytes), 16), dtype=torch.uint8, device=self.device)
        pn = (C.c_void_p * max(layers, 1))(*[t.data_ptr() for t in outs_n])
        pw = (C.c_void_p * max(layers, 1))(*[t.data_ptr() for t in outs_w])
        pt = (C.c_void_p * max(layers, 1))(*[t.data_ptr() for t in outs_t])
        pd = (C.c_void_p * max(len(dense), 1))(*[t.data_ptr() for t in dense])
        with self._on_device():
            check(lib().euler_gpu_sample_fanout_with_feature(
                self._h, _stream(), self.seed, self._take_call_ids(layers, call_id),
                _ptr(nodes), n, et_p, k, cnt_p, layers, int(default_node), pn, pw, pt, _ptr(ws),
                fid_p, dim_p, nd, pd))
        return [nodes] + outs_n, outs_w, outs_t, dense

    def sage_blocks(self, nodes, edge_types, fanouts, default_node=-1, add_self_loops=True,
                    call_id=None, sync=True):
        """The whole SageDataFlow (tf_euler/python/dataflow/sage_dataflow.py +
        UniqueDataFlow.produce_subgraph) as ONE enqueue, no host round trip between
        the hops (euler_gpu_sage_blocks).  Returns (blocks, counts): blocks[h] =
        (n_id, res_n_id, edge_src, edge_dst, edge_index) - FIVE entries since round 4: edge_index
        is the [:, :e] view of the [2, cap] tensor whose rows are edge_src / edge_dst (rows
        contiguous, the tensor as a whole not: .contiguous() before .view(-1) / data_ptr() /
        torch.save, which would otherwise serialise the cap-sized storage) - and
        counts = the layer sizes - sliced to
        their true sizes after one read of the counts when sync is True, else padded
        to the worst case with `counts` left on the device (uint32 [layers + 1])."""
        nodes = _as_i64_cuda(nodes, self.device).reshape(-1)
        layers = len(fanouts)
        et = np.asarray(edge_types, dtype=np.int32).reshape(layers, -1)
        et, et_p, _ = _i32_array(et)
        k = et.size // layers if layers else 0
        fan, fan_p, _ = _i32_array(fanouts)
        n = nodes.numel()
        caps = [n]
        for c in fanouts:
            caps.append(caps[-1] * (int(c) + 1))
        dev = self.device
        n_ids = [torch.empty(max(caps[h + 1], 1), dtype=torch.int64, device=dev) for h in range(layers)]
        res = [torch.empty(max(caps[h], 1), dtype=torch.int64, device=dev) for h in range(layers)]
        # edge_src / edge_dst of a hop are the two rows of ONE [2, cap] tensor: the block's
        # edge_index is then a view of it ([:, :e]) instead of a stacked copy
        eidx = [torch.empty((2, max(caps[h + 1], 1)), dtype=torch.int64, device=dev) for h in range(layers)]
        esrc = [t[0] for t in eidx]
        edst = [t[1] for t in eidx]
        counts = torch.zeros(layers + 1, dtype=torch.int32, device=dev)
        ws = torch.empty(max(int(lib().euler_gpu_sage_blocks_workspace(n, fan_p, layers)), 16),
                         dtype=torch.uint8, device=dev)
        arr = lambda ts: (C.c_void_p * layers)(*[t.data_ptr() for t in ts])
        with self._on_device():
            check(lib().euler_gpu_sage_blocks(
                self._h, _stream(), self.seed, self._take_call_ids(layers, call_id), _ptr(nodes), n,
                et_p, k, fan_p, layers, int(default_node), 1 if add_self_loops else 0, _ptr(ws),
                arr(n_ids), arr(res), arr(esrc), arr(edst), _ptr(counts)))
        if not sync:
            return list(zip(n_ids, res, esrc, edst, eidx)), counts
        cnt = _read_counts(counts)                             # the one host read
        blocks = []
        for h in range(layers):
            e = cnt[h] * int(fanouts[h]) + (cnt[h] if add_self_loops else 0)
            blocks.append((n_ids[h][:cnt[h + 1]], res[h][:cnt[h]], esrc[h][:e], edst[h][:e], eidx[h][:, :e]))
        return blocks, cnt

    def sage_blocks_multi(self, nodes, edge_types, fanouts, default_node=-1, add_self_loops=True,
                          call_id=None, sync=True):
        """M minibatches' SageDataFlows as ONE enqueue (euler_gpu_sage_blocks_multi): nodes
        [M, n] int64 -> a list of M (blocks, counts) pairs, each what sage_blocks returns for
        that minibatch - minibatch b draws with the call ids of the b-th of M consecutive
        sage_blocks calls, and the result equals theirs bit for bit.  sync=False: (padded
        tensors per hop - n_id [M, cap], res_n_id [M, cap], edge_src / edge_dst [M, cap] - and the
        device-side counts [M, layers + 1]), no host read."""
        nodes = _as_i64_cuda(nodes, self.device)
        assert nodes.dim() == 2, "sage_blocks_multi: nodes must be [M, n]"
        M, n = int(nodes.shape[0]), int(nodes.shape[1])
        nodes = nodes.contiguous()
        layers = len(fanouts)
        et = np.asarray(edge_types, dtype=np.int32).reshape(layers, -1)
        et, et_p, _ = _i32_array(et)
        k = et.size // layers if layers else 0
        fan, fan_p, _ = _i32_array(fanouts)
        caps = [n]
        for c in fanouts:
            caps.append(caps[-1] * (int(c) + 1))
        dev = self.device
        n_ids = [torch.empty((M, max(caps[h + 1], 1)), dtype=torch.int64, device=dev) for h in range(layers)]
        res = [torch.empty((M, max(caps[h], 1)), dtype=torch.int64, device=dev) for h in range(layers)]
        esrc = [torch.empty((M, max(caps[h + 1], 1)), dtype=torch.int64, device=dev) for h in range(layers)]
        edst = [torch.empty((M, max(caps[h + 1], 1)), dtype=torch.int64, device=dev) for h in range(layers)]
        counts = torch.zeros((M, layers + 1), dtype=torch.int32, device=dev)
        ws = torch.empty(max(int(lib().euler_gpu_sage_blocks_multi_workspace(M, n, fan_p, layers)), 16),
                         dtype=torch.uint8, device=dev)
        arr = lambda ts: (C.c_void_p * layers)(*[t.data_ptr() for t in ts])
        with self._on_device():
            check(lib().euler_gpu_sage_blocks_multi(
                self._h, _stream(), self.seed, self._take_call_ids(layers * M, call_id), layers, M,
                _ptr(nodes), n, et_p, k, fan_p, layers, int(default_node), 1 if add_self_loops else 0,
                _ptr(ws), arr(n_ids), arr(res), arr(esrc), arr(edst), _ptr(counts)))
        if not sync:
            return list(zip(n_ids, res, esrc, edst)), counts
        cnt_all = counts.cpu().numpy().astype(np.int64)          # the one host read
        out = []
        for b in range(M):
            cnt = [int(x) for x in cnt_all[b]]
            blocks = []
            for h in range(layers):
                e = cnt[h] * int(fanouts[h]) + (cnt[h] if add_self_loops else 0)
                s_, d_ = esrc[h][b, :e], edst[h][b, :e]
                blocks.append((n_ids[h][b, :cnt[h + 1]], res[h][b, :cnt[h]], s_, d_, torch.stack([s_, d_], 0)))
            out.append((blocks, cnt))
        return out

    def full_blocks(self, nodes, edge_types, edge_caps, add_self_loops=True, with_types=False):
        """GCNDataFlow / RelationDataFlow block construction as ONE enqueue
        (euler_gpu_full_blocks): every hop = full neighbours of the listed types of the
        nodes so far + first-occurrence unique + res_n_id + edge_index.  edge_caps[h] =
        capacity of hop h's edge list.  Returns (blocks, counts) with blocks[h] = (n_id,
        res_n_id, edge_src, edge_dst, e_type or None) sliced to their true sizes after the
        one host read of the counts, or None when a hop overflowed its capacity (counts then
        tells the sizes that were reached)."""
        nodes = _as_i64_cuda(nodes, self.device).reshape(-1)
        layers = len(edge_types)
        et = np.asarray(edge_types, dtype=np.int32).reshape(layers, -1)
        et, et_p, _ = _i32_array(et)
        k = et.size // layers if layers else 0
        n = nodes.numel()
        caps_e = [int(c) for c in edge_caps]
        caps_n = [n]
        for c in caps_e:
            caps_n.append(caps_n[-1] + c)
        dev = self.device
        ecap = (C.c_int64 * layers)(*caps_e)
        n_ids = [torch.empty(max(caps_n[h + 1], 1), dtype=torch.int64, device=dev) for h in range(layers)]
        res = [torch.empty(max(caps_n[h], 1), dtype=torch.int64, device=dev) for h in range(layers)]
        esrc = [torch.empty(max(caps_e[h] + caps_n[h], 1), dtype=torch.int64, device=dev) for h in range(layers)]
        edst = [torch.empty(max(caps_e[h] + caps_n[h], 1), dtype=torch.int64, device=dev) for h in range(layers)]
        etyp = [torch.empty(max(caps_e[h], 1), dtype=torch.int32, device=dev) for h in range(layers)] \
            if with_types else None
        counts = torch.zeros(2 * layers + 2, dtype=torch.int32, device=dev)
        ws = torch.empty(max(int(lib().euler_gpu_full_blocks_workspace(n, ecap, layers)), 16),
                         dtype=torch.uint8, device=dev)
        arr = lambda ts: (C.c_void_p * layers)(*[t.data_ptr() for t in ts])
        with self._on_device():
            check(lib().euler_gpu_full_blocks(
                self._h, _stream(), _ptr(nodes), n, et_p, k, layers, 1 if add_self_loops else 0, ecap,
                _ptr(ws), arr(n_ids), arr(res), arr(esrc), arr(edst),
                arr(etyp) if with_types else None, _ptr(counts)))
        cnt = _read_counts(counts)                             # the one host read
        if cnt[2 * layers + 1]:
            return None, cnt
        blocks = []
        for h in range(layers):
            m_nb = cnt[layers + 1 + h]
            e = m_nb + (cnt[h] if add_self_loops else 0)
            blocks.append((n_ids[h][:cnt[h + 1]], res[h][:cnt[h]], esrc[h][:e], edst[h][:e],
                           etyp[h][:m_nb] if with_types else None))
        return blocks, cnt

    def set_node_sampler(self, ids=None, node_types=None, node_weights=None, n_node_types=None):
        """Graph::BuildGlobalSampler (core/graph/graph.cc:333-370) for a graph that has no
        global node sampler yet (a synthetic one), or in place of the one it has: the nodes
        in the order the sampler enumerates them (ids None = the rows in row order), their
        types (None = 0) and weights (None = 1.0).  Host arrays; host work."""
        n = self.num_nodes if ids is None else len(ids)
        ids_a = None if ids is None else _np(ids, np.uint64)
        ty_a = None if node_types is None else _np(node_types, np.int32)
        w_a = None if node_weights is None else _np(node_weights, np.float32)
        for a in (ty_a, w_a):
            if a is not None and len(a) != n:
                raise ValueError("set_node_sampler: array lengths differ")
        if n_node_types is None:
            n_node_types = int(ty_a.max()) + 1 if ty_a is not None and n else 1
        check(lib().euler_gpu_graph_set_node_sampler(
            self._h, int(n),
            None if ids_a is None else ids_a.ctypes.data_as(_lib.u64p),
            None if ty_a is None else ty_a.ctypes.data_as(_lib.i32p),
            None if w_a is None else w_a.ctypes.data_as(_lib.f32p), int(n_node_types)))

    def sample_node(self, count, node_type=-1, call_id=None):
        """tf_euler sample_node (tf_euler/kernels/sample_node_op.cc:39-98):
        [count] int64 ids drawn by node weight within the type(s)."""
        nt, nt_p, k = _i32_array(np.atleast_1d(node_type))
        out = torch.empty(int(count), dtype=torch.int64, device=self.device)
        with self._on_device():
            check(lib().euler_gpu_sample_node(
                self._h, _stream(), self.seed, self._take_call_ids(1, call_id),
                nt_p, k, int(count), _ptr(out)))
        return out

    def id_range(self):
        """(largest node id of this graph / shard, ids are base + stride * row)."""
        mx = C.c_uint64(0)
        ident = C.c_int32(0)
        check(lib().euler_gpu_graph_id_range(self._h, C.byref(mx), C.byref(ident)))
        return int(mx.value), bool(ident.value)

    @property
    def num_float_features(self):
        return lib().euler_gpu_graph_num_float_features(self._h)

    def get_dense_feature(self, nodes, feature_ids, dimensions):
        """tf_euler get_dense_feature (euler_ops/feature_ops.py:108-122 over
        tf_euler/kernels/get_dense_feature_op.cc): nodes [n] int64 -> list of
        [n, dim] float32 tensors, one per feature id; unknown nodes and missing
        features are zero rows."""
        nodes = _as_i64_cuda(nodes, self.device).reshape(-1)
        n = nodes.numel()
        outs = []
        with self._on_device():
            for fid, dim in zip(feature_ids, dimensions):
                out = torch.empty((n, int(dim)), dtype=torch.float32, device=self.device)
                check(lib().euler_gpu_get_dense_feature(
                    self._h, _stream(), _ptr(nodes), n, int(fid), int(dim), _ptr(out)))
                outs.append(out)
        return outs

    def num_u64_features(self):
        return lib().euler_gpu_graph_num_u64_features(self._h)

    def get_sparse_feature(self, nodes, feature_ids, default_values=None):
        """tf_euler get_sparse_feature (euler_ops/feature_ops.py:57-73 over
        tf_euler/kernels/get_sparse_feature_op.cc): nodes [n] int64 -> one
        SparseTensor triple (indices [nnz, 2] int64, values [nnz] int64,
        dense_shape [n, max_len]) per uint64 feature id; a node without values
        contributes the single entry (row, 0) = default value."""
        nodes = _as_i64_cuda(nodes, self.device).reshape(-1)
        n = nodes.numel()
        if default_values is None:
            default_values = [0] * len(feature_ids)
        outs = []
        with self._on_device():
            for fid, dv in zip(feature_ids, default_values):
                row_off = torch.empty(n + 1, dtype=torch.int64, device=self.device)
                nnz, max_len = C.c_int64(0), C.c_int64(0)
                check(lib().euler_gpu_get_sparse_feature(
                    self._h, _stream(), _ptr(nodes), n, int(fid), int(dv), _ptr(row_off),
                    C.byref(nnz), C.byref(max_len), None, None))
                ind = torch.empty((int(nnz.value), 2), dtype=torch.int64, device=self.device)
                val = torch.empty(int(nnz.value), dtype=torch.int64, device=self.device)
                if nnz.value:
                    check(lib().euler_gpu_get_sparse_feature(
                        self._h, _stream(), _ptr(nodes), n, int(fid), int(dv),
                        _ptr(row_off), C.byref(nnz), C.byref(max_len), _ptr(ind),
                        _ptr(val)))
                outs.append((ind, val, [n, int(max_len.value)] if n else [0, 0]))
        return outs

    def get_sparse_feature_core(self, nodes, fid):
        """One uint64 feature slot in the GQL `values()` layout: (idx [n, 2]
        int32 offsets, values int64), no default entries."""
        nodes = _as_i64_cuda(nodes, self.device).reshape(-1)
        n = nodes.numel()
        idx = torch.empty((n, 2), dtype=torch.int32, device=self.device)
        total = C.c_int64(0)
        with self._on_device():
            check(lib().euler_gpu_get_sparse_feature_core(
                self._h, _stream(), _ptr(nodes), n, int(fid), _ptr(idx), C.byref(total),
                None))
            vals = torch.empty(int(total.value), dtype=torch.int64, device=self.device)
            if total.value:
                check(lib().euler_gpu_get_sparse_feature_core(
                    self._h, _stream(), _ptr(nodes), n, int(fid), _ptr(idx),
                    C.byref(total), _ptr(vals)))
        return idx, vals

    _ORDER = {None: 0, "": 0, "id": 1, "weight": 2}

    def get_full_neighbor(self, nodes, edge_types, order_by=None, desc=False,
                          limit=None):
        """GQL `v(nodes).outV(edge_types)[.order_by(f, asc|desc)][.limit(k)]`
        result (idx [n,2] int32, ids int64, weights f32, types int32),
        core/kernels/get_neighbor_op.cc (post-process :117-168)."""
        nodes = _as_i64_cuda(nodes, self.device).reshape(-1)
        n = nodes.numel()
        et, et_p, k = _i32_array(edge_types)
        idx = torch.empty((n, 2), dtype=torch.int32, device=self.device)
        total = C.c_int64(0)
        with self._on_device():
            check(lib().euler_gpu_get_full_neighbor(
                self._h, _stream(), _ptr(nodes), n, et_p, k, _ptr(idx),
                C.byref(total), None, None, None))
            tot = int(total.value)
            ids = torch.empty(tot, dtype=torch.int64, device=self.device)
            w = torch.empty(tot, dtype=torch.float32, device=self.device)
            t = torch.empty(tot, dtype=torch.int32, device=self.device)
            if n:
                check(lib().euler_gpu_get_full_neighbor(
                    self._h, _stream(), _ptr(nodes), n, et_p, k, _ptr(idx),
                    C.byref(total), _ptr(ids), _ptr(w), _ptr(t)))
            if n and (self._ORDER[order_by] or limit is not None):
                new_total = C.c_int64(0)
                check(lib().euler_gpu_neighbor_post_process(
                    _stream(), n, _ptr(idx), tot, _ptr(ids), _ptr(w), _ptr(t),
                    self._ORDER[order_by], 1 if desc else 0,
                    -1 if limit is None else int(limit), C.byref(new_total)))
                m = int(new_total.value)
                ids, w, t = ids[:m], w[:m], t[:m]
        return idx, ids, w, t

    def get_sorted_full_neighbor(self, nodes, edge_types):
        """tf_euler get_sorted_full_neighbor: rows ordered by neighbour id
        (tf_euler/kernels/get_sorted_full_neighbor_op.cc:43-46)."""
        return self.get_full_neighbor(nodes, edge_types, order_by="id")

    def get_top_k_neighbor(self, nodes, edge_types, k, default_node=-1):
        """tf_euler get_top_k_neighbor (tf_euler/kernels/get_top_k_neighbor_op.cc):
        the k heaviest neighbours per node as dense [n, k] tensors (ids int64,
        weights f32, types int32), default_node / 0.0 / -1 where a node has
        fewer."""
        nodes = _as_i64_cuda(nodes, self.device).reshape(-1)
        n = nodes.numel()
        et, et_p, k_types = _i32_array(edge_types)
        out_n = torch.empty((n, k), dtype=torch.int64, device=self.device)
        out_w = torch.empty((n, k), dtype=torch.float32, device=self.device)
        out_t = torch.empty((n, k), dtype=torch.int32, device=self.device)
        with self._on_device():
            check(lib().euler_gpu_get_top_k_neighbor(
                self._h, _stream(), _ptr(nodes), n, et_p, k_types, int(k),
                int(default_node), _ptr(out_n), _ptr(out_w), _ptr(out_t)))
        return out_n, out_w, out_t

    def get_node_type(self, nodes):
        """tf_euler get_node_type (tf_euler/kernels/get_node_type_op.cc): int32
        type of every node, INT32_MIN for an unknown id."""
        nodes = _as_i64_cuda(nodes, self.device).reshape(-1)
        n = nodes.numel()
        out = torch.empty(n, dtype=torch.int32, device=self.device)
        with self._on_device():
            check(lib().euler_gpu_get_node_type(self._h, _stream(), _ptr(nodes), n,
                                                _ptr(out)))
        return out

    def sample_n_with_types(self, count, types, call_id=None):
        """tf_euler sample_n_with_types (tf_euler/kernels/
        sample_n_with_types_op.cc): [len(types), count] int64, row i drawn from
        the global sampler of node type types[i] (-1: all types).  Raises where
        the reference aborts (unknown / zero-weight type)."""
        types = torch.as_tensor(types, dtype=torch.int32, device=self.device) \
            .reshape(-1).contiguous()
        n = types.numel()
        out = torch.empty((n, int(count)), dtype=torch.int64, device=self.device)
        with self._on_device():
            check(lib().euler_gpu_sample_n_with_types(
                self._h, _stream(), self.seed, self._take_call_ids(1, call_id),
                _ptr(types), n, int(count), _ptr(out)))
        return out

    # ------------------------------------------------- layerwise sampling
    def get_edge_sum_weight(self, nodes, edge_types):
        """API_GET_EDGE_SUM_WEIGHT (core/kernels/get_edge_sum_weight_op.cc):
        f32 sum of each node's out-edge weights over the listed types."""
        nodes = _as_i64_cuda(nodes, self.device).reshape(-1)
        n = nodes.numel()
        et, et_p, k = _i32_array(edge_types)
        out = torch.empty(n, dtype=torch.float32, device=self.device)
        with self._on_device():
            check(lib().euler_gpu_get_edge_sum_weight(
                self._h, _stream(), _ptr(nodes), n, et_p, k, _ptr(out)))
        return out

    def sample_root(self, roots, weights, m, default_node=-1, call_id=None):
        """API_SAMPLE_ROOT (core/kernels/sample_root_op.cc): roots / weights
        [batch, n] -> [batch, m] draws (alias method per batch row)."""
        roots = _as_i64_cuda(roots, self.device)
        batch, n = roots.shape
        weights = torch.as_tensor(weights, dtype=torch.float32,
                                  device=self.device).reshape(batch, n).contiguous()
        out = torch.empty((batch, int(m)), dtype=torch.int64, device=self.device)
        with self._on_device():
            check(lib().euler_gpu_sample_root(
                _stream(), self.seed, self._take_call_ids(1, call_id), _ptr(roots),
                _ptr(weights), batch, n, int(m), int(default_node), _ptr(out)))
        return out

    def sample_layer(self, roots, edge_types, default_node=-1, call_id=None, positions=None):
        """API_SAMPLE_L (core/kernels/sample_layer_op.cc): one neighbour per
        listed root (position-keyed RNG) or (default_node, 0.0, 0).  positions:
        the RNG stream of every root (default: its index) - what the shard of a
        multi-GPU hop receives from the requester."""
        roots = _as_i64_cuda(roots, self.device).reshape(-1)
        n = roots.numel()
        if positions is not None:
            positions = _as_i64_cuda(positions, self.device).reshape(-1)
            assert positions.numel() == n
        et, et_p, k = _i32_array(edge_types)
        oid = torch.empty(n, dtype=torch.int64, device=self.device)
        ow = torch.empty(n, dtype=torch.float32, device=self.device)
        ot = torch.empty(n, dtype=torch.int32, device=self.device)
        with self._on_device():
            check(lib().euler_gpu_sample_layer_at(
                self._h, _stream(), self.seed, self._take_call_ids(1, call_id),
                _ptr(roots), _ptr(positions), n, et_p, k, int(default_node), _ptr(oid),
                _ptr(ow), _ptr(ot)))
        return oid, ow, ot

    def local_sample_layer(self, idx, ids, w, t, batch, n, m, weight_func="sqrt",
                           default_node=-1, call_id=None):
        """API_LOCAL_SAMPLE_L (core/kernels/local_sample_layer_op.cc): m draws
        per batch row from the distinct (id, type) full neighbours of its n
        nodes, weights of duplicates added and transformed by weight_func;
        (ids, weights, types) each [batch * m]."""
        idx = idx.to(torch.int32).contiguous()
        ids = ids.to(torch.int64).contiguous()
        w = w.to(torch.float32).contiguous()
        t = t.to(torch.int32).contiguous()
        total = ids.numel()
        oid = torch.empty(batch * m, dtype=torch.int64, device=self.device)
        ow = torch.empty(batch * m, dtype=torch.float32, device=self.device)
        ot = torch.empty(batch * m, dtype=torch.int32, device=self.device)
        with self._on_device():
            check(lib().euler_gpu_local_sample_layer(
                _stream(), self.seed, self._take_call_ids(1, call_id), _ptr(idx),
                _ptr(ids), _ptr(w), _ptr(t), total, batch, n, m,
                str(weight_func).encode(), int(default_node), _ptr(oid), _ptr(ow),
                _ptr(ot)))
        return oid, ow, ot

    def sparse_get_adj_core(self, roots, l_nb, n, m, edge_types):
        """API_SPARSE_GEN_ADJ + API_SPARSE_GET_ADJ (core/kernels/
        sparse_get_adj_op.cc): (idx [batch*n, 2] int32, ids int64)."""
        roots = _as_i64_cuda(roots, self.device).reshape(-1)
        l_nb = _as_i64_cuda(l_nb, self.device).reshape(-1)
        n, m = int(n), int(m)
        batch = roots.numel() // n if n else 0
        assert roots.numel() == batch * n and l_nb.numel() == batch * m
        et, et_p, k = _i32_array(edge_types)
        idx = torch.empty((batch * n, 2), dtype=torch.int32, device=self.device)
        ws = self._adj_workspace(batch, n, m)
        total = C.c_int64(0)
        with self._on_device():
            check(lib().euler_gpu_sparse_get_adj(
                self._h, _stream(), _ptr(roots), _ptr(l_nb), batch, n, m, et_p, k,
                _ptr(ws), _ptr(idx), C.byref(total), None))
            vals = torch.empty(int(total.value), dtype=torch.int64, device=self.device)
            if total.value:
                check(lib().euler_gpu_sparse_get_adj(
                    self._h, _stream(), _ptr(roots), _ptr(l_nb), batch, n, m, et_p, k,
                    _ptr(ws), _ptr(idx), C.byref(total), _ptr(vals)))
        return idx, vals

    def _adj_workspace(self, batch, n, m):
        """Scratch of a SparseGetAdj query (offsets + one bit per pair), kept
        between its two calls."""
        nbytes = int(lib().euler_gpu_sparse_get_adj_workspace(batch, n, m))
        return torch.empty((nbytes + 7) // 8, dtype=torch.int64, device=self.device)

    def sparse_get_adj(self, nodes, nb_nodes, edge_types, n=-1, m=-1):
        """tf_euler sparse_get_adj (tf_euler/kernels/sparse_get_adj_op.cc):
        COO (indices [nnz, 3] int64, values [nnz] int64, dense_shape) of the
        [batch, n, m] adjacency between nodes [batch*n] and nb_nodes
        [batch*m]; n / m = -1 take the whole input as one batch row."""
        nodes = _as_i64_cuda(nodes, self.device).reshape(-1)
        nb_nodes = _as_i64_cuda(nb_nodes, self.device).reshape(-1)
        n = nodes.numel() if n == -1 else int(n)
        m = nb_nodes.numel() if m == -1 else int(m)
        batch = nodes.numel() // n if n else 0
        if nodes.numel() != batch * n or nb_nodes.numel() < batch * m:
            raise ValueError("sparse_get_adj: nodes / nb_nodes do not match n, m")
        et, et_p, k = _i32_array(edge_types)
        row_off = self._adj_workspace(batch, n, m)
        nnz = C.c_int64(0)
        with self._on_device():
            check(lib().euler_gpu_sparse_get_adj_tf(
                self._h, _stream(), _ptr(nodes), _ptr(nb_nodes), batch, n, m, et_p, k,
                _ptr(row_off), C.byref(nnz), None, None))
            ind = torch.empty((int(nnz.value), 3), dtype=torch.int64, device=self.device)
            val = torch.empty(int(nnz.value), dtype=torch.int64, device=self.device)
            if nnz.value:
                check(lib().euler_gpu_sparse_get_adj_tf(
                    self._h, _stream(), _ptr(nodes), _ptr(nb_nodes), batch, n, m, et_p,
                    k, _ptr(row_off), C.byref(nnz), _ptr(ind), _ptr(val)))
        shape = [batch, n, m] if nnz.value else [0, 0, 0]
        return ind, val, shape

    def sparse_adj_mask(self, nodes, nb_nodes, batch, n, m, edge_types):
        """The hit mask of SparseGetAdj on THIS graph (shard): int64 [batch * n, (m + 63) // 64],
        bit c of a source = candidate c of its batch row is in the source's row; sources
        without a row here give zeros (euler_gpu_sparse_adj_mask)."""
        nodes = _as_i64_cuda(nodes, self.device).reshape(-1)
        nb_nodes = _as_i64_cuda(nb_nodes, self.device).reshape(-1)
        words = (int(m) + 63) // 64
        mask = torch.zeros((int(batch) * int(n), words), dtype=torch.int64, device=self.device)
        et, et_p, k = _i32_array(edge_types)
        if mask.numel():
            with self._on_device():
                check(lib().euler_gpu_sparse_adj_mask(
                    self._h, _stream(), _ptr(nodes), _ptr(nb_nodes), int(batch), int(n), int(m),
                    et_p, k, _ptr(mask)))
        return mask

    @staticmethod
    def adj_from_mask(mask, batch, n, m):
        """TF SparseGetAdj triple (indices [nnz, 3], values, dense_shape) from a hit mask
        (euler_gpu_sparse_adj_from_mask_tf) - needs no graph."""
        dev = mask.device
        batch, n, m = int(batch), int(n), int(m)
        mask = mask.contiguous()
        ws = torch.empty(8 * (batch * n + 1) + 16, dtype=torch.uint8, device=dev)
        nnz = C.c_int64(0)
        with torch.cuda.device(dev):
            check(lib().euler_gpu_sparse_adj_from_mask_tf(
                _stream(), _ptr(mask), batch, n, m, _ptr(ws), C.byref(nnz), None, None))
            ind = torch.empty((int(nnz.value), 3), dtype=torch.int64, device=dev)
            val = torch.empty(int(nnz.value), dtype=torch.int64, device=dev)
            if nnz.value:
                check(lib().euler_gpu_sparse_adj_from_mask_tf(
                    _stream(), _ptr(mask), batch, n, m, _ptr(ws), C.byref(nnz), _ptr(ind), _ptr(val)))
        return ind, val, ([batch, n, m] if nnz.value else [0, 0, 0])

    def sample_neighbor_layerwise(self, nodes, edge_types, count, default_node=-1,
                                  weight_func='', call_id=None):
        """tf_euler sample_neighbor_layerwise (euler_ops/neighbor_ops.py:72-77
        over tf_euler/kernels/sample_neighbor_layerwise_with_adj_op.cc):
        nodes [batch, n] -> (neighbors [batch, count] int64, (indices, values,
        dense_shape) of the [batch, n, count] adjacency).  weight_func == ''
        draws a node of the layer by edge weight sum, then one of its
        neighbours; 'sqrt' draws from the layer's distinct neighbours by the
        square root of their accumulated weight."""
        nodes = _as_i64_cuda(nodes, self.device)
        if nodes.dim() != 2:
            raise ValueError("sample_neighbor_layerwise: nodes must be [batch, n]")
        batch, n = nodes.shape
        et, et_p, k = _i32_array(edge_types)
        if weight_func:
            # sampleLNB(edge_types, n, m, weight_func, default_node): API_GET_NB_NODE
            # -> API_LOCAL_SAMPLE_L (parser/translator.cc:388-441)
            idx, ids, w, t = self.get_full_neighbor(nodes.reshape(-1), et)
            out, _w, _t = self.local_sample_layer(idx, ids, w, t, batch, n, int(count),
                                                  weight_func, default_node, call_id)
            out = out.reshape(batch, int(count))
            return out, self.sparse_get_adj(nodes, out, et, n, int(count))
        out = torch.empty((batch, int(count)), dtype=torch.int64, device=self.device)
        with self._on_device():
            check(lib().euler_gpu_sample_neighbor_layerwise(
                self._h, _stream(), self.seed, self._take_call_ids(1, call_id),
                _ptr(nodes), batch, n, et_p, k, int(count), int(default_node),
                _ptr(out)))
        return out, self.sparse_get_adj(nodes, out, et, n, int(count))

    def random_walk(self, nodes, edge_types, p=1.0, q=1.0, default_node=-1,
                    call_id=None):
        """tf_euler random_walk (tf_euler/kernels/random_walk_op.cc):
        edge_types = list (length walk_len) of per-step edge type lists;
        returns [n, walk_len+1] int64."""
        nodes = _as_i64_cuda(nodes, self.device).reshape(-1)
        walk_len = len(edge_types)
        et = np.asarray(edge_types, dtype=np.int32).reshape(walk_len, -1) \
            if walk_len else np.zeros((0, 0), np.int32)
        k = et.shape[1] if walk_len else 0
        et, et_p, _ = _i32_array(et)
        n = nodes.numel()
        out = torch.empty((n, walk_len + 1), dtype=torch.int64, device=self.device)
        with self._on_device():
            check(lib().euler_gpu_random_walk(
                self._h, _stream(), self.seed,
                self._take_call_ids(max(walk_len, 1), call_id), _ptr(nodes), n,
                et_p, k, walk_len, float(p), float(q), int(default_node),
                _ptr(out)))
        return out
