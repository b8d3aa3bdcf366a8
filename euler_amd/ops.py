"""Graph-independent device operators of libeuler_gpu.so on torch tensors:
message passing (MPGather / MPScatterAdd / MPScatterMax), GenPair, the GQL
helper ops (ID_UNIQUE / IDX_GATHER / DATA_GATHER) and the shard split/merge
ops.  All tensors must live on a CUDA (ROCm) device; nothing here has a CPU
path."""
import contextlib
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import check, lib


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    """The current HIP stream of the current device as a pointer (torch.cuda.current_stream()
    builds a Stream object for it: 8 us a call, a quarter of a small op's enqueue cost)."""
    if _RAW_STREAM is not None:
        return C.c_void_p(_RAW_STREAM(torch.cuda.current_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


_NULL_CTX = contextlib.nullcontext()


def _on(device):
    """Context of `device`; nothing to switch (5 us less per op) when it is the current one."""
    if device.index is None or torch.cuda.current_device() == device.index:
        return _NULL_CTX
    return torch.cuda.device(device)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _need_cuda(*ts):
    for t in ts:
        if not t.is_cuda:
            raise RuntimeError("euler_amd ops need tensors in GPU memory "
                               "(no CPU fallback)")


def _gather_raw(params, indices):
    params = params.contiguous()
    indices = indices.to(torch.int32).contiguous()
    _need_cuda(params, indices)
    e, d = indices.numel(), params.shape[1]
    out = torch.empty((e, d), dtype=torch.float32, device=params.device)
    with _on(params.device):
        check(lib().euler_gpu_gather(_stream(), _ptr(params), _ptr(indices), e, d,
                                     params.shape[0], _ptr(out)))
    return out


def _scatter_raw(fn, updates, indices, size):
    updates = updates.contiguous()
    indices = indices.to(torch.int32).contiguous()
    _need_cuda(updates, indices)
    e, d = updates.shape
    out = torch.empty((int(size), d), dtype=torch.float32, device=updates.device)
    with _on(updates.device):
        check(fn(_stream(), _ptr(updates), _ptr(indices), e, d, int(size), _ptr(out)))
    return out


# Gradients as registered in tf_euler/python/euler_ops/mp_ops.py:39-62.
class _Gather(torch.autograd.Function):
    @staticmethod
    def forward(ctx, params, indices):
        ctx.save_for_backward(indices)
        ctx.n = params.shape[0]
        return _gather_raw(params, indices)

    @staticmethod
    def backward(ctx, grad):
        (indices,) = ctx.saved_tensors
        return _scatter_raw(lib().euler_gpu_scatter_add, grad, indices, ctx.n), None


class _ScatterAdd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, updates, indices, size):
        ctx.save_for_backward(indices)
        return _scatter_raw(lib().euler_gpu_scatter_add, updates, indices, size)

    @staticmethod
    def backward(ctx, grad):
        (indices,) = ctx.saved_tensors
        return _gather_raw(grad, indices), None, None


class _ScatterMax(torch.autograd.Function):
    @staticmethod
    def forward(ctx, updates, indices, size):
        out = _scatter_raw(lib().euler_gpu_scatter_max, updates, indices, size)
        ctx.save_for_backward(updates, indices, out)
        ctx.size = size
        return out

    @staticmethod
    def backward(ctx, grad):
        updates, indices, out = ctx.saved_tensors
        indicators = (updates == _gather_raw(out, indices)).to(updates.dtype)
        num_selected = _scatter_raw(lib().euler_gpu_scatter_add, indicators,
                                    indices, ctx.size)
        indicators = indicators / _gather_raw(num_selected, indices)
        return indicators * _gather_raw(grad, indices), None, None


def gather(params, indices):
    """MPGather: out[i,:] = params[indices[i],:] (fp32, int32 indices)."""
    return _Gather.apply(params, indices)


def scatter_add(updates, indices, size):
    """MPScatterAdd: out[indices[i],:] += updates[i,:], zero init, [size,D]."""
    return _ScatterAdd.apply(updates, indices, int(size))


def scatter_max(updates, indices, size):
    """MPScatterMax: element-wise max per destination, -1e9 for empty rows."""
    return _ScatterMax.apply(updates, indices, int(size))


class _ScatterMean(torch.autograd.Function):
    """scatter_add(x) / (scatter_add(ones) + 1e-7) in one pass (euler_gpu_scatter_mean);
    the gradient of the composition: grad / (count + 1e-7) gathered back."""

    @staticmethod
    def forward(ctx, updates, indices, size):
        ctx.save_for_backward(indices)
        ctx.size = size
        return _scatter_raw(lib().euler_gpu_scatter_mean, updates, indices, size)

    @staticmethod
    def backward(ctx, grad):
        (indices,) = ctx.saved_tensors
        ones = torch.ones((indices.numel(), 1), dtype=torch.float32, device=grad.device)
        count = _scatter_raw(lib().euler_gpu_scatter_add, ones, indices, ctx.size) + 1e-7
        return _gather_raw(grad / count, indices), None, None


def scatter_mean(updates, indices, size):
    """mp_ops.py:65-69."""
    if updates.shape[0] < (1 << 24):
        return _ScatterMean.apply(updates, indices, int(size))
    out = scatter_add(updates, indices, size)
    ep = 1e-7
    ones = torch.ones((updates.shape[0], 1), dtype=torch.float32,
                      device=updates.device)
    count = scatter_add(ones, indices, size) + ep
    return out / count


_GS_MODE = {"add": 0, "max": 1, "mean": 2}


def _check_rows(name, gi, n_rows):
    """opt-in (a device round trip): the kernels index params with these as they are"""
    if gi.numel() and (int(gi.min()) < 0 or int(gi.max()) >= n_rows):
        raise IndexError("%s: gather index out of range" % name)


def _gather_scatter_raw(mode, params, gather_indices, scatter_indices, size, validate=False):
    params = params.contiguous()
    gi = gather_indices.to(torch.int32).contiguous()
    si = scatter_indices.to(torch.int32).contiguous()
    _need_cuda(params, gi)
    _need_cuda(params, si)
    if gi.numel() != si.numel():
        raise ValueError("gather_scatter: one gather index and one scatter index per edge")
    if validate:
        _check_rows("gather_scatter", gi, params.shape[0])
    e, d = gi.numel(), params.shape[1]
    out = torch.empty((int(size), d), dtype=torch.float32, device=params.device)
    with _on(params.device):
        check(lib().euler_gpu_gather_scatter(_stream(), mode, _ptr(params), _ptr(gi), _ptr(si), e, d,
                                             int(size), _ptr(out)))
    return out


class _GatherScatter(torch.autograd.Function):
    """scatter_(op, gather(params, gi), si, size) in one pass (euler_gpu_gather_scatter);
    the gradient is the composition's: the scatter's gradient (mp_ops.py:39-62) per edge,
    scatter-added into the rows of params the edges read."""

    @staticmethod
    def forward(ctx, params, gather_indices, scatter_indices, size, op, validate):
        out = _gather_scatter_raw(_GS_MODE[op], params, gather_indices, scatter_indices, size,
                                  validate)
        ctx.save_for_backward(params, gather_indices, scatter_indices, out)
        ctx.size, ctx.op = size, op
        return out

    @staticmethod
    def backward(ctx, grad):
        params, gi, si, out = ctx.saved_tensors
        if ctx.op == "add":
            per_edge = _gather_raw(grad, si)
        elif ctx.op == "mean":
            ones = torch.ones((si.numel(), 1), dtype=torch.float32, device=grad.device)
            count = _scatter_raw(lib().euler_gpu_scatter_add, ones, si, ctx.size) + 1e-7
            per_edge = _gather_raw(grad / count, si)
        else:
            updates = _gather_raw(params, gi)
            indicators = (updates == _gather_raw(out, si)).to(updates.dtype)
            num_selected = _scatter_raw(lib().euler_gpu_scatter_add, indicators, si, ctx.size)
            per_edge = indicators / _gather_raw(num_selected, si) * _gather_raw(grad, si)
        return (_scatter_raw(lib().euler_gpu_scatter_add, per_edge, gi, params.shape[0]),
                None, None, None, None, None)


def gather_scatter(op, params, gather_indices, scatter_indices, size, validate=False):
    """scatter_(op, gather(params, gather_indices), scatter_indices, size) for op in
    "add" / "max" / "mean" - the aggregation of a message-passing step whose message is
    the neighbour's row (SAGE mean / max, GCN after its normalisation) - without
    materialising the gathered [E, D] block: same bits, a third of the HBM traffic.
    Like `gather`, the kernel trusts the gather indices; validate=True checks them first
    (one device round trip)."""
    if op not in _GS_MODE:
        raise ValueError("gather_scatter: op is add, max or mean")
    if op == "mean" and gather_indices.numel() >= (1 << 24):
        return scatter_mean(gather(params, gather_indices), scatter_indices, size)
    return _GatherScatter.apply(params, gather_indices, scatter_indices, int(size), op, bool(validate))


def _segment_dst(seg_ptr, count, size, device):
    """destination of every update of a segmented block (for the gradient)"""
    if seg_ptr is None:
        return torch.arange(size, device=device, dtype=torch.int32).repeat_interleave(int(count))
    lens = (seg_ptr[1:] - seg_ptr[:-1]).to(torch.int64)
    return torch.repeat_interleave(torch.arange(size, device=device, dtype=torch.int32), lens)


class _GatherSegmentReduce(torch.autograd.Function):
    @staticmethod
    def forward(ctx, params, gather_indices, seg_ptr, count, size, op, validate):
        params = params.contiguous()
        # int64 ids (what the samplers return) are read in place: index = low word of the id,
        # the int32 a cast would give, without the cast's pass (euler_gpu_gather_segment_reduce_ids)
        as_ids = gather_indices.dtype == torch.int64 and not validate
        gi = gather_indices.contiguous() if as_ids else gather_indices.to(torch.int32).contiguous()
        _need_cuda(params, gi)
        sp = None
        if seg_ptr is not None:
            sp = seg_ptr.to(torch.int64).contiguous()
            _need_cuda(params, sp)
            if sp.numel() != size + 1:
                raise ValueError("gather_segment_reduce: seg_ptr has size + 1 entries")
        elif gi.numel() != size * count:
            raise ValueError("gather_segment_reduce: size * count gather indices")
        if validate:
            _check_rows("gather_segment_reduce", gi, params.shape[0])
        out = torch.empty((int(size), params.shape[1]), dtype=torch.float32, device=params.device)
        with _on(params.device):
            if as_ids:      # (ids past the table read its last row: euler_gpu.h)
                check(lib().euler_gpu_gather_segment_reduce_ids(
                    _stream(), _GS_MODE[op], _ptr(params), int(params.shape[0]), _ptr(gi),
                    _ptr(sp) if sp is not None else None, int(count), params.shape[1], int(size), _ptr(out)))
            else:
                check(lib().euler_gpu_gather_segment_reduce(
                    _stream(), _GS_MODE[op], _ptr(params), _ptr(gi), _ptr(sp) if sp is not None else None,
                    int(count), params.shape[1], int(size), _ptr(out)))
        if as_ids and ctx.needs_input_grad[0]:
            # the gradient kernels take int32 indices.  The forward kernel reads an id past the
            # table - or a negative one, e.g. default_node = -1: unsigned there - from the table's
            # LAST row (euler_gpu.h); the gradient must flow to that same row, so the saved
            # indices get the same clamp (the raw low word would be an out-of-range or
            # negative scatter key)
            last = int(params.shape[0]) - 1            # min((uint32) low word, rows - 1), as the kernel
            gi = torch.clamp(gi & 0xFFFFFFFF, max=last).to(torch.int32)
        ctx.save_for_backward(params, gi, sp if sp is not None else torch.empty(0), out)
        ctx.has_ptr, ctx.count, ctx.size, ctx.op = sp is not None, int(count), int(size), op
        return out

    @staticmethod
    def backward(ctx, grad):
        params, gi, sp, out = ctx.saved_tensors
        si = _segment_dst(sp if ctx.has_ptr else None, ctx.count, ctx.size, grad.device)
        if ctx.op == "add":
            per_edge = _gather_raw(grad, si)
        elif ctx.op == "mean":
            ones = torch.ones((si.numel(), 1), dtype=torch.float32, device=grad.device)
            cnt = _scatter_raw(lib().euler_gpu_scatter_add, ones, si, ctx.size) + 1e-7
            per_edge = _gather_raw(grad / cnt, si)
        else:
            updates = _gather_raw(params, gi)
            indicators = (updates == _gather_raw(out, si)).to(updates.dtype)
            num_selected = _scatter_raw(lib().euler_gpu_scatter_add, indicators, si, ctx.size)
            per_edge = indicators / _gather_raw(num_selected, si) * _gather_raw(grad, si)
        return (_scatter_raw(lib().euler_gpu_scatter_add, per_edge, gi, params.shape[0]),
                None, None, None, None, None, None)


def gather_segment_reduce(op, params, gather_indices, size, seg_ptr=None, count=None, validate=False):
    """The aggregation of a sampled block: destination r reduces (op = "add" / "max" /
    "mean") the rows params[gather_indices[p]] for p in [seg_ptr[r], seg_ptr[r + 1]) - or
    its `count` consecutive indices when seg_ptr is None (SampleNeighbor's fixed fan-out)
    - in that order.  The bits of scatter_(op, gather(params, gather_indices), dst, size)
    with dst = the destination of every index, in one pass and without the scatter's
    look at its key column (no host wait; validate=True checks the gather indices first,
    which is one)."""
    if op not in _GS_MODE:
        raise ValueError("gather_segment_reduce: op is add, max or mean")
    if (seg_ptr is None) == (count is None):
        raise ValueError("gather_segment_reduce: pass seg_ptr or count")
    return _GatherSegmentReduce.apply(params, gather_indices, seg_ptr, 0 if count is None else int(count),
                                      int(size), op, bool(validate))


def scatter_softmax(updates, indices, size):
    """mp_ops.py:76-79."""
    updates = updates - gather(scatter_max(updates, indices, size), indices)
    updates = torch.exp(updates)
    return updates / gather(scatter_add(updates, indices, size), indices)


def scatter_(op, updates, indices, size):
    """mp_ops.py:72-73."""
    return {"add": scatter_add, "max": scatter_max, "mean": scatter_mean,
            "softmax": scatter_softmax}[op](updates, indices, size)


def gen_pair(paths, left_win_size, right_win_size):
    """GenPair (tf_euler/kernels/gen_pair_op.cc): [batch, path_len] int64 ->
    [batch, pair_count, 2]."""
    paths = paths.to(torch.int64).contiguous()
    _need_cuda(paths)
    b, l = paths.shape
    pc = lib().euler_gpu_gen_pair_count(l, left_win_size, right_win_size)
    out = torch.empty((b, pc, 2), dtype=torch.int64, device=paths.device)
    with _on(paths.device):
        check(lib().euler_gpu_gen_pair(_stream(), _ptr(paths), b, l, left_win_size,
                                       right_win_size, _ptr(out)))
    return out


def node2vec_step(seed, call_id, c_row, c_idx, c_ids, c_w, p_row, p_idx, p_ids, parent_ids,
                  p, q, default_node=-1):
    """One node2vec step over explicit neighbour lists (euler_gpu_node2vec_step;
    tf_euler/kernels/random_walk_op.cc:83-168): walker i's child list is row c_row[i] of
    (c_idx [rows, 2], c_ids, c_w), its parent's list row p_row[i] of (p_idx, p_ids) -
    p_row None on the first step.  Returns the next nodes [n] int64."""
    c_row = c_row.to(torch.int32).contiguous()
    c_idx = c_idx.to(torch.int32).contiguous()
    c_ids = c_ids.to(torch.int64).contiguous()
    c_w = c_w.to(torch.float32).contiguous()
    parent_ids = parent_ids.to(torch.int64).contiguous()
    _need_cuda(c_row, c_idx, c_ids, c_w, parent_ids)
    if p_row is not None:
        p_row = p_row.to(torch.int32).contiguous()
        p_idx = p_idx.to(torch.int32).contiguous()
        p_ids = p_ids.to(torch.int64).contiguous()
        _need_cuda(p_row, p_idx, p_ids)
    n = c_row.numel()
    out = torch.empty(n, dtype=torch.int64, device=c_row.device)
    null = C.c_void_p(0)
    with _on(c_row.device):
        check(lib().euler_gpu_node2vec_step(
            _stream(), int(seed), int(call_id), n, _ptr(c_row), _ptr(c_idx), _ptr(c_ids), _ptr(c_w),
            int(c_w.numel()),
            _ptr(p_row) if p_row is not None else null,
            _ptr(p_idx) if p_row is not None else null,
            _ptr(p_ids) if p_row is not None else null,
            _ptr(parent_ids), float(p), float(q), int(default_node), _ptr(out)))
    return out


def id_unique(ids):
    """ID_UNIQUE: (unique ids in first-occurrence order, gather_idx int32)."""
    ids = ids.to(torch.int64).contiguous().reshape(-1)
    _need_cuda(ids)
    n = ids.numel()
    uq = torch.empty(n, dtype=torch.int64, device=ids.device)
    gi = torch.empty(n, dtype=torch.int32, device=ids.device)
    nu = C.c_int64(0)
    with _on(ids.device):
        check(lib().euler_gpu_id_unique(_stream(), _ptr(ids), n, _ptr(uq), _ptr(gi),
                                        C.byref(nu)))
    return uq[:nu.value], gi


def idx_gather(idx, gather_idx):
    """IDX_GATHER: re-based [n,2] offsets of the gathered segments."""
    idx = idx.to(torch.int32).contiguous()
    gi = gather_idx.to(torch.int32).contiguous()
    _need_cuda(idx, gi)
    n = gi.numel()
    out = torch.empty((n, 2), dtype=torch.int32, device=idx.device)
    total = C.c_int64(0)
    with _on(idx.device):
        check(lib().euler_gpu_idx_gather(_stream(), _ptr(idx), _ptr(gi), n,
                                         _ptr(out), C.byref(total)))
    return out, int(total.value)


def data_gather(data, idx, gather_idx):
    """DATA_GATHER: concatenated segments data[idx[g]] for g in gather_idx."""
    data = data.contiguous()
    idx = idx.to(torch.int32).contiguous()
    gi = gather_idx.to(torch.int32).contiguous()
    _need_cuda(data, idx, gi)
    out_idx, total = idx_gather(idx, gi)
    out = torch.empty(total, dtype=data.dtype, device=data.device)
    with _on(data.device):
        check(lib().euler_gpu_data_gather(_stream(), _ptr(data), data.element_size(),
                                          _ptr(idx), _ptr(gi), _ptr(out_idx),
                                          gi.numel(), _ptr(out)))
    return out


def sparse_gather(gather_idx, indices, values, dense_shape):
    """tf_euler sparse_gather (tf_euler/kernels/sparse_gather_op.cc:32-240, op
    tf_euler/ops/util_ops.cc:37-55): `gather` on a SparseTensor whose entries are sorted by row.
    Row g of the result = row gather_idx[g] of the input: its entries in order, first index
    column rewritten to g, the other columns and the values copied; dense_shape[0] = len(
    gather_idx).  Returns (out_indices [nnz', cols] int64, out_values, out_dense_shape).  The
    rows' extents come from a binary search of the first index column (the reference's
    GatherWithBinarySearch; its GatherWithIndex branch answers the same on inputs where every
    gathered row has an entry - what get_sparse_feature produces: it inserts a default entry
    for empty rows), the copies are the DATA_GATHER kernel's.  A row without entries gathers
    nothing; an index >= dense_shape[0] raises IndexError as the reference's InvalidArgument."""
    gi = gather_idx.reshape(-1).to(torch.int64)
    indices = indices.to(torch.int64).contiguous()
    _need_cuda(indices, values)
    if indices.dim() != 2 or values.dim() != 1 or values.numel() != indices.shape[0]:
        raise ValueError("sparse_gather: indices [nnz, cols] and values [nnz]")
    shape = [int(x) for x in (dense_shape.tolist() if hasattr(dense_shape, "tolist") else dense_shape)]
    if len(shape) != indices.shape[1]:
        raise ValueError("sparse_gather: dense_shape and indices shape mismatch")
    rows = shape[0]
    if gi.numel() and (int(gi.max()) >= rows or int(gi.min()) < 0):
        raise IndexError("sparse_gather: gather idx out of range")
    dev = indices.device
    col0 = indices[:, 0].contiguous()
    bounds = torch.searchsorted(col0, torch.arange(rows + 1, device=dev, dtype=torch.int64))
    idx = torch.stack([bounds[:-1], bounds[1:]], dim=1).to(torch.int32)
    g32 = gi.to(torch.int32).to(dev)
    out_values = data_gather(values, idx, g32)
    lens = (idx[:, 1] - idx[:, 0]).to(torch.int64)[gi.to(dev)]
    cols = [torch.repeat_interleave(torch.arange(gi.numel(), device=dev, dtype=torch.int64), lens)]
    for k in range(1, indices.shape[1]):
        cols.append(data_gather(indices[:, k].contiguous(), idx, g32))
    out_indices = torch.stack(cols, dim=1) if out_values.numel() else \
        torch.empty((0, indices.shape[1]), dtype=torch.int64, device=dev)
    out_shape = torch.tensor([gi.numel()] + shape[1:], dtype=torch.int64)
    return out_indices, out_values, out_shape


def inflate_idx(idx):
    """tf_euler inflate_idx (tf_euler/kernels/inflate_idx_op.cc:34-66, op tf_euler/ops/util_ops.cc):
    idx is a 1-D int32 vector whose values are exactly 0 .. U-1; out[i] = the place of entry i
    after a stable sort by value.  A value outside [0, number of distinct values) raises
    ValueError (the reference's InvalidArgument), a non-vector too."""
    if idx.dim() != 1:
        raise ValueError("InflateIdx expects a 1-D vector.")
    idx = idx.to(torch.int32).contiguous()
    _need_cuda(idx)
    out = torch.empty_like(idx)
    with _on(idx.device):
        rc = lib().euler_gpu_inflate_idx(_stream(), _ptr(idx), idx.numel(), _ptr(out))
    if rc == -1:
        raise ValueError("InflateIdx: expect input idx in [0,unique_cnt).")
    check(rc)
    return out


def id_split(ids, partitions, shards):
    """ID_SPLIT: stable bucket by owner(id) = (id % partitions) % shards.
    Returns (shard_off list[shards+1], shard_ids int64 [n], merge_idx int32 [n])."""
    ids = ids.to(torch.int64).contiguous().reshape(-1)
    _need_cuda(ids)
    n = ids.numel()
    off = (C.c_int64 * (shards + 1))()
    sid = torch.empty(n, dtype=torch.int64, device=ids.device)
    mi = torch.empty(n, dtype=torch.int32, device=ids.device)
    with _on(ids.device):
        check(lib().euler_gpu_id_split(_stream(), _ptr(ids), n, partitions, shards,
                                       off, _ptr(sid), _ptr(mi)))
    return list(off), sid, mi


def dedup_split(ids, partitions, shards, root_mask=None, root_group=1, dense_table=None):
    """Distinct ids bucketed by owner + the bucketed index of every position
    (ID_UNIQUE + ID_SPLIT in one call, one host sync).  Returns (shard_off
    list[shards+1], shard_ids int64 [m], pos int32 [n]) with m = shard_off[-1] <= n.
    root_mask ([ceil(n / root_group)] uint8): marked groups count as id 0.
    dense_table (optional int32 [limit + 1] scratch kept by the caller, contents
    irrelevant): every id of the graph is < limit, so duplicates are found in a
    table indexed by the id itself instead of by hashing."""
    ids = ids.to(torch.int64).contiguous().reshape(-1)
    _need_cuda(ids)
    n = ids.numel()
    if root_mask is not None:
        root_mask = root_mask.to(torch.uint8).contiguous()
    limit = 0
    if dense_table is not None:
        _need_cuda(dense_table)
        assert dense_table.dtype == torch.int32 and dense_table.is_contiguous()
        limit = dense_table.numel() - 1
    off = (C.c_int64 * (shards + 1))()
    sid = torch.empty(n, dtype=torch.int64, device=ids.device)
    pos = torch.empty(n, dtype=torch.int32, device=ids.device)
    with _on(ids.device):
        check(lib().euler_gpu_dedup_split(_stream(), _ptr(ids), n, _ptr(root_mask),
                                          int(root_group), partitions, shards,
                                          _ptr(dense_table), limit, off, _ptr(sid),
                                          _ptr(pos)))
    off = list(off)
    return off, sid[:off[-1]], pos


class FrontHandle(object):
    """One dedup_split call in flight (euler_gpu_front): begin() enqueues the
    front end and returns, end() waits for the bucket sizes only."""

    def __init__(self):
        h = C.c_void_p()
        check(lib().euler_gpu_front_create(C.byref(h)))
        self._h = h
        self._token = None

    def begin(self, ids, partitions, shards, root_mask=None, root_group=1, dense_table=None):
        ids = ids.to(torch.int64).contiguous().reshape(-1)
        _need_cuda(ids)
        n = ids.numel()
        if root_mask is not None:
            root_mask = root_mask.to(torch.uint8).contiguous()
        limit = dense_table.numel() - 1 if dense_table is not None else 0
        sid = torch.empty(n, dtype=torch.int64, device=ids.device)
        pos = torch.empty(n, dtype=torch.int32, device=ids.device)
        with _on(ids.device):
            check(lib().euler_gpu_dedup_split_begin(
                self._h, _stream(), _ptr(ids), n, _ptr(root_mask), int(root_group), partitions,
                shards, _ptr(dense_table), limit, _ptr(sid), _ptr(pos)))
        # the inputs must outlive the enqueued kernels: keep them with the token
        self._token = (shards, sid, pos, ids, root_mask)
        return self

    def end(self):
        shards, sid, pos, _ids, _mask = self._token
        self._token = None
        off = (C.c_int64 * (shards + 1))()
        check(lib().euler_gpu_dedup_split_end(self._h, off))
        off = list(off)
        return off, sid[:off[-1]], pos

    def __del__(self):
        try:
            if self._h is not None and self._h.value:
                lib().euler_gpu_front_destroy(self._h)
                self._h = None
        except Exception:
            pass


def expand_rows(pos, ids, w, t, mask, count):
    """Row pos[i] of the sampled rows of the distinct roots -> position i:
    (ids [n,count] int64, w f32, t int32, mask [n] uint8)."""
    pos = pos.to(torch.int32).contiguous()
    ids = ids.contiguous(); w = w.contiguous(); t = t.contiguous()
    mask = mask.to(torch.uint8).contiguous()
    _need_cuda(pos, ids, w, t, mask)
    n = pos.numel()
    dev = pos.device
    o_id = torch.empty((n, count), dtype=torch.int64, device=dev)
    o_w = torch.empty((n, count), dtype=torch.float32, device=dev)
    o_t = torch.empty((n, count), dtype=torch.int32, device=dev)
    o_m = torch.empty(n, dtype=torch.uint8, device=dev)
    with _on(dev):
        check(lib().euler_gpu_expand_rows(_stream(), _ptr(pos), n, int(count), _ptr(ids),
                                          _ptr(w), _ptr(t), _ptr(mask), _ptr(o_id),
                                          _ptr(o_w), _ptr(o_t), _ptr(o_m)))
    return o_id, o_w, o_t, o_m


def packed_words(count, single_type=None):
    """int32 words per wire row: ids (2 each) | weights | [types] | mask, pad."""
    return ((3 if single_type is not None else 4) * int(count) + 2 + 1) & ~1     # even: 8-byte rows


def pack_rows(ids, w, t, mask, count, single_type=None):
    """Sampler outputs [m, count] (+ mask [m]) -> wire rows [m, packed_words] int32.
    single_type (the one listed edge type of the call): the type column is left
    off the wire."""
    ids = ids.contiguous(); w = w.contiguous(); t = t.contiguous()
    mask = mask.to(torch.uint8).contiguous()
    _need_cuda(ids, w, t, mask)
    m = mask.numel()
    out = torch.empty((m, packed_words(count, single_type)), dtype=torch.int32,
                      device=ids.device)
    with _on(ids.device):
        check(lib().euler_gpu_pack_rows(_stream(), _ptr(ids), _ptr(w), _ptr(t),
                                        _ptr(mask), m, int(count),
                                        -1 if single_type is None else int(single_type),
                                        _ptr(out)))
    return out


def expand_packed(pos, packed, count, single_type=None):
    """Wire rows -> (ids [n,count] int64, w f32, t int32, mask [n] uint8) per
    position: row pos[i] of `packed` is position i's row."""
    pos = pos.to(torch.int32).contiguous()
    packed = packed.contiguous()
    _need_cuda(pos, packed)
    assert packed.shape[-1] == packed_words(count, single_type), "wire row width"
    n = pos.numel()
    dev = pos.device
    o_id = torch.empty((n, count), dtype=torch.int64, device=dev)
    o_w = torch.empty((n, count), dtype=torch.float32, device=dev)
    o_t = torch.empty((n, count), dtype=torch.int32, device=dev)
    o_m = torch.empty(n, dtype=torch.uint8, device=dev)
    with _on(dev):
        check(lib().euler_gpu_expand_packed(_stream(), _ptr(pos), n, int(count),
                                            -1 if single_type is None else int(single_type),
                                            _ptr(packed), _ptr(o_id), _ptr(o_w),
                                            _ptr(o_t), _ptr(o_m)))
    return o_id, o_w, o_t, o_m


def merge_rows(rows, merge_idx, n_rows=None):
    """IDX_MERGE / DATA_MERGE for fixed-size rows: out[merge_idx[j]] = rows[j]."""
    rows = rows.contiguous()
    mi = merge_idx.to(torch.int32).contiguous()
    _need_cuda(rows, mi)
    n = mi.numel()
    out = torch.empty_like(rows) if n_rows is None else torch.empty(
        (n_rows,) + tuple(rows.shape[1:]), dtype=rows.dtype, device=rows.device)
    row_bytes = rows.element_size() * (rows.numel() // max(n, 1)) if n else 4
    with _on(rows.device):
        check(lib().euler_gpu_merge_rows(_stream(), _ptr(rows), _ptr(mi), n,
                                         row_bytes, _ptr(out)))
    return out


def sample_node_split(seed, call_id, count, shard_weight):
    """SAMPLE_NODE_SPLIT (host): per-shard counts proportional to the shards'
    type weight sums; shard_weight has shards+1 entries (last = total)."""
    sw = np.ascontiguousarray(shard_weight, dtype=np.float32)
    shards = len(sw) - 1
    out = np.zeros(shards, np.int32)
    check(lib().euler_gpu_sample_node_split(
        seed, call_id, count, sw.ctypes.data_as(_lib.f32p), shards,
        out.ctypes.data_as(_lib.i32p)))
    return out
