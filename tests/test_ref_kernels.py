"""The reference's OWN op kernels of the hot path that need nothing but the plugin API -
core/kernels/{id_unique_op,idx_gather_op,data_gather_op}.cc - compiled UNMODIFIED against the
plugin-API mirror (include/euler_op_framework.h, through the forwarding headers of oracle/shim/)
into oracle/_ref/libeuler_ref_kernels.so and run through the mirror's registry as
"REF:ID_UNIQUE" / "REF:IDX_GATHER" / "REF:DATA_GATHER":

  * CPU: they equal the oracle's restatement (oracle/euler_oracle.c) - the restatement is pinned
    to the reference's sources, not only to its test expectations;
  * GPU: the kernels libeuler_gpu.so registers under the plain names, run through the SAME
    harness and registry, equal them bit for bit.

That these sources compile at all is the source-compatibility claim of the mirror: `const
DAGNodeProto&`, `node_def.inputs(i)`, `ctx->tensor(..)`, `ctx->Allocate(..)`, `OutputName`,
`Tensor::Raw<T>()`, `TensorShape({..})`, `REGISTER_OP_KERNEL` as the reference spells them
(core/framework/op_kernel.h:38-130)."""
import ctypes as C
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "oracle", "_ref", "libeuler_ref_kernels.so")

u64p, i32p = C.POINTER(C.c_uint64), C.POINTER(C.c_int32)


def _lib():
    if not os.path.exists(SO):
        pytest.skip("oracle/_ref/libeuler_ref_kernels.so not built (make -C oracle ref_kernels)")
    from euler_amd import _lib as L           # libeuler_gpu.so first: it holds the registry
    L.lib()
    K = C.CDLL(SO, mode=C.RTLD_GLOBAL)
    K.refk_id_unique.restype = C.c_int64
    K.refk_id_unique.argtypes = [C.c_char_p, u64p, C.c_int64, u64p, i32p]
    K.refk_idx_gather.restype = C.c_int64
    K.refk_idx_gather.argtypes = [C.c_char_p, i32p, C.c_int64, i32p, C.c_int64, i32p]
    K.refk_data_gather.restype = C.c_int64
    K.refk_data_gather.argtypes = [C.c_char_p, C.c_void_p, C.c_int64, C.c_int32, i32p, C.c_int64, i32p,
                                   C.c_int64, C.c_void_p, C.c_int64]
    return K


def _p(a, t):
    return a.ctypes.data_as(t)


def _unique(K, op, ids):
    ids = np.ascontiguousarray(ids, np.uint64)
    uq = np.zeros(max(len(ids), 1), np.uint64)
    gi = np.zeros(max(len(ids), 1), np.int32)
    n = K.refk_id_unique(op, _p(ids, u64p), len(ids), _p(uq, u64p), _p(gi, i32p))
    assert n >= 0, (op, n)
    return uq[:n], gi[:len(ids)]


def _idx_gather(K, op, idx, gi):
    idx = np.ascontiguousarray(idx, np.int32).reshape(-1, 2)
    gi = np.ascontiguousarray(gi, np.int32)
    out = np.zeros((max(len(gi), 1), 2), np.int32)
    n = K.refk_idx_gather(op, _p(idx, i32p), len(idx), _p(gi, i32p), len(gi), _p(out, i32p))
    assert n == len(gi), (op, n)
    return out[:len(gi)]


_DT = {np.dtype(np.int32): 2, np.dtype(np.uint64): 7, np.dtype(np.float32): 8}


def _data_gather(K, op, data, idx, gi):
    data = np.ascontiguousarray(data)
    idx = np.ascontiguousarray(idx, np.int32).reshape(-1, 2)
    gi = np.ascontiguousarray(gi, np.int32)
    cap = int((idx[gi, 1] - idx[gi, 0]).sum()) if len(gi) else 0
    out = np.zeros(max(cap, 1), data.dtype)
    n = K.refk_data_gather(op, data.ctypes.data, len(data), _DT[data.dtype], _p(idx, i32p), len(idx),
                           _p(gi, i32p), len(gi), out.ctypes.data, len(out))
    assert n == cap, (op, n, cap)
    return out[:cap]


def _cases():
    rng = np.random.default_rng(20240930)
    for n, hi in ((1, 5), (7, 3), (1000, 50), (5000, 10 ** 12), (4096, 2)):
        ids = rng.integers(0, hi, n).astype(np.uint64)
        if n == 5000:
            ids[::7] = np.uint64(2 ** 63 + 5)
        yield ids


def test_reference_kernels_equal_the_oracle_restatement(O):
    K = _lib()
    for ids in _cases():
        uq, gi = _unique(K, b"REF:ID_UNIQUE", ids)
        ouq, ogi = O.id_unique(ids)
        assert np.array_equal(uq, ouq) and np.array_equal(gi, ogi)
        # the rows of the unique ids: ragged (a row per unique id), then gathered back per position
        rng = np.random.default_rng(len(ids))
        lens = rng.integers(0, 6, len(uq)).astype(np.int32)
        end = np.cumsum(lens).astype(np.int32)
        idx = np.stack([end - lens, end], 1).astype(np.int32)
        assert np.array_equal(_idx_gather(K, b"REF:IDX_GATHER", idx, gi), O.idx_gather(idx, gi))
        for dt in (np.uint64, np.float32, np.int32):
            data = rng.integers(0, 2 ** 31, int(end[-1]) if len(end) else 0).astype(dt)
            assert np.array_equal(_data_gather(K, b"REF:DATA_GATHER", data, idx, gi), O.data_gather(data, idx, gi))
    # an unknown op name is an error of the registry, not a crash
    assert K.refk_id_unique(b"REF:NO_SUCH_OP", _p(np.zeros(1, np.uint64), u64p), 1,
                            _p(np.zeros(1, np.uint64), u64p), _p(np.zeros(1, np.int32), i32p)) == -2


@pytest.mark.gpu
def test_gpu_kernels_equal_the_reference_kernels(torch_cuda):
    """ID_UNIQUE / IDX_GATHER / DATA_GATHER as libeuler_gpu.so registers them (HIP kernels behind
    the plugin API) against the reference's own kernels, both through the mirror's registry."""
    K = _lib()
    for ids in _cases():
        uq, gi = _unique(K, b"REF:ID_UNIQUE", ids)
        guq, ggi = _unique(K, b"ID_UNIQUE", ids)
        assert np.array_equal(uq, guq) and np.array_equal(gi, ggi)
        rng = np.random.default_rng(len(ids))
        lens = rng.integers(0, 6, len(uq)).astype(np.int32)
        end = np.cumsum(lens).astype(np.int32)
        idx = np.stack([end - lens, end], 1).astype(np.int32)
        assert np.array_equal(_idx_gather(K, b"REF:IDX_GATHER", idx, gi), _idx_gather(K, b"IDX_GATHER", idx, gi))
        for dt in (np.uint64, np.float32, np.int32):
            data = rng.integers(0, 2 ** 31, int(end[-1]) if len(end) else 0).astype(dt)
            assert np.array_equal(_data_gather(K, b"REF:DATA_GATHER", data, idx, gi),
                                  _data_gather(K, b"DATA_GATHER", data, idx, gi))
