"""ORACLE python bindings (test infrastructure, NOT product code).

ctypes wrappers over
  * oracle/libeuler_oracle.so   - the plain-C restatement (euler_oracle.c)
  * oracle/_ref/libeuler_ref.so - the reference sampler compiled from
    /root/reference with the RNG seam (ref_harness.cc), when it has been built

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this module.  The product package (euler_amd) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "libeuler_oracle.so")
REF_SO = os.path.join(HERE, "_ref", "libeuler_ref.so")

DOMAIN_NEIGHBOR, DOMAIN_NODE, DOMAIN_WALK, DOMAIN_SPLIT = 0, 1, 2, 3

_u64p = C.POINTER(C.c_uint64)
_i64p = C.POINTER(C.c_int64)
_i32p = C.POINTER(C.c_int32)
_f32p = C.POINTER(C.c_float)


def build(ref=True):
    """Compile the C restatement (and oracle/_ref when /root/reference exists)."""
    subprocess.check_call(["make", "-s", "-C", HERE, "oracle"])
    if ref:
        subprocess.check_call(["make", "-s", "-C", HERE, "ref"])


def _p(a, typ):
    if a is None:
        return None
    return a.ctypes.data_as(typ)


def _arr(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


_lib = None
_ref = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(ORACLE_SO):
            build(ref=False)
        _lib = C.CDLL(ORACLE_SO)
        L = _lib
        L.eo_graph_create.restype = C.c_void_p
        L.eo_graph_create.argtypes = [C.c_int64, C.c_int32, _u64p, _i64p, _i32p,
                                      _u64p, _f32p, _f32p]
        L.eo_graph_destroy.argtypes = [C.c_void_p]
        L.eo_graph_find_row.restype = C.c_int64
        L.eo_graph_find_row.argtypes = [C.c_void_p, C.c_uint64]
        L.eo_uniform_at.restype = C.c_double
        L.eo_uniform_at.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32,
                                    C.c_uint64, C.c_uint64]
        L.eo_random_select.restype = C.c_int64
        L.eo_random_select.argtypes = [_f32p, C.c_uint64, C.c_uint64, C.c_double]
        L.eo_sample_neighbor_core.restype = C.c_int64
        L.eo_sample_neighbor_core.argtypes = [
            C.c_void_p, C.c_uint64, C.c_uint32, _u64p, C.c_int64, _i32p,
            C.c_int32, C.c_int32, _i32p, _u64p, _f32p, _i32p]
        L.eo_sample_neighbor_tf.argtypes = [
            C.c_void_p, C.c_uint64, C.c_uint32, _i64p, C.c_int64, _i32p,
            C.c_int32, C.c_int32, C.c_int64, _i64p, _f32p, _i32p]
        L.eo_sample_fanout_tf.argtypes = [
            C.c_void_p, C.c_uint64, C.c_uint32, _i64p, C.c_int64, _i32p,
            C.c_int32, _i32p, C.c_int32, C.c_int64, C.POINTER(_i64p),
            C.POINTER(_f32p), C.POINTER(_i32p)]
        L.eo_neighbor_post_process.restype = C.c_int64
        L.eo_neighbor_post_process.argtypes = [C.c_int64, _i32p, _u64p, _f32p, _i32p,
                                               C.c_int32, C.c_int32, C.c_int64]
        L.eo_neighbor_to_dense.argtypes = [C.c_int64, _i32p, _u64p, _f32p, _i32p,
                                           C.c_int32, C.c_int64, _i64p, _f32p, _i32p]
        L.eo_get_dense_feature.restype = C.c_int
        L.eo_get_dense_feature.argtypes = [C.c_void_p, C.c_void_p, _u64p, C.c_int64,
                                           C.c_int32, C.c_int32, _f32p]
        L.eo_get_full_neighbor.restype = C.c_int64
        L.eo_get_full_neighbor.argtypes = [C.c_void_p, _u64p, C.c_int64, _i32p,
                                           C.c_int32, _i32p, _u64p, _f32p, _i32p]
        L.eo_id_unique.restype = C.c_int64
        L.eo_id_unique.argtypes = [_u64p, C.c_int64, _u64p, _i32p]
        L.eo_idx_gather.argtypes = [_i32p, _i32p, C.c_int64, _i32p]
        L.eo_data_gather.restype = C.c_int64
        L.eo_data_gather.argtypes = [C.c_void_p, C.c_int32, _i32p, _i32p,
                                     C.c_int64, C.c_void_p]
        L.eo_alias_init.argtypes = [_f32p, C.c_int64, _f32p, _i64p]
        L.eo_node_sampler_create.restype = C.c_void_p
        L.eo_node_sampler_create.argtypes = [C.c_int64, _u64p, _i32p, _f32p,
                                             C.c_int32]
        L.eo_node_sampler_destroy.argtypes = [C.c_void_p]
        L.eo_sample_node.restype = C.c_int64
        L.eo_sample_node.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, _i32p,
                                     C.c_int32, C.c_int32, _u64p]
        L.eo_random_walk.argtypes = [
            C.c_void_p, C.c_uint64, C.c_uint32, _i64p, C.c_int64, _i32p,
            C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_int64, _i64p]
        L.eo_gen_pair_count.restype = C.c_int64
        L.eo_gen_pair_count.argtypes = [C.c_int64, C.c_int32, C.c_int32]
        L.eo_gen_pair.argtypes = [_i64p, C.c_int64, C.c_int64, C.c_int32,
                                  C.c_int32, _i64p]
        for f in (L.eo_scatter_add, L.eo_scatter_max):
            f.argtypes = [_f32p, _i32p, C.c_int64, C.c_int64, C.c_int32, _f32p]
        L.eo_gather.argtypes = [_f32p, _i32p, C.c_int64, C.c_int64, _f32p]
        L.eo_shard_of.restype = C.c_int32
        L.eo_shard_of.argtypes = [C.c_uint64, C.c_int32, C.c_int32]
        L.eo_id_split.argtypes = [_u64p, C.c_int64, C.c_int32, C.c_int32, _i64p,
                                  _u64p, _i32p]
        L.eo_sample_node_split.argtypes = [C.c_uint64, C.c_uint32, C.c_int32,
                                           _f32p, C.c_int32, _i32p]
        L.eo_build_prefix.argtypes = [C.c_int64, C.c_int32, _i64p, _f32p, _i64p,
                                      _i32p, _f32p, _f32p]
        L.eo_bench_fanout.restype = C.c_double
        L.eo_bench_fanout.argtypes = [C.c_void_p, C.c_uint64, _u64p, C.c_int64,
                                      C.c_int32, _i32p, C.c_int32, C.c_int32,
                                      _i64p]
        L.eo_synth_fill_table.argtypes = [C.c_void_p]
        L.eo_synth_external_id.restype = C.c_uint64
        L.eo_synth_external_id.argtypes = [C.c_void_p, C.c_uint64]
        L.eo_synth_degree.restype = C.c_int64
        L.eo_synth_degree.argtypes = [C.c_void_p, C.c_uint64]
        L.eo_synth_neighbor.restype = C.c_uint64
        L.eo_synth_neighbor.argtypes = [C.c_void_p, C.c_uint64, C.c_int64]
        L.eo_synth_weight.restype = C.c_float
        L.eo_synth_weight.argtypes = [C.c_void_p, C.c_uint64, C.c_int64]
        L.eo_synth_build.restype = C.c_int64
        L.eo_synth_build.argtypes = [C.c_void_p, C.c_int64, C.c_int64, _i64p,
                                     _i32p, _u64p, _f32p, _f32p]
        L.eo_philox_kat.argtypes = [C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                                    C.POINTER(C.c_uint32)]
        L.eo_std_hash_bytes.restype = C.c_uint64
        L.eo_std_hash_bytes.argtypes = [C.c_char_p, C.c_uint64]
        L.eo_umap_iteration_order.argtypes = [_u64p, C.c_int64, _i64p]
        L.eo_local_sample_layer.argtypes = [
            C.c_uint64, C.c_uint32, _i32p, C.c_int64, _u64p, _f32p, _i32p, C.c_int32,
            C.c_int32, C.c_int32, C.c_int64, _u64p, _f32p, _i32p]
        L.eo_get_sparse_feature.restype = C.c_int64
        L.eo_get_sparse_feature.argtypes = [C.c_void_p, C.c_void_p, _u64p, C.c_int64,
                                            C.c_int32, C.c_int64, _i64p, _i64p, _i64p]
        L.eo_sample_n_with_types.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, _i32p,
                                             C.c_int64, C.c_int32, _u64p]
        L.eo_get_edge_sum_weight.argtypes = [C.c_void_p, _u64p, C.c_int64, _i32p,
                                             C.c_int32, _f32p]
        L.eo_sample_root.argtypes = [C.c_uint64, C.c_uint32, _u64p, _f32p, C.c_int64,
                                     C.c_int32, C.c_int32, C.c_int64, _u64p]
        L.eo_sample_layer.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, _u64p,
                                      C.c_int64, _i32p, C.c_int32, C.c_int64, _u64p,
                                      _f32p, _i32p]
        L.eo_sample_layer_at.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, _u64p, _i64p,
                                         C.c_int64, _i32p, C.c_int32, C.c_int64, _u64p,
                                         _f32p, _i32p]
        L.eo_sparse_get_adj.restype = C.c_int64
        L.eo_sparse_get_adj.argtypes = [C.c_void_p, _u64p, _u64p, C.c_int64, C.c_int32,
                                        C.c_int32, _i32p, C.c_int32, _i32p, _u64p]
        L.eo_adj_to_sparse.restype = C.c_int64
        L.eo_adj_to_sparse.argtypes = [_u64p, _u64p, C.c_int64, C.c_int32, C.c_int32,
                                       _i32p, _u64p, _i64p, _i64p, _i64p]
    return _lib


def have_ref():
    return os.path.exists(REF_SO)


def ref():
    global _ref
    if _ref is None:
        _ref = C.CDLL(REF_SO)
        R = _ref
        R.euler_ref_graph_build.argtypes = [C.c_int64, _u64p, _i32p, _f32p,
                                            C.c_int32, C.c_int32, _i64p, _u64p,
                                            _f32p]
        R.euler_ref_graph_load.argtypes = [C.c_char_p]
        R.euler_ref_num_nodes.restype = C.c_int64
        R.euler_ref_node_order.restype = C.c_int64
        R.euler_ref_node_order.argtypes = [_u64p]
        R.euler_ref_node_info.argtypes = [_u64p, C.c_int64, _i32p, _f32p]
        R.euler_ref_row_shape.argtypes = [C.c_uint64, _i32p, _i64p]
        R.euler_ref_export_rows.restype = C.c_int64
        R.euler_ref_export_rows.argtypes = [_u64p, C.c_int64, C.c_int32, _i64p,
                                            _i32p, _u64p, _f32p, _f32p]
        R.euler_ref_sample_neighbor.restype = C.c_int64
        R.euler_ref_sample_neighbor.argtypes = [
            C.c_uint64, C.c_uint32, _u64p, C.c_int64, _i32p, C.c_int32,
            C.c_int32, _i32p, _u64p, _f32p, _i32p]
        R.euler_ref_sample_node.restype = C.c_int64
        R.euler_ref_sample_node.argtypes = [C.c_uint64, C.c_uint32, _i32p,
                                            C.c_int32, C.c_int32, _u64p]
        R.euler_ref_alias_size.restype = C.c_int64
        R.euler_ref_alias_size.argtypes = [C.c_int32]
        R.euler_ref_alias_table.argtypes = [C.c_int32, _u64p, _f32p, _f32p, _i64p,
                                            _f32p]
        R.euler_ref_get_full_neighbor.restype = C.c_int64
        R.euler_ref_get_full_neighbor.argtypes = [_u64p, C.c_int64, _i32p,
                                                  C.c_int32, _i32p, _u64p, _f32p,
                                                  _i32p]
        R.euler_ref_random_walk.argtypes = [
            C.c_uint64, C.c_uint32, _i64p, C.c_int64, _i32p, C.c_int32,
            C.c_int32, C.c_float, C.c_float, C.c_int64, _i64p]
        R.euler_ref_bench_fanout.restype = C.c_double
        R.euler_ref_bench_fanout.argtypes = [C.c_uint64, _u64p, C.c_int64,
                                             C.c_int32, _i32p, C.c_int32,
                                             C.c_int32, _i64p]
        R.euler_ref_set_rng.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32,
                                        C.c_uint64]
        if hasattr(R, "euler_ref_bench_fanout_dag"):
            R.euler_ref_bench_fanout_dag.restype = C.c_int64
            R.euler_ref_bench_fanout_dag.argtypes = [
                C.c_uint64, _u64p, C.c_int64, C.c_int64, _i32p, C.c_int32, C.c_int32,
                C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_double)]
            R.euler_ref_graph_build_mt.argtypes = [
                C.c_int64, _u64p, _i32p, _f32p, C.c_int32, C.c_int32, _i64p, _u64p, _f32p,
                C.c_int32, C.c_int32]
        R.euler_ref_get_neighbor.restype = C.c_int64
        R.euler_ref_get_neighbor.argtypes = [_u64p, C.c_int64, _i32p, C.c_int32,
                                             C.c_int32, C.c_int32, C.c_int64, _i32p,
                                             _u64p, _f32p, _i32p]
        R.euler_ref_set_float_features.argtypes = [_u64p, C.c_int64, C.c_int32,
                                                   _i64p, _i32p, _f32p]
        R.euler_ref_export_float_features.restype = C.c_int64
        R.euler_ref_export_float_features.argtypes = [_u64p, C.c_int64, C.c_int32,
                                                      _i64p, _i32p, _f32p]
        R.euler_ref_num_float_features.restype = C.c_int32
        R.euler_ref_get_dense_feature.argtypes = [_u64p, C.c_int64, C.c_int32,
                                                  C.c_int32, _f32p]
        R.euler_ref_set_u64_features.argtypes = [_u64p, C.c_int64, C.c_int32, _i64p,
                                                 _i32p, _u64p]
        R.euler_ref_export_u64_features.restype = C.c_int64
        R.euler_ref_export_u64_features.argtypes = [_u64p, C.c_int64, C.c_int32, _i64p,
                                                    _i32p, _u64p]
        R.euler_ref_num_u64_features.restype = C.c_int32
        R.euler_ref_get_sparse_feature.restype = C.c_int64
        R.euler_ref_get_sparse_feature.argtypes = [_u64p, C.c_int64, C.c_int32, C.c_int64,
                                                   _i64p, _i64p, _i64p]
        R.euler_ref_sample_n_with_types.argtypes = [C.c_uint64, C.c_uint32, _i32p,
                                                    C.c_int64, C.c_int32, _u64p]
        R.euler_ref_get_node_type.argtypes = [_u64p, C.c_int64, _i32p]
        R.euler_ref_umap_order.restype = C.c_int64
        R.euler_ref_umap_order.argtypes = [C.c_char_p, _i32p, C.c_int64, _i64p, _u64p]
        R.euler_ref_local_sample_layer.argtypes = [
            C.c_uint64, C.c_uint32, _i32p, C.c_int64, _u64p, _f32p, _i32p, C.c_int32,
            C.c_int32, C.c_char_p, C.c_int64, _u64p, _f32p, _i32p]
        R.euler_ref_graph_load_all.argtypes = [C.c_char_p]
        R.euler_ref_add_edges_from_adjacency.restype = C.c_int64
        R.euler_ref_num_edges.restype = C.c_int64
        R.euler_ref_edge_exist.argtypes = [C.c_uint64, C.c_uint64, C.c_int32]
        R.euler_ref_get_edge_sum_weight.argtypes = [_u64p, C.c_int64, _i32p, C.c_int32,
                                                    _f32p]
        R.euler_ref_sample_root.argtypes = [C.c_uint64, C.c_uint32, _u64p, _f32p,
                                            C.c_int64, C.c_int32, C.c_int32, C.c_int64,
                                            _u64p]
        R.euler_ref_sample_layer.argtypes = [C.c_uint64, C.c_uint32, _u64p, C.c_int64,
                                             _i32p, C.c_int32, C.c_int64, _u64p, _f32p,
                                             _i32p]
        R.euler_ref_sparse_get_adj.restype = C.c_int64
        R.euler_ref_sparse_get_adj.argtypes = [_u64p, _u64p, C.c_int64, C.c_int32,
                                               C.c_int32, _i32p, C.c_int32, _i32p, _u64p]
        R.euler_ref_adj_to_sparse.restype = C.c_int64
        R.euler_ref_adj_to_sparse.argtypes = [_u64p, _u64p, C.c_int64, C.c_int32,
                                              C.c_int32, _i32p, _u64p, _i64p, _i64p,
                                              _i64p]
    return _ref


# --------------------------------------------------------------------------
# Graph containers
# --------------------------------------------------------------------------
class CSR:
    """Reference-format adjacency (node.h:49-57) concatenated over rows."""

    def __init__(self, row_id, row_ptr, type_end, nbr, prefix_w, type_prefix,
                 n_types, node_type=None, node_weight=None):
        self.row_id = _arr(row_id, np.uint64)
        self.row_ptr = _arr(row_ptr, np.int64)
        self.type_end = _arr(type_end, np.int32).reshape(-1)
        self.nbr = _arr(nbr, np.uint64)
        self.prefix_w = _arr(prefix_w, np.float32)
        self.type_prefix = _arr(type_prefix, np.float32).reshape(-1)
        self.n_types = int(n_types)
        self.n_rows = len(self.row_id)
        self.node_type = (_arr(node_type, np.int32) if node_type is not None
                          else np.zeros(self.n_rows, np.int32))
        self.node_weight = (_arr(node_weight, np.float32)
                            if node_weight is not None
                            else np.ones(self.n_rows, np.float32))


ORDER = {None: 0, "": 0, "id": 1, "weight": 2}


def neighbor_post_process(idx, ids, w, t, order_by=None, desc=False, limit=None):
    """API_GET_NB_NODE post-process on a GQL-layout result (copies)."""
    idx = _arr(idx, np.int32).copy()
    ids = _arr(ids, np.uint64).copy()
    w = _arr(w, np.float32).copy()
    t = _arr(t, np.int32).copy()
    tot = lib().eo_neighbor_post_process(len(idx), _p(idx, _i32p), _p(ids, _u64p),
                                         _p(w, _f32p), _p(t, _i32p), ORDER[order_by],
                                         1 if desc else 0,
                                         -1 if limit is None else int(limit))
    return idx, ids[:tot], w[:tot], t[:tot]


def neighbor_to_dense(idx, ids, w, t, k, default_node=-1):
    idx = _arr(idx, np.int32)
    n = len(idx)
    ids = _arr(ids, np.uint64); w = _arr(w, np.float32); t = _arr(t, np.int32)
    oi = np.zeros((n, k), np.int64); ow = np.zeros((n, k), np.float32)
    ot = np.zeros((n, k), np.int32)
    lib().eo_neighbor_to_dense(n, _p(idx, _i32p), _p(ids, _u64p), _p(w, _f32p),
                               _p(t, _i32p), k, default_node, _p(oi, _i64p),
                               _p(ow, _f32p), _p(ot, _i32p))
    return oi, ow, ot


class Features(C.Structure):
    _fields_ = [("n_float", C.c_int32), ("feat_ptr", _i64p), ("feat_idx", _i32p),
                ("feat_val", _f32p)]


class DenseFeatures:
    """Per-node float features in the reference's storage form
    (float_features_idx_ = cumulative ends per slot, float_features_)."""

    def __init__(self, n_float, feat_ptr, feat_idx, feat_val):
        self.n_float = int(n_float)
        self.feat_ptr = _arr(feat_ptr, np.int64)
        self.feat_idx = _arr(feat_idx, np.int32).reshape(-1)
        self.feat_val = _arr(feat_val, np.float32)

    @staticmethod
    def from_lists(per_node):
        """per_node[i] = list (one entry per slot) of value lists."""
        F = max((len(x) for x in per_node), default=0)
        ptr, idx, val = [0], [], []
        for slots in per_node:
            end = 0
            for f in range(F):
                if f < len(slots):
                    val.extend(slots[f]); end += len(slots[f])
                idx.append(end)
            ptr.append(len(val))
        return DenseFeatures(F, ptr, idx, val)

    def c_struct(self):
        return Features(self.n_float, _p(self.feat_ptr, _i64p),
                        _p(self.feat_idx, _i32p), _p(self.feat_val, _f32p))


class U64Features(C.Structure):
    _fields_ = [("n_u64", C.c_int32), ("feat_ptr", _i64p), ("feat_idx", _i32p),
                ("feat_val", _u64p)]


class SparseFeatures:
    """uint64 features of every row in the reference's per-node form
    (uint64_features_idx_ / uint64_features_, node.h), concatenated."""

    def __init__(self, n_u64, feat_ptr, feat_idx, feat_val):
        self.n_u64 = int(n_u64)
        self.feat_ptr = _arr(feat_ptr, np.int64)
        self.feat_idx = _arr(feat_idx, np.int32).reshape(-1)
        self.feat_val = _arr(feat_val, np.uint64)

    @staticmethod
    def from_lists(per_node):
        U = max((len(x) for x in per_node), default=0)
        ptr, idx, val = [0], [], []
        for slots in per_node:
            end = 0
            for f in range(U):
                if f < len(slots):
                    val.extend(slots[f]); end += len(slots[f])
                idx.append(end)
            ptr.append(len(val))
        return SparseFeatures(U, ptr, idx, np.array(val, np.uint64))

    def c_struct(self):
        return U64Features(self.n_u64, _p(self.feat_ptr, _i64p),
                           _p(self.feat_idx, _i32p), _p(self.feat_val, _u64p))


def _sparse_feature_with(fn, head, nodes, fid, default_value):
    nodes = _arr(nodes, np.uint64)
    shape = np.zeros(2, np.int64)
    nnz = fn(*head, _p(nodes, _u64p), len(nodes), fid, default_value, None, None,
             _p(shape, _i64p))
    ind = np.zeros((nnz, 2), np.int64)
    val = np.zeros(nnz, np.int64)
    fn(*head, _p(nodes, _u64p), len(nodes), fid, default_value, _p(ind, _i64p),
       _p(val, _i64p), _p(shape, _i64p))
    return ind, val, shape


def csr_from_raw(row_id, seg_ptr, nbr, w, n_types, node_type=None,
                 node_weight=None):
    """RAW weights per (node,type) segment -> reference-format CSR via the
    restated Node::Init running sums."""
    row_id = _arr(row_id, np.uint64)
    seg_ptr = _arr(seg_ptr, np.int64)
    nbr = _arr(nbr, np.uint64)
    w = _arr(w, np.float32)
    n = len(row_id)
    row_ptr = np.zeros(n + 1, np.int64)
    type_end = np.zeros(n * n_types, np.int32)
    prefix = np.zeros(len(w), np.float32)
    tpre = np.zeros(n * n_types, np.float32)
    lib().eo_build_prefix(n, n_types, _p(seg_ptr, _i64p), _p(w, _f32p),
                          _p(row_ptr, _i64p), _p(type_end, _i32p),
                          _p(prefix, _f32p), _p(tpre, _f32p))
    return CSR(row_id, row_ptr, type_end, nbr, prefix, tpre, n_types,
               node_type, node_weight)



# --------------------------------------------------------------------------
# Layerwise sampling (sampleLNB without a weight function): the op chain
# API_GET_EDGE_SUM_WEIGHT -> API_SAMPLE_ROOT -> API_SAMPLE_L ->
# API_SPARSE_GET_ADJ + the TF kernel's sparse assembly.  `_LayerwiseMixin`
# composes the chain from the four primitives of whichever backend (C
# restatement / compiled reference) the class provides.
# --------------------------------------------------------------------------
def _adj_to_sparse_with(fn, nodes, nb_nodes, batch, n, m, idx, vals):
    nodes = _arr(np.asarray(nodes).reshape(-1), np.uint64)
    nb_nodes = _arr(np.asarray(nb_nodes).reshape(-1), np.uint64)
    idx = _arr(idx, np.int32)
    vals = _arr(vals, np.uint64)
    shape = np.zeros(3, np.int64)
    nnz = fn(_p(nodes, _u64p), _p(nb_nodes, _u64p), batch, n, m, _p(idx, _i32p),
             _p(vals, _u64p), None, None, _p(shape, _i64p))
    indices = np.zeros((nnz, 3), np.int64)
    values = np.zeros(nnz, np.int64)
    fn(_p(nodes, _u64p), _p(nb_nodes, _u64p), batch, n, m, _p(idx, _i32p),
       _p(vals, _u64p), _p(indices, _i64p), _p(values, _i64p), _p(shape, _i64p))
    return indices, values, shape


def adj_to_sparse(nodes, nb_nodes, batch, n, m, idx, vals):
    return _adj_to_sparse_with(lib().eo_adj_to_sparse, nodes, nb_nodes, batch, n, m,
                               idx, vals)


def sample_root(seed, call_id, roots, weights, n, m, default_node=-1):
    roots = _arr(np.asarray(roots).reshape(-1), np.uint64)
    weights = _arr(np.asarray(weights).reshape(-1), np.float32)
    batch = len(roots) // n
    out = np.zeros(batch * m, np.uint64)
    lib().eo_sample_root(seed, call_id, _p(roots, _u64p), _p(weights, _f32p), batch,
                         n, m, default_node, _p(out, _u64p))
    return out


def std_hash(key):
    """libstdc++ std::hash<std::string> restated (oracle/eo_umap.c)."""
    key = key if isinstance(key, bytes) else key.encode()
    return int(lib().eo_std_hash_bytes(key, len(key)))


def umap_iteration_order(keys):
    """Restated iteration order of std::unordered_map<std::string, ...> after
    inserting the DISTINCT keys in order."""
    h = np.array([std_hash(k) for k in keys], np.uint64)
    order = np.zeros(len(keys), np.int64)
    lib().eo_umap_iteration_order(_p(h, _u64p), len(keys), _p(order, _i64p))
    return order


def ref_umap_iteration_order(keys):
    """The real container inside oracle/_ref: (order over the distinct keys,
    std::hash of every key)."""
    bs = [k if isinstance(k, bytes) else k.encode() for k in keys]
    lens = np.array([len(b) for b in bs], np.int32)
    order = np.zeros(len(bs), np.int64)
    hashes = np.zeros(len(bs), np.uint64)
    d = ref().euler_ref_umap_order(b"".join(bs), _p(lens, _i32p), len(bs), _p(order, _i64p),
                                   _p(hashes, _u64p))
    return order[:d], hashes


class _LayerwiseMixin:
    def sparse_get_adj_tf(self, nodes, nb_nodes, edge_types, n=-1, m=-1):
        """TF SparseGetAdj (tf_euler/kernels/sparse_get_adj_op.cc:43-134):
        (indices [nnz,3], values [nnz], dense_shape [3])."""
        nodes = np.asarray(nodes).reshape(-1)
        nb_nodes = np.asarray(nb_nodes).reshape(-1)
        if n == -1:
            n = len(nodes)
        if m == -1:
            m = len(nb_nodes)
        batch = len(nodes) // n if n else 0
        idx, vals = self.sparse_get_adj(nodes, nb_nodes, batch, n, m, edge_types)
        return self._adj_to_sparse(nodes, nb_nodes, batch, n, m, idx, vals)

    def sample_neighbor_layerwise_func(self, seed, call_id, nodes, edge_types, count,
                                       weight_func, default_node=-1):
        """sampleLNB with a weight function: API_GET_NB_NODE -> API_LOCAL_SAMPLE_L
        -> adjacency (translator.cc:388-441,489-527)."""
        nodes = np.asarray(nodes)
        batch, n = nodes.shape
        flat = _arr(nodes.reshape(-1), np.uint64)
        idx, ids, w, t = self.get_full_neighbor(flat, edge_types)
        l_nb, l_w, l_t = self.local_sample_layer(seed, call_id, idx, ids, w, t, n, count,
                                                 weight_func, default_node)
        aidx, avals = self.sparse_get_adj(flat, l_nb, batch, n, count, edge_types)
        ind, val, shape = self._adj_to_sparse(flat, l_nb, batch, n, count, aidx, avals)
        return (l_nb.view(np.int64).reshape(batch, count), l_w.reshape(batch, count),
                l_t.reshape(batch, count), ind, val, shape)

    def sample_neighbor_layerwise(self, seed, call_id, nodes, edge_types, count,
                                  default_node=-1):
        """TF SampleNeighborLayerwiseWithAdj with weight_func == '':
        nodes [batch, n] -> (neighbors [batch, count] int64, indices, values,
        dense_shape)."""
        nodes = np.asarray(nodes)
        batch, n = nodes.shape
        flat = _arr(nodes.reshape(-1), np.uint64)
        w = self.get_edge_sum_weight(flat, edge_types)
        l_root = self._sample_root(seed, call_id, flat, w, n, count, default_node)
        l_nb, _, _ = self.sample_layer(seed, call_id, l_root, edge_types, default_node)
        idx, vals = self.sparse_get_adj(flat, l_nb, batch, n, count, edge_types)
        ind, val, shape = self._adj_to_sparse(flat, l_nb, batch, n, count, idx, vals)
        return l_nb.view(np.int64).reshape(batch, count), ind, val, shape

class OracleGraph(_LayerwiseMixin):
    """The C restatement bound to one CSR."""

    def __init__(self, csr):
        self.csr = csr
        c = csr
        self.h = lib().eo_graph_create(
            c.n_rows, c.n_types, _p(c.row_id, _u64p), _p(c.row_ptr, _i64p),
            _p(c.type_end, _i32p), _p(c.nbr, _u64p), _p(c.prefix_w, _f32p),
            _p(c.type_prefix, _f32p))
        self._sampler = None

    def __del__(self):
        try:
            if self._sampler:
                lib().eo_node_sampler_destroy(self._sampler)
            lib().eo_graph_destroy(self.h)
        except Exception:
            pass

    def get_dense_feature(self, feats, nodes, feature_ids, dimensions):
        """tf_euler get_dense_feature: list of [n, dim] f32 arrays; `feats` is
        a DenseFeatures aligned with the CSR rows."""
        ids = _arr(nodes, np.int64).astype(np.uint64)
        fs = feats.c_struct()
        outs = []
        for fid, dim in zip(feature_ids, dimensions):
            out = np.zeros((len(ids), dim), np.float32)
            rc = lib().eo_get_dense_feature(self.h, C.byref(fs), _p(ids, _u64p), len(ids),
                                            int(fid), int(dim), _p(out, _f32p))
            assert rc == 0, rc
            outs.append(out)
        return outs

    def sample_neighbor_core(self, seed, call_id, ids, edge_types, count):
        ids = _arr(ids, np.uint64)
        et = _arr(edge_types, np.int32)
        n = len(ids)
        idx = np.zeros((n, 2), np.int32)
        oid = np.zeros(n * count, np.uint64)
        ow = np.zeros(n * count, np.float32)
        ot = np.zeros(n * count, np.int32)
        lib().eo_sample_neighbor_core(self.h, seed, call_id, _p(ids, _u64p), n,
                                      _p(et, _i32p), len(et), count,
                                      _p(idx, _i32p), _p(oid, _u64p),
                                      _p(ow, _f32p), _p(ot, _i32p))
        return idx, oid, ow, ot

    def sample_neighbor(self, seed, call_id, nodes, edge_types, count,
                        default_node=-1):
        nodes = _arr(nodes, np.int64)
        et = _arr(edge_types, np.int32)
        n = len(nodes)
        on = np.zeros((n, count), np.int64)
        ow = np.zeros((n, count), np.float32)
        ot = np.zeros((n, count), np.int32)
        lib().eo_sample_neighbor_tf(self.h, seed, call_id, _p(nodes, _i64p), n,
                                    _p(et, _i32p), len(et), count, default_node,
                                    _p(on, _i64p), _p(ow, _f32p), _p(ot, _i32p))
        return on, ow, ot

    def sample_fanout(self, seed, call_id, nodes, edge_types, counts,
                      default_node=-1):
        nodes = _arr(nodes, np.int64).reshape(-1)
        et = _arr(edge_types, np.int32)
        layers = len(counts)
        et = et.reshape(layers, -1)
        cnt = _arr(counts, np.int32)
        n = len(nodes)
        outs_n, outs_w, outs_t = [], [], []
        m = n
        for c in counts:
            m *= c
            outs_n.append(np.zeros(m, np.int64))
            outs_w.append(np.zeros(m, np.float32))
            outs_t.append(np.zeros(m, np.int32))
        pn = (_i64p * layers)(*[_p(a, _i64p) for a in outs_n])
        pw = (_f32p * layers)(*[_p(a, _f32p) for a in outs_w])
        pt = (_i32p * layers)(*[_p(a, _i32p) for a in outs_t])
        lib().eo_sample_fanout_tf(self.h, seed, call_id, _p(nodes, _i64p), n,
                                  _p(et, _i32p), et.shape[1], _p(cnt, _i32p),
                                  layers, default_node, pn, pw, pt)
        return outs_n, outs_w, outs_t

    def get_full_neighbor(self, ids, edge_types):
        ids = _arr(ids, np.uint64)
        et = _arr(edge_types, np.int32)
        n = len(ids)
        tot = lib().eo_get_full_neighbor(self.h, _p(ids, _u64p), n,
                                         _p(et, _i32p), len(et), None, None,
                                         None, None)
        idx = np.zeros((n, 2), np.int32)
        oid = np.zeros(tot, np.uint64)
        ow = np.zeros(tot, np.float32)
        ot = np.zeros(tot, np.int32)
        lib().eo_get_full_neighbor(self.h, _p(ids, _u64p), n, _p(et, _i32p),
                                   len(et), _p(idx, _i32p), _p(oid, _u64p),
                                   _p(ow, _f32p), _p(ot, _i32p))
        return idx, oid, ow, ot

    def get_sparse_feature(self, feats, nodes, feature_ids, default_values=None):
        """[(indices, values, dense_shape)] per feature id (TF GetSparseFeature)."""
        st = feats.c_struct()
        dv = [0] * len(feature_ids) if default_values is None else default_values
        return [_sparse_feature_with(lib().eo_get_sparse_feature, (self.h, C.byref(st)),
                                     nodes, int(f), int(d))
                for f, d in zip(feature_ids, dv)]

    def get_node_type(self, ids):
        """euler::GetNodeType (api.cc:50-61): DEFAULT_INT32 for unknown ids."""
        ids = _arr(ids, np.uint64)
        out = np.full(len(ids), -2 ** 31, np.int32)
        for i, v in enumerate(ids):
            r = lib().eo_graph_find_row(self.h, int(v))
            if r >= 0:
                out[i] = self.csr.node_type[r]
        return out

    def sample_n_with_types(self, seed, call_id, types, count):
        """[len(types), count] ids, or None where the TF kernel aborts."""
        types = _arr(types, np.int32)
        out = np.zeros((len(types), count), np.uint64)
        rc = lib().eo_sample_n_with_types(self._sampler, seed, call_id,
                                          _p(types, _i32p), len(types), count,
                                          _p(out, _u64p))
        return out if rc == 0 else None

    def local_sample_layer(self, seed, call_id, idx, ids, w, t, n, m, weight_func="sqrt",
                           default_node=-1):
        """API_LOCAL_SAMPLE_L with libstdc++'s container order restated in C."""
        idx = _arr(idx, np.int32).reshape(-1)
        ids, w, t = _arr(ids, np.uint64), _arr(w, np.float32), _arr(t, np.int32)
        batch = len(idx) // (2 * n)
        oid = np.zeros(batch * m, np.uint64)
        ow = np.zeros(batch * m, np.float32)
        ot = np.zeros(batch * m, np.int32)
        lib().eo_local_sample_layer(seed, call_id, _p(idx, _i32p), len(idx), _p(ids, _u64p),
                                    _p(w, _f32p), _p(t, _i32p), n, m,
                                    1 if weight_func == "sqrt" else 0, default_node,
                                    _p(oid, _u64p), _p(ow, _f32p), _p(ot, _i32p))
        return oid, ow, ot

    # ---- layerwise primitives (C restatement)
    _adj_to_sparse = staticmethod(adj_to_sparse)
    _sample_root = staticmethod(sample_root)

    def get_edge_sum_weight(self, ids, edge_types):
        ids = _arr(ids, np.uint64)
        et = _arr(edge_types, np.int32)
        out = np.zeros(len(ids), np.float32)
        lib().eo_get_edge_sum_weight(self.h, _p(ids, _u64p), len(ids), _p(et, _i32p),
                                     len(et), _p(out, _f32p))
        return out

    def sample_layer(self, seed, call_id, roots, edge_types, default_node=-1,
                     positions=None):
        roots = _arr(roots, np.uint64)
        et = _arr(edge_types, np.int32)
        n = len(roots)
        pos = None if positions is None else _arr(positions, np.int64)
        oid = np.zeros(n, np.uint64)
        ow = np.zeros(n, np.float32)
        ot = np.zeros(n, np.int32)
        lib().eo_sample_layer_at(self.h, seed, call_id, _p(roots, _u64p),
                                 None if pos is None else _p(pos, _i64p), n,
                                 _p(et, _i32p), len(et), default_node, _p(oid, _u64p),
                                 _p(ow, _f32p), _p(ot, _i32p))
        return oid, ow, ot

    def sparse_get_adj(self, roots, l_nb, batch, n, m, edge_types):
        roots = _arr(np.asarray(roots).reshape(-1), np.uint64)
        l_nb = _arr(np.asarray(l_nb).reshape(-1), np.uint64)
        et = _arr(edge_types, np.int32)
        idx = np.zeros((batch * n, 2), np.int32)
        tot = lib().eo_sparse_get_adj(self.h, _p(roots, _u64p), _p(l_nb, _u64p), batch,
                                      n, m, _p(et, _i32p), len(et), _p(idx, _i32p),
                                      None)
        vals = np.zeros(tot, np.uint64)
        lib().eo_sparse_get_adj(self.h, _p(roots, _u64p), _p(l_nb, _u64p), batch, n, m,
                                _p(et, _i32p), len(et), _p(idx, _i32p),
                                _p(vals, _u64p))
        return idx, vals

    def random_walk(self, seed, call_id, nodes, edge_types, walk_len, p=1.0,
                    q=1.0, default_node=-1):
        nodes = _arr(nodes, np.int64)
        et = _arr(edge_types, np.int32).reshape(walk_len, -1)
        n = len(nodes)
        out = np.zeros((n, walk_len + 1), np.int64)
        lib().eo_random_walk(self.h, seed, call_id, _p(nodes, _i64p), n,
                             _p(et, _i32p), et.shape[1], walk_len, p, q,
                             default_node, _p(out, _i64p))
        return out

    def build_node_sampler(self, order=None):
        """order: node ids in the order the reference iterates node_map_
        (defaults to row order)."""
        c = self.csr
        if order is None:
            ids, types, weights = c.row_id, c.node_type, c.node_weight
        else:
            order = _arr(order, np.uint64)
            pos = {int(v): i for i, v in enumerate(c.row_id)}
            sel = np.array([pos[int(v)] for v in order], np.int64)
            ids = order
            types = _arr(c.node_type[sel], np.int32)
            weights = _arr(c.node_weight[sel], np.float32)
        n_types = int(types.max()) + 1 if len(types) else 1
        if self._sampler:
            lib().eo_node_sampler_destroy(self._sampler)
        self._keep = (ids, types, weights)
        self._sampler = lib().eo_node_sampler_create(
            len(ids), _p(ids, _u64p), _p(types, _i32p), _p(weights, _f32p),
            n_types)
        self.n_node_types = n_types

    def sample_node(self, seed, call_id, node_types, count):
        nt = _arr(np.atleast_1d(node_types), np.int32)
        out = np.zeros(max(count, 1), np.uint64)
        got = lib().eo_sample_node(self._sampler, seed, call_id, _p(nt, _i32p),
                                   len(nt), count, _p(out, _u64p))
        return out[:max(got, 0)]

    def bench_fanout(self, seed, roots, batch, iters, counts, threads):
        roots = _arr(roots, np.uint64)
        cnt = _arr(counts, np.int32)
        edges = C.c_int64(0)
        secs = lib().eo_bench_fanout(self.h, seed, _p(roots, _u64p), batch,
                                     iters, _p(cnt, _i32p), len(cnt), threads,
                                     C.byref(edges))
        return secs, edges.value


# --------------------------------------------------------------------------
# Stateless helpers of the restatement
# --------------------------------------------------------------------------
def uniform_at(seed, call_id, domain, stream, draw_idx):
    return lib().eo_uniform_at(seed, call_id, domain, stream, draw_idx)


def philox(ctr, key):
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    lib().eo_philox_kat(c, k, o)
    return list(o)


def random_select(sum_weights, begin, end, u):
    sw = _arr(sum_weights, np.float32)
    return lib().eo_random_select(_p(sw, _f32p), begin, end, u)


def node2vec_step_lists(seed, call_id, c_row, c_idx, c_ids, c_w, p_row, p_idx, p_ids,
                        parent_ids, p, q, default_node=-1):
    """One node2vec step over explicit neighbour lists, the reference's client code restated
    (tf_euler/kernels/random_walk_op.cc:83-138 RWCallback::operator(), :140-168
    BuildWeights, euler/common/compact_weighted_collection.h:84-152 Init / Sample over
    RandomSelect :30-52).  Pure Python: small cases.  Walker i's child list is row c_row[i]
    of (c_idx, c_ids, c_w), its parent's list row p_row[i] of (p_idx, p_ids) (p_row None:
    no parent lists, the first step); draw = eo_uniform_at(seed, call_id, WALK = 2,
    stream i, draw 0), as the seam defines it for the walk."""
    c_row = np.asarray(c_row).reshape(-1)
    c_idx = np.asarray(c_idx).reshape(-1, 2)
    c_ids = np.asarray(c_ids).astype(np.int64).reshape(-1)
    c_w = np.asarray(c_w, dtype=np.float32).reshape(-1)
    parent_ids = np.asarray(parent_ids).astype(np.int64).reshape(-1)
    if p_row is not None:
        p_row = np.asarray(p_row).reshape(-1)
        p_idx = np.asarray(p_idx).reshape(-1, 2)
        p_ids = np.asarray(p_ids).astype(np.int64).reshape(-1)
    pf, qf = np.float32(p), np.float32(q)
    out = np.full(len(c_row), default_node, np.int64)
    for i in range(len(c_row)):
        b, e = int(c_idx[c_row[i], 0]), int(c_idx[c_row[i], 1])
        if e <= b:
            continue
        cn = c_ids[b:e]
        w = c_w[b:e].copy()
        pn = p_ids[int(p_idx[p_row[i], 0]):int(p_idx[p_row[i], 1])] if p_row is not None \
            else np.zeros(0, np.int64)
        parent = parent_ids[i]
        j = k = 0                                   # BuildWeights, :140-168
        while j < len(cn) and k < len(pn):
            if cn[j] < pn[k]:
                w[j] = w[j] / (qf if cn[j] != parent else pf)
                j += 1
            elif cn[j] == pn[k]:
                k += 1
                j += 1
            else:
                k += 1
        while j < len(cn):
            w[j] = w[j] / (qf if cn[j] != parent else pf)
            j += 1
        sums = np.zeros(len(cn), np.float32)        # CompactWeightedCollection::Init
        acc = np.float32(0)
        for x in range(len(cn)):
            acc = np.float32(acc + w[x])
            sums[x] = acc
        u = uniform_at(seed, call_id, 2, i, 0)
        out[i] = cn[random_select(sums, 0, len(cn) - 1, u)]
    return out


def id_unique(ids):
    ids = _arr(ids, np.uint64)
    uq = np.zeros(len(ids), np.uint64)
    gi = np.zeros(len(ids), np.int32)
    n = lib().eo_id_unique(_p(ids, _u64p), len(ids), _p(uq, _u64p),
                           _p(gi, _i32p))
    return uq[:n], gi


def idx_gather(idx, gather_idx):
    idx = _arr(idx, np.int32)
    gi = _arr(gather_idx, np.int32)
    out = np.zeros((len(gi), 2), np.int32)
    lib().eo_idx_gather(_p(idx, _i32p), _p(gi, _i32p), len(gi), _p(out, _i32p))
    return out


def data_gather(data, idx, gather_idx):
    data = np.ascontiguousarray(data)
    idx = _arr(idx, np.int32)
    gi = _arr(gather_idx, np.int32)
    tot = lib().eo_data_gather(data.ctypes.data, data.itemsize, _p(idx, _i32p),
                               _p(gi, _i32p), len(gi), None)
    out = np.zeros(tot, data.dtype)
    lib().eo_data_gather(data.ctypes.data, data.itemsize, _p(idx, _i32p),
                         _p(gi, _i32p), len(gi), out.ctypes.data)
    return out


def sparse_gather(gather_idx, indices, values, dense_shape):
    """tf_euler/kernels/sparse_gather_op.cc:136-184 (GatherWithBinarySearch) restated in numpy:
    rows of a SparseTensor sorted by row, gathered; (out_indices, out_values, out_dense_shape)."""
    gi = np.asarray(gather_idx, np.int64).reshape(-1)
    ind = np.asarray(indices, np.int64)
    val = np.asarray(values)
    rows = int(dense_shape[0])
    out_i, out_v = [], []
    for g, r in enumerate(gi):
        if r >= rows:
            raise IndexError("SparseGather: gather idx out of range.")
        lo = int(np.searchsorted(ind[:, 0], r, side="left"))
        hi = int(np.searchsorted(ind[:, 0], r + 1, side="left"))
        for j in range(lo, hi):
            out_i.append([g] + [int(x) for x in ind[j, 1:]])
            out_v.append(val[j])
    oi = np.asarray(out_i, np.int64).reshape(-1, ind.shape[1])
    ov = np.asarray(out_v, val.dtype)
    return oi, ov, np.asarray([len(gi)] + [int(x) for x in dense_shape[1:]], np.int64)


def inflate_idx(idx):
    """tf_euler/kernels/inflate_idx_op.cc:34-66 restated: count per value, exclusive prefix sums,
    places handed out in input order.  ValueError where the reference returns InvalidArgument."""
    a = [int(x) for x in np.asarray(idx, np.int64).reshape(-1)]
    unique_cnt = len(set(a))
    sub_cnt = [0] * unique_cnt
    for v in a:
        if not 0 <= v < unique_cnt:
            raise ValueError("expect input idx in [0,unique_cnt).")
        sub_cnt[v] += 1
    off = [0] * unique_cnt
    for i in range(1, unique_cnt):
        off[i] = off[i - 1] + sub_cnt[i - 1]
    out = []
    for v in a:
        out.append(off[v])
        off[v] += 1
    return np.asarray(out, np.int32)


def alias_init(weights):
    w = _arr(weights, np.float32)
    prob = np.zeros(len(w), np.float32)
    alias = np.zeros(len(w), np.int64)
    lib().eo_alias_init(_p(w, _f32p), len(w), _p(prob, _f32p), _p(alias, _i64p))
    return prob, alias


def gen_pair(paths, left, right):
    paths = _arr(paths, np.int64)
    b, l = paths.shape
    pc = lib().eo_gen_pair_count(l, left, right)
    out = np.zeros((b, pc, 2), np.int64)
    lib().eo_gen_pair(_p(paths, _i64p), b, l, left, right, _p(out, _i64p))
    return out


def scatter_add(updates, indices, size):
    u = _arr(updates, np.float32)
    i = _arr(indices, np.int32)
    out = np.zeros((size, u.shape[1]), np.float32)
    lib().eo_scatter_add(_p(u, _f32p), _p(i, _i32p), u.shape[0], u.shape[1],
                         size, _p(out, _f32p))
    return out


def scatter_max(updates, indices, size):
    u = _arr(updates, np.float32)
    i = _arr(indices, np.int32)
    out = np.zeros((size, u.shape[1]), np.float32)
    lib().eo_scatter_max(_p(u, _f32p), _p(i, _i32p), u.shape[0], u.shape[1],
                         size, _p(out, _f32p))
    return out


def gather(params, indices):
    p = _arr(params, np.float32)
    i = _arr(indices, np.int32)
    out = np.zeros((len(i), p.shape[1]), np.float32)
    lib().eo_gather(_p(p, _f32p), _p(i, _i32p), len(i), p.shape[1],
                    _p(out, _f32p))
    return out


def scatter_mean(updates, indices, size):
    """mp_ops.py:65-69: add / (count + 1e-7), all fp32."""
    u = _arr(updates, np.float32)
    out = scatter_add(u, indices, size)
    cnt = scatter_add(np.ones((u.shape[0], 1), np.float32), indices, size)
    return out / (cnt + np.float32(1e-7))


def scatter_softmax(updates, indices, size):
    """mp_ops.py:76-79."""
    u = _arr(updates, np.float32)
    u = u - gather(scatter_max(u, indices, size), indices)
    u = np.exp(u)
    return u / gather(scatter_add(u, indices, size), indices)


def shard_of(ids, partitions, shards):
    ids = _arr(ids, np.uint64)
    return ((ids % np.uint64(partitions)) % np.uint64(shards)).astype(np.int32)


def id_split(ids, partitions, shards):
    ids = _arr(ids, np.uint64)
    off = np.zeros(shards + 1, np.int64)
    sid = np.zeros(len(ids), np.uint64)
    mi = np.zeros(len(ids), np.int32)
    lib().eo_id_split(_p(ids, _u64p), len(ids), partitions, shards,
                      _p(off, _i64p), _p(sid, _u64p), _p(mi, _i32p))
    return off, sid, mi


def sample_node_split(seed, call_id, count, shard_weight):
    sw = _arr(shard_weight, np.float32)
    shards = len(sw) - 1
    out = np.zeros(shards, np.int32)
    lib().eo_sample_node_split(seed, call_id, count, _p(sw, _f32p), shards,
                               _p(out, _i32p))
    return out


# --------------------------------------------------------------------------
# Synthetic graph (eo_synth.c)
# --------------------------------------------------------------------------
class SynthParams(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("n_nodes", C.c_int64),
                ("n_edges_target", C.c_int64), ("scale", C.c_int32),
                ("n_types", C.c_int32), ("weighted", C.c_int32),
                ("hashed_ids", C.c_int32), ("deg_table", C.c_double * 64)]


_M64 = (1 << 64) - 1


def synth_external_ids(p, x):
    """External ids of the internal node numbers x (uint64 array): x itself, or - hashed_ids -
    mix64(x), eo_synth.c's bijection (numpy restatement, checked against the C in the tests)."""
    x = np.asarray(x, dtype=np.uint64)
    if not p.hashed_ids:
        return x.copy()
    with np.errstate(over="ignore"):
        z = x.copy()
        z ^= z >> np.uint64(30); z *= np.uint64(0xbf58476d1ce4e5b9)
        z ^= z >> np.uint64(27); z *= np.uint64(0x94d049bb133111eb)
        z ^= z >> np.uint64(31)
    return z


def synth_internal_id(p, ext):
    """Inverse of synth_external_ids for one id (Python ints)."""
    z = int(ext) & _M64
    if not p.hashed_ids:
        return z

    def unxorshift(v, s):
        r = v
        for _ in range(64 // s + 1):
            r = v ^ (r >> s)
        return r
    z = unxorshift(z, 31)
    z = (z * pow(0x94d049bb133111eb, -1, 1 << 64)) & _M64
    z = unxorshift(z, 27)
    z = (z * pow(0xbf58476d1ce4e5b9, -1, 1 << 64)) & _M64
    z = unxorshift(z, 30)
    return z


def synth_params(seed, n_nodes, n_edges, n_types=1, weighted=True, scale=None, hashed_ids=False):
    p = SynthParams()
    p.hashed_ids = 1 if hashed_ids else 0
    p.seed = seed
    p.n_nodes = n_nodes
    p.n_edges_target = n_edges
    if scale is None:
        scale = max(1, int(np.ceil(np.log2(max(n_nodes, 2)))))
    p.scale = scale
    p.n_types = n_types
    p.weighted = 1 if weighted else 0
    lib().eo_synth_fill_table(C.byref(p))
    return p


def synth_csr(p, row_begin=0, row_end=None, threads=1):
    """Rows [row_begin,row_end) of the synthetic graph as a CSR (ids row+1).
    threads > 1: row ranges are generated concurrently (every quantity is a pure
    function of (seed, node, slot); ctypes releases the GIL)."""
    if row_end is None:
        row_end = p.n_nodes
    n = row_end - row_begin
    if threads > 1 and n >= 4 * threads:
        from concurrent.futures import ThreadPoolExecutor
        L = lib()
        cuts = [row_begin + n * i // threads for i in range(threads + 1)]
        with ThreadPoolExecutor(threads) as ex:
            tots = list(ex.map(lambda i: L.eo_synth_build(C.byref(p), cuts[i], cuts[i + 1], None,
                                                          None, None, None, None),
                               range(threads)))
            offs = np.concatenate([[0], np.cumsum(tots)]).astype(np.int64)
            tot = int(offs[-1])
            T = p.n_types
            row_ptr = np.zeros(n + 1, np.int64)
            type_end = np.zeros(n * T, np.int32)
            nbr = np.zeros(tot, np.uint64)
            prefix = np.zeros(tot, np.float32)
            tpre = np.zeros(n * T, np.float32)
            tmp_ptr = [np.zeros(cuts[i + 1] - cuts[i] + 1, np.int64) for i in range(threads)]

            def fill(i):
                r0 = cuts[i] - row_begin
                m = cuts[i + 1] - cuts[i]
                L.eo_synth_build(C.byref(p), cuts[i], cuts[i + 1], _p(tmp_ptr[i], _i64p),
                                 _p(type_end[r0 * T:(r0 + m) * T], _i32p),
                                 _p(nbr[offs[i]:offs[i + 1]], _u64p),
                                 _p(prefix[offs[i]:offs[i + 1]], _f32p),
                                 _p(tpre[r0 * T:(r0 + m) * T], _f32p))
                row_ptr[r0:r0 + m] = tmp_ptr[i][:m] + offs[i]
            list(ex.map(fill, range(threads)))
        row_ptr[n] = tot
        row_id = synth_external_ids(p, np.arange(row_begin + 1, row_end + 1, dtype=np.uint64))
        return CSR(row_id, row_ptr, type_end, nbr, prefix, tpre, p.n_types)
    tot = lib().eo_synth_build(C.byref(p), row_begin, row_end, None, None, None,
                               None, None)
    row_ptr = np.zeros(n + 1, np.int64)
    type_end = np.zeros(n * p.n_types, np.int32)
    nbr = np.zeros(tot, np.uint64)
    prefix = np.zeros(tot, np.float32)
    tpre = np.zeros(n * p.n_types, np.float32)
    lib().eo_synth_build(C.byref(p), row_begin, row_end, _p(row_ptr, _i64p),
                         _p(type_end, _i32p), _p(nbr, _u64p), _p(prefix, _f32p),
                         _p(tpre, _f32p))
    row_id = synth_external_ids(p, np.arange(row_begin + 1, row_end + 1, dtype=np.uint64))
    return CSR(row_id, row_ptr, type_end, nbr, prefix, tpre, p.n_types)


# --------------------------------------------------------------------------
# Reference library (oracle/_ref)
# --------------------------------------------------------------------------
class RefGraph(_LayerwiseMixin):
    """The REFERENCE graph singleton, loaded through the harness."""

    @staticmethod
    def build_raw(row_id, seg_ptr, nbr, w, n_types, node_type=None,
                  node_weight=None, threads=1, build_sampler=True):
        """threads > 1: the Node objects are constructed by that many threads
        (euler_ref_graph_build_mt; same graph)."""
        row_id = _arr(row_id, np.uint64)
        n = len(row_id)
        nt = (_arr(node_type, np.int32) if node_type is not None
              else np.zeros(n, np.int32))
        nw = (_arr(node_weight, np.float32) if node_weight is not None
              else np.ones(n, np.float32))
        seg_ptr = _arr(seg_ptr, np.int64)
        nbr = _arr(nbr, np.uint64)
        w = _arr(w, np.float32)
        n_node_types = int(nt.max()) + 1 if n else 1
        if threads > 1 or not build_sampler:
            rc = ref().euler_ref_graph_build_mt(
                n, _p(row_id, _u64p), _p(nt, _i32p), _p(nw, _f32p), n_types, n_node_types,
                _p(seg_ptr, _i64p), _p(nbr, _u64p), _p(w, _f32p), int(threads),
                1 if build_sampler else 0)
        else:
            rc = ref().euler_ref_graph_build(n, _p(row_id, _u64p), _p(nt, _i32p),
                                             _p(nw, _f32p), n_types, n_node_types,
                                             _p(seg_ptr, _i64p), _p(nbr, _u64p),
                                             _p(w, _f32p))
        assert rc == 0
        return RefGraph(n_types)

    @staticmethod
    def load(path, n_types):
        rc = ref().euler_ref_graph_load(path.encode())
        assert rc == 0
        return RefGraph(n_types)

    def __init__(self, n_types):
        self.n_types = n_types

    def set_u64_features(self, row_id, feats):
        row_id = _arr(row_id, np.uint64)
        rc = ref().euler_ref_set_u64_features(
            _p(row_id, _u64p), len(row_id), feats.n_u64, _p(feats.feat_ptr, _i64p),
            _p(feats.feat_idx, _i32p), _p(feats.feat_val, _u64p))
        assert rc == 0

    def export_u64_features(self, ids):
        ids = _arr(ids, np.uint64)
        U = ref().euler_ref_num_u64_features()
        n = len(ids)
        tot = ref().euler_ref_export_u64_features(_p(ids, _u64p), n, U, None, None, None)
        ptr = np.zeros(n + 1, np.int64)
        idx = np.zeros(n * max(U, 1), np.int32)
        val = np.zeros(max(tot, 1), np.uint64)
        ref().euler_ref_export_u64_features(_p(ids, _u64p), n, U, _p(ptr, _i64p),
                                            _p(idx, _i32p), _p(val, _u64p))
        return SparseFeatures(U, ptr, idx[:n * U], val[:tot])

    def get_sparse_feature(self, nodes, feature_ids, default_values=None):
        dv = [0] * len(feature_ids) if default_values is None else default_values
        return [_sparse_feature_with(ref().euler_ref_get_sparse_feature, (), nodes,
                                     int(f), int(d))
                for f, d in zip(feature_ids, dv)]

    def get_node_type(self, ids):
        ids = _arr(ids, np.uint64)
        out = np.zeros(len(ids), np.int32)
        ref().euler_ref_get_node_type(_p(ids, _u64p), len(ids), _p(out, _i32p))
        return out

    def sample_n_with_types(self, seed, call_id, types, count):
        types = _arr(types, np.int32)
        out = np.zeros((len(types), count), np.uint64)
        rc = ref().euler_ref_sample_n_with_types(seed, call_id, _p(types, _i32p),
                                                 len(types), count, _p(out, _u64p))
        return out if rc == 0 else None

    # ---- layerwise primitives (reference code through the harness)
    @staticmethod
    def load_all(path, n_types):
        """Node/ and Edge/ partitions through the reference's own loader."""
        rc = ref().euler_ref_graph_load_all(path.encode())
        assert rc == 0
        return RefGraph(n_types)

    def add_edges_from_adjacency(self):
        return ref().euler_ref_add_edges_from_adjacency()

    def num_edges(self):
        return ref().euler_ref_num_edges()

    def edge_exist(self, src, dst, etype):
        return bool(ref().euler_ref_edge_exist(int(src), int(dst), int(etype)))

    def local_sample_layer(self, seed, call_id, idx, ids, w, t, n, m, weight_func="sqrt",
                           default_node=-1):
        """API_LOCAL_SAMPLE_L over the reference's own containers."""
        idx = _arr(idx, np.int32).reshape(-1)
        ids, w, t = _arr(ids, np.uint64), _arr(w, np.float32), _arr(t, np.int32)
        batch = len(idx) // (2 * n)
        oid = np.zeros(batch * m, np.uint64)
        ow = np.zeros(batch * m, np.float32)
        ot = np.zeros(batch * m, np.int32)
        ref().euler_ref_local_sample_layer(seed, call_id, _p(idx, _i32p), len(idx),
                                           _p(ids, _u64p), _p(w, _f32p), _p(t, _i32p), n,
                                           m, weight_func.encode(), default_node,
                                           _p(oid, _u64p), _p(ow, _f32p), _p(ot, _i32p))
        return oid, ow, ot

    @staticmethod
    def _adj_to_sparse(nodes, nb_nodes, batch, n, m, idx, vals):
        return _adj_to_sparse_with(ref().euler_ref_adj_to_sparse, nodes, nb_nodes,
                                   batch, n, m, idx, vals)

    @staticmethod
    def _sample_root(seed, call_id, roots, weights, n, m, default_node=-1):
        roots = _arr(np.asarray(roots).reshape(-1), np.uint64)
        weights = _arr(np.asarray(weights).reshape(-1), np.float32)
        batch = len(roots) // n
        out = np.zeros(batch * m, np.uint64)
        ref().euler_ref_sample_root(seed, call_id, _p(roots, _u64p), _p(weights, _f32p),
                                    batch, n, m, default_node, _p(out, _u64p))
        return out

    def get_edge_sum_weight(self, ids, edge_types):
        ids = _arr(ids, np.uint64)
        et = _arr(edge_types, np.int32)
        out = np.zeros(len(ids), np.float32)
        ref().euler_ref_get_edge_sum_weight(_p(ids, _u64p), len(ids), _p(et, _i32p),
                                            len(et), _p(out, _f32p))
        return out

    def sample_layer(self, seed, call_id, roots, edge_types, default_node=-1):
        roots = _arr(roots, np.uint64)
        et = _arr(edge_types, np.int32)
        n = len(roots)
        oid = np.zeros(n, np.uint64)
        ow = np.zeros(n, np.float32)
        ot = np.zeros(n, np.int32)
        ref().euler_ref_sample_layer(seed, call_id, _p(roots, _u64p), n, _p(et, _i32p),
                                     len(et), default_node, _p(oid, _u64p),
                                     _p(ow, _f32p), _p(ot, _i32p))
        return oid, ow, ot

    def sparse_get_adj(self, roots, l_nb, batch, n, m, edge_types):
        roots = _arr(np.asarray(roots).reshape(-1), np.uint64)
        l_nb = _arr(np.asarray(l_nb).reshape(-1), np.uint64)
        et = _arr(edge_types, np.int32)
        idx = np.zeros((batch * n, 2), np.int32)
        tot = ref().euler_ref_sparse_get_adj(_p(roots, _u64p), _p(l_nb, _u64p), batch,
                                             n, m, _p(et, _i32p), len(et),
                                             _p(idx, _i32p), None)
        vals = np.zeros(tot, np.uint64)
        ref().euler_ref_sparse_get_adj(_p(roots, _u64p), _p(l_nb, _u64p), batch, n, m,
                                       _p(et, _i32p), len(et), _p(idx, _i32p),
                                       _p(vals, _u64p))
        return idx, vals

    def get_neighbor(self, ids, edge_types, order_by=None, desc=False, limit=None):
        """The reference's GetFullNeighbor + the post-process of
        API_GET_NB_NODE with the reference's comparators and std::sort."""
        ids = _arr(ids, np.uint64)
        et = _arr(edge_types, np.int32)
        n = len(ids)
        a = (_p(ids, _u64p), n, _p(et, _i32p), len(et), ORDER[order_by],
             1 if desc else 0, -1 if limit is None else int(limit))
        tot = ref().euler_ref_get_neighbor(*a, None, None, None, None)
        idx = np.zeros((n, 2), np.int32)
        oid = np.zeros(max(tot, 1), np.uint64)
        ow = np.zeros(max(tot, 1), np.float32)
        ot = np.zeros(max(tot, 1), np.int32)
        ref().euler_ref_get_neighbor(*a, _p(idx, _i32p), _p(oid, _u64p), _p(ow, _f32p),
                                     _p(ot, _i32p))
        return idx, oid[:tot], ow[:tot], ot[:tot]

    def set_float_features(self, row_id, feats):
        ids = _arr(row_id, np.uint64)
        rc = ref().euler_ref_set_float_features(
            _p(ids, _u64p), len(ids), feats.n_float, _p(feats.feat_ptr, _i64p),
            _p(feats.feat_idx, _i32p), _p(feats.feat_val, _f32p))
        assert rc == 0

    def export_float_features(self, ids):
        ids = _arr(ids, np.uint64)
        F = ref().euler_ref_num_float_features()
        n = len(ids)
        tot = ref().euler_ref_export_float_features(_p(ids, _u64p), n, F, None, None, None)
        ptr = np.zeros(n + 1, np.int64)
        idx = np.zeros(n * max(F, 1), np.int32)
        val = np.zeros(max(tot, 1), np.float32)
        ref().euler_ref_export_float_features(_p(ids, _u64p), n, F, _p(ptr, _i64p),
                                              _p(idx, _i32p), _p(val, _f32p))
        return DenseFeatures(F, ptr, idx[:n * F], val[:tot])

    def get_dense_feature(self, nodes, feature_ids, dimensions):
        ids = _arr(nodes, np.int64).astype(np.uint64)
        outs = []
        for fid, dim in zip(feature_ids, dimensions):
            out = np.zeros((len(ids), dim), np.float32)
            rc = ref().euler_ref_get_dense_feature(_p(ids, _u64p), len(ids), int(fid),
                                                   int(dim), _p(out, _f32p))
            assert rc == 0, rc
            outs.append(out)
        return outs

    def node_order(self):
        n = ref().euler_ref_num_nodes()
        out = np.zeros(n, np.uint64)
        ref().euler_ref_node_order(_p(out, _u64p))
        return out

    def export_csr(self, ids=None):
        if ids is None:
            ids = np.sort(self.node_order())
        ids = _arr(ids, np.uint64)
        n = len(ids)
        T = self.n_types
        tot = ref().euler_ref_export_rows(_p(ids, _u64p), n, T, None, None, None,
                                          None, None)
        assert tot >= 0, tot
        row_ptr = np.zeros(n + 1, np.int64)
        type_end = np.zeros(n * T, np.int32)
        nbr = np.zeros(tot, np.uint64)
        prefix = np.zeros(tot, np.float32)
        tpre = np.zeros(n * T, np.float32)
        rc = ref().euler_ref_export_rows(_p(ids, _u64p), n, T, _p(row_ptr, _i64p),
                                         _p(type_end, _i32p), _p(nbr, _u64p),
                                         _p(prefix, _f32p), _p(tpre, _f32p))
        assert rc >= 0, rc
        ntype = np.zeros(n, np.int32)
        nweight = np.zeros(n, np.float32)
        ref().euler_ref_node_info(_p(ids, _u64p), n, _p(ntype, _i32p),
                                  _p(nweight, _f32p))
        return CSR(ids, row_ptr, type_end, nbr, prefix, tpre, T, ntype, nweight)

    def sample_neighbor_core(self, seed, call_id, ids, edge_types, count):
        ids = _arr(ids, np.uint64)
        et = _arr(edge_types, np.int32)
        n = len(ids)
        idx = np.zeros((n, 2), np.int32)
        oid = np.zeros(n * count, np.uint64)
        ow = np.zeros(n * count, np.float32)
        ot = np.zeros(n * count, np.int32)
        ref().euler_ref_sample_neighbor(seed, call_id, _p(ids, _u64p), n,
                                        _p(et, _i32p), len(et), count,
                                        _p(idx, _i32p), _p(oid, _u64p),
                                        _p(ow, _f32p), _p(ot, _i32p))
        return idx, oid, ow, ot

    def sample_node(self, seed, call_id, node_types, count):
        nt = _arr(np.atleast_1d(node_types), np.int32)
        out = np.zeros(max(count, 1), np.uint64)
        got = ref().euler_ref_sample_node(seed, call_id, _p(nt, _i32p), len(nt),
                                          count, _p(out, _u64p))
        return out[:got]

    def alias_table(self, node_type):
        n = ref().euler_ref_alias_size(node_type)
        ids = np.zeros(n, np.uint64)
        w = np.zeros(n, np.float32)
        prob = np.zeros(n, np.float32)
        alias = np.zeros(n, np.int64)
        s = C.c_float(0)
        ref().euler_ref_alias_table(node_type, _p(ids, _u64p), _p(w, _f32p),
                                    _p(prob, _f32p), _p(alias, _i64p),
                                    C.byref(s))
        return ids, w, prob, alias, s.value

    def get_full_neighbor(self, ids, edge_types):
        ids = _arr(ids, np.uint64)
        et = _arr(edge_types, np.int32)
        n = len(ids)
        tot = ref().euler_ref_get_full_neighbor(_p(ids, _u64p), n, _p(et, _i32p),
                                                len(et), None, None, None, None)
        idx = np.zeros((n, 2), np.int32)
        oid = np.zeros(tot, np.uint64)
        ow = np.zeros(tot, np.float32)
        ot = np.zeros(tot, np.int32)
        ref().euler_ref_get_full_neighbor(_p(ids, _u64p), n, _p(et, _i32p),
                                          len(et), _p(idx, _i32p),
                                          _p(oid, _u64p), _p(ow, _f32p),
                                          _p(ot, _i32p))
        return idx, oid, ow, ot

    def random_walk(self, seed, call_id, nodes, edge_types, walk_len, p=1.0,
                    q=1.0, default_node=-1):
        nodes = _arr(nodes, np.int64)
        et = _arr(edge_types, np.int32).reshape(walk_len, -1)
        out = np.zeros((len(nodes), walk_len + 1), np.int64)
        ref().euler_ref_random_walk(seed, call_id, _p(nodes, _i64p), len(nodes),
                                    _p(et, _i32p), et.shape[1], walk_len, p, q,
                                    default_node, _p(out, _i64p))
        return out

    def bench_fanout_dag(self, seed, roots, batch, counts, threads, mode, dedup=True,
                         warmup=5, timed=30):
        """SURVEY 8(d) CPU-baseline protocol (euler_ref_bench_fanout_dag): the
        ID_UNIQUE -> API_SAMPLE_NB -> DATA_GATHER DAG per hop.  mode 0 = `threads`
        concurrent single-threaded queries per round ("as shipped"), mode 1 = one
        query per round with the OpenMP batch loop over `threads` threads.
        Returns (seconds of every timed round [timed], sampled edges per round)."""
        roots = _arr(roots, np.uint64)
        cnt = _arr(counts, np.int32)
        n_batches = len(roots) // batch
        assert n_batches >= 1
        secs = np.zeros(timed, np.float64)
        edges = ref().euler_ref_bench_fanout_dag(
            seed, _p(roots, _u64p), batch, n_batches, _p(cnt, _i32p), len(cnt), int(threads),
            int(mode), 1 if dedup else 0, int(warmup), int(timed),
            secs.ctypes.data_as(C.POINTER(C.c_double)))
        return secs, int(edges)

    def bench_fanout(self, seed, roots, batch, iters, counts, threads):
        roots = _arr(roots, np.uint64)
        cnt = _arr(counts, np.int32)
        edges = C.c_int64(0)
        secs = ref().euler_ref_bench_fanout(seed, _p(roots, _u64p), batch, iters,
                                            _p(cnt, _i32p), len(cnt), threads,
                                            C.byref(edges))
        return secs, edges.value
