"""ctypes binding of euler_amd/lib/libeuler_gpu.so (the C ABI in
include/euler_gpu.h).  There is NO fallback: if the HIP library is missing the
import fails loudly - build it with `python -c "import __graft_entry__ as g;
g.build()"` or `make -C euler_amd/csrc`."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("EULER_GPU_LIB_PATH") or os.path.join(HERE, "lib", "libeuler_gpu.so")   # (the variable: A/B of experimental builds, tools/)

OK, EINVAL, ENOMEM, EHIP, ENOGRAPH, EIO, EEMPTY = 0, -1, -2, -3, -4, -5, -6
LAYOUT_CORE, LAYOUT_TF = 0, 1

u64p = C.POINTER(C.c_uint64)
i64p = C.POINTER(C.c_int64)
i32p = C.POINTER(C.c_int32)
f32p = C.POINTER(C.c_float)
u8p = C.POINTER(C.c_uint8)
vp = C.c_void_p


class HostCSR(C.Structure):
    _fields_ = [("n_rows", C.c_int64), ("n_edge_types", C.c_int32),
                ("n_node_types", C.c_int32), ("row_id", u64p),
                ("row_ptr", i64p), ("type_end", i32p), ("nbr", u64p),
                ("prefix_w", f32p), ("type_prefix", f32p),
                ("node_type", i32p), ("node_weight", f32p),
                ("sampler_order", u64p),
                ("n_float_features", C.c_int32), ("pad0", C.c_int32),
                ("feat_ptr", i64p), ("feat_idx", i32p), ("feat_val", f32p),
                ("n_u64_features", C.c_int32), ("pad1", C.c_int32),
                ("ufeat_ptr", i64p), ("ufeat_idx", i32p), ("ufeat_val", u64p)]


class SynthParams(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("n_nodes", C.c_int64),
                ("n_edges_target", C.c_int64), ("scale", C.c_int32),
                ("n_types", C.c_int32), ("weighted", C.c_int32),
                ("hashed_ids", C.c_int32), ("deg_table", C.c_double * 64)]


# name -> (restype, argtypes); also the list of symbols include/euler_gpu.h declares
SIGNATURES = {
    "euler_gpu_last_error": (C.c_char_p, []),
    "euler_gpu_version": (C.c_char_p, []),
    "euler_gpu_device_count": (C.c_int, []),
    "euler_gpu_graph_create": (C.c_int, [C.POINTER(HostCSR), C.c_int, C.POINTER(vp)]),
    "euler_gpu_graph_create_shard": (C.c_int, [C.POINTER(HostCSR), C.c_int, C.c_int32,
                                               C.c_int32, C.c_int32, C.POINTER(vp)]),
    "euler_gpu_graph_create_synthetic": (C.c_int, [C.POINTER(SynthParams), C.c_int,
                                                   C.c_int32, C.c_int32, C.c_int32,
                                                   C.POINTER(vp)]),
    "euler_gpu_graph_load": (C.c_int, [C.c_char_p, C.c_int, C.c_int32, C.c_int32,
                                       C.POINTER(vp)]),
    "euler_gpu_dat_open": (C.c_int, [C.c_char_p, C.c_int32, C.c_int32,
                                     C.POINTER(HostCSR), i32p, C.POINTER(vp)]),
    "euler_gpu_dat_close": (None, [vp]),
    "euler_gpu_dat_verify_edges": (C.c_int, [C.c_char_p, C.c_int32, C.c_int32, i64p, i64p,
                                             i64p]),
    "euler_gpu_graph_destroy": (None, [vp]),
    "euler_gpu_graph_num_nodes": (C.c_int64, [vp]),
    "euler_gpu_graph_num_edges": (C.c_int64, [vp]),
    "euler_gpu_graph_num_edge_types": (C.c_int32, [vp]),
    "euler_gpu_graph_num_node_types": (C.c_int32, [vp]),
    "euler_gpu_graph_device": (C.c_int, [vp]),
    "euler_gpu_graph_bytes": (C.c_int64, [vp]),
    "euler_gpu_graph_node_weight_sums": (C.c_int, [vp, f32p]),
    "euler_gpu_graph_set_node_sampler": (C.c_int, [vp, C.c_int64, u64p, i32p, f32p, C.c_int32]),
    "euler_gpu_last_fanout_kernel": (C.c_char_p, []),
    "euler_gpu_graph_index_overflow_rows": (C.c_int, [vp, u64p, C.c_int64, i64p]),
    "euler_gpu_graph_export_rows": (C.c_int, [vp, u64p, C.c_int64, i64p, i32p, u64p,
                                              f32p, f32p]),
    "InitQueryProxy": (C.c_bool, [C.c_char_p]),
    "euler_gpu_default_graph": (vp, []),
    "euler_gpu_graph_partitions": (C.c_int32, [vp]),
    "euler_gpu_sample_neighbor": (C.c_int, [vp, vp, C.c_uint64, C.c_uint32, vp,
                                            C.c_int64, vp, C.c_int32, i32p, C.c_int32,
                                            C.c_int32, C.c_int32, C.c_int64, vp, vp,
                                            vp, vp]),
    "euler_gpu_sample_fanout_workspace": (C.c_size_t, [C.c_int64, i32p, C.c_int32]),
    "euler_gpu_sample_fanout": (C.c_int, [vp, vp, C.c_uint64, C.c_uint32, vp,
                                          C.c_int64, i32p, C.c_int32, i32p, C.c_int32,
                                          C.c_int64, C.POINTER(vp), C.POINTER(vp),
                                          C.POINTER(vp), vp]),
    "euler_gpu_sample_fanout_multi": (C.c_int, [vp, vp, C.c_uint64, C.c_uint32, C.c_uint32, vp,
                                                C.c_int32, vp, C.c_int64, i32p, C.c_int32, i32p,
                                                C.c_int32, C.c_int64, C.POINTER(vp), C.POINTER(vp),
                                                C.POINTER(vp), vp]),
    "euler_gpu_sample_neighbor_sets": (C.c_int, [vp, vp, C.c_uint64, C.c_uint32, vp, C.c_int64, i32p,
                                                 i32p, C.c_int32, C.c_int32, C.c_int64, vp, vp, vp]),
    "euler_gpu_sample_aggregate_sets": (C.c_int, [vp, vp, C.c_uint64, C.c_uint32, vp, C.c_int64, i32p,
                                                  i32p, C.c_int32, C.c_int32, C.c_int64, C.c_int32, vp,
                                                  C.c_int64, C.c_int64, vp, vp, vp, vp]),
    "euler_gpu_sample_node": (C.c_int, [vp, vp, C.c_uint64, C.c_uint32, i32p,
                                        C.c_int32, C.c_int32, vp]),
    "euler_gpu_get_full_neighbor": (C.c_int, [vp, vp, vp, C.c_int64, i32p, C.c_int32,
                                              vp, i64p, vp, vp, vp]),
    "euler_gpu_graph_num_u64_features": (C.c_int32, [vp]),
    "euler_gpu_get_sparse_feature": (C.c_int, [vp, vp, vp, C.c_int64, C.c_int32, C.c_int64,
                                               vp, i64p, i64p, vp, vp]),
    "euler_gpu_get_sparse_feature_core": (C.c_int, [vp, vp, vp, C.c_int64, C.c_int32, vp,
                                                    i64p, vp]),
    "euler_gpu_get_node_type": (C.c_int, [vp, vp, vp, C.c_int64, vp]),
    "euler_gpu_sample_n_with_types": (C.c_int, [vp, vp, C.c_uint64, C.c_uint32, vp,
                                                C.c_int64, C.c_int32, vp]),
    "euler_gpu_get_edge_sum_weight": (C.c_int, [vp, vp, vp, C.c_int64, i32p, C.c_int32, vp]),
    "euler_gpu_sample_root": (C.c_int, [vp, C.c_uint64, C.c_uint32, vp, vp, C.c_int64,
                                        C.c_int32, C.c_int32, C.c_int64, vp]),
    "euler_gpu_sample_layer": (C.c_int, [vp, vp, C.c_uint64, C.c_uint32, vp, C.c_int64,
                                         i32p, C.c_int32, C.c_int64, vp, vp, vp]),
    "euler_gpu_sample_layer_at": (C.c_int, [vp, vp, C.c_uint64, C.c_uint32, vp, vp, C.c_int64,
                                            i32p, C.c_int32, C.c_int64, vp, vp, vp]),
    "euler_gpu_sample_neighbor_layerwise": (C.c_int, [vp, vp, C.c_uint64, C.c_uint32, vp,
                                                      C.c_int64, C.c_int32, i32p, C.c_int32,
                                                      C.c_int32, C.c_int64, vp]),
    "euler_gpu_local_sample_layer": (C.c_int, [vp, C.c_uint64, C.c_uint32, vp, vp, vp, vp,
                                               C.c_int64, C.c_int64, C.c_int32, C.c_int32,
                                               C.c_char_p, C.c_int64, vp, vp, vp]),
    "euler_gpu_sparse_get_adj_workspace": (C.c_size_t, [C.c_int64, C.c_int32, C.c_int32]),
    "euler_gpu_sparse_get_adj": (C.c_int, [vp, vp, vp, vp, C.c_int64, C.c_int32, C.c_int32,
                                           i32p, C.c_int32, vp, vp, i64p, vp]),
    "euler_gpu_sparse_get_adj_tf": (C.c_int, [vp, vp, vp, vp, C.c_int64, C.c_int32,
                                              C.c_int32, i32p, C.c_int32, vp, i64p, vp, vp]),
    "euler_gpu_get_top_k_neighbor": (C.c_int, [vp, vp, vp, C.c_int64, i32p, C.c_int32,
                                               C.c_int32, C.c_int64, vp, vp, vp]),
    "euler_gpu_random_walk": (C.c_int, [vp, vp, C.c_uint64, C.c_uint32, vp, C.c_int64,
                                        i32p, C.c_int32, C.c_int32, C.c_float,
                                        C.c_float, C.c_int64, vp]),
    "euler_gpu_gen_pair_count": (C.c_int64, [C.c_int64, C.c_int32, C.c_int32]),
    "euler_gpu_gen_pair": (C.c_int, [vp, vp, C.c_int64, C.c_int64, C.c_int32,
                                     C.c_int32, vp]),
    "euler_gpu_id_unique": (C.c_int, [vp, vp, C.c_int64, vp, vp, i64p]),
    "euler_gpu_idx_gather": (C.c_int, [vp, vp, vp, C.c_int64, vp, i64p]),
    "euler_gpu_data_gather": (C.c_int, [vp, vp, C.c_int32, vp, vp, vp, C.c_int64, vp]),
    "euler_gpu_scatter_add": (C.c_int, [vp, vp, vp, C.c_int64, C.c_int64, C.c_int32, vp]),
    "euler_gpu_scatter_max": (C.c_int, [vp, vp, vp, C.c_int64, C.c_int64, C.c_int32, vp]),
    "euler_gpu_scatter_mean": (C.c_int, [vp, vp, vp, C.c_int64, C.c_int64, C.c_int32, vp]),
    "euler_gpu_gather": (C.c_int, [vp, vp, vp, C.c_int64, C.c_int64, C.c_int64, vp]),
    "euler_gpu_gather_segment_reduce": (C.c_int, [vp, C.c_int32, vp, vp, vp, C.c_int64, C.c_int64,
                                                  C.c_int32, vp]),
    "euler_gpu_gather_segment_reduce_ids": (C.c_int, [vp, C.c_int32, vp, C.c_int64, vp, vp, C.c_int64,
                                                      C.c_int64, C.c_int32, vp]),
    "euler_gpu_gather_scatter": (C.c_int, [vp, C.c_int32, vp, vp, vp, C.c_int64, C.c_int64, C.c_int32,
                                           vp]),
    "euler_gpu_id_split": (C.c_int, [vp, vp, C.c_int64, C.c_int32, C.c_int32, i64p,
                                     vp, vp]),
    "euler_gpu_merge_rows": (C.c_int, [vp, vp, vp, C.c_int64, C.c_int64, vp]),
    "euler_gpu_inflate_idx": (C.c_int, [vp, vp, C.c_int64, vp]),
    "euler_gpu_sample_node_split": (C.c_int, [C.c_uint64, C.c_uint32, C.c_int32, f32p,
                                              C.c_int32, i32p]),
    "euler_gpu_time_sample_neighbor": (C.c_int, [vp, vp, C.c_uint64, vp, C.c_int64,
                                                 i32p, C.c_int32, C.c_int32,
                                                 C.c_int32, vp, vp, vp, C.c_int32,
                                                 f32p]),
    "euler_gpu_sample_neighbor_algo_bytes": (C.c_int, [vp, vp, vp, C.c_int64, i32p,
                                                       C.c_int32, C.c_int32,
                                                       C.POINTER(C.c_double)]),
    "euler_gpu_set_tuning": (C.c_int, [C.c_int32, C.c_int32]),
    "euler_gpu_set_index_budget": (C.c_int, [C.c_int64, C.c_double]),
    "euler_gpu_set_debug_buffer": (C.c_int, [vp]),
    "euler_gpu_graph_num_float_features": (C.c_int32, [vp]),
    "euler_gpu_sample_neighbor_distinct": (C.c_int, [vp, vp, C.c_uint64, C.c_uint32, vp,
                                                     C.c_int64, i32p, C.c_int32,
                                                     C.c_int32, C.c_int32, C.c_int64, vp,
                                                     vp, vp, vp]),
    "euler_gpu_sample_neighbor_packed": (C.c_int, [vp, vp, C.c_uint64, C.c_uint32, vp,
                                                   C.c_int64, i32p, C.c_int32, C.c_int32,
                                                   C.c_int64, vp]),
    "euler_gpu_sample_neighbor_sets_packed": (C.c_int, [vp, vp, C.c_uint64, C.c_uint32, vp, C.c_int64, i32p,
                                                        i32p, C.c_int32, C.c_int32, C.c_int64, vp]),
    "euler_gpu_dedup_split": (C.c_int, [vp, vp, C.c_int64, vp, C.c_int32, C.c_int32,
                                        C.c_int32, vp, C.c_int64, C.POINTER(C.c_int64),
                                        vp, vp]),
    "euler_gpu_front_create": (C.c_int, [C.POINTER(vp)]),
    "euler_gpu_front_destroy": (None, [vp]),
    "euler_gpu_dedup_split_begin": (C.c_int, [vp, vp, vp, C.c_int64, vp, C.c_int32, C.c_int32,
                                              C.c_int32, vp, C.c_int64, vp, vp]),
    "euler_gpu_dedup_split_end": (C.c_int, [vp, C.POINTER(C.c_int64)]),
    "euler_gpu_graph_id_range": (C.c_int, [vp, u64p, i32p]),
    "euler_shm_open": (C.c_int, [C.c_char_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(vp)]),
    "euler_shm_alltoall_i64": (C.c_int, [vp, i64p, i64p, C.c_int32, C.c_int64]),
    "euler_shm_attached": (C.c_int32, [vp]),
    "euler_shm_unlink": (C.c_int, [vp]),
    "euler_shm_close": (None, [vp]),
    "euler_gpu_pack_rows": (C.c_int, [vp, vp, vp, vp, vp, C.c_int64, C.c_int32, C.c_int32,
                                      vp]),
    "euler_gpu_expand_packed": (C.c_int, [vp, vp, C.c_int64, C.c_int32, C.c_int32, vp, vp, vp,
                                          vp, vp]),
    "euler_gpu_expand_rows": (C.c_int, [vp, vp, C.c_int64, C.c_int32, vp, vp, vp, vp,
                                        vp, vp, vp, vp]),
    "euler_gpu_neighbor_post_process": (C.c_int, [vp, C.c_int64, vp, C.c_int64, vp, vp, vp,
                                                  C.c_int32, C.c_int32, C.c_int64,
                                                  C.POINTER(C.c_int64)]),
    "euler_gpu_neighbor_to_dense": (C.c_int, [vp, C.c_int64, vp, vp, vp, vp, C.c_int32,
                                              C.c_int64, vp, vp, vp]),
    "euler_gpu_get_dense_feature": (C.c_int, [vp, vp, vp, C.c_int64, C.c_int32,
                                              C.c_int32, vp]),
    "euler_gpu_time_sample_neighbor_phases": (C.c_int, [vp, vp, C.c_uint64, vp, C.c_int64,
                                                        i32p, C.c_int32, C.c_int32,
                                                        C.c_int32, C.c_int32, vp, vp, vp,
                                                        C.c_int32, f32p,
                                                        C.POINTER(C.c_int64)]),
    "euler_gpu_node2vec_step": (C.c_int, [vp, C.c_uint64, C.c_uint32, C.c_int64, vp, vp, vp, vp, C.c_int64,
                                          vp, vp, vp, vp, C.c_float, C.c_float, C.c_int64, vp]),
    "euler_gpu_sample_fanout_with_feature": (C.c_int, [vp, vp, C.c_uint64, C.c_uint32, vp, C.c_int64,
                                                       i32p, C.c_int32, i32p, C.c_int32, C.c_int64,
                                                       C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), vp,
                                                       i32p, i32p, C.c_int32, C.POINTER(vp)]),
    "euler_gpu_sample_fanout_unique": (C.c_int, [vp, vp, C.c_uint64, C.c_uint32, vp, C.c_int64, i32p, i32p,
                                                 C.c_int64, vp, vp, vp, vp, vp, vp, vp, vp]),
    "euler_gpu_full_blocks_workspace": (C.c_size_t, [C.c_int64, C.POINTER(C.c_int64), C.c_int32]),
    "euler_gpu_full_blocks": (C.c_int, [vp, vp, vp, C.c_int64, i32p, C.c_int32, C.c_int32, C.c_int32,
                                        C.POINTER(C.c_int64), vp, C.POINTER(vp), C.POINTER(vp),
                                        C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), vp]),
    "euler_gpu_sparse_adj_mask": (C.c_int, [vp, vp, vp, vp, C.c_int64, C.c_int32, C.c_int32, i32p,
                                            C.c_int32, vp]),
    "euler_gpu_sparse_adj_from_mask_tf": (C.c_int, [vp, vp, C.c_int64, C.c_int32, C.c_int32, vp,
                                                    C.POINTER(C.c_int64), vp, vp]),
    "euler_gpu_time_sample_fanout": (C.c_int, [vp, vp, C.c_uint64, vp, C.c_int64,
                                               i32p, C.c_int32, i32p, C.c_int32,
                                               C.c_int64, C.POINTER(vp),
                                               C.POINTER(vp), C.POINTER(vp), vp,
                                               C.c_int32, C.POINTER(C.c_float)]),
    "euler_gpu_time_sample_fanout_phases": (C.c_int, [vp, vp, C.c_uint64, vp, C.c_int64,
                                                      i32p, C.c_int32, i32p, C.c_int32,
                                                      C.c_int64, C.POINTER(vp),
                                                      C.POINTER(vp), C.POINTER(vp), vp,
                                                      C.c_int32, f32p,
                                                      C.POINTER(C.c_int64)]),
    "euler_op_registered": (C.c_int, [C.c_char_p]),
    "euler_op_run_get_nb": (C.c_int64, [vp, u64p, C.c_int64, i32p, C.c_int32, C.c_char_p,
                                        C.c_int64, i32p, u64p, f32p, i32p]),
    "euler_op_run_sample_lnb": (C.c_int64, [vp, C.c_uint64, C.c_uint32, u64p, C.c_int64,
                                            C.c_int32, i32p, C.c_int32, C.c_int32,
                                            C.c_char_p, C.c_int64, C.c_int64, i32p, u64p,
                                            u64p]),
    "euler_op_run_sample_nb": (C.c_int64, [vp, C.c_uint64, u64p, C.c_int64, i32p,
                                           C.c_int32, C.c_int32, i32p, u64p, f32p,
                                           i32p]),
    "euler_op_run_sample_nb_post": (C.c_int64, [vp, C.c_uint64, C.c_uint32, u64p, C.c_int64,
                                                i32p, C.c_int32, C.c_int32, C.c_char_p, i32p,
                                                u64p, f32p, i32p]),
    "euler_gpu_random_walk_stats": (C.c_int, [C.POINTER(C.c_uint64), C.c_int32]),
    "euler_gpu_random_walk_algo_bytes": (C.c_int, [vp, vp, vp, C.c_int64, i32p, C.c_int32, C.c_int32,
                                                   C.c_float, C.c_float, C.POINTER(C.c_double)]),
    "euler_gpu_sage_blocks_workspace": (C.c_size_t, [C.c_int64, i32p, C.c_int32]),
    "euler_gpu_sage_blocks": (C.c_int, [vp, vp, C.c_uint64, C.c_uint32, vp, C.c_int64, i32p,
                                        C.c_int32, i32p, C.c_int32, C.c_int64, C.c_int32, vp,
                                        vp, vp, vp, vp, vp]),
    "euler_gpu_sage_blocks_multi_workspace": (C.c_size_t, [C.c_int32, C.c_int64, i32p, C.c_int32]),
    "euler_gpu_sage_blocks_multi": (C.c_int, [vp, vp, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int32, vp,
                                              C.c_int64, i32p, C.c_int32, i32p, C.c_int32, C.c_int64,
                                              C.c_int32, vp, vp, vp, vp, vp, vp]),
    "euler_gpu_sharded_random_walk": (C.c_int, [vp, vp, vp, C.c_uint64, C.c_uint32, vp, C.c_int64,
                                                 i32p, C.c_int32, C.c_int32, C.c_int64, C.c_int32,
                                                 C.c_int32, vp, C.c_int64, vp, i64p]),
    "euler_gpu_sharded_node2vec_walk": (C.c_int, [vp, vp, vp, C.c_uint64, C.c_uint32, vp, C.c_int64,
                                                   i32p, C.c_int32, C.c_int32, C.c_float, C.c_float,
                                                   C.c_int64, C.c_int32, vp, C.c_int64, vp, i64p]),
    "euler_gpu_transport_rccl": (C.c_int, [vp, C.c_int32, C.c_int32, vp, vp]),
    "euler_gpu_transport_rccl_release": (None, [vp]),
    "euler_gpu_sharded_sample_neighbor": (C.c_int, [vp, vp, vp, C.c_uint64, C.c_uint32, vp,
                                                     C.c_int64, vp, C.c_int32, i32p, C.c_int32,
                                                     C.c_int32, C.c_int64, C.c_int32, vp, vp, vp,
                                                     vp]),
    "euler_gpu_sharded_sample_fanout": (C.c_int, [vp, vp, vp, C.c_uint64, C.c_uint32, vp,
                                                   C.c_int64, i32p, C.c_int32, i32p, C.c_int32,
                                                   C.c_int64, C.c_int32, vp, vp, vp, vp]),
    # include/euler_query.h
    "euler_query_run": (C.c_int64, [C.c_char_p, C.c_int32, C.POINTER(C.c_char_p), i32p,
                                    C.POINTER(C.c_int64), C.POINTER(C.c_void_p), C.c_char_p,
                                    C.c_void_p, C.c_int64]),
    "euler_query_set_seed": (None, [C.c_uint64]),
    "euler_query_set_graph": (None, [vp]),
}

_lib = None


class EulerGpuError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("euler_gpu error %d: %s" % (code, msg))
        self.code = code


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "euler_amd: %s is missing - the HIP library must be built "
                "(make -C euler_amd/csrc); there is no CPU fallback" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc):
    if rc != OK:
        raise EulerGpuError(rc, lib().euler_gpu_last_error().decode())
    return rc
