/*
 * ORACLE (test infrastructure, NOT product code): CPU restatement, in plain C,
 * of the reference algorithms on Euler's minibatch-construction hot path.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * use it; the product (euler_amd/) never links, imports or falls back to it.
 *
 * Parity status: PINNED.  Every function here is checked (tests/, -m "not
 * gpu") against (1) oracle/_ref = the reference's own sources compiled with
 * the RNG seam, on random graphs and on the reference's 6-node fixture, and
 * (2) the reference's exact golden vectors (mp_ops_test.py, walk_ops_test.py,
 * unique_gather_test.cc, *_merge_op_test.cc) committed under tests/golden/.
 * Sampled ids have no golden in the reference (its sampling tests are
 * statistical); they are pinned by (1).
 * Third-party arithmetic: API_LOCAL_SAMPLE_L depends on the iteration order of
 * libstdc++'s std::unordered_map<std::string, ...> (GCC 11.4.0, GLIBCXX_3.4.30);
 * eo_umap.c restates it and is pinned against the real container in oracle/_ref.
 */
#ifndef EULER_ORACLE_H_
#define EULER_ORACLE_H_

#include <stdint.h>

#include "eo_rng.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Adjacency exactly as the reference stores it per node (node.h:49-57),
 * concatenated over rows. */
typedef struct eo_graph {
  int64_t n_rows;
  int32_t n_types;            /* edge-type groups per node                  */
  const uint64_t* row_id;     /* [n_rows]   node id of the row              */
  const int64_t* row_ptr;     /* [n_rows+1] offsets into nbr / prefix_w     */
  const int32_t* type_end;    /* [n_rows*T] neighbor_groups_idx (cumulative,
                                 row-relative)                              */
  const uint64_t* nbr;        /* [E] neighbors, grouped by type             */
  const float* prefix_w;      /* [E] neighbors_weight: running f32 sums
                                 across ALL types of the row                */
  const float* type_prefix;   /* [n_rows*T] edge_group_collection running
                                 sums                                       */
  /* id -> row open-addressing index (built by eo_graph_create) */
  uint64_t hash_cap;
  uint64_t* hash_key;
  int64_t* hash_row;
} eo_graph;

eo_graph* eo_graph_create(int64_t n_rows, int32_t n_types,
                          const uint64_t* row_id, const int64_t* row_ptr,
                          const int32_t* type_end, const uint64_t* nbr,
                          const float* prefix_w, const float* type_prefix);
void eo_graph_destroy(eo_graph* g);
int64_t eo_graph_find_row(const eo_graph* g, uint64_t id);

/* Build reference-format rows from RAW per-(node,type) weights the way
 * Node::Init does (node.cc:37-96): sequential f32 running sums. */
void eo_build_prefix(int64_t n_rows, int32_t n_types, const int64_t* seg_ptr,
                     const float* w, int64_t* row_ptr, int32_t* type_end,
                     float* prefix_w, float* type_prefix);

void eo_philox_kat(const uint32_t ctr[4], const uint32_t key[2],
                   uint32_t out[4]);
double eo_uniform_at(uint64_t seed, uint32_t call_id, uint32_t domain,
                     uint64_t stream, uint64_t draw_idx);

int64_t eo_random_select(const float* sum_weights, uint64_t begin_pos,
                         uint64_t end_pos, double u);

int64_t eo_sample_neighbor_core(const eo_graph* g, uint64_t seed,
                                uint32_t call_id, const uint64_t* ids,
                                int64_t n, const int32_t* edge_types,
                                int32_t k, int32_t count, int32_t* idx,
                                uint64_t* out_id, float* out_w,
                                int32_t* out_t);

void eo_sample_neighbor_tf(const eo_graph* g, uint64_t seed, uint32_t call_id,
                           const int64_t* nodes, int64_t n,
                           const int32_t* edge_types, int32_t k,
                           int32_t count, int64_t default_node,
                           int64_t* out_n, float* out_w, int32_t* out_t);

void eo_sample_fanout_tf(const eo_graph* g, uint64_t seed, uint32_t call_id,
                         const int64_t* nodes, int64_t n,
                         const int32_t* edge_types, int32_t k,
                         const int32_t* counts, int32_t layers,
                         int64_t default_node, int64_t** out_n,
                         float** out_w, int32_t** out_t);

/* Dense float features in the reference's per-node form (node.h
 * float_features_ / float_features_idx_), concatenated over rows. */
typedef struct eo_features {
  int32_t n_float;            /* feature slots per node                       */
  const int64_t* feat_ptr;    /* [n_rows+1] offset of each row's values        */
  const int32_t* feat_idx;    /* [n_rows*F] cumulative ends, row-relative      */
  const float* feat_val;
} eo_features;

/* TF GetDenseFeature (tf_euler/kernels/get_dense_feature_op.cc:63-125):
 * out [n, dim] zero filled, then the stored values of feature fid. Returns -2
 * if a node stores more than dim values (the reference would overrun). */
int eo_get_dense_feature(const eo_graph* g, const eo_features* f,
                         const uint64_t* ids, int64_t n, int32_t fid,
                         int32_t dim, float* out);

/* uint64 ("sparse") features, same layout as eo_features */
typedef struct eo_u64_features {
  int32_t n_u64;
  const int64_t* feat_ptr;
  const int32_t* feat_idx;
  const uint64_t* feat_val;
} eo_u64_features;
int64_t eo_get_sparse_feature(const eo_graph* g, const eo_u64_features* f,
                              const uint64_t* ids, int64_t n, int32_t fid,
                              int64_t default_value, int64_t* indices,
                              int64_t* values, int64_t* shape);

int64_t eo_get_full_neighbor(const eo_graph* g, const uint64_t* ids,
                             int64_t n, const int32_t* edge_types, int32_t k,
                             int32_t* idx, uint64_t* out_id, float* out_w,
                             int32_t* out_t);

/* Post-process of API_GET_NB_NODE (core/kernels/get_neighbor_op.cc:117-168) on
 * a FillNeighbor-layout result, in place: order_by (0 none, 1 id, 2 weight;
 * desc != 0 reverses) then limit (< 0 none).  Rows are re-packed, idx
 * rewritten; returns the new total.  Ties keep storage order (the reference
 * sorts with a non-strict comparator, so its order of equal keys is undefined). */
int64_t eo_neighbor_post_process(int64_t n, int32_t* idx, uint64_t* ids, float* w,
                                 int32_t* t, int32_t order_by, int32_t desc,
                                 int64_t limit);
/* TF GetTopKNeighbor dense fill (tf_euler/kernels/get_top_k_neighbor_op.cc:
 * 70-75,101-109). */
void eo_neighbor_to_dense(int64_t n, const int32_t* idx, const uint64_t* ids,
                          const float* w, const int32_t* t, int32_t k,
                          int64_t default_node, int64_t* out_id, float* out_w,
                          int32_t* out_t);

/* libstdc++ internals restated for API_LOCAL_SAMPLE_L (oracle/eo_umap.c) */
uint64_t eo_std_hash_bytes(const void* ptr, uint64_t len);
void eo_umap_iteration_order(const uint64_t* hash, int64_t n, int64_t* order);
void eo_local_sample_layer(uint64_t seed, uint32_t call_id, const int32_t* idx,
                           int64_t idx_elems, const uint64_t* ids, const float* w,
                           const int32_t* t, int32_t n, int32_t m, int32_t take_sqrt,
                           int64_t default_node, uint64_t* o_nb, float* o_w, int32_t* o_t);

/* layerwise sampling (sampleLNB without a weight function) */
void eo_get_edge_sum_weight(const eo_graph* g, const uint64_t* ids, int64_t n,
                            const int32_t* edge_types, int32_t k, float* out_w);
void eo_sample_root(uint64_t seed, uint32_t call_id, const uint64_t* roots,
                    const float* weights, int64_t batch, int32_t n, int32_t m,
                    int64_t default_node, uint64_t* out);
void eo_sample_layer(const eo_graph* g, uint64_t seed, uint32_t call_id,
                     const uint64_t* roots, int64_t n, const int32_t* edge_types,
                     int32_t k, int64_t default_node, uint64_t* out_id,
                     float* out_w, int32_t* out_t);
void eo_sample_layer_at(const eo_graph* g, uint64_t seed, uint32_t call_id,
                        const uint64_t* roots, const int64_t* pos, int64_t n,
                        const int32_t* edge_types, int32_t k, int64_t default_node,
                        uint64_t* out_id, float* out_w, int32_t* out_t);
int64_t eo_sparse_get_adj(const eo_graph* g, const uint64_t* roots,
                          const uint64_t* l_nb, int64_t batch, int32_t n,
                          int32_t m, const int32_t* edge_types, int32_t k,
                          int32_t* idx, uint64_t* out_id);
int64_t eo_adj_to_sparse(const uint64_t* nodes, const uint64_t* nb_nodes,
                         int64_t batch, int32_t n, int32_t m, const int32_t* idx,
                         const uint64_t* vals, int64_t* indices, int64_t* values,
                         int64_t* shape);

int64_t eo_id_unique(const uint64_t* ids, int64_t n, uint64_t* unique_ids,
                     int32_t* gather_idx);
void eo_idx_gather(const int32_t* idx, const int32_t* gather_idx, int64_t n,
                   int32_t* out);
int64_t eo_data_gather(const void* data, int32_t elem_size,
                       const int32_t* idx, const int32_t* gather_idx,
                       int64_t n, void* out);

/* Alias method (alias_method.cc:23-78). */
void eo_alias_init(const float* weights, int64_t n, float* prob,
                   int64_t* alias);
/* Global node sampler = Graph::BuildGlobalSampler (graph.cc:333-370) over a
 * node list given in the order the reference iterates its node_map_. */
typedef struct eo_node_sampler {
  int32_t n_types;
  int64_t* type_off;      /* [T+1] */
  uint64_t* ids;          /* per type, concatenated */
  float* prob;
  int64_t* alias;
  float* type_sum;        /* [T] node_weight_sums_ */
  float* sampler_sum;     /* [T] FastWeightedCollection::sum_weight_ */
  float* tc_prob;         /* [T] node_type_collection_ alias table */
  int64_t* tc_alias;
  float tc_sum;
} eo_node_sampler;
eo_node_sampler* eo_node_sampler_create(int64_t n, const uint64_t* ids,
                                        const int32_t* types,
                                        const float* weights,
                                        int32_t n_types);
void eo_node_sampler_destroy(eo_node_sampler* s);
int64_t eo_sample_node(const eo_node_sampler* s, uint64_t seed,
                       uint32_t call_id, const int32_t* node_types, int32_t k,
                       int32_t count, uint64_t* out);
int eo_sample_n_with_types(const eo_node_sampler* s, uint64_t seed,
                           uint32_t call_id, const int32_t* types, int64_t n,
                           int32_t count, uint64_t* out);

int eo_random_walk(const eo_graph* g, uint64_t seed, uint32_t call_id,
                   const int64_t* nodes, int64_t n, const int32_t* edge_types,
                   int32_t k, int32_t walk_len, float p, float q,
                   int64_t default_node, int64_t* out);

int64_t eo_gen_pair_count(int64_t path_len, int32_t left, int32_t right);
void eo_gen_pair(const int64_t* paths, int64_t batch, int64_t path_len,
                 int32_t left, int32_t right, int64_t* out);

void eo_scatter_add(const float* updates, const int32_t* indices, int64_t e,
                    int64_t d, int32_t size, float* out);
void eo_scatter_max(const float* updates, const int32_t* indices, int64_t e,
                    int64_t d, int32_t size, float* out);
void eo_gather(const float* params, const int32_t* indices, int64_t e,
               int64_t d, float* out);

/* Shard ops (distributed mode semantics). */
int32_t eo_shard_of(uint64_t id, int32_t partitions, int32_t shards);
void eo_id_split(const uint64_t* ids, int64_t n, int32_t partitions,
                 int32_t shards, int64_t* shard_off, uint64_t* shard_ids,
                 int32_t* merge_idx);
void eo_sample_node_split(uint64_t seed, uint32_t call_id, int32_t count,
                          const float* shard_weight, int32_t shards,
                          int32_t* split_cnt);

/* CPU-baseline timing loop over the restatement ("port"). */
double eo_bench_fanout(const eo_graph* g, uint64_t seed,
                       const uint64_t* roots, int64_t batch, int32_t iters,
                       const int32_t* counts, int32_t hops, int32_t threads,
                       int64_t* edges);

/* Deterministic synthetic power-law graph (eo_synth.c), mirrored bit-for-bit
 * by the device generator in euler_amd/csrc/graph_build.hip. */
typedef struct eo_synth_params {
  uint64_t seed;
  int64_t n_nodes;        /* ids are 1..n_nodes */
  int64_t n_edges_target; /* expected edge total */
  int32_t scale;          /* RMAT scale: ids drawn from [0, 2^scale) */
  int32_t n_types;        /* edge types (edges split by hash) */
  int32_t weighted;       /* 0: all 1.0f; 1: uniform [0.5, 8) */
  int32_t hashed_ids;        /* 1: external id of node x = eo_synth_external_id(x) (a bijection of u64) */
  double deg_table[64];   /* expected extra degree by popcount(id-1) */
} eo_synth_params;
uint64_t eo_synth_external_id(const eo_synth_params* p, uint64_t node_id);
void eo_synth_fill_table(eo_synth_params* p);
int64_t eo_synth_degree(const eo_synth_params* p, uint64_t node_id);
uint64_t eo_synth_neighbor(const eo_synth_params* p, uint64_t node_id,
                           int64_t j);
float eo_synth_weight(const eo_synth_params* p, uint64_t node_id, int64_t j);
int32_t eo_synth_type(const eo_synth_params* p, uint64_t node_id, int64_t j);
int64_t eo_synth_build(const eo_synth_params* p, int64_t row_begin,
                       int64_t row_end, int64_t* row_ptr, int32_t* type_end,
                       uint64_t* nbr, float* prefix_w, float* type_prefix);

#ifdef __cplusplus
}
#endif
#endif  /* EULER_ORACLE_H_ */
