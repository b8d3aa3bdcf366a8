/*
 * euler_gpu.h - C ABI of the MI355X-native sampling / aggregation backend for
 * Euler's minibatch-construction hot path.
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++ or torch
 * types.  Each entry point names the reference interface it replaces
 * (paths relative to the alibaba/euler tree).  INTEGRATION.md shows the
 * binding a maintainer would add on the reference side.
 *
 * Conventions
 *   - Every function returns 0 on success or a negative EULER_GPU_E* code; the
 *     message is available from euler_gpu_last_error() (thread-local).  This
 *     mirrors the reference's "log and produce no output" error behaviour
 *     (core/kernels/sample_node_op.cc:118-122) with a code the caller can test.
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream).  All
 *     device work is enqueued on it; nothing synchronises unless documented.
 *   - Pointers named *_dev are device (HBM) pointers, *_host are host
 *     pointers.  Outputs are caller-allocated; their sizes are static
 *     (n * count ...), exactly the shapes of the reference ops.
 *   - Node ids are uint64 (euler::common::NodeID); the TF-level ops read and
 *     write them as int64 bit patterns (tf_euler/kernels/sample_neighbor_op.cc:110).
 *   - Random numbers: counter-based Philox4x32-10 keyed by (seed, call_id,
 *     node id | sample index, draw index) - see DESIGN.md "RNG contract".  The
 *     same (seed, call_id) always reproduces the same ids, independent of batch
 *     order, duplication, launch geometry or sharding.
 */
#ifndef EULER_GPU_H_
#define EULER_GPU_H_

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EULER_GPU_OK 0
#define EULER_GPU_EINVAL (-1)   /* bad argument                                */
#define EULER_GPU_ENOMEM (-2)   /* device or host allocation failed            */
#define EULER_GPU_EHIP (-3)     /* HIP runtime error (message has the detail)  */
#define EULER_GPU_ENOGRAPH (-4) /* no graph initialised / bad handle           */
#define EULER_GPU_EIO (-5)      /* data_path unreadable / malformed .dat       */
#define EULER_GPU_EEMPTY (-6)   /* sampler has zero total weight: no output    */

typedef struct euler_gpu_graph euler_gpu_graph;

/* Result layouts of the neighbour sampler. */
#define EULER_GPU_LAYOUT_CORE 0 /* API_SAMPLE_NB / GQL layout: empty rows are
                                   count x (0, 0.0f, 0)
                                   (core/kernels/sample_neighbor_op.cc:134-143) */
#define EULER_GPU_LAYOUT_TF 1   /* tf_euler dense layout: rows whose first id
                                   is the sentinel 0 are default_node/0.0/-1
                                   (tf_euler/kernels/sample_neighbor_op.cc:79-81,
                                   114-122)                                     */

/* Host description of a graph in the reference's own per-node storage
 * (euler/core/graph/node.h:49-57) concatenated over rows. */
typedef struct euler_gpu_host_csr {
  int64_t n_rows;
  int32_t n_edge_types;        /* edge-type groups per node                    */
  int32_t n_node_types;
  const uint64_t* row_id;      /* [n_rows] node id of each row                 */
  const int64_t* row_ptr;      /* [n_rows+1]                                   */
  const int32_t* type_end;     /* [n_rows*T] neighbor_groups_idx (row-relative)*/
  const uint64_t* nbr;         /* [E] neighbors                                */
  const float* prefix_w;       /* [E] neighbors_weight (running f32 sums)      */
  const float* type_prefix;    /* [n_rows*T] edge_group_collection running sums*/
  const int32_t* node_type;    /* [n_rows] or NULL (all type 0)                */
  const float* node_weight;    /* [n_rows] or NULL (all 1.0f)                  */
  const uint64_t* sampler_order; /* [n_rows] node ids in the order the global
                                  node sampler enumerates them
                                  (Graph::BuildGlobalSampler, graph.cc:349) or
                                  NULL = row order                             */
  /* dense float features, the reference's per-node float_features_idx_ /
   * float_features_ (core/graph/node.h): optional (n_float_features = 0) */
  int32_t n_float_features;    /* feature slots per node                       */
  int32_t pad0;
  const int64_t* feat_ptr;     /* [n_rows+1] offset of each row's values        */
  const int32_t* feat_idx;     /* [n_rows*F] cumulative ends per slot,
                                  row-relative; a missing slot repeats the
                                  previous end                                 */
  const float* feat_val;       /* [feat_ptr[n_rows]]                            */
  /* sparse (uint64) features, the reference's uint64_features_idx_ /
   * uint64_features_ (core/graph/node.h), same layout: optional */
  int32_t n_u64_features;
  int32_t pad1;
  const int64_t* ufeat_ptr;    /* [n_rows+1]                                    */
  const int32_t* ufeat_idx;    /* [n_rows*U] cumulative ends per slot           */
  const uint64_t* ufeat_val;   /* [ufeat_ptr[n_rows]]                           */
} euler_gpu_host_csr;

/* Parameters of the deterministic synthetic power-law graph (benchmarks). */
typedef struct euler_gpu_synth_params {
  uint64_t seed;
  int64_t n_nodes;             /* ids 1..n_nodes                               */
  int64_t n_edges_target;
  int32_t scale;               /* RMAT scale (bits)                            */
  int32_t n_types;
  int32_t weighted;            /* 0: all weights 1.0f, 1: uniform [0.5, 8)     */
  int32_t hashed_ids;          /* 1: node x (1..n_nodes) carries the external id mix64(x), a
                                * bijection of u64 - arbitrary ids, hash id map - instead of x  */
  double deg_table[64];        /* expected extra degree by popcount(id-1)      */
} euler_gpu_synth_params;

/* ---- library / error ---------------------------------------------------- */
const char* euler_gpu_last_error(void);
const char* euler_gpu_version(void);
int euler_gpu_device_count(void);

/* ---- graph life cycle ---------------------------------------------------
 * Replaces QueryProxy::Init local branch -> Graph::Init -> GraphBuilder::Build
 * -> BuildGlobalSampler (client/query_proxy.cc:145-190, core/graph/graph.cc:
 * 72-120,333-370): the graph becomes an immutable CSR + alias tables in HBM. */
int euler_gpu_graph_create(const euler_gpu_host_csr* csr, int device,
                           euler_gpu_graph** out);
/* Only rows owned by `shard_index` under owner(id) = (id % partitions) % shards
 * (core/kernels/id_split_op.cc:46-49) are kept (Graph::Init file filter,
 * core/graph/graph.cc:90-98). */
int euler_gpu_graph_create_shard(const euler_gpu_host_csr* csr, int device,
                                 int32_t partitions, int32_t shard_index,
                                 int32_t shards, euler_gpu_graph** out);
/* Synthetic graph generated directly in HBM.  rows are [row_begin,row_end) of
 * the global graph when sharded by contiguous ranges is wanted; pass
 * partitions/shards = 1 for the whole graph.  With shards > 1 the shard keeps
 * the nodes with owner(id) == shard_index. */
int euler_gpu_graph_create_synthetic(const euler_gpu_synth_params* p,
                                     int device, int32_t partitions,
                                     int32_t shard_index, int32_t shards,
                                     euler_gpu_graph** out);
/* Load a directory written by euler/tools (euler.meta + Node/<x>_<partition>.dat), the
 * input of the reference's Graph::Init (core/graph/graph_builder.cc:57-158,
 * core/graph/node.cc:414-526). */
int euler_gpu_graph_load(const char* data_path, int device,
                         int32_t shard_index, int32_t shards,
                         euler_gpu_graph** out);
/* Host-only view of the same reader (no GPU needed): parses the directory
 * into malloc'ed arrays described by *csr; free with euler_gpu_dat_close().
 * *partitions receives euler.meta's partitions_num. */
int euler_gpu_dat_open(const char* data_path, int32_t shard_index,
                       int32_t shards, euler_gpu_host_csr* csr,
                       int32_t* partitions, void** owner);
void euler_gpu_dat_close(void* owner);
/* Host-only check of a dataset (no GPU needed): every record of the Edge partition files (the
 * (src, dst, type) the reference's EdgeExist consults, core/api/api.cc:46-48,
 * core/graph/edge.cc:136-153) against the node rows this backend answers
 * SparseGetAdj from.  The two agree exactly when *not_in_rows == 0 and
 * *edge_records == *row_triples (distinct (src, dst, type) entries of the rows). */
int euler_gpu_dat_verify_edges(const char* data_path, int32_t shard_index, int32_t shards,
                               int64_t* edge_records, int64_t* not_in_rows,
                               int64_t* row_triples);
void euler_gpu_graph_destroy(euler_gpu_graph* g);

int64_t euler_gpu_graph_num_nodes(const euler_gpu_graph* g);
int64_t euler_gpu_graph_num_edges(const euler_gpu_graph* g);
int32_t euler_gpu_graph_num_edge_types(const euler_gpu_graph* g);
int32_t euler_gpu_graph_num_node_types(const euler_gpu_graph* g);
int euler_gpu_graph_device(const euler_gpu_graph* g);
/* Total device bytes held by the graph. */
int64_t euler_gpu_graph_bytes(const euler_gpu_graph* g);
/* euler.meta's partitions_num of a graph loaded with euler_gpu_graph_load (0 for a
 * graph built from arrays): a multi-GPU sampler must route ids with
 * owner(id) = (id % partitions) % shards using THIS value, because the loader kept
 * the partition files with file_idx % shards == shard_index
 * (core/graph/graph.cc:90-98). */
int32_t euler_gpu_graph_partitions(const euler_gpu_graph* g);
/* Per node type weight sums (Graph::GetNodeWeightSums, used by
 * SAMPLE_NODE_SPLIT); out_host has n_node_types floats. */
int euler_gpu_graph_node_weight_sums(const euler_gpu_graph* g, float* out_host);
/* Graph::BuildGlobalSampler (core/graph/graph.cc:333-370) as a call of its own: the global
 * node sampler (per node type an alias table over weight / type sum, FastWeightedCollection;
 * a type table over the type sums) of a graph that has none - a synthetic one - or in place
 * of the one it has.  The n nodes are taken in the order given, which is the order the
 * reference enumerates node_map_: ids_host NULL = the graph's rows in row order
 * (strided-identity id maps only), types_host NULL = all type 0, weights_host NULL = all
 * 1.0f.  Host work, O(n); not to be called while SampleNode calls are in flight. */
int euler_gpu_graph_set_node_sampler(euler_gpu_graph* g, int64_t n, const uint64_t* ids_host,
                                     const int32_t* types_host, const float* weights_host,
                                     int32_t n_node_types);
/* Copy rows of the device CSR back to the host for the listed ids, in the
 * euler_gpu_host_csr layout (spot checks at sizes no CPU structure can hold).
 * Call with nbr_host == NULL to obtain row_ptr_host (n+1) first. */
int euler_gpu_graph_export_rows(const euler_gpu_graph* g, const uint64_t* ids_host,
                                int64_t n, int64_t* row_ptr_host,
                                int32_t* type_end_host, uint64_t* nbr_host,
                                float* prefix_w_host, float* type_prefix_host);

/* The reference's one process-level C entry (tf_euler/utils/
 * init_query_proxy.cc:19-37, loaded by euler_ops/base.py:33-67).  Accepts the
 * same "k=v;k=v" string (mode=local; data_path; sampler_type; data_type) plus
 * `device`, `shard_idx`, `shard_num` and `verify_edges` (1: refuse a dataset whose
 * Edge records are not exactly the entries of its node rows, see
 * euler_gpu_dat_verify_edges).  Installs the process-wide default graph returned
 * by euler_gpu_default_graph(). */
bool InitQueryProxy(const char* conf);
euler_gpu_graph* euler_gpu_default_graph(void);

/* Largest node id of the graph (shard) and whether its id -> row map is the
 * strided identity (ids = base + stride * row).  A multi-GPU sampler reduces
 * max_id over the shards to size euler_gpu_dedup_split's dense table. */
int euler_gpu_graph_id_range(const euler_gpu_graph* g, uint64_t* max_id_host,
                             int32_t* identity_host);

/* ---- SampleNeighbor ------------------------------------------------------
 * Replaces euler::SampleNeighbor (core/api/api.cc:223-236) ->
 * Node::SampleNeighbor (core/graph/node.cc:98-167) -> RandomSelect
 * (common/compact_weighted_collection.h:30-52), the empty-row fill of
 * API_SAMPLE_NB, FillNeighbor (core/kernels/common.cc:275-334) and, with
 * LAYOUT_TF, the dense repack of the TF SampleNeighbor kernel.
 *   roots_dev      [n] node ids
 *   root_mask_dev  optional [ceil(n / root_group)] bytes; a non-zero byte means
 *                  "this group of roots came from a missing row": they sample
 *                  as node id 0, the id the reference's core tensors carry
 *                  (used to chain hops, tf_euler/kernels/sample_fanout_op.cc)
 *   edge_types_host[k] (k may be 0 = all types)
 *   out_*_dev      [n*count]; out_row_mask_dev optional [n] (1 = default row)
 */
int euler_gpu_sample_neighbor(const euler_gpu_graph* g, void* stream,
                              uint64_t seed, uint32_t call_id,
                              const uint64_t* roots_dev, int64_t n,
                              const uint8_t* root_mask_dev, int32_t root_group,
                              const int32_t* edge_types_host, int32_t k,
                              int32_t count, int32_t layout,
                              int64_t default_node, uint64_t* out_id_dev,
                              float* out_w_dev, int32_t* out_t_dev,
                              uint8_t* out_row_mask_dev);

/* Same call for roots the caller knows to be (mostly) distinct - e.g. the ids a
 * shard receives after euler_gpu_dedup_split: skips the on-device duplicate
 * detection of euler_gpu_sample_neighbor.  Results are identical. */
int euler_gpu_sample_neighbor_distinct(const euler_gpu_graph* g, void* stream,
                                       uint64_t seed, uint32_t call_id,
                                       const uint64_t* roots_dev, int64_t n,
                                       const int32_t* edge_types_host, int32_t k,
                                       int32_t count, int32_t layout,
                                       int64_t default_node, uint64_t* out_id_dev,
                                       float* out_w_dev, int32_t* out_t_dev,
                                       uint8_t* out_row_mask_dev);
/* Several edge-type SETS of one batch of roots in ONE launch - what a heterogeneous
 * (RGCN-style) model issues per minibatch as n_sets SampleNeighbor ops over the same nodes
 * (tf_euler/kernels/sample_neighbor_op.cc:29-132 once per relation set).  Set s lists
 * set_k_host[s] types, taken in order from edge_types_host, and draws with call_id + s; TF
 * layout; out_*_dev hold [n_sets][n][count] elements.  The result equals n_sets calls of
 * euler_gpu_sample_neighbor bit for bit (Node::__SampleNeighbor's type modes,
 * core/graph/node.cc:98-161: one listed type, a sub-collection in the listed order, all
 * groups).  Graphs the one-launch kernel does not serve (running sums that decrease, the
 * id-0 sentinel rule) get the separate calls. */
int euler_gpu_sample_neighbor_sets(const euler_gpu_graph* g, void* stream, uint64_t seed,
                                   uint32_t call_id, const uint64_t* roots_dev, int64_t n,
                                   const int32_t* edge_types_host, const int32_t* set_k_host,
                                   int32_t n_sets, int32_t count, int64_t default_node,
                                   uint64_t* out_id_dev, float* out_w_dev, int32_t* out_t_dev);
/* ... followed, in the same enqueue, by the aggregation of the sampled neighbours' feature
 * rows per (set, root): out_agg_dev [n_sets][n][d] = add / max / mean (mode 0 / 1 / 2) over
 * the `count` rows feat_dev[id] of every root, in sample order - the bits of
 * scatter_(aggr, gather(feat, ids)) (tf_euler/kernels/gather_op.cc, scatter_op.cc;
 * euler_gpu_gather_segment_reduce_ids).  feat_dev is a [feat_rows, d] f32 table indexed by
 * node id; default_node must be one of its rows. */
int euler_gpu_sample_aggregate_sets(const euler_gpu_graph* g, void* stream, uint64_t seed,
                                    uint32_t call_id, const uint64_t* roots_dev, int64_t n,
                                    const int32_t* edge_types_host, const int32_t* set_k_host,
                                    int32_t n_sets, int32_t count, int64_t default_node,
                                    int32_t mode, const float* feat_dev, int64_t feat_rows, int64_t d,
                                    uint64_t* out_id_dev, float* out_w_dev, int32_t* out_t_dev,
                                    float* out_agg_dev);
/* The shard side of a multi-GPU hop in one call: sample the (distinct) ids this
 * shard received, TF layout, and write the wire rows euler_gpu_expand_packed
 * consumes - euler_gpu_pack_rows' format: 4 * count + 2 int32 words per root
 * (ids | weights | types | row mask, pad), 3 * count + 2 (padded to even) without
 * the type column when k == 1.  Single-type calls on
 * graphs the pivot kernels serve write the rows straight from the sampling
 * kernel; every other call samples into scratch arrays and packs. */
int euler_gpu_sample_neighbor_packed(const euler_gpu_graph* g, void* stream,
                                     uint64_t seed, uint32_t call_id,
                                     const uint64_t* roots_dev, int64_t n,
                                     const int32_t* edge_types_host, int32_t k,
                                     int32_t count, int64_t default_node,
                                     int32_t* packed_dev);
/* ... for several edge-type SETS over the same ids in ONE launch (euler_gpu_sample_neighbor_sets'
 * kernel writing wire rows: the typed hops of a heterogeneous minibatch on a sharded graph).
 * Set s draws with call_id + s and its n rows begin sum over s' < s of n * words(s') int32 words into
 * packed_dev, words(s) = 4 * count + 2, or 3 * count + 2 padded to even for a set of ONE type (no
 * type column) - the rows of n_sets euler_gpu_sample_neighbor_packed calls, bit for bit in every
 * word euler_gpu_expand_packed reads.  Graphs the one-launch kernel does not serve take those calls. */
int euler_gpu_sample_neighbor_sets_packed(const euler_gpu_graph* g, void* stream, uint64_t seed,
                                          uint32_t call_id, const uint64_t* roots_dev, int64_t n,
                                          const int32_t* edge_types_host, const int32_t* set_k_host,
                                          int32_t n_sets, int32_t count, int64_t default_node,
                                          int32_t* packed_dev);

/* TF SampleFanout (tf_euler/kernels/sample_fanout_op.cc:32-148): `layers` hops
 * chained on device; hop h uses call_id + h, edge_types_host[h*k .. h*k+k) and
 * counts_host[h].  out_*_dev[h] has n * prod(counts[0..h]) elements.
 * workspace_dev must hold euler_gpu_sample_fanout_workspace() bytes. */
size_t euler_gpu_sample_fanout_workspace(int64_t n, const int32_t* counts_host,
                                         int32_t layers);
/* M minibatches of one SampleFanout in ONE enqueue - the regime of the reference's callers
 * (tf_euler/python/dataflow/sage_dataflow.py:35-50 issues one sample_fanout per minibatch of
 * a few hundred roots; its client keeps 8 such queries in flight, client/query_proxy.cc:
 * 205-210).  roots_dev holds m * n roots, minibatch b = roots [b n, (b + 1) n) draws with
 * call id call_ids_dev[b] (a DEVICE array, 2-hop single-type fanouts only) or, when that is
 * NULL, call_id + b * call_stride (hop h: + h).  out_*_dev[h] has m * n * prod(counts[0..h])
 * elements, minibatch-major.  Draws are keyed by (call id, node id), so the result equals
 * m calls of euler_gpu_sample_fanout bit for bit.  A 2-hop fanout of single listed types
 * is one launch (a workgroup per root below 8 192 roots in all, the wave-per-4-roots kernel
 * of fanout_local.h above); other shapes enqueue the minibatches one after the other.
 * workspace_dev: euler_gpu_sample_fanout_workspace(m * n, ...) bytes. */
int euler_gpu_sample_fanout_multi(const euler_gpu_graph* g, void* stream, uint64_t seed,
                                  uint32_t call_id, uint32_t call_stride,
                                  const uint32_t* call_ids_dev, int32_t m,
                                  const uint64_t* roots_dev, int64_t n,
                                  const int32_t* edge_types_host, int32_t k,
                                  const int32_t* counts_host, int32_t layers,
                                  int64_t default_node, uint64_t* const* out_id_dev,
                                  float* const* out_w_dev, int32_t* const* out_t_dev,
                                  void* workspace_dev);
int euler_gpu_sample_fanout(const euler_gpu_graph* g, void* stream,
                            uint64_t seed, uint32_t call_id,
                            const uint64_t* roots_dev, int64_t n,
                            const int32_t* edge_types_host, int32_t k,
                            const int32_t* counts_host, int32_t layers,
                            int64_t default_node, uint64_t* const* out_id_dev,
                            float* const* out_w_dev, int32_t* const* out_t_dev,
                            void* workspace_dev);

/* ---- SampleNode -----------------------------------------------------------
 * Replaces euler::SampleNode (core/api/api.cc:32-37) -> Graph::SampleNode
 * (core/graph/graph.cc:221-275) -> AliasMethod::Next (common/alias_method.cc:
 * 66-78).  node_types_host[k]: k == 1 && type == -1 samples over all types.
 * Returns EULER_GPU_EEMPTY (and writes nothing) when the weight sum is 0. */
int euler_gpu_sample_node(const euler_gpu_graph* g, void* stream, uint64_t seed,
                          uint32_t call_id, const int32_t* node_types_host,
                          int32_t k, int32_t count, uint64_t* out_dev);

/* API_GET_NODE_T / TF GetNodeType (core/kernels/get_node_type_op.cc:35-62,
 * tf_euler/kernels/get_node_type_op.cc:33-57): the node type of every id;
 * INT32_MIN (euler::common::DEFAULT_INT32) for an unknown node. */
int euler_gpu_get_node_type(const euler_gpu_graph* g, void* stream,
                            const uint64_t* ids_dev, int64_t n, int32_t* out_dev);

/* API_SAMPLE_N_WITH_TYPES / TF SampleNWithTypes (core/kernels/
 * sample_n_with_types_op.cc:33-75, tf_euler/kernels/sample_n_with_types_op.cc:
 * 38-84; euler_ops sample_node_with_src): row i of out_dev [n, count] = one
 * Graph::SampleNode(types_dev[i], count) call (type -1 = all types).  RNG:
 * domain 1, stream = i.  A type that is out of range or has zero weight makes
 * the reference's TF kernel abort ("samples size error, invalid node types!");
 * here the call returns EULER_GPU_EEMPTY.  Synchronises the stream. */
int euler_gpu_sample_n_with_types(const euler_gpu_graph* g, void* stream, uint64_t seed,
                                  uint32_t call_id, const int32_t* types_dev, int64_t n,
                                  int32_t count, uint64_t* out_dev);

/* ---- full neighbours ------------------------------------------------------
 * euler::GetFullNeighbor (core/api/api.cc:208-221) -> Node::GetFullNeighbor
 * (core/graph/node.cc:175-197) in the FillNeighbor layout.  Two calls: first
 * with out_id_dev == NULL fills idx_dev [n,2] (int32 offsets) and *total_host
 * (synchronises the stream); the second call writes the values (out_t_dev may be
 * NULL: the edge types are not written - the sharded node2vec walk does not read them). */
int euler_gpu_get_full_neighbor(const euler_gpu_graph* g, void* stream,
                                const uint64_t* ids_dev, int64_t n,
                                const int32_t* edge_types_host, int32_t k,
                                int32_t* idx_dev, int64_t* total_host,
                                uint64_t* out_id_dev, float* out_w_dev,
                                int32_t* out_t_dev);

/* Post-process of API_GET_NB_NODE (core/kernels/get_neighbor_op.cc:117-168) on
 * the result of euler_gpu_get_full_neighbor, in place: order_by (0 none, 1 id,
 * 2 weight; desc != 0 = descending) then limit (< 0 = none).  Rows are
 * re-packed, idx_dev rewritten, *total_host receives the new total (stream
 * sync).  This is what the TF ops GetSortedFullNeighbor (`order_by(id, asc)`,
 * tf_euler/kernels/get_sorted_full_neighbor_op.cc:43-46) and GetTopKNeighbor
 * (`order_by(weight, desc).limit(k)`, get_top_k_neighbor_op.cc:36-44) run.
 * Equal keys keep storage order (undefined in the reference: std::sort with a
 * non-strict comparator). */
int euler_gpu_neighbor_post_process(void* stream, int64_t n, int32_t* idx_dev,
                                    int64_t total, uint64_t* id_dev, float* w_dev,
                                    int32_t* t_dev, int32_t order_by, int32_t desc,
                                    int64_t limit, int64_t* total_host);
/* tf_euler get_top_k_neighbor (tf_euler/kernels/get_top_k_neighbor_op.cc: GQL
 * `v(nodes).outV(edge_types).order_by(weight, desc).limit(k)`, dense fill) in
 * one kernel: out_*_dev [n, k] = the k heaviest neighbours of every node over
 * the listed edge types (ties keep the order of Node::GetFullNeighbor,
 * core/graph/node.cc:175-197), default_node / 0.0 / -1 where a node has fewer.
 * Same result as euler_gpu_get_full_neighbor + euler_gpu_neighbor_post_process
 * (weight, descending, limit k) + euler_gpu_neighbor_to_dense. */
int euler_gpu_get_top_k_neighbor(const euler_gpu_graph* g, void* stream,
                                 const uint64_t* ids_dev, int64_t n,
                                 const int32_t* edge_types_host, int32_t k_types,
                                 int32_t k, int64_t default_node, uint64_t* out_id_dev,
                                 float* out_w_dev, int32_t* out_t_dev);

/* Dense [n, k] fill of the TF GetTopKNeighbor kernel
 * (tf_euler/kernels/get_top_k_neighbor_op.cc:70-75,101-109): default_node /
 * 0.0 / -1 where a row has fewer than k entries. */
int euler_gpu_neighbor_to_dense(void* stream, int64_t n, const int32_t* idx_dev,
                                const uint64_t* id_dev, const float* w_dev,
                                const int32_t* t_dev, int32_t k, int64_t default_node,
                                int64_t* out_id_dev, float* out_w_dev,
                                int32_t* out_t_dev);

/* ---- layerwise sampling (GQL sampleLNB without a weight function) -----------
 * The DAG euler/parser/translator.cc:338-386,489-527 builds for
 * `v(nodes).sampleLNB(edge_types, n, m, default_node)`:
 * API_GET_EDGE_SUM_WEIGHT -> API_SAMPLE_ROOT -> API_SAMPLE_L ->
 * API_SPARSE_GEN_ADJ -> API_SPARSE_GET_ADJ -> API_GATHER_RESULT.
 * With a weight function the first three are API_GET_NB_NODE ->
 * API_LOCAL_SAMPLE_L (euler_gpu_local_sample_layer below). */

/* API_GET_EDGE_SUM_WEIGHT (core/kernels/get_edge_sum_weight_op.cc:33-66):
 * out_w_dev[i] = f32 sum, in Node::GetFullNeighbor order (core/graph/node.cc:
 * 175-197), of the weights of ids_dev[i]'s out-edges of the listed types; 0 for
 * an unknown node. */
int euler_gpu_get_edge_sum_weight(const euler_gpu_graph* g, void* stream,
                                  const uint64_t* ids_dev, int64_t n,
                                  const int32_t* edge_types_host, int32_t k,
                                  float* out_w_dev);

/* API_SAMPLE_ROOT (core/kernels/sample_root_op.cc:33-88): for every batch row b,
 * a FastWeightedCollection (common/fast_weighted_collection.h:55-75 over
 * AliasMethod::Init, common/alias_method.cc:23-63) of roots_dev[b*n .. b*n+n)
 * with weights_dev[b*n ..]; m draws per row -> out_dev [batch * m]; rows whose
 * f32 weight sum is 0 give default_node.  RNG: domain 4, stream = b, draws 2j
 * (column) and 2j+1 (coin) for draw j. */
int euler_gpu_sample_root(void* stream, uint64_t seed, uint32_t call_id,
                          const uint64_t* roots_dev, const float* weights_dev,
                          int64_t batch, int32_t n, int32_t m, int64_t default_node,
                          uint64_t* out_dev);

/* API_SAMPLE_L (core/kernels/sample_layer_op.cc:32-72): position i receives ONE
 * neighbour of roots_dev[i] (euler::SampleNeighbor({root}, edge_types, 1)) or
 * (default_node, 0.0, 0) when that yields nothing.  RNG: domain 5, stream = the
 * position i (a root listed twice is sampled twice, as in the reference).
 * out_w_dev / out_t_dev may be NULL. */
int euler_gpu_sample_layer(const euler_gpu_graph* g, void* stream, uint64_t seed,
                           uint32_t call_id, const uint64_t* roots_dev, int64_t n,
                           const int32_t* edge_types_host, int32_t k,
                           int64_t default_node, uint64_t* out_id_dev,
                           float* out_w_dev, int32_t* out_t_dev);

/* The same draw with explicit RNG streams: root i samples as position
 * pos_dev[i].  A shard of a multi-GPU hop answers the requester's positions, so
 * the sharded result equals the single-GPU one. */
int euler_gpu_sample_layer_at(const euler_gpu_graph* g, void* stream, uint64_t seed,
                              uint32_t call_id, const uint64_t* roots_dev,
                              const int64_t* pos_dev, int64_t n,
                              const int32_t* edge_types_host, int32_t k,
                              int64_t default_node, uint64_t* out_id_dev,
                              float* out_w_dev, int32_t* out_t_dev);

/* The three ops above chained on the stream = output 0 of the TF kernel
 * SampleNeighborLayerwiseWithAdj (tf_euler/kernels/
 * sample_neighbor_layerwise_with_adj_op.cc:56-150, weight_func == ""):
 * nodes_dev [batch, n] -> out_dev [batch, count].  Its sparse adjacency outputs
 * are euler_gpu_sparse_get_adj_tf(nodes_dev, out_dev, batch, n, count, ...). */
int euler_gpu_sample_neighbor_layerwise(const euler_gpu_graph* g, void* stream,
                                        uint64_t seed, uint32_t call_id,
                                        const uint64_t* nodes_dev, int64_t batch,
                                        int32_t n, const int32_t* edge_types_host,
                                        int32_t k, int32_t count,
                                        int64_t default_node, uint64_t* out_dev);

/* API_LOCAL_SAMPLE_L (core/kernels/local_sample_layer_op.cc:43-146), the
 * sampler of `sampleLNB(edge_types, n, m, weight_func, default_node)`: per batch
 * row the distinct (neighbour id, edge type) pairs among the full neighbours of
 * its n nodes (idx_dev [batch*n, 2] / ids_dev / w_dev / t_dev [total]: the
 * API_GET_NB_NODE result, e.g. euler_gpu_get_full_neighbor), weights of
 * duplicates added, then `sqrt` when weight_func is "sqrt" (any other string
 * leaves them as they are, :86-95), then m draws by CDF inversion -> out_*_dev
 * [batch * m].  The candidate ORDER is the iteration order of the op's
 * std::unordered_map<std::string, ...>; the tables are therefore built on the
 * host with that very container (a stream sync each way), the draws run on the
 * device.  Rows with no candidates or zero weight are memset like the op does:
 * every BYTE of the ids = default_node's low byte.  RNG: domain 6, stream = b. */
int euler_gpu_local_sample_layer(void* stream, uint64_t seed, uint32_t call_id,
                                 const int32_t* idx_dev, const uint64_t* ids_dev,
                                 const float* w_dev, const int32_t* t_dev, int64_t total,
                                 int64_t batch, int32_t n, int32_t m,
                                 const char* weight_func, int64_t default_node,
                                 uint64_t* out_id_dev, float* out_w_dev,
                                 int32_t* out_t_dev);

/* API_SPARSE_GET_ADJ (core/kernels/sparse_get_adj_op.cc:35-92) with the
 * root_batch tensor API_SPARSE_GEN_ADJ builds (sparse_gen_adj_op.cc:52-61:
 * root r belongs to batch row r / n): for every root the candidates
 * l_nb_dev[b*m .. b*m+m) it has an edge of a listed type to, in candidate
 * order.  FillNeighbor-style result: idx_dev [batch*n, 2] int32 offsets +
 * out_id_dev [total].  Two calls like euler_gpu_get_full_neighbor: with
 * out_id_dev == NULL it answers the query (one bit per (root, candidate) kept
 * in workspace_dev), fills idx_dev and *total_host (stream sync); the second
 * call writes the ids from the workspace.  workspace_dev:
 * euler_gpu_sparse_get_adj_workspace(batch, n, m) bytes, 8-byte aligned, kept
 * by the caller between the two calls.  EdgeExist (core/api/api.cc:46-48) is
 * answered from the adjacency rows: the Edge records the reference would
 * consult hold the same (src, dst, type) triples in data written by its
 * converter. */
size_t euler_gpu_sparse_get_adj_workspace(int64_t batch, int32_t n, int32_t m);
int euler_gpu_sparse_get_adj(const euler_gpu_graph* g, void* stream,
                             const uint64_t* roots_dev, const uint64_t* l_nb_dev,
                             int64_t batch, int32_t n, int32_t m,
                             const int32_t* edge_types_host, int32_t k,
                             void* workspace_dev, int32_t* idx_dev,
                             int64_t* total_host, uint64_t* out_id_dev);

/* TF SparseGetAdj (tf_euler/kernels/sparse_get_adj_op.cc:43-134) and the
 * adjacency outputs of SampleNeighborLayerwiseWithAdj: COO triples (b, j, c) in
 * row-major order with value 1 where nodes[b, j] has a listed-type edge to
 * nb_nodes[b, c], plus the kernel's explicit 0 at (b, n-1, m-1) when that pair
 * is no edge (so dense_shape = [batch, n, m]).  Two calls over the same
 * workspace (see above): indices_dev == NULL answers the query and returns
 * *nnz_host (stream sync); then indices_dev [nnz, 3] / values_dev [nnz] int64. */
int euler_gpu_sparse_get_adj_tf(const euler_gpu_graph* g, void* stream,
                                const uint64_t* nodes_dev, const uint64_t* nb_nodes_dev,
                                int64_t batch, int32_t n, int32_t m,
                                const int32_t* edge_types_host, int32_t k,
                                void* workspace_dev, int64_t* nnz_host,
                                int64_t* indices_dev, int64_t* values_dev);

/* SparseGetAdj on a SHARDED graph, owner side and requester side.  Every shard computes the
 * hit mask of the sources it owns - mask_dev [batch * n][(m + 63) / 64] uint64, bit c of a
 * source = candidate c of its batch row is in the source's row; sources without a row on
 * this shard give zeros - and the masks of all shards OR to the mask of the whole graph.  The
 * requester turns the combined mask into the TF triple (two calls as
 * euler_gpu_sparse_get_adj_tf; workspace 8 * (batch * n + 1) bytes).  Replaces shipping the
 * sources' whole rows (core/kernels/sparse_get_adj_op.cc:35-92 is the per-shard op). */
int euler_gpu_sparse_adj_mask(const euler_gpu_graph* g, void* stream, const uint64_t* nodes_dev,
                              const uint64_t* nb_nodes_dev, int64_t batch, int32_t n, int32_t m,
                              const int32_t* edge_types_host, int32_t k, uint64_t* mask_dev);
int euler_gpu_sparse_adj_from_mask_tf(void* stream, const uint64_t* mask_dev, int64_t batch,
                                      int32_t n, int32_t m, void* workspace_dev,
                                      int64_t* nnz_host, int64_t* indices_dev,
                                      int64_t* values_dev);

/* ---- dense features --------------------------------------------------------
 * TF GetDenseFeature (tf_euler/kernels/get_dense_feature_op.cc:63-125) over
 * Node::GetFloat32Feature (core/graph/node.cc:330-394): out_dev is [n, dim]
 * float32, zero filled; row j receives the stored values of feature slot `fid`
 * of node nodes_dev[j] (unknown node / slot: zeros).  A node that stores more
 * than `dim` values would overrun the reference's output row; here the row is
 * truncated to dim. */
int32_t euler_gpu_graph_num_float_features(const euler_gpu_graph* g);
int euler_gpu_get_dense_feature(const euler_gpu_graph* g, void* stream,
                                const uint64_t* nodes_dev, int64_t n, int32_t fid,
                                int32_t dim, float* out_dev);

/* ---- sparse (uint64) features ----------------------------------------------
 * TF GetSparseFeature (tf_euler/kernels/get_sparse_feature_op.cc:52-131) over
 * Node::GetUint64Feature (core/graph/node.cc:330-372), one feature slot per
 * call: the COO triple of the [n, max_len] SparseTensor - for node j the
 * entries (j, 0..len-1) = its stored values, or ONE entry (j, 0) =
 * default_value when it stores none (unknown node / slot included).
 * Two calls: with indices_dev == NULL it fills row_off_dev [n + 1] (int64
 * scratch the second call reads), *nnz_host and *max_len_host (dense_shape =
 * [n, max_len]; stream sync); then indices_dev [nnz, 2] / values_dev [nnz]. */
int32_t euler_gpu_graph_num_u64_features(const euler_gpu_graph* g);
int euler_gpu_get_sparse_feature(const euler_gpu_graph* g, void* stream,
                                 const uint64_t* nodes_dev, int64_t n, int32_t fid,
                                 int64_t default_value, int64_t* row_off_dev,
                                 int64_t* nnz_host, int64_t* max_len_host,
                                 int64_t* indices_dev, int64_t* values_dev);

/* The same feature slot in the GQL `values(...)` layout of API_GET_P
 * (core/kernels/get_feature_op.cc:34-70): idx_dev [n, 2] int32 offsets + the
 * packed values, no default entries - what a shard returns to the requester in
 * the multi-GPU path.  Two calls like euler_gpu_get_full_neighbor. */
int euler_gpu_get_sparse_feature_core(const euler_gpu_graph* g, void* stream,
                                      const uint64_t* nodes_dev, int64_t n, int32_t fid,
                                      int32_t* idx_dev, int64_t* total_host,
                                      uint64_t* values_dev);

/* ---- block construction (SageDataFlow) on the device ---------------------------
 * tf_euler/python/dataflow/sage_dataflow.py:35-50 + neighbor_dataflow.py:84-110
 * (UniqueDataFlow.produce_subgraph) for `layers` hops, enqueued on `stream` with NO
 * host round trip in between: hop h samples fanouts_host[h] neighbours
 * (edge_types_host[h*k .. h*k+k), call id call_id + h, default_node) of the layer's
 * nodes, takes the first-occurrence unique (tf.unique) of [neighbours | nodes] as
 * the next layer and emits the block.  Arrays are sized for the worst case
 * cap_0 = n, cap_{h+1} = cap_h * (fanouts[h] + 1); the true sizes are
 * counts_dev [layers + 1] (uint32: counts[0] = n, counts[h+1] = nodes of layer
 * h + 1), which the host may read after the call or never.  Block h:
 *   n_id_dev[h]      [cap_{h+1}]                 new_n_id (valid: counts[h+1])
 *   res_n_id_dev[h]  [cap_h]                     index of every node of layer h in
 *                                                new_n_id (valid: counts[h])
 *   edge_src_dev[h], edge_dst_dev[h]  [cap_h * (fanouts[h] + 1)]   edge_index rows;
 *                    valid: counts[h] * fanouts[h] (+ counts[h] self loops when
 *                    add_self_loops != 0)
 * workspace_dev: euler_gpu_sage_blocks_workspace(n, fanouts_host, layers) bytes. */
size_t euler_gpu_sage_blocks_workspace(int64_t n, const int32_t* fanouts_host, int32_t layers);
int euler_gpu_sage_blocks(const euler_gpu_graph* g, void* stream, uint64_t seed,
                          uint32_t call_id, const uint64_t* roots_dev, int64_t n,
                          const int32_t* edge_types_host, int32_t k,
                          const int32_t* fanouts_host, int32_t layers, int64_t default_node,
                          int32_t add_self_loops, void* workspace_dev,
                          uint64_t* const* n_id_dev, int64_t* const* res_n_id_dev,
                          int64_t* const* edge_src_dev, int64_t* const* edge_dst_dev,
                          uint32_t* counts_dev);

/* n_mb minibatches of n roots each as ONE enqueue - what n_mb consecutive euler_gpu_sage_blocks
 * calls produce, bit for bit: the reference's GraphSAGE callers build a flow per minibatch of a
 * few hundred roots (sage_dataflow.py:35-50; examples/graphsage/run_graphsage.py:35 defaults to
 * 32), which neither fills a GPU nor amortises the dozen launches of a flow.  roots_dev
 * [n_mb * n], minibatch-major; minibatch b draws hop h with call_id + b * call_stride + h (what
 * consecutive calls take from a counter with call_stride = layers).  Every output is n_mb copies
 * of the single call's array, each sized for the worst case and b * cap apart: n_id_dev[h]
 * [n_mb][cap_{h+1}], res_n_id_dev[h] [n_mb][cap_h], edge_src_dev[h] / edge_dst_dev[h]
 * [n_mb][cap_{h+1}], counts_dev [n_mb][layers + 1].  Hops that list ONE edge type on a graph
 * without the id-0 sentinel rule run as one launch per kernel of the flow (a minibatch per
 * blockIdx.y); other shapes are served by the separate calls, one after the other.
 * workspace_dev: euler_gpu_sage_blocks_multi_workspace(n_mb, n, fanouts_host, layers) bytes. */
size_t euler_gpu_sage_blocks_multi_workspace(int32_t n_mb, int64_t n, const int32_t* fanouts_host,
                                             int32_t layers);
int euler_gpu_sage_blocks_multi(const euler_gpu_graph* g, void* stream, uint64_t seed,
                                uint32_t call_id, uint32_t call_stride, int32_t n_mb,
                                const uint64_t* roots_dev, int64_t n,
                                const int32_t* edge_types_host, int32_t k,
                                const int32_t* fanouts_host, int32_t layers, int64_t default_node,
                                int32_t add_self_loops, void* workspace_dev,
                                uint64_t* const* n_id_dev, int64_t* const* res_n_id_dev,
                                int64_t* const* edge_src_dev, int64_t* const* edge_dst_dev,
                                uint32_t* counts_dev);

/* The full-neighbour dataflows (tf_euler/python/dataflow/gcn_dataflow.py:33-47 GCNDataFlow,
 * relation_dataflow.py:30-72 RelationDataFlow: every hop takes ALL neighbours of the listed
 * edge types of the nodes seen so far, then tf.unique of [neighbours | nodes], res_n_id and
 * edge_index) enqueued without a host round trip between the hops.  Row lengths are
 * data: the caller gives the capacity of every hop's edge list, edge_caps_host[h]; layer
 * h + 1 then holds at most cap_n[h] + edge_caps[h] nodes (cap_n[0] = n).  A hop whose rows
 * do not fit sets the overflow word and contributes no edges - the host sees it in its one
 * read of counts_dev and repeats the flow with larger capacities (or op by op).
 *   n_id_dev[h]      [cap_n[h + 1]]              nodes of layer h + 1 (first-occurrence order)
 *   res_n_id_dev[h]  [cap_n[h]]                  index of layer h's nodes in layer h + 1
 *   edge_src_dev[h], edge_dst_dev[h]  [edge_caps[h] + cap_n[h]]   edges, then (add_self_loops) one
 *                                                self loop per node of layer h
 *   e_type_dev[h]    [edge_caps[h]] or e_type_dev == NULL   the edge type of every edge (RGCN's e_id)
 *   counts_dev       uint32 [2 * layers + 2]: nodes per layer [layers + 1], edges per hop
 *                    [layers] (self loops not counted), overflow [1] */
size_t euler_gpu_full_blocks_workspace(int64_t n, const int64_t* edge_caps_host, int32_t layers);
int euler_gpu_full_blocks(const euler_gpu_graph* g, void* stream, const uint64_t* roots_dev,
                          int64_t n, const int32_t* edge_types_host, int32_t k, int32_t layers,
                          int32_t add_self_loops, const int64_t* edge_caps_host,
                          void* workspace_dev, uint64_t* const* n_id_dev,
                          int64_t* const* res_n_id_dev, int64_t* const* edge_src_dev,
                          int64_t* const* edge_dst_dev, int32_t* const* e_type_dev,
                          uint32_t* counts_dev);

/* ---- RandomWalk -------------------------------------------------------------
 * TF RandomWalk kernel (tf_euler/kernels/random_walk_op.cc:172-291):
 * |p-1|,|q-1| <= 1e-6 -> chain of count=1 SampleNeighbor hops (:207-247), else
 * node2vec with BuildWeights (:83-168).  edge_types_host is [walk_len, k];
 * out_dev is [n, walk_len+1] int64. */
int euler_gpu_random_walk(const euler_gpu_graph* g, void* stream, uint64_t seed,
                          uint32_t call_id, const int64_t* nodes_dev, int64_t n,
                          const int32_t* edge_types_host, int32_t k,
                          int32_t walk_len, float p, float q,
                          int64_t default_node, int64_t* out_dev);

/* One node2vec step over explicit neighbour lists, as the reference's client runs it
 * (tf_euler/kernels/random_walk_op.cc:83-168: RWCallback::operator(), BuildWeights,
 * CompactWeightedCollection::Sample) on the lists the `v(nodes).outV(edge_types)` query
 * returned - no graph: on a sharded graph the requester calls this on fetched rows.
 * Walker i's child list is row c_row[i] of the packed rows (c_idx [rows, 2] int32 offsets,
 * c_ids, c_w - c_entries ids / weights in all), its parent's list row p_row[i] of (p_idx,
 * p_ids) (p_row NULL on the first
 * step: no parent lists yet), parent_ids[i] its previous node (the start node on the first
 * step).  out[i] = the next node, default_node for an empty child list.  RNG: domain WALK,
 * stream = walker index i, draw 0 of call_id (the single-GPU random_walk uses call_id +
 * step).  One wave per walker runs the single-GPU walk's two-cursor merge with integer
 * running sums (csrc/n2v_kernels.h) over the fetched rows; tuning key 7 = 0 selects the
 * lane-per-walker reference loop. */
int euler_gpu_node2vec_step(void* stream, uint64_t seed, uint32_t call_id, int64_t n,
                            const int32_t* c_row_dev, const int32_t* c_idx_dev,
                            const uint64_t* c_ids_dev, const float* c_w_dev, int64_t c_entries,
                            const int32_t* p_row_dev, const int32_t* p_idx_dev,
                            const uint64_t* p_ids_dev, const int64_t* parent_ids_dev,
                            float p, float q, int64_t default_node, int64_t* out_dev);

/* GenPair (tf_euler/kernels/gen_pair_op.cc:42-95): paths [batch,path_len] ->
 * pairs [batch, pair_count, 2]. */
int64_t euler_gpu_gen_pair_count(int64_t path_len, int32_t left_win,
                                 int32_t right_win);
int euler_gpu_gen_pair(void* stream, const int64_t* paths_dev, int64_t batch,
                       int64_t path_len, int32_t left_win, int32_t right_win,
                       int64_t* out_dev);

/* ---- ID_UNIQUE / IDX_GATHER / DATA_GATHER ----------------------------------
 * core/kernels/id_unique_op.cc:35-64 (first-occurrence order),
 * idx_gather_op.cc:33-55, data_gather_op.cc:33-80.
 * euler_gpu_id_unique synchronises the stream to return *n_unique_host. */
int euler_gpu_id_unique(void* stream, const uint64_t* ids_dev, int64_t n,
                        uint64_t* unique_dev, int32_t* gather_idx_dev,
                        int64_t* n_unique_host);
int euler_gpu_idx_gather(void* stream, const int32_t* idx_dev,
                         const int32_t* gather_idx_dev, int64_t n,
                         int32_t* out_idx_dev, int64_t* total_host);
int euler_gpu_data_gather(void* stream, const void* data_dev, int32_t elem_size,
                          const int32_t* idx_dev, const int32_t* gather_idx_dev,
                          const int32_t* out_idx_dev, int64_t n, void* out_dev);

/* ---- message passing -----------------------------------------------------
 * MPScatterAdd / MPScatterMax / MPGather (tf_euler/kernels/scatter_op.cc:27-105,
 * gather_op.cc:26-59).  fp32 data, int32 indices.  Scatter results are
 * order-faithful: each output element adds its updates in input order, so
 * they are bit-identical to the reference's sequential loop. */
int euler_gpu_scatter_add(void* stream, const float* updates_dev,
                          const int32_t* indices_dev, int64_t e, int64_t d,
                          int32_t size, float* out_dev);
int euler_gpu_scatter_max(void* stream, const float* updates_dev,
                          const int32_t* indices_dev, int64_t e, int64_t d,
                          int32_t size, float* out_dev);
/* scatter_mean of euler_ops/mp_ops.py:65-69 - scatter_add(x) / (scatter_add(ones) +
 * 1e-7) - in ONE pass over the segments (a destination's count is its segment
 * length); same bits as the composition.  e < 2^24. */
int euler_gpu_scatter_mean(void* stream, const float* updates_dev,
                           const int32_t* indices_dev, int64_t e, int64_t d,
                           int32_t size, float* out_dev);
/* scatter_{add,max,mean}(gather(params, gather_indices), scatter_indices, size) of a
 * message-passing step (tf_euler/python/convolution: x_j = gather(x, edge_index[1]);
 * scatter_(aggr, x_j, edge_index[0])) in ONE pass: a destination's updates are read
 * straight from the rows of `params` they name, in input order - the bits of the
 * composition, without writing and re-reading the e x d block of gathered rows (a third
 * of the composition's HBM traffic).  mode: 0 add, 1 max, 2 mean.  Indices are int32,
 * gather indices must be valid rows of params. */
int euler_gpu_gather_scatter(void* stream, int32_t mode, const float* params_dev,
                             const int32_t* gather_indices_dev,
                             const int32_t* scatter_indices_dev, int64_t e, int64_t d,
                             int32_t size, float* out_dev);
/* The same reduction when the caller knows the segments (a sampled block: `count`
 * neighbours per destination, or CSR offsets): out[r] = reduce over p in
 * [seg_ptr[r], seg_ptr[r+1]) - or [r * count, (r+1) * count) when seg_ptr_dev is NULL -
 * of params[gather_indices[p]] (gather_indices_dev NULL: of params[p]), in that order.
 * No sortedness check of a key column, no host wait.  mean divides by (length + 1e-7)
 * as scatter_mean does. */
int euler_gpu_gather_segment_reduce(void* stream, int32_t mode, const float* params_dev,
                                    const int32_t* gather_indices_dev,
                                    const int64_t* seg_ptr_dev, int64_t count, int64_t d,
                                    int32_t size, float* out_dev);
/* ... with the gather indices given as the int64 ids a sampler returned (SampleNeighbor's
 * output feeding the aggregation of a block): index p is the low 32 bits of
 * gather_ids_dev[p] - the int32 MPGather would receive after a cast
 * (tf_euler/kernels/gather_op.cc:26-59 takes int32 indices) - read in place, without the
 * cast's pass over the ids.  params_rows = the rows of params_dev (< 2^31): an id at or past
 * it - a neighbour that is not a node of this graph - reads the LAST row, never memory outside
 * the table (0 = the caller vouches for the ids). */
int euler_gpu_gather_segment_reduce_ids(void* stream, int32_t mode, const float* params_dev,
                                        int64_t params_rows, const int64_t* gather_ids_dev,
                                        const int64_t* seg_ptr_dev, int64_t count, int64_t d,
                                        int32_t size, float* out_dev);
int euler_gpu_gather(void* stream, const float* params_dev,
                     const int32_t* indices_dev, int64_t e, int64_t d,
                     int64_t n_params, float* out_dev);

/* ---- shard ops (multi-GPU) --------------------------------------------------
 * ID_SPLIT (core/kernels/id_split_op.cc:46-99): stable bucket of ids by
 * owner(id) = (id % partitions) % shards.  shard_off_host [shards+1] is
 * returned after a stream sync; shard_ids_dev / merge_idx_dev have n entries.
 * euler_gpu_merge_rows is IDX_MERGE/DATA_MERGE (idx_merge_op.cc:61-77,
 * data_merge_op.cc:44-67) for fixed-size rows: out[merge_idx[j]] = in[j]. */
int euler_gpu_id_split(void* stream, const uint64_t* ids_dev, int64_t n,
                       int32_t partitions, int32_t shards,
                       int64_t* shard_off_host, uint64_t* shard_ids_dev,
                       int32_t* merge_idx_dev);
/* InflateIdx (tf_euler/kernels/inflate_idx_op.cc:34-66, op InflateIdx of tf_euler/ops/util_ops.cc):
 * idx_dev holds n values that are exactly 0 .. U-1 for some U; out_dev[i] = the place of entry
 * i after a stable sort by value (entries with a smaller value + equal entries before i).
 * EULER_GPU_EINVAL when a value lies outside [0, number of distinct values) - the reference's
 * InvalidArgument; out_dev is then unspecified.  Synchronizes the stream (the check). */
int euler_gpu_inflate_idx(void* stream, const int32_t* idx_dev, int64_t n, int32_t* out_dev);
/* Split sizes between the ranks of one node without the GPU: an all-to-all of
 * up to 8 int64 per peer through a POSIX shared-memory mailbox.  Replaces what
 * the reference carries inside its per-shard gRPC requests / replies
 * (core/kernels/remote_op.cc:62-142); RCCL needs both sides' counts on the host
 * before the data moves.  One rank creates the region (create = 1, the name is
 * new), the others open it afterwards; every rank then calls
 * euler_shm_alltoall_i64 in the same order (send / recv: [world * width], peer
 * p's message at p * width).  timeout_ms <= 0 means 60 s. */
typedef struct euler_shm euler_shm;
int euler_shm_open(const char* name, int32_t rank, int32_t world, int32_t create,
                   euler_shm** out);
int euler_shm_alltoall_i64(euler_shm* s, const int64_t* send, int64_t* recv,
                           int32_t width, int64_t timeout_ms);
int32_t euler_shm_attached(const euler_shm* s);
int euler_shm_unlink(euler_shm* s);
void euler_shm_close(euler_shm* s);

/* Front end of a multi-GPU hop in one call: the DISTINCT ids of the batch
 * (ID_UNIQUE, which the reference's optimizer puts before ID_SPLIT:
 * parser/compiler.cc:76-90) bucketed by owner into shard_ids_dev (<= n entries,
 * shard s at [shard_off_host[s], shard_off_host[s+1])), and pos_dev[i] = index
 * in shard_ids_dev of ids_dev[i].  The shards answer in the order they were
 * asked, so row pos_dev[i] of the concatenated answers is position i's row:
 * euler_gpu_expand_rows finishes the hop (IDX_MERGE / DATA_MERGE / DATA_GATHER
 * in one pass).  An id may occur more than once in shard_ids_dev (hash-slot
 * collisions are not resolved); results are unaffected.  Synchronises once.
 * dense_owner_dev (optional): scratch of dense_limit + 1 uint32 the caller keeps
 * between calls (contents irrelevant) when every id of the graph is below
 * dense_limit (euler_gpu_graph_id_range): duplicates are then found with one
 * store and one load per id in a table indexed by the id itself instead of two
 * rounds of hashing; ids >= dense_limit are "no such node" and share one
 * representative (every owner answers the default row for them).
 * root_mask_dev (optional, [ceil(n / root_group)] bytes) has the meaning it has
 * in euler_gpu_sample_neighbor: marked groups sample as node id 0. */
int euler_gpu_dedup_split(void* stream, const uint64_t* ids_dev, int64_t n,
                          const uint8_t* root_mask_dev, int32_t root_group,
                          int32_t partitions, int32_t shards, uint32_t* dense_owner_dev,
                          int64_t dense_limit, int64_t* shard_off_host,
                          uint64_t* shard_ids_dev, int32_t* pos_dev);
/* The same call in two halves, for callers that keep several minibatches in
 * flight from one host thread: _begin enqueues everything (the bucket sizes end
 * in a pinned buffer of the handle, an event marks their arrival) and returns;
 * _end waits for that event only and hands out shard_off_host [shards + 1].  One
 * call in flight per handle; shard_ids_dev / pos_dev are valid in stream order
 * after _begin. */
typedef struct euler_gpu_front euler_gpu_front;
int euler_gpu_front_create(euler_gpu_front** out);
void euler_gpu_front_destroy(euler_gpu_front* f);
int euler_gpu_dedup_split_begin(euler_gpu_front* f, void* stream, const uint64_t* ids_dev,
                                int64_t n, const uint8_t* root_mask_dev, int32_t root_group,
                                int32_t partitions, int32_t shards,
                                uint32_t* dense_owner_dev, int64_t dense_limit,
                                uint64_t* shard_ids_dev, int32_t* pos_dev);
int euler_gpu_dedup_split_end(euler_gpu_front* f, int64_t* shard_off_host);
/* Wire format of the result exchange: one int32 row per root = [ids (2 words
 * each) | weights | types | mask | pad], 4 * count + 2 words; with ONE listed
 * edge type (single_type >= 0) the type column stays off the wire - 3 * count +
 * 2 words, padded to an even number (rows stay 8-byte aligned) - and is rebuilt on arrival (single_type, or -1 for a masked row).
 * pack_rows writes the rows from the sampler's outputs [m, count] (+ row mask
 * [m]); expand_packed reads the concatenated answers back per position through
 * pos_dev (merge + gather + unpack in one pass).  single_type = -1: full rows. */
int euler_gpu_pack_rows(void* stream, const uint64_t* id_dev, const float* w_dev,
                        const int32_t* t_dev, const uint8_t* mask_dev, int64_t m,
                        int32_t count, int32_t single_type, int32_t* packed_dev);
int euler_gpu_expand_packed(void* stream, const int32_t* pos_dev, int64_t n,
                            int32_t count, int32_t single_type, const int32_t* packed_dev,
                            uint64_t* out_id_dev, float* out_w_dev, int32_t* out_t_dev,
                            uint8_t* out_mask_dev);
/* out row i = row pos_dev[i] of (row_id, row_w, row_t [m, count], row_mask [m]). */
int euler_gpu_expand_rows(void* stream, const int32_t* pos_dev, int64_t n,
                          int32_t count, const uint64_t* row_id_dev,
                          const float* row_w_dev, const int32_t* row_t_dev,
                          const uint8_t* row_mask_dev, uint64_t* out_id_dev,
                          float* out_w_dev, int32_t* out_t_dev, uint8_t* out_mask_dev);
int euler_gpu_merge_rows(void* stream, const void* in_dev,
                         const int32_t* merge_idx_dev, int64_t n_rows,
                         int64_t row_bytes, void* out_dev);
/* ---- multi-GPU hop for C / C++ hosts ------------------------------------------
 * What ID_SPLIT -> REMOTE -> IDX_MERGE / DATA_MERGE does over gRPC in the
 * reference (core/kernels/id_split_op.cc:46-99, remote_op.cc:60-142,
 * idx_merge_op.cc:32-78, data_merge_op.cc:44-67), as ONE call per rank: every
 * rank (one process per GPU, holding the shard owner(id) = (id % partitions) %
 * world == rank) calls it with its own batch, and the calls exchange ids and wire
 * rows through `tr`.  The result equals the unsharded euler_gpu_sample_neighbor /
 * euler_gpu_sample_fanout (TF layout) bit for bit.  All ranks must make the same
 * sequence of calls (also a rank whose batch is empty: n = 0).
 *
 * The transport is two callbacks over one opaque pointer:
 *   alltoall_counts  host-side all-to-all of one int64 per peer (send[p] = rows this
 *                    rank will send to p -> recv[p] = rows it gets from p);
 *   alltoallv        all-to-all(v) of DEVICE rows of row_bytes each, enqueued on
 *                    `stream` (send_rows / recv_rows per peer, buffers packed in
 *                    peer order).
 * euler_gpu_transport_rccl fills it for an ncclComm_t (ncclSend / ncclRecv groups
 * over xGMI; the RCCL the host process already loaded is resolved at run time, this
 * library does not link it); `counts` = an euler_shm mailbox the ranks opened, or
 * NULL = the counts travel through the communicator (a device sync per hop).
 *
 * Memory: the per-call scratch of the euler_gpu_sharded_* entries (bucketed ids, wire rows, a
 * walk's levels, the rows a node2vec step fetches) comes from blocks the library keeps per
 * (device, stream) once it has allocated them - at most 32 GB per process, beyond that a
 * released block goes back to the driver.  (Stream-ordered allocations of gigabyte blocks of
 * ever different sizes stalled for seconds once in a few hundred calls.) */
typedef struct euler_gpu_transport {
  int32_t rank, world;
  void* user;
  int (*alltoall_counts)(void* user, const int64_t* send, int64_t* recv);
  int (*alltoallv)(void* user, const void* send_dev, const int64_t* send_rows,
                   void* recv_dev, const int64_t* recv_rows, int64_t row_bytes,
                   void* stream);
} euler_gpu_transport;
int euler_gpu_transport_rccl(void* nccl_comm, int32_t rank, int32_t world,
                             euler_shm* counts, euler_gpu_transport* out);
void euler_gpu_transport_rccl_release(euler_gpu_transport* t);
/* One hop; arguments as euler_gpu_sample_neighbor (TF layout), out_mask_dev [n]
 * optional. */
int euler_gpu_sharded_sample_neighbor(const euler_gpu_graph* shard,
                                      const euler_gpu_transport* tr, void* stream,
                                      uint64_t seed, uint32_t call_id,
                                      const uint64_t* roots_dev, int64_t n,
                                      const uint8_t* root_mask_dev, int32_t root_group,
                                      const int32_t* edge_types_host, int32_t k,
                                      int32_t count, int64_t default_node,
                                      int32_t partitions, uint64_t* out_id_dev,
                                      float* out_w_dev, int32_t* out_t_dev,
                                      uint8_t* out_mask_dev);
/* The fanout; arguments as euler_gpu_sample_fanout (workspace_dev:
 * euler_gpu_sample_fanout_workspace bytes). */
int euler_gpu_sharded_sample_fanout(const euler_gpu_graph* shard,
                                    const euler_gpu_transport* tr, void* stream,
                                    uint64_t seed, uint32_t call_id,
                                    const uint64_t* roots_dev, int64_t n,
                                    const int32_t* edge_types_host, int32_t k,
                                    const int32_t* counts_host, int32_t layers,
                                    int64_t default_node, int32_t partitions,
                                    uint64_t* const* out_id_dev, float* const* out_w_dev,
                                    int32_t* const* out_t_dev, void* workspace_dev);

/* TF RandomWalk with p = q = 1 over the sharded graph
 * (tf_euler/kernels/random_walk_op.cc:207-247: one sampleNB(edge_types, 1) query per step,
 * each ID_UNIQUE -> ID_SPLIT -> REMOTE -> MERGE -> GATHER): starts_dev [n] int64 ->
 * out_dev [n, walk_len + 1] int64 (missing neighbour -> default_node), bit-identical to
 * euler_gpu_random_walk on the unsharded graph.  The walk runs over LEVELS of distinct nodes
 * (walkers that meet stay together: the draw is keyed by the node), one front end + id
 * exchange + owners' draw + answer exchange per step, the walkers' paths written once at
 * the end.  The whole walk is ENQUEUED (round 6): a level and its per-owner buckets live in
 * slabs of a fixed stride (the largest walker count of a rank + 1, rounded up to 2 048 words:
 * a header word = the entries behind it, the entries, padding), their sizes stay on the device,
 * every message has the slab's size - so the host exchanges ONE number per peer per CALL (the
 * ranks' walker counts, through tr->alltoall_counts) and then only enqueues: per step the
 * front end (mark + one kernel that ranks and places; from step 16 on the level is sent as it
 * is), tr->alltoallv of world equal messages, the owners' draw over the slabs received, the
 * answers back into the next level.  The wire carries world x stride words per exchange
 * instead of the level's distinct nodes.  euler_gpu_set_tuning(63, 0) (euler_gpu_measure.h)
 * keeps the polled form: exchanges sized from the bucket sizes, one host wait per step and
 * cohort.  cohorts (1..16, the same on every rank): the walkers are split into that many
 * independent walks whose steps alternate on `stream` (the polled form hides its waits with
 * them; the enqueued form has none to hide).  dense_owner_dev / dense_limit: the id-indexed
 * table of euler_gpu_dedup_split (NULL / 0 = hashing).  stats_host (optional, int64[4]):
 * host waits, level entries summed over the steps, ids sent to other ranks, cohorts - asking
 * for them makes the enqueued call wait for the stream once at its end (the sizes never
 * reached the host otherwise).  All ranks must call it together (also with n = 0). */
int euler_gpu_sharded_random_walk(const euler_gpu_graph* shard, const euler_gpu_transport* tr,
                                  void* stream, uint64_t seed, uint32_t call_id,
                                  const int64_t* starts_dev, int64_t n,
                                  const int32_t* edge_types_host, int32_t k, int32_t walk_len,
                                  int64_t default_node, int32_t partitions, int32_t cohorts,
                                  uint32_t* dense_owner_dev, int64_t dense_limit,
                                  int64_t* out_dev, int64_t* stats_host);

/* TF RandomWalk with p or q != 1 (node2vec) over the sharded graph, as the reference's client
 * runs it (tf_euler/kernels/random_walk_op.cc:83-168: per step one `v(nodes).outV(edge_types)`
 * query for the walkers' current nodes - ID_UNIQUE -> ID_SPLIT -> REMOTE -> MERGE, one row per
 * distinct node - the previous step's rows kept as the parents', BuildWeights and the draw on
 * the client): starts_dev [n] int64 -> out_dev [n, walk_len + 1] int64, bit-identical to
 * euler_gpu_random_walk(p, q) on the unsharded graph (the draw of step s is keyed by the
 * walker's index and call_id + s).  Per step and rank: front end (distinct nodes by owner) ->
 * ids to the owners -> their full rows (euler_gpu_get_full_neighbor) -> row lengths, ids and
 * weights back (all-to-all(v)s sized from the lengths) -> euler_gpu_node2vec_step on the
 * fetched rows.  The host waits three times per step (bucket sizes, the owners' row offsets,
 * the value counts per peer).  stats_host (optional, int64[4]): host waits, rows asked, row
 * entries received (summed over the steps), ids sent to other ranks.  All ranks call it
 * together (also with n = 0). */
int euler_gpu_sharded_node2vec_walk(const euler_gpu_graph* shard, const euler_gpu_transport* tr,
                                    void* stream, uint64_t seed, uint32_t call_id,
                                    const int64_t* starts_dev, int64_t n,
                                    const int32_t* edge_types_host, int32_t k, int32_t walk_len,
                                    float p, float q, int64_t default_node, int32_t partitions,
                                    uint32_t* dense_owner_dev, int64_t dense_limit,
                                    int64_t* out_dev, int64_t* stats_host);

/* SAMPLE_NODE_SPLIT (core/kernels/sample_node_split_op.cc:57-85), host only:
 * shard_weight_host[shards+1] (last = total) -> split_cnt_host[shards]. */
int euler_gpu_sample_node_split(uint64_t seed, uint32_t call_id, int32_t count,
                                const float* shard_weight_host, int32_t shards,
                                int32_t* split_cnt_host);

/* ---- index memory ------------------------------------------------------------
 * The weight-bucket index (csrc/wb_index.h) is built on the first sampling / walk / block call
 * of a graph with non-uniform, non-decreasing running sums: 128 bytes per 4 edges of rows
 * with more than 10 edges + 8 + 8 T bytes per row (about 40 GB for 100M nodes / 1B edges; a graph
 * it serves does not get the 13 bytes per edge of EdgeBlocks the pivot-level search needs - those
 * are built, also on first use, only for graphs without a serving index).  It is an
 * optimisation that is DECLINED - the pivot-level search then serves, same results - when
 * it would take more than `max_free_fraction` of the HBM that is free at that moment
 * (default 0.5) or more than `max_bytes` (default: no absolute cap; 0 = never build it),
 * or when anything fails while it is built.  Process-wide; affects graphs whose index has
 * not been built yet.  Negative arguments leave a value unchanged.  The environment sets
 * the defaults: EULER_GPU_WB_INDEX=0, EULER_GPU_WB_INDEX_MAX_GB, EULER_GPU_WB_INDEX_MAX_FRACTION. */
int euler_gpu_set_index_budget(int64_t max_bytes, double max_free_fraction);

/* The 2-hop fanout of single listed types in the (unique rows, index) form - the GQL result
 * before DATA_GATHER (core/kernels/data_gather_op.cc:33-80; ID_UNIQUE: id_unique_op.cc:35-64):
 * hop 1 dense ([n, counts[0]] as euler_gpu_sample_fanout), hop 2 as DISTINCT rows:
 * rows_{id,w,t}_dev are [n * counts[0], counts[1]] row slots, row_index_dev [n * counts[0]]
 * (uint32) names the row of every hop-1 sample - the dense hop-2 tensors are
 * rows[row_index].  Roots are processed in groups of 4 (tuning key 28); a group fills the
 * first D of its 4 * counts[0] row slots, D = the distinct children it drew, the rest stay
 * untouched.  Plain graphs only (one edge-type group per node, identity id map, no neighbour
 * id 0, counts[1] even): EULER_GPU_EINVAL otherwise.  workspace: as euler_gpu_sample_fanout. */
int euler_gpu_sample_fanout_unique(const euler_gpu_graph* g, void* stream, uint64_t seed,
                                   uint32_t call_id, const uint64_t* roots_dev, int64_t n,
                                   const int32_t* edge_types_host, const int32_t* counts_host,
                                   int64_t default_node, uint64_t* out_id1_dev, float* out_w1_dev,
                                   int32_t* out_t1_dev, uint32_t* row_index_dev,
                                   uint64_t* rows_id_dev, float* rows_w_dev, int32_t* rows_t_dev,
                                   void* workspace_dev);
/* TF SampleFanoutWithFeature (tf_euler/kernels/sample_fanout_with_feature_op.cc:135-233;
 * op tf_euler/ops/neighbor_ops.cc:282-321): euler_gpu_sample_fanout plus the dense
 * features of every layer's nodes (layer 0 = the roots), enqueued back to back on `stream`
 * - no host round trip.  dense_out_dev[(layer) * n_dense + j] is a [m_layer, dims[j]]
 * float32 buffer (m_0 = n, m_{l+1} = m_l * counts[l]); zero rows for default_node,
 * unknown nodes and missing slots.  (The sparse features of the op need their sizes on
 * the host: euler_gpu_get_sparse_feature per layer, euler_amd/euler_ops/neighbor_ops.py.) */
int euler_gpu_sample_fanout_with_feature(const euler_gpu_graph* g, void* stream, uint64_t seed,
                                         uint32_t call_id, const uint64_t* roots_dev, int64_t n,
                                         const int32_t* edge_types_host, int32_t k,
                                         const int32_t* counts_host, int32_t layers,
                                         int64_t default_node, uint64_t* const* out_id_dev,
                                         float* const* out_w_dev, int32_t* const* out_t_dev,
                                         void* workspace_dev, const int32_t* dense_fids_host,
                                         const int32_t* dense_dims_host, int32_t n_dense,
                                         float* const* dense_out_dev);

/* Tuning keys, phase timers and byte counters (A/B measurements, bench.py's roofline leg) are
 * not part of the product surface: include/euler_gpu_measure.h. */

#ifdef __cplusplus
}
#endif
#endif  /* EULER_GPU_H_ */
