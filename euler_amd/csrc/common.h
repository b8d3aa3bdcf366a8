// Internal definitions shared by the HIP translation units of libeuler_gpu.so.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/euler_gpu.h"
#include "../../include/euler_gpu_measure.h"
#include "philox.h"

namespace euler_gpu {

// ----------------------------------------------------------------- errors
void SetError(const std::string& msg);
int Fail(int code, const std::string& msg);

#define EG_HIP(expr)                                                        \
  do {                                                                      \
    hipError_t _e = (expr);                                                 \
    if (_e != hipSuccess)                                                   \
      return ::euler_gpu::Fail(EULER_GPU_EHIP,                              \
                               std::string(#expr) + ": " +                  \
                                   hipGetErrorString(_e));                  \
  } while (0)

// ------------------------------------------------------------ HBM layout
//
// The reference keeps one heap object per node (unordered_map<id, Node*> ->
// three std::vectors, node.h:49-57; 424 B + allocations per node).  Here the
// whole graph is four flat arrays, sized for one 288 GB HBM stack:
//
//   row_meta [n_rows] records of (8 + 8*T) bytes:
//        int64  row_ptr              offset of the row in nbr / prefix_w
//        int32  type_end[T]          neighbor_groups_idx (cumulative, row-relative)
//        float  type_prefix[T]       edge_group_collection running sums
//     one record = everything Node::SampleNeighbor needs before the search,
//     fetched with a single 16-byte load when T == 1 (one 64 B sector per root
//     instead of three dependent pointer chases).
//   prefix_w [E] float  neighbors_weight: the running f32 sums the reference
//     binary-searches (kept bit-identical; 4 B/edge keeps a row's search
//     inside 1-2 sectors for typical degrees)
//   nbr      [E] uint64 neighbor ids, read once per sample at the found slot
//   id map   identity/strided (id -> (id - base) / stride, no memory at all)
//            or an open-addressing table of 16-byte {key,row} slots.
constexpr int kPivotLevels = 10;   // levels 1..10: rows of up to 4*5^9*4 edges

struct GraphView {
  int64_t n_rows;
  int64_t n_edges;
  const uint8_t* row_meta;
  const uint64_t* nbr;
  const float* prefix_w;
  const uint64_t* row_id;       // [n_rows] or nullptr when ids are implicit
  int32_t T;                    // edge-type groups per node
  int32_t meta_stride;          // 8 + 8*T
  int32_t map_mode;             // 0 = strided identity, 1 = hash table
  int32_t has_zero_nbr;         // a neighbor id equals the sentinel 0 (Q1)
  int32_t monotone;             // every row's prefix_w is non-decreasing, >= 0
                                // and NaN-free (true for non-negative weights):
                                // licence for the single-load search of K1
  int32_t uniform_w;            // every edge weight is exactly 1.0f (running sums 1, 2, 3, ...
                                // per row, degrees < 2^24): the draw r = u * span + begin
                                // lands on edge floor(r) - no search, no sums read
  int32_t total_in_meta;        // T == 1 and every row's type_prefix[0] has the
                                // bits of its last running sum (checked on device
                                // at build): the segment limit comes with the row
                                // record, no separate load
  uint64_t id_base;             // identity: row = (id - id_base) / id_stride
  uint64_t id_stride;
  const uint64_t* hash_slots;   // [2 * (hash_mask + 1)] = {key, row} pairs
  uint64_t hash_mask;
  // dense float features (node.h float_features_idx_ / float_features_):
  // values of row r live at feat_val[feat_ptr[r] ...], slot f ends at
  // feat_idx[r * n_float + f] (row-relative).  When every row has the same
  // slot layout (the usual fixed-dimension table) feat_uniform = 1 and
  // feat_ptr[r] = r * feat_stride with the ends of row 0 valid for all rows.
  const int64_t* feat_ptr;
  const int32_t* feat_idx;
  const float* feat_val;
  int32_t n_float;
  int32_t feat_uniform;
  int64_t feat_stride;
  // pivot levels (K1 default search): level 1 entry q = prefix_w[4 q + 3],
  // level k+1 entry q = level k entry 5 q + 4 (indices clamped to the level's
  // end).  One unaligned 16-byte load reads the <= 4 candidate entries a
  // search step needs, so a sample costs ~log5(deg) loads instead of
  // ~log2(deg).  All levels live in one allocation, level k at pivots +
  // piv_off[k] (floats); +1.25 B per edge.
  const float* pivots;
  int64_t piv_off[kPivotLevels + 1];
  // block pivot levels (K1 variant 6): level 1 = skip1 (one entry per EdgeBlock),
  // level k+1 entry q = level k entry 5q+4; bpiv + bpiv_off[k] for k >= 2
  const float* bpiv;
  int64_t bpiv_off[kPivotLevels + 1];
  const struct EdgeBlock* blk;
  const float* skip1;           // [n_blk] running sum of the last edge of every EdgeBlock
  int64_t n_blk;
  // weight-bucket index (wb_index.h; built on first use for plain graphs: one edge-type
  // group, identity id map, monotone non-uniform weights, < 2^32 edges): a 16-byte record per
  // row and one 128-byte block per bucket of a row's running-sum range
  const struct WbRec* wrec;     // plain graphs only (the lean kernels' 16-byte record)
  const struct EdgeBlock* wb;
  int64_t n_wb;
  // every graph the index serves (monotone non-uniform weights, < 2^32 edges): per row
  // {uint32 wb_lo; uint32 row_lo; int32 type_end[T]; float lim[T]; float type_sum[T] (T > 1)} -
  // the row's first block, its first edge, the ends of its edge-type groups, the running sum
  // at the end of each (lim[T-1] = the row's total; limit_begin of group t = lim[t-1]) and the
  // row record's cumulative type sums (what a type draw selects by): everything a draw needs
  // before its block, in one record - a cold root costs one line, not the row record's and
  // this one's, and a segment's limits are not two more dependent loads
  const uint8_t* wbg;
  int32_t wbg_stride;           // 16 (T = 1), else 8 + 12 T
  // the same records for the kernels that only need the ROW part (first edge, group ends, their
  // running sums, type sums) - the typed hops of the one-kernel fanout: = wbg on graphs with
  // the index; graphs of uniform weights and several edge-type groups (every dataset the
  // reference ships: weight 1.0, 'train' / 'train_removed') get the records alone (wb_lo = 0)
  const uint8_t* trec;
  int32_t trec_stride;
  // hash id map + at most two edge-type groups: 64-byte hash slots that carry the row's record
  // {u64 key | u32 wb_lo, row_lo | i32 te[2], f32 lim[2] | f32 ts[2], i64 row | pad}, slot h =
  // slot h of hash_slots (same probe sequence); an empty slot has te[1] = -1.  The general
  // builds of the one-kernel fanout (fanout_local.h: FatFind) find a root's record with ONE
  // cold line instead of slot -> record.  nullptr: not built (T > 2, identity map, no room).
  const uint8_t* fat;
  int32_t wb_lean_ok;           // at most 2 buckets in a thousand overflow their block (counted at
                                // build): the lean kernels - whose second chance is the reference's
                                // bisection - draw through the index; otherwise they keep the pivot
                                // levels, and only the kernels that can fall back to the levels
                                // (k1_search.h: BlockPivotSample) use it
};

// A sampling view (SamplingView) can serve the block-pivot kernels: through the EdgeBlocks and
// their pivot levels, or through the weight-bucket index alone - a graph whose index the lean
// kernels may use (wb_lean_ok) does not get the 13 bytes per edge of EdgeBlocks at all, and the
// few draws its buckets do not bracket bisect the flat arrays (k1_search.h: BlockPivotSample).
inline bool HasBlockSearch(const GraphView& v) { return v.blk != nullptr || v.wbg != nullptr; }

// Edge block of the sampling index: 10 consecutive edges of the flat arrays
// (global edge indices [10 i, 10 i + 10)) with their running sums AND their
// neighbour ids inside ONE 128-byte line - the fetch unit of the L2 / fabric
// on gfx950.  The last search steps and the id read of a sample therefore cost
// one line instead of two (measured: random 128 B lines beyond the L2 are the
// scarce resource of K1, ~54 G lines/s for the whole chip).
constexpr int kEdgesPerBlock = 10;
struct alignas(128) EdgeBlock {
  float pw[kEdgesPerBlock];        // prefix_w of the 10 edges
  float prev_last;                 // prefix_w of the edge before the block (0 for block 0)
  uint32_t pad;
  uint64_t nbr[kEdgesPerBlock];    // their neighbour ids
};
static_assert(sizeof(EdgeBlock) == 128, "EdgeBlock must be one 128-byte line");

// A 256-thread workgroup is one wave per SIMD, and a SIMD admits
// min(8, 800 / (ceil(sgpr / 16) * 16 + 16)) of them (MI355X_MICROARCH.md,
// "Residency").  Kernels that take a GraphView by value have their arguments
// hoisted into SGPRs and, at the compiler's default budget, use 106: SIX waves.
// The sampling kernels wait on dependent loads most of the time, so they are
// declared __launch_bounds__(256, kWavesPerSimd): the compiler then keeps the
// SGPR count at <= 80 (the rest is spilled to VGPR lanes) and all eight waves
// fit (measured on the metric workload: hop-1 kernel -7 %, distinct-root pass
// -4 %).
constexpr int kWavesPerSimd = 8;

constexpr int kMaxListedTypes = 32;
constexpr int kMaxNodeTypes = 32;

// Alias-table entry of the global node sampler: one 32-byte record resolves a
// draw (column hit or alias) with a single memory access.
struct AliasEntry {
  uint64_t id_self;
  uint64_t id_alias;
  float prob;
  uint32_t pad0;
  uint64_t pad1;
};
static_assert(sizeof(AliasEntry) == 32, "AliasEntry must be 32 bytes");

struct NodeSamplerView {
  const AliasEntry* entries;          // all types, concatenated
  int32_t n_types;
  int32_t pad;
  int64_t type_off[kMaxNodeTypes + 1];
  float type_sum[kMaxNodeTypes];      // node_weight_sums_
  float sampler_sum[kMaxNodeTypes];   // FastWeightedCollection::sum_weight_
  float tc_prob[kMaxNodeTypes];       // node_type_collection_ alias table
  int32_t tc_alias[kMaxNodeTypes];
  float tc_sum;
};

EG_HD uint64_t Mix64(uint64_t z) {
  z ^= z >> 30; z *= 0xbf58476d1ce4e5b9ULL;
  z ^= z >> 27; z *= 0x94d049bb133111ebULL;
  z ^= z >> 31;
  return z;
}

}  // namespace euler_gpu

// The opaque handle of the C ABI.
struct euler_gpu_graph {
  int device = 0;
  euler_gpu::GraphView view{};
  euler_gpu::NodeSamplerView sampler{};
  bool has_sampler = false;
  int32_t n_node_types = 1;
  int64_t bytes = 0;
  uint64_t max_id = 0;          // largest node id of this graph (shard)
  int32_t partitions = 0;       // euler.meta's partitions_num for a loaded dataset, else 0
  std::vector<void*> allocations;     // every hipMalloc owned by the graph
  std::vector<float> node_weight_sums;
  const int32_t* node_type_dev = nullptr;   // [n_rows] node types; nullptr = all 0
  // sparse (uint64) features in the layout of the float ones: values of row r at
  // ufeat_val[ufeat_ptr[r] ...], slot f ends at ufeat_idx[r * n_u64 + f]
  const int64_t* ufeat_ptr = nullptr;
  const int32_t* ufeat_idx = nullptr;
  const uint64_t* ufeat_val = nullptr;
  int32_t n_u64 = 0;
  bool feat_slot_aligned = false;     // uniform feature table: every slot begins at a
                                      // multiple of 4 floats (16-byte lanes allowed)
  // scratch of the sampling launcher (dedup table, unique rows), one buffer
  // per stream: calls on one stream are ordered, calls on different streams
  // never share a buffer
  mutable std::mutex ws_mu;
  // held while one call ENQUEUES its kernels on a stream's scratch: two host
  // threads sharing a stream then cannot interleave their launches (the stream
  // runs one call's kernels to completion before the next call's touch the
  // scratch)
  mutable std::recursive_mutex launch_mu;   // a fanout holds it across its hops
  mutable std::map<void*, std::pair<void*, size_t>> ws;
  // which counter of the stream's pair the next duplicate-root call uses
  // (DedupNumberKernel clears the other one); reset when the scratch is reallocated
  mutable std::map<void*, int> ws_parity;
  // the stream of the previous sampling call: a caller that alternates streams keeps
  // several minibatches in flight, and the launcher then sizes its K1 grids so that the
  // kernels of two streams fit on the chip together (sample_kernels.hip: ConcurrentCall)
  mutable std::atomic<void*> last_stream{nullptr};
  mutable std::atomic<int> wb_tried{0};     // EnsureWbIndex ran (whatever it decided)
  std::vector<uint32_t> wb_overflow_rows;   // rows with a bucket that overflows its block (the first 4 096 found)
  mutable std::atomic<int> blk_ready{0};    // EnsureBlockedIndex built the EdgeBlocks (view.blk / skip1 / bpiv)
  // Block construction (dataflow_kernels.hip): first-occurrence unique of a hop's node list
  // through a table indexed by graph ROW, one per stream, kept across calls: 8 bytes per row
  // {~epoch, smallest position} + 4 bytes {rank}.  Every hop of every call takes a new epoch
  // and the table is cleared once - an atomicMin with a newer epoch beats whatever an older
  // one left.  Guarded by ws_mu.
  struct FlowTableDense { void* p = nullptr; size_t rows = 0; uint32_t next_epoch = 1; };
  mutable std::map<void*, FlowTableDense> flow_tables;
};

namespace euler_gpu {

// graph_build.hip
int BuildGraphFromHost(const euler_gpu_host_csr* csr, int device,
                       int32_t partitions, int32_t shard_index, int32_t shards,
                       euler_gpu_graph** out);
int BuildGraphSynthetic(const euler_gpu_synth_params* p, int device,
                        int32_t partitions, int32_t shard_index, int32_t shards,
                        euler_gpu_graph** out);
int EnsureBlockedIndex(const euler_gpu_graph* g);   // EdgeBlocks + block pivots, on first use
// weight-bucket index, on first use; leaves view.wb == nullptr (and returns OK) for graphs it
// does not serve
int EnsureWbIndex(const euler_gpu_graph* g);
// sample_kernels.hip: the view a sampling launcher hands to its kernels - the search indexes
// built on first use, the weight-bucket fields nulled when tuning key 45 = 0 (the kernels
// then walk the pivot levels)
int SamplingView(const euler_gpu_graph* g, GraphView* out);
// sample_kernels.hip: TF-layout SampleNeighbor over the first *n_dev roots of a list
// sized for `cap` (dataflow_kernels.hip)
int LaunchSampleNeighborCounted(const euler_gpu_graph* g, hipStream_t stream, uint64_t seed,
                                uint32_t call_id, const uint64_t* roots, int64_t cap,
                                const uint32_t* n_dev, const int32_t* edge_types, int32_t k,
                                int32_t count, int64_t default_node, uint64_t* out_id,
                                float* out_w, int32_t* out_t);
// walk_kernels.hip: the pieces of the sharded DeepWalk (csrc/sharded.cc) - the edge-type table
// of a walk on the device (stream-ordered allocation, the caller frees it with hipFreeAsync),
// the owners' draw of step `step` for the ids asked (0 = no neighbour), and the walkers' paths
// from the levels' (ids, next) arrays (HOST tables of device pointers)
int WalkEdgeTypes(hipStream_t st, const int32_t* edge_types_host, int32_t k, int32_t walk_len,
                  int32_t** et_dev);
int WalkOwnedStep(const euler_gpu_graph* g, hipStream_t st, uint64_t seed, uint32_t call_id,
                  const int32_t* et_dev, int32_t k, int32_t walk_len, int32_t step,
                  const uint64_t* ids_dev, int64_t n, uint64_t* out_dev);
int WalkPathsFromLevels(hipStream_t st, const int64_t* starts_dev, int64_t n, int32_t walk_len,
                        const uint64_t* const* level_ids_host, const int32_t* const* level_next_host,
                        int64_t default_node, int64_t* out_dev);
// The walk enqueued without host waits: levels and buckets in SLAB layout (a slab per owner:
// header word = the number of entries, then the entries, then padding), sizes on the device.
// mp_kernels.hip: the front end over such a level; walk_kernels.hip: the owners' draw over the
// slabs received (the header of each says how many of its words hold ids; lens_dev not null: that
// array instead - a lone rank draws on the slabs its own front end filled)
int FrontSlabs(hipStream_t st, const uint64_t* ids_dev, int64_t n_pos, const uint32_t* in_lens_dev,
               uint32_t in_stride, int32_t partitions, int32_t shards, uint32_t* dense_owner_dev,
               int64_t dense_limit, uint64_t* out_slabs_dev, uint32_t out_stride, uint32_t* out_lens_dev,
               bool write_headers, bool dedup, int32_t* pos_dev);
int WalkOwnedSlabs(const euler_gpu_graph* g, hipStream_t st, uint64_t seed, uint32_t call_id,
                   const int32_t* et_dev, int32_t k, int32_t walk_len, int32_t step,
                   const uint64_t* slabs_dev, const uint32_t* lens_dev, int32_t n_slabs, uint32_t stride,
                   uint64_t* out_dev);
int WalkPathsFromLevelsTail(hipStream_t st, const int64_t* starts_dev, int64_t n, int32_t walk_len,
                            const uint64_t* const* level_ids_host, const int32_t* const* level_next_host,
                            int64_t default_node, int64_t* out_dev, int32_t T, const uint32_t* lens_T_dev,
                            uint32_t stride, int32_t n_slabs, void* scratch_dev);
// ... and of the sharded node2vec walk: row lengths <-> FillNeighbor offsets, one column of
// the [n, walk_len + 1] result, the value offsets at the peers' row bounds
int N2vRowLens(hipStream_t st, const int32_t* idx_dev, int64_t m, int32_t* lens_dev);
int N2vIdxFromLens(hipStream_t st, const int32_t* lens_dev, int64_t m, int32_t* tmp_ends_dev, int32_t* idx_dev);
int N2vStoreColumn(hipStream_t st, const int64_t* src_dev, int64_t n, int64_t stride, int64_t col, int64_t* out_dev);
int N2vBoundOffsets(hipStream_t st, const int32_t* idx_dev, const int64_t* bounds_dev, int32_t w, int64_t* out_dev);
// dat_reader.cc
struct DatGraph {
  std::vector<uint64_t> row_id, nbr;
  std::vector<int64_t> row_ptr, feat_ptr;
  std::vector<int32_t> type_end, node_type, feat_idx;
  std::vector<float> prefix_w, type_prefix, node_weight, feat_val;
  std::vector<int64_t> ufeat_ptr;
  std::vector<int32_t> ufeat_idx;
  std::vector<uint64_t> ufeat_val;
  int32_t n_edge_types = 0, n_node_types = 0, partitions = 1, n_float = 0, n_u64 = 0;
  void Describe(euler_gpu_host_csr* c) const;
};
int LoadDatDirectory(const char* data_path, int32_t shard_index, int32_t shards,
                     DatGraph* out);
int VerifyEdgeFiles(const char* data_path, int32_t shard_index, int32_t shards,
                    const DatGraph& g, int64_t* edge_records, int64_t* not_in_rows,
                    int64_t* row_triples);

// Grid sizing for HBM-bound kernels: enough workgroups to fill 256 CUs x 8
// resident blocks, grid-stride beyond that.
inline int GridFor(int64_t work_items, int block) {
  int64_t blocks = (work_items + block - 1) / block;
  const int64_t cap = 256 * 16;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  return (int)blocks;
}

}  // namespace euler_gpu
