// Arguments and device-side gates shared by the sample_neighbor kernels
// (sample_kernels.hip: the single-type pivot kernels and the duplicate-root path;
// k1_variants.hip: the reference loop and the type-draw kernel) and the tuning
// switches of euler_gpu_set_tuning.
#ifndef EULER_AMD_CSRC_K1_ARGS_H_
#define EULER_AMD_CSRC_K1_ARGS_H_

#include <hip/hip_runtime.h>

#include "device_fns.h"

namespace euler_gpu {

// ------------------------------------------------------------------------
// K1  sample_neighbor
//
// One lane per SAMPLE (root r, slot j): 64 consecutive lanes cover
// consecutive slots, so the id/weight/type stores are fully coalesced and the
// `count` lanes of one root issue identical addresses for the root's metadata
// and the first probes of its binary search (served as one request per wave,
// then from L1/L2).  Degree skew costs nothing at the scheduling level: a hub
// row only deepens that lane's search (<= ceil(log2 deg) probes into an
// L2-resident, hot prefix array).  The RNG is addressed by (node id, j), never
// by position, so duplicate roots produce identical rows - the result of the
// reference's ID_UNIQUE -> sample -> GATHER rewrite (parser/compiler.cc:76-90)
// without running it.
// ------------------------------------------------------------------------
struct SampleNbArgs {
  GraphView g;
  uint64_t seed;
  const uint64_t* roots;
  const uint8_t* root_mask;
  uint64_t* out_id;
  float* out_w;
  int32_t* out_t;
  uint8_t* out_row_mask;
  int64_t n;
  int64_t default_node;
  uint32_t call_id;
  int32_t root_group;
  int32_t k;
  int32_t count;
  int32_t layout;
  int32_t dd_role;              // 0 no gate, 1 pass over the given roots, 2 pass
                                // over the unique roots (see DedupGate)
  const uint32_t* dd_counter;   // [0] = number of unique roots (device)
  int64_t dd_n_in;              // roots of the call
  uint32_t* mark_owner;         // not null: the outputs are the next hop's roots -
                                // enter them into its owner table (MarkNextHop)
  int32_t* packed;              // not null (pivot kernels): write wire rows of
                                // 4 * count + 2 words instead of out_id / out_w /
                                // out_t / out_row_mask (see PackRowsKernel)
  int32_t packed_tcol;          // ... with (1) or without (0: single-type call) the types
  int32_t cold_roots;           // hint: the roots are distinct (one sample per lane)
  int32_t et[kMaxListedTypes];
};

// ------------------------------------------------------------------------
// Duplicate roots.  Rows are a pure function of (seed, call_id, node id), so
// sampling a node once and copying its row to every position that asked for it
// is exactly the reference's ID_UNIQUE -> sample -> GATHER rewrite
// (parser/compiler.cc:76-90, core/kernels/id_unique_op.cc, data_gather_op.cc).
// The second hop of a fanout is where it pays: on the metric workload 3.28 M
// hop-2 roots are 286 K distinct nodes (8.7 %).  Nothing returns to the host:
// the insert kernel counts the unique roots on device and every later kernel
// of the call reads that count and either runs or exits:
//   unique * 4 <= roots * 3   -> sample the unique roots, then expand;
//   otherwise                 -> sample the given roots directly.
// ------------------------------------------------------------------------
__device__ __forceinline__ bool DedupActive(const uint32_t* counter, int64_t n_in) {
  return (int64_t)(*counter) * 4 <= n_in * 3;
}

// false = this launch has nothing to do; *n = number of roots it processes
__device__ __forceinline__ bool DedupGate(const SampleNbArgs& a, int64_t* n) {
  *n = a.n;
  if (a.dd_role == 0) return true;
  const bool dedup = DedupActive(a.dd_counter, a.dd_n_in);
  if (a.dd_role == 1) return !dedup;
  *n = (int64_t)(*a.dd_counter);
  return dedup;
}

// owner-table slot of a root key: its row, or n_rows for "no such node"
__device__ __forceinline__ uint32_t OwnerSlot(const GraphView& g, uint64_t key) {
  const int64_t row = FindRow(g, key);
  return row < 0 ? (uint32_t)g.n_rows : (uint32_t)row;
}

// DedupMarkKernel of the NEXT hop, done by the kernel that writes this hop's
// ids (fanout only, identity id map): output position s will be root s of the
// next hop, and a masked row stands for node id 0 there.
__device__ __forceinline__ void MarkNextHop(const GraphView& g, uint32_t* owner,
                                            uint64_t id, bool row_valid, int64_t s) {
  owner[OwnerSlot(g, row_valid ? id : 0)] = (uint32_t)s;
}

// int32 words of one wire row (multi-GPU result exchange, see PackRowsKernel):
// ids (2 each) | weights | [types] | mask | pad to an even count, so that every
// row starts 8-byte aligned: the ids are read and written as 8-byte words.
__host__ __device__ __forceinline__ int32_t PackedWords(int32_t count, int32_t tcol) {
  return ((3 + tcol) * count + 2 + 1) & ~1;
}

// ---- tuning switches (euler_gpu_set_tuning; defined in sample_kernels.hip) ----
// thread_local: a knob set by one host thread changes the launches THAT thread
// enqueues and nobody else's (the reference's client pool runs 8 query threads
// side by side; tests toggle variants in-process)
constexpr int64_t kK1GridCap = 32768;
extern thread_local int g_k1_variant, g_k1_ablate, g_k1_grid_cap, g_k1_pair, g_k1_dedup,
    g_k1_fuse_mark, g_k1_dual, g_expand_steps, g_expand_const_type, g_expand_grid_cap,
    g_n2v_wave, g_dedup_block_numbering, g_dedup_resolve_in_expand, g_fanout_fused,
    g_full_nb_balanced, g_n2v_big, g_k1_typed_pivot, g_fl_wb, g_k1_sets_lds;

// k1_variants.hip: launches the kernel variant `g_k1_variant` selects for calls
// the pivot kernels do not serve (grid = workgroups for one sample per lane)
int LaunchK1Variant(const euler_gpu_graph* g, hipStream_t stream, const SampleNbArgs& a,
                    int grid);

// k1_variants.hip: several edge-type sets over the same roots in one launch; false = not this
// way (the caller issues the separate calls)
bool LaunchSampleNeighborSets(const euler_gpu_graph* g, hipStream_t stream, uint64_t seed,
                              uint32_t call_id, const uint64_t* roots, int64_t n,
                              const int32_t* edge_types, const int32_t* set_k, int32_t n_sets,
                              int32_t count, int64_t default_node, uint64_t* out_id, float* out_w,
                              int32_t* out_t, int* rc_out, int32_t* packed = nullptr);

}  // namespace euler_gpu

#endif  // EULER_AMD_CSRC_K1_ARGS_H_
