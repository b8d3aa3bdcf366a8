// Sampling kernels: SampleNeighbor / SampleFanout / SampleNode /
// GetFullNeighbor / RandomWalk for gfx950, plus their C-ABI entry points.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <cmath>
#include <vector>

#include "device_fns.h"

extern "C" int g_feature_vec4;   // mp_kernels.hip

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

// Finding the duplicates without atomics.  A hash table filled with
// compare-and-swap was measured at 0.6 ms for the 3.28 M hop-2 roots of the
// metric workload (device-scope atomics execute at the memory side on this
// multi-die part: ~0.45 ns per CAS, ~9 ns per same-address add).  Rows give a
// dense key instead: every position stores its own index into owner[row] with a
// plain 4-byte store (a benign race: one of the positions naming the row
// survives, and every referenced row is written by this call, so the table
// never needs clearing); reading it back tells who survived - the row's
// representative; an exclusive scan of the representative flags (evaluated
// inside the scan's loads) numbers the unique roots and counts them.  All unknown ids share the slot n_rows:
// their rows are the same default fill.
struct DedupArgs {
  GraphView g;
  const uint64_t* roots;
  const uint8_t* root_mask;
  int64_t n;
  int32_t root_group;
  int32_t pad;
  uint32_t* owner;           // [n_rows + 1] row -> a position that names it
  uint32_t* row_slot;        // [n] row of every position (n_rows = no such node)
  uint32_t* pos;             // [n + 1] exclusive scan of flag; pos[n] = unique count
  uint64_t* uniq;            // [<= n] unique roots (node ids), by first representative
  uint32_t* uidx_of;         // [n] index into uniq of every position
  uint32_t* counter;         // [0] unique count
};

__device__ __forceinline__ uint64_t DedupKey(const DedupArgs& a, int64_t i) {
  uint64_t key = a.roots[i];
  if (a.root_mask != nullptr && a.root_mask[i / a.root_group]) key = 0;
  return key;
}

__global__ __launch_bounds__(256) void DedupMarkKernel(const DedupArgs a) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n;
       i += stride) {
    const int64_t row = FindRow(a.g, DedupKey(a, i));
    const uint32_t slot = row < 0 ? (uint32_t)a.g.n_rows : (uint32_t)row;
    a.row_slot[i] = slot;
    a.owner[slot] = (uint32_t)i;
  }
}

// flag[i] = 1 when position i is its row's representative; evaluated inside
// the scan's loads (no flag array, no separate pass)
struct DedupFlagOp {
  const uint32_t* owner;
  const uint32_t* row_slot;
  int64_t n;
  __host__ __device__ __forceinline__ uint32_t operator()(const uint32_t& i) const {
    return ((int64_t)i < n && owner[row_slot[i]] == i) ? 1u : 0u;
  }
};

__global__ __launch_bounds__(256) void DedupIndexKernel(const DedupArgs a) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t first = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (first == 0) a.counter[0] = a.pos[a.n];
  for (int64_t i = first; i < a.n; i += stride) {
    const uint32_t rep = a.owner[a.row_slot[i]];
    a.uidx_of[i] = a.pos[rep];
    if (rep == (uint32_t)i) a.uniq[a.pos[i]] = DedupKey(a, i);
  }
}

// Premarked form (the previous hop of a fanout filled `owner`, identity id
// map): the slot is arithmetic on the key, so there is no row_slot array.
struct IdentityMap {
  uint64_t id_base, id_stride;
  int64_t n_rows;
  __host__ __device__ __forceinline__ uint32_t Slot(uint64_t id) const {
    if (id < id_base) return (uint32_t)n_rows;
    const uint64_t d = id - id_base;
    const uint64_t r = id_stride == 1 ? d : d / id_stride;
    return (r * id_stride == d && r < (uint64_t)n_rows) ? (uint32_t)r : (uint32_t)n_rows;
  }
};

struct DedupFlagPremarkedOp {
  const uint32_t* owner;
  const uint64_t* roots;
  const uint8_t* root_mask;
  IdentityMap map;
  int64_t n;
  int32_t root_group;
  __host__ __device__ __forceinline__ uint32_t operator()(const uint32_t& i) const {
    if ((int64_t)i >= n) return 0u;
    uint64_t key = roots[i];
    if (root_mask != nullptr && root_mask[i / root_group]) key = 0;
    return owner[map.Slot(key)] == i ? 1u : 0u;
  }
};

__global__ __launch_bounds__(256) void DedupIndexPremarkedKernel(const DedupArgs a,
                                                                 const IdentityMap map) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t first = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (first == 0) a.counter[0] = a.pos[a.n];
  for (int64_t i = first; i < a.n; i += stride) {
    const uint64_t key = DedupKey(a, i);
    const uint32_t rep = a.owner[map.Slot(key)];
    a.uidx_of[i] = a.pos[rep];
    if (rep == (uint32_t)i) a.uniq[a.pos[i]] = key;
  }
}

struct ExpandArgs {
  const uint32_t* counter;
  const uint32_t* uidx_of;
  const uint64_t* t_id;
  const float* t_w;
  const int32_t* t_t;
  const uint8_t* t_mask;
  uint64_t* out_id;
  float* out_w;
  int32_t* out_t;
  uint8_t* out_mask;
  int64_t n;
  int32_t count;
  uint32_t* mark_owner;      // see SampleNbArgs::mark_owner
  IdentityMap map;
  int32_t type0;             // CT kernels: type of every valid sample ...
  int32_t masked_type;       // ... and of the samples of a masked row
};

// out row i = sampled row of unique root uidx_of[i].  U adjacent samples per
// lane (U = 2: 16-byte id stores, needs an even count); V grid-stride steps of
// a lane are in flight together.  The kernel is a gather-copy: on the metric's
// hop 2 it reads 0.39 GB of rows (L2 / infinity cache; the 34 MB of distinct
// rows do not fit one XCD's L2) and writes 0.52 GB - 6.9 TB/s of fabric
// traffic at 0.134 ms, where writing the same three arrays alone takes 0.087 ms
// (tools/ubench_fill.hip).  Measured (tools/ab_expand.py): V = 2 is the best
// (V = 4 is slower: not a latency-bound kernel), and so is CT - every sample of
// a single-type call has the same type, so the type column is not gathered but
// rebuilt from the row mask (-5 %).
template <int U, int V, bool CT>
__global__ __launch_bounds__(256) void DedupExpandKernel(const ExpandArgs a,
                                                         const int64_t stride_rows,
                                                         const int32_t stride_slots) {
  typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  typedef int i32x2 __attribute__((ext_vector_type(2)));
  if (!DedupActive(a.counter, a.n)) return;
  const int64_t total = a.n * (int64_t)a.count;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x * U;
  int64_t s = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * U;
  if (s >= total) return;
  int64_t i = s / a.count;
  int32_t j = (int32_t)(s - i * a.count);
  const bool need_mask = CT || a.mark_owner != nullptr;
  while (s < total) {
    int64_t sv[V], iv[V];
    int32_t jv[V];
    bool live[V];
#pragma unroll
    for (int v = 0; v < V; ++v) {
      sv[v] = s; iv[v] = i; jv[v] = j;
      live[v] = s < total;
      s += stride;
      i += stride_rows;
      j += stride_slots;
      if (j >= a.count) { j -= a.count; ++i; }
    }
    // all index loads, then all row loads, then the stores: steps past the end
    // load element 0 (harmless) so that the loads stay unconditional
    int64_t uv[V];
#pragma unroll
    for (int v = 0; v < V; ++v) uv[v] = (int64_t)a.uidx_of[live[v] ? iv[v] : 0];
    uint64_t id[V][U];
    float w[V][U];
    int32_t t[V][U];
    uint8_t m[V];
#pragma unroll
    for (int v = 0; v < V; ++v) {
      const int64_t src = live[v] ? uv[v] * a.count + jv[v] : 0;   // even when count is
      if (U == 1) {
        id[v][0] = a.t_id[src];
        w[v][0] = a.t_w[src];
        if (!CT) t[v][0] = a.t_t[src];
      } else {
        const u64x2 i2 = *reinterpret_cast<const u64x2*>(a.t_id + src);
        const f32x2 w2 = *reinterpret_cast<const f32x2*>(a.t_w + src);
        id[v][0] = i2.x; id[v][U - 1] = i2.y;
        w[v][0] = w2.x; w[v][U - 1] = w2.y;
        if (!CT) {
          const i32x2 t2 = *reinterpret_cast<const i32x2*>(a.t_t + src);
          t[v][0] = t2.x; t[v][U - 1] = t2.y;
        }
      }
      m[v] = (need_mask || (jv[v] == 0 && a.out_mask != nullptr)) ? a.t_mask[uv[v]] : 0;
      if (CT) {
#pragma unroll
        for (int x = 0; x < U; ++x) t[v][x] = m[v] ? a.masked_type : a.type0;
      }
    }
#pragma unroll
    for (int v = 0; v < V; ++v) {
      if (!live[v]) continue;
      const int64_t d = sv[v];
      // the outputs are written once and not read by this call: non-temporal
      // stores keep the rows of the distinct roots (re-read ~11x) in the caches
      if (U == 1) {
        __builtin_nontemporal_store(id[v][0], a.out_id + d);
        __builtin_nontemporal_store(w[v][0], a.out_w + d);
        __builtin_nontemporal_store(t[v][0], a.out_t + d);
      } else {
        const u64x2 i2 = {id[v][0], id[v][U - 1]};
        const f32x2 w2 = {w[v][0], w[v][U - 1]};
        const i32x2 t2 = {t[v][0], t[v][U - 1]};
        __builtin_nontemporal_store(i2, reinterpret_cast<u64x2*>(a.out_id + d));
        __builtin_nontemporal_store(w2, reinterpret_cast<f32x2*>(a.out_w + d));
        __builtin_nontemporal_store(t2, reinterpret_cast<i32x2*>(a.out_t + d));
      }
      if (a.mark_owner != nullptr) {
#pragma unroll
        for (int x = 0; x < U; ++x)
          a.mark_owner[a.map.Slot(m[v] ? 0 : id[v][x])] = (uint32_t)(d + x);
      }
      if (jv[v] == 0 && a.out_mask != nullptr) a.out_mask[iv[v]] = m[v];
    }
  }
}

// Back end of a multi-GPU hop: position i takes row pos[i] of the packed
// answers (4 * count + 2 int32 words per row: ids, weights, types, mask, pad).
// U = 2 (even count, aligned outputs): a lane moves two adjacent samples -
// rows are 8-byte aligned, so the ids are two 8-byte loads and one 16-byte
// store; weights and types one 8-byte load and store each.
template <int U>
__global__ __launch_bounds__(256) void ExpandPackedKernel(
    const int32_t* __restrict__ pos, const int32_t* __restrict__ packed, int64_t n,
    int32_t count, uint64_t* __restrict__ out_id, float* __restrict__ out_w,
    int32_t* __restrict__ out_t, uint8_t* __restrict__ out_mask,
    const int64_t stride_rows, const int32_t stride_slots) {
  typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  typedef int i32x2 __attribute__((ext_vector_type(2)));
  const int32_t words = 4 * count + 2;
  const int64_t total = n * (int64_t)count;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x * U;
  int64_t s = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * U;
  if (s >= total) return;
  int64_t i = s / count;
  int32_t j = (int32_t)(s - i * count);
  for (; s < total; s += stride) {
    const int32_t* row = packed + (int64_t)pos[i] * words;
    if (U == 1) {
      const uint64_t id = *reinterpret_cast<const uint64_t*>(row + 2 * j);
      __builtin_nontemporal_store(id, out_id + s);
      __builtin_nontemporal_store(__int_as_float(row[2 * count + j]), out_w + s);
      __builtin_nontemporal_store(row[3 * count + j], out_t + s);
    } else {
      const uint64_t* idp = reinterpret_cast<const uint64_t*>(row + 2 * j);
      const u64x2 id2 = {idp[0], idp[1]};
      const f32x2 w2 = *reinterpret_cast<const f32x2*>(row + 2 * count + j);
      const i32x2 t2 = *reinterpret_cast<const i32x2*>(row + 3 * count + j);
      __builtin_nontemporal_store(id2, reinterpret_cast<u64x2*>(out_id + s));
      __builtin_nontemporal_store(w2, reinterpret_cast<f32x2*>(out_w + s));
      __builtin_nontemporal_store(t2, reinterpret_cast<i32x2*>(out_t + s));
    }
    if (j == 0 && out_mask != nullptr) out_mask[i] = (uint8_t)row[4 * count];
    i += stride_rows;
    j += stride_slots;
    if (j >= count) { j -= count; ++i; }
  }
}

__global__ __launch_bounds__(256) void SampleNeighborKernel(const SampleNbArgs a) {
  int64_t n_roots;
  if (!DedupGate(a, &n_roots)) return;
  const int64_t total = n_roots * (int64_t)a.count;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < total;
       s += stride) {
    const int64_t r = s / a.count;
    const int32_t j = (int32_t)(s - r * a.count);
    uint64_t node = a.roots[r];
    if (a.root_mask != nullptr && a.root_mask[r / a.root_group]) node = 0;
    RowSampler rs;
    InitRowSampler(rs, a.g, FindRow(a.g, node), a.et, a.k);
    uint64_t id = 0;
    float w = 0.f;
    int32_t t = 0;
    bool masked = !rs.valid;
    if (rs.valid) {
      SampleAt(rs, a.seed, a.call_id, node, j, &id, &w, &t);
      if (a.layout == EULER_GPU_LAYOUT_TF) {
        // tf_euler/kernels/sample_neighbor_op.cc:114-122: the row is kept only
        // if its FIRST id is not the sentinel.  Only graphs that contain the
        // id 0 as a neighbour can have a live row that starts with 0.
        if (j == 0) {
          masked = id == 0;
        } else if (a.g.has_zero_nbr) {
          uint64_t id0; float w0; int32_t t0;
          SampleAt(rs, a.seed, a.call_id, node, 0, &id0, &w0, &t0);
          masked = id0 == 0;
        }
      }
    }
    if (a.layout == EULER_GPU_LAYOUT_TF) {
      if (masked) { id = (uint64_t)a.default_node; w = 0.f; t = -1; }
    } else if (masked) {
      id = 0; w = 0.f; t = 0;   // core/kernels/sample_neighbor_op.cc:134-143
    }
    a.out_id[s] = id;
    a.out_w[s] = w;
    a.out_t[s] = t;
    if (j == 0 && a.out_row_mask != nullptr) a.out_row_mask[r] = masked ? 1 : 0;
  }
}

// ------------------------------------------------------------------------
// K1 fast path: single listed edge type (the GraphSAGE / DeepWalk case) on a
// graph whose prefix sums are monotone (GraphView::monotone).
//
// Same lane-per-sample mapping, but the per-sample instruction stream is cut
// to what the hardware needs:
//   * (root, slot) advance incrementally through the grid-stride loop - one
//     64-bit division per lane per launch instead of one per sample;
//   * 32-bit row-relative indices;
//   * the search is a single-load upper bound (first m with sw[m] > r).  With
//     non-decreasing sums the interval that holds r is unique, so this is the
//     index the reference's bisection returns (compact_weighted_collection.h:
//     37-50); when NO interval holds r (r rounded up to the segment's end, Q3)
//     the lane replays the reference's exact probe sequence instead.
// ------------------------------------------------------------------------
template <bool TF_LAYOUT, bool ZERO_CHECK>
__device__ __forceinline__ void FastSampleOne(const GraphView& g,
                                              const float* __restrict__ nw,
                                              const uint64_t* __restrict__ nbr,
                                              int32_t b, int32_t e, double u,
                                              uint64_t* out_id, float* out_w) {
  const float limit_begin = b == 0 ? 0.f : nw[b - 1];
  const float limit_end = nw[e];
  const double r = ScaleDraw(u, limit_begin, limit_end);
  int32_t lo = b, hi = e + 1;
  while (lo < hi) {
    const int32_t mid = (int32_t)(((uint32_t)lo + (uint32_t)hi) >> 1);
    if ((double)nw[mid] > r) hi = mid; else lo = mid + 1;
  }
  int32_t m = lo;
  float pre;
  if (m <= e) {
    pre = m == 0 ? 0.f : nw[m - 1];
  } else {
    // fall-through of RandomSelect: replay the reference probe sequence
    m = (int32_t)RandomSelect(nw, (uint64_t)b, (uint64_t)e, u);
    pre = m == 0 ? 0.f : nw[m - 1];
  }
  *out_id = nbr[m];
  *out_w = __fsub_rn(nw[m], pre);
}

template <bool TF_LAYOUT, bool ZERO_CHECK>
__global__ __launch_bounds__(256) void SampleNeighborFastKernel(
    const SampleNbArgs a, const int64_t stride_rows, const int32_t stride_slots) {
  int64_t n_roots;
  if (!DedupGate(a, &n_roots)) return;
  const int64_t total = n_roots * (int64_t)a.count;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= total) return;
  int64_t r = s / a.count;
  int32_t j = (int32_t)(s - r * a.count);
  const int32_t t = a.et[0];
  const int32_t T = a.g.T;
  for (; s < total; s += stride) {
    uint64_t node = a.roots[r];
    if (a.root_mask != nullptr && a.root_mask[r / a.root_group]) node = 0;
    const int64_t row = FindRow(a.g, node);
    uint64_t id = 0;
    float w = 0.f;
    bool valid = false;
    if (row >= 0 && t >= 0 && t < T) {
      const uint8_t* rec = a.g.row_meta + row * (int64_t)a.g.meta_stride;
      int64_t row_ptr;
      int32_t b, e;
      if (T == 1) {
        // {row_ptr, type_end[0], type_prefix[0]} in one 16-byte load
        const uint4 q = *reinterpret_cast<const uint4*>(rec);
        row_ptr = (int64_t)(((uint64_t)q.y << 32) | q.x);
        b = 0;
        e = (int32_t)q.z - 1;
      } else {
        row_ptr = *reinterpret_cast<const int64_t*>(rec);
        const int32_t* te = reinterpret_cast<const int32_t*>(rec + 8);
        b = t == 0 ? 0 : te[t - 1];
        e = te[t] - 1;
      }
      if (e >= b) {                                       // node.cc:133-135
        valid = true;
        const Philox4 blk = RngBlock(a.seed, a.call_id, kDomainNeighbor, node,
                                     ((uint32_t)j) >> 1);
        const double u = (j & 1) ? UnitFromWords(blk.w[2], blk.w[3])
                                 : UnitFromWords(blk.w[0], blk.w[1]);
        const float* nw = a.g.prefix_w + row_ptr;
        const uint64_t* nbr = a.g.nbr + row_ptr;
        FastSampleOne<TF_LAYOUT, ZERO_CHECK>(a.g, nw, nbr, b, e, u, &id, &w);
        if (TF_LAYOUT && ZERO_CHECK) {
          // Q1: the row is dropped when its FIRST sample is the sentinel id 0
          uint64_t id0 = id;
          if (j != 0) {
            const Philox4 b0 = RngBlock(a.seed, a.call_id, kDomainNeighbor, node, 0);
            float w0;
            FastSampleOne<TF_LAYOUT, ZERO_CHECK>(a.g, nw, nbr, b, e,
                                                 UnitFromWords(b0.w[0], b0.w[1]),
                                                 &id0, &w0);
          }
          valid = id0 != 0;
        }
      }
    }
    int32_t ot = t;
    if (!valid) {
      if (TF_LAYOUT) { id = (uint64_t)a.default_node; w = 0.f; ot = -1; }
      else { id = 0; w = 0.f; ot = 0; }
    }
    a.out_id[s] = id;
    a.out_w[s] = w;
    a.out_t[s] = ot;
    if (j == 0 && a.out_row_mask != nullptr) a.out_row_mask[r] = valid ? 0 : 1;
    // advance (root, slot) by the grid stride without dividing
    r += stride_rows;
    j += stride_slots;
    if (j >= a.count) { j -= a.count; ++r; }
  }
}

// ------------------------------------------------------------------------
// K1 ILP path: the same search as the fast path, U independent samples per
// lane advanced in lock step.  The kernel is bound by the latency of its chain
// of dependent loads (root -> row record -> limit -> ~log2(deg) probes -> id;
// SQ_WAIT_ANY = 87 % of wave time at full occupancy, 29 VGPRs), not by any
// one memory unit, so the lever is memory-level parallelism: U chains per
// lane keep U times as many requests in flight at the same occupancy.
// Values seen by the probes are carried along (nw[m], nw[m-1] are always among
// them), which removes the three trailing re-loads of the fast path.
// ------------------------------------------------------------------------
template <int U, bool TF_LAYOUT>
__global__ __launch_bounds__(256) void SampleNeighborIlpKernel(const SampleNbArgs a) {
  const int64_t total = a.n * (int64_t)a.count;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int32_t t = a.et[0];
  const int32_t T = a.g.T;
  const bool type_ok = t >= 0 && t < T;
  const bool small = total < (int64_t)0x7fffffff;
  for (int64_t s0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s0 < total;
       s0 += stride * U) {
    int64_t s[U], r[U];
    int32_t j[U], lo[U], hi[U], b[U], e[U];
    uint64_t node[U];
    const float* nw[U];
    const uint64_t* nbr[U];
    float vlo[U], vhi[U];
    double rr[U], u01[U];
    bool in[U], valid[U], replay[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      s[u] = s0 + (int64_t)u * stride;
      in[u] = s[u] < total;
      const int64_t sc = in[u] ? s[u] : 0;
      if (small) {
        const uint32_t q = (uint32_t)sc / (uint32_t)a.count;
        r[u] = q;
        j[u] = (int32_t)((uint32_t)sc - q * (uint32_t)a.count);
      } else {
        r[u] = sc / a.count;
        j[u] = (int32_t)(sc - r[u] * a.count);
      }
      node[u] = a.roots[r[u]];
    }
    if (a.root_mask != nullptr) {
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (a.root_mask[r[u] / a.root_group]) node[u] = 0;
    }
    int64_t row[U];
#pragma unroll
    for (int u = 0; u < U; ++u) row[u] = in[u] ? FindRow(a.g, node[u]) : -1;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      valid[u] = false;
      b[u] = 0; e[u] = -1;
      nw[u] = a.g.prefix_w; nbr[u] = a.g.nbr;
      if (row[u] >= 0 && type_ok) {
        const uint8_t* rec = a.g.row_meta + row[u] * (int64_t)a.g.meta_stride;
        int64_t row_ptr;
        if (T == 1) {
          const uint4 q = *reinterpret_cast<const uint4*>(rec);
          row_ptr = (int64_t)(((uint64_t)q.y << 32) | q.x);
          e[u] = (int32_t)q.z - 1;
        } else {
          row_ptr = *reinterpret_cast<const int64_t*>(rec);
          const int32_t* te = reinterpret_cast<const int32_t*>(rec + 8);
          b[u] = t == 0 ? 0 : te[t - 1];
          e[u] = te[t] - 1;
        }
        nw[u] = a.g.prefix_w + row_ptr;
        nbr[u] = a.g.nbr + row_ptr;
        valid[u] = e[u] >= b[u];                           // node.cc:133-135
      }
    }
    // limits of the searched segment (compact_weighted_collection.h:32-36)
#pragma unroll
    for (int u = 0; u < U; ++u) {
      vlo[u] = 0.f; vhi[u] = 0.f;
      if (valid[u]) {
        vhi[u] = nw[u][e[u]];
        if (b[u] != 0) vlo[u] = nw[u][b[u] - 1];
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const Philox4 blk = RngBlock(a.seed, a.call_id, kDomainNeighbor, node[u],
                                   ((uint32_t)j[u]) >> 1);
      u01[u] = (j[u] & 1) ? UnitFromWords(blk.w[2], blk.w[3])
                          : UnitFromWords(blk.w[0], blk.w[1]);
      rr[u] = ScaleDraw(u01[u], vlo[u], vhi[u]);
      // first m in [b, e] with nw[m] > r; nw[e] > r unless r was rounded up to
      // the segment's end (Q3) - those lanes replay the reference probes below
      replay[u] = valid[u] && !((double)vhi[u] > rr[u]);
      lo[u] = b[u];
      hi[u] = (valid[u] && !replay[u]) ? e[u] : b[u];
    }
    bool any = true;
    while (any) {
      any = false;
      float v[U];
      int32_t mid[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        mid[u] = (int32_t)(((uint32_t)lo[u] + (uint32_t)hi[u]) >> 1);
        if (lo[u] < hi[u]) v[u] = nw[u][mid[u]];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (lo[u] < hi[u]) {
          if ((double)v[u] > rr[u]) { hi[u] = mid[u]; vhi[u] = v[u]; }
          else { lo[u] = mid[u] + 1; vlo[u] = v[u]; }
          any |= lo[u] < hi[u];
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      uint64_t id = 0;
      float w = 0.f;
      int32_t ot = t;
      if (valid[u]) {
        int32_t m = lo[u];
        if (replay[u]) {
          m = (int32_t)RandomSelect(nw[u], (uint64_t)b[u], (uint64_t)e[u], u01[u]);
          vhi[u] = nw[u][m];
          vlo[u] = m == 0 ? 0.f : nw[u][m - 1];
        }
        id = nbr[u][m];
        w = __fsub_rn(vhi[u], vlo[u]);
      } else if (TF_LAYOUT) {
        id = (uint64_t)a.default_node; ot = -1;
      } else {
        ot = 0;
      }
      if (in[u]) {
        a.out_id[s[u]] = id;
        a.out_w[s[u]] = w;
        a.out_t[s[u]] = ot;
        if (j[u] == 0 && a.out_row_mask != nullptr)
          a.out_row_mask[r[u]] = valid[u] ? 0 : 1;
      }
    }
  }
}

namespace {
constexpr int64_t kK1GridCap = 32768;
int g_k1_dedup = 1;     // 0 = never, 1 = automatic for >= 16384 roots, 2 = always try
int g_n2v_wave = 1;     // node2vec: 1 = one wave per walker (LDS-staged lists), 0 = one lane
int g_k1_group = 0;     // block-pivot kernel: five adjacent samples per lane for odd counts
                        // that are a multiple of 5 - measured 8 % SLOWER on the metric's
                        // first hop (it is bound by the dependent-load chain per lane, not
                        // by instruction count), kept selectable
int g_k1_pair = 1;      // pivot kernel: two adjacent samples per lane when count is even
int g_k1_grid_cap = 0;  // measurement only: override the workgroup cap of K1
int g_k1_fuse_mark = 1; // fanout: a hop's kernels fill the next hop's owner table
int g_k1_dual = 1;      // duplicate-root call: both gated passes in one launch
int g_expand_steps = 2;        // DedupExpandKernel: grid-stride steps in flight per lane (1, 2, 4)
int g_expand_const_type = 1;   // ... rebuild the type column of single-type calls from the mask
int g_expand_grid_cap = 0;     // ... workgroup cap (0 = kK1GridCap)
int g_k1_ablate = 0;   // measurement only: skip parts of the blocked kernel
int g_k1_ilp = 4;      // samples per lane of the ILP kernel (1, 2, 4, 8)
int g_k1_variant = 6;   // 6 = block pivots, 5 = pivot levels, 4 = wave-staged (count >= 8) else blocked,
                        // 3 = blocked index,
                        // 2 = ILP, 1 = fast path, 0 = generic
}

// ------------------------------------------------------------------------
// K1 blocked path (default): the search runs on the sampling index of
// common.h (EdgeBlock + skip levels) instead of the flat arrays.
//
// Measured on the metric workload (profiles/r1_*): the flat-array kernels are
// bound by the vector-memory pipeline, time ~= L1 accesses x 0.5 clk + L2 line
// fills x 2.3 clk + lines from beyond the L2 x 11.5 clk per CU, and the last
// term is the largest: every sample ends in two cold 128-byte lines, one of
// prefix_w (the last ~5 probes) and one of nbr (8 useful bytes).  Here both
// live in the same EdgeBlock line, and the upper probes walk skip arrays that
// are 10x / 320x / 10240x smaller than prefix_w (the last two stay in the L2),
// with every level confined to one line.  More samples per lane do not help
// (tools/ab_k1.py: U = 2/4/8 chains per lane are slower) - the kernel is
// bound by line throughput, not by latency.
//
// Search contract (same as the fast path): m = first index of [b, e] with
// nw[m] > r; rows are non-decreasing (GraphView::monotone), so that is the
// index RandomSelect returns; r >= nw[e] (Q3) replays the reference loop.
// ------------------------------------------------------------------------
// first x in [lo, hi) with a[x] > r, else hi
__device__ __forceinline__ int32_t UpperBound32(const float* __restrict__ a,
                                                int32_t lo, int32_t hi, double r) {
  while (lo < hi) {
    const int32_t mid = (int32_t)(((uint32_t)lo + (uint32_t)hi) >> 1);
    if ((double)a[mid] > r) hi = mid; else lo = mid + 1;
  }
  return lo;
}

// Global edge index of the first edge in [lo, hi] whose running sum exceeds r
// (the caller guarantees sum(hi) > r).  *pw_m / *nbr_m receive that edge.
__device__ __forceinline__ int64_t BlockedSearch(const GraphView& g,
                                                 int64_t row_start, int64_t lo,
                                                 int64_t hi, double r,
                                                 float* pw_m, float* pw_prev,
                                                 uint64_t* nbr_m, int mode = 0) {
  const int32_t b_lo = (int32_t)(lo / kEdgesPerBlock);
  const int32_t b_hi = (int32_t)(hi / kEdgesPerBlock);
  // answer block = first B in [x_lo, x_hi) with skip1[B] > r, else x_hi
  int32_t x_lo = b_lo, x_hi = b_hi;
  if (mode == 10) {          // measurement only: no skip search, random block
    x_lo = b_lo + (int32_t)((uint32_t)(r * 7919.0) % (uint32_t)(b_hi - b_lo + 1));
    x_hi = x_lo;
  } else
  if (x_hi - x_lo > kSkipFanout) {
    int32_t y_lo = x_lo / kSkipFanout, y_hi = (x_hi - 1) / kSkipFanout;
    if (y_hi - y_lo > kSkipFanout) {
      const int32_t z_lo = y_lo / kSkipFanout, z_hi = (y_hi - 1) / kSkipFanout;
      const int32_t zm = UpperBound32(g.skip3, z_lo, z_hi, r);
      y_lo = max(y_lo, zm * kSkipFanout);
      y_hi = min(y_hi, zm * kSkipFanout + kSkipFanout);
    }
    const int32_t ym = UpperBound32(g.skip2, y_lo, y_hi, r);
    x_lo = max(x_lo, ym * kSkipFanout);
    x_hi = min(x_hi, ym * kSkipFanout + kSkipFanout);
  }
  int32_t bm;
  const bool short_row = x_hi - x_lo <= 2;
  if (short_row) {
    // short rows: probe the candidate blocks themselves (one of them is the
    // answer's line anyway) instead of touching a skip1 line
    bm = x_lo;
    while (bm < x_hi && !((double)g.blk[bm].pw[kEdgesPerBlock - 1] > r)) ++bm;
  } else {
    bm = UpperBound32(g.skip1, x_lo, x_hi, r);
  }
  if (mode == 11) {          // measurement only: no leaf
    *pw_m = (float)bm; *pw_prev = 0.f; *nbr_m = (uint64_t)bm;
    return bm;
  }
  const EdgeBlock* bk = g.blk + bm;
  const int64_t base = (int64_t)bm * kEdgesPerBlock;
  const int32_t i_lo = lo > base ? (int32_t)(lo - base) : 0;
  const int32_t i_hi = hi - base < kEdgesPerBlock - 1 ? (int32_t)(hi - base)
                                                      : kEdgesPerBlock - 1;
  const int32_t i = UpperBound32(bk->pw, i_lo, i_hi, r);
  *pw_m = bk->pw[i];
  *nbr_m = bk->nbr[i];
  // `mid ? nw[mid-1] : 0` is row-relative: the first edge of a row subtracts
  // 0, not the previous row's last sum.  The previous block's last sum is read
  // from whichever line the search has already touched.
  if (base + i == row_start) *pw_prev = 0.f;
  else if (i > 0) *pw_prev = bk->pw[i - 1];
  else *pw_prev = short_row ? g.blk[bm - 1].pw[kEdgesPerBlock - 1] : g.skip1[bm - 1];
  return base + i;
}

__device__ __forceinline__ float BlockedPw(const GraphView& g, int64_t m) {
  const int64_t bi = m / kEdgesPerBlock;
  return g.blk[bi].pw[(int32_t)(m - bi * kEdgesPerBlock)];
}

template <bool TF_LAYOUT>
__global__ __launch_bounds__(256) void SampleNeighborBlockedKernel(
    const SampleNbArgs a, const int64_t stride_rows, const int32_t stride_slots,
    const int32_t ablate) {
  const int64_t total = a.n * (int64_t)a.count;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= total) return;
  int64_t r = s / a.count;
  int32_t j = (int32_t)(s - r * a.count);
  const int32_t t = a.et[0];
  const int32_t T = a.g.T;
  for (; s < total; s += stride) {
    uint64_t node = a.roots[r];
    if (a.root_mask != nullptr && a.root_mask[r / a.root_group]) node = 0;
    const int64_t row = FindRow(a.g, node);
    uint64_t id = 0;
    float w = 0.f;
    bool valid = false;
    if (ablate >= 5 && ablate < 10) {
      id = node;
    } else if (row >= 0 && t >= 0 && t < T) {
      const uint8_t* rec = a.g.row_meta + row * (int64_t)a.g.meta_stride;
      int64_t row_ptr;
      int32_t b, e;
      if (ablate >= 4 && ablate < 10) {
        row_ptr = row * 10; b = 0; e = 9;
      } else if (T == 1) {
        const uint4 q = *reinterpret_cast<const uint4*>(rec);
        row_ptr = (int64_t)(((uint64_t)q.y << 32) | q.x);
        b = 0;
        e = (int32_t)q.z - 1;
      } else {
        row_ptr = *reinterpret_cast<const int64_t*>(rec);
        const int32_t* te = reinterpret_cast<const int32_t*>(rec + 8);
        b = t == 0 ? 0 : te[t - 1];
        e = te[t] - 1;
      }
      if (e >= b) {                                       // node.cc:133-135
        valid = true;
        double u;
        if (ablate >= 3 && ablate < 10) {
          u = (double)j * 0.03 + (double)(node & 1023) * 1e-4;
        } else {
          const Philox4 blk = RngBlock(a.seed, a.call_id, kDomainNeighbor, node,
                                       ((uint32_t)j) >> 1);
          u = (j & 1) ? UnitFromWords(blk.w[2], blk.w[3])
                      : UnitFromWords(blk.w[0], blk.w[1]);
        }
        float limit_begin = 0.f, limit_end = 1.f;
        if (ablate < 2 || ablate >= 10) {
          limit_begin = b == 0 ? 0.f : BlockedPw(a.g, row_ptr + b - 1);
          limit_end = BlockedPw(a.g, row_ptr + e);
        }
        const double rr = ScaleDraw(u, limit_begin, limit_end);
        if (ablate >= 1 && ablate < 10) {
          id = (uint64_t)row_ptr + (uint64_t)(int64_t)(rr * 1000.0);
          w = (float)rr;
        } else if ((double)limit_end > rr) {
          float pw_m, pw_prev;
          BlockedSearch(a.g, row_ptr, row_ptr + b, row_ptr + e, rr, &pw_m, &pw_prev,
                        &id, ablate);
          w = __fsub_rn(pw_m, pw_prev);
        } else {
          // Q3: r rounded up to the end of the segment - replay the reference
          const float* nw = a.g.prefix_w + row_ptr;
          const int32_t m = (int32_t)RandomSelect(nw, (uint64_t)b, (uint64_t)e, u);
          id = a.g.nbr[row_ptr + m];
          w = __fsub_rn(nw[m], m == 0 ? 0.f : nw[m - 1]);
        }
      }
    }
    int32_t ot = t;
    if (!valid) {
      if (TF_LAYOUT) { id = (uint64_t)a.default_node; w = 0.f; ot = -1; }
      else { id = 0; w = 0.f; ot = 0; }
    }
    a.out_id[s] = id;
    a.out_w[s] = w;
    a.out_t[s] = ot;
    if (j == 0 && a.out_row_mask != nullptr) a.out_row_mask[r] = valid ? 0 : 1;
    r += stride_rows;
    j += stride_slots;
    if (j >= a.count) { j -= a.count; ++r; }
  }
}

// ------------------------------------------------------------------------
// K1 wave-staged path (default for count >= 8).
//
// Ablation of the blocked kernel on the metric workload (tools/ab_k1.py, hop 2,
// 32.8 M samples): stores + root ids 0.14 ms, row record + limits + Philox
// 0.15 ms, skip-level probes 0.31 ms, leaf probes + id 0.29 ms.  Halving the L2
// and HBM traffic (flat -> blocked index) did not move the time; what it tracks
// is the number of lane-divergent vector-memory instructions (TCP busy ~100 %,
// 55-65 % of its cycles stalled on hits to pending lines): ~18 per sample.  A
// wave instruction whose 64 lanes touch 64 different lines costs the L1 about
// as much as eight instructions that fetch eight whole lines each.  So this
// kernel never lets a lane chase pointers on its own:
//
//   * one wave = 64 consecutive samples = the `count` samples of <= 10 roots;
//   * OWNER lanes (one per root) read the root id, the row record and the
//     segment limits once, pick the coarsest index level whose candidate range
//     fits 64 entries, and publish a descriptor in LDS;
//   * the wave copies each root's <= 64 candidate entries into LDS with one
//     coalesced load (the per-wavefront frontier buffer) and every sample
//     bisects them there;
//   * each remaining level is one 128-byte line per sample: 8 lanes fetch a
//     sample's skip node (3 lanes its leaf EdgeBlock sums) into LDS, again
//     searched in LDS; the only lane-divergent global load left is the 8-byte
//     neighbour id.
// Search results are those of BlockedSearch (same candidate ranges, same
// first-greater rule); Q3 lanes and rows beyond 64*32*32 blocks take the
// per-lane paths.
// ------------------------------------------------------------------------
constexpr int kStage = 64;           // staged candidate entries per root
constexpr int kMaxWaveRoots = 10;    // 64 / count + 2 for count >= 8
constexpr int kNodeStride = 36;      // floats per staged skip node (32 + pad)

struct RootDesc {
  int64_t row_ptr;      // first edge of the row (global index)
  int64_t lo, hi;       // searched segment [lo, hi], global edge indices
  uint64_t node;
  float limit_begin, limit_end;
  int32_t b_lo, b_hi;   // blocks of lo / hi
  int32_t level;        // staged level 1..3, 0 = per-lane search, -1 = invalid row
  int32_t x_lo;         // first staged entry
  int32_t cnt;          // staged entries: candidates [x_lo, x_lo + cnt), else x_lo + cnt
  int32_t pad;
};

struct alignas(16) WaveLds {
  float buf[32 * kNodeStride];               // 4.5 KB: skip nodes / leaf sums
  float top[kMaxWaveRoots][kStage];          // 2.5 KB: staged candidate entries
  RootDesc desc[kMaxWaveRoots];
};

__device__ __forceinline__ void WaveSync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// candidate range of a row at skip level `level` (see BlockedSearch)
__device__ __forceinline__ void LevelRange(int32_t level, int32_t b_lo, int32_t b_hi,
                                           int32_t* lo, int32_t* hi) {
  int32_t l = b_lo, h = b_hi;
  for (int32_t x = 1; x < level; ++x) {
    l = l / kSkipFanout;
    h = (h - 1) / kSkipFanout;
  }
  *lo = l; *hi = h;
}

template <bool TF_LAYOUT>
__global__ __launch_bounds__(256) void SampleNeighborWaveKernel(const SampleNbArgs a) {
  __shared__ WaveLds lds_all[4];
  WaveLds& L = lds_all[threadIdx.x >> 6];
  const int32_t lane = threadIdx.x & 63;
  const int64_t total = a.n * (int64_t)a.count;
  const int64_t n_chunks = (total + 63) >> 6;
  const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const int32_t t = a.et[0];
  const int32_t T = a.g.T;
  const bool small = total < (int64_t)0x7fffffff;
  for (int64_t chunk = wave0; chunk < n_chunks; chunk += n_waves) {
    const int64_t s_base = chunk << 6;
    const int64_t s = s_base + lane;
    const bool in = s < total;
    const int64_t s_last = s_base + 63 < total ? s_base + 63 : total - 1;
    int64_t r_first, r_last;
    if (small) {
      r_first = (uint32_t)s_base / (uint32_t)a.count;
      r_last = (uint32_t)s_last / (uint32_t)a.count;
    } else {
      r_first = s_base / a.count;
      r_last = s_last / a.count;
    }
    const int32_t nr = (int32_t)(r_last - r_first) + 1;
    // ---- owner phase: lane q < nr owns root r_first + q -------------------
    if (lane < nr) {
      RootDesc d;
      const int64_t r = r_first + lane;
      uint64_t node = a.roots[r];
      if (a.root_mask != nullptr && a.root_mask[r / a.root_group]) node = 0;
      d.node = node;
      d.level = -1;
      d.row_ptr = 0; d.lo = 0; d.hi = -1; d.limit_begin = 0.f; d.limit_end = 0.f;
      d.b_lo = 0; d.b_hi = 0; d.x_lo = 0; d.cnt = 0; d.pad = 0;
      const int64_t row = FindRow(a.g, node);
      if (row >= 0 && t >= 0 && t < T) {
        const uint8_t* rec = a.g.row_meta + row * (int64_t)a.g.meta_stride;
        int64_t row_ptr;
        int32_t b, e;
        if (T == 1) {
          const uint4 q = *reinterpret_cast<const uint4*>(rec);
          row_ptr = (int64_t)(((uint64_t)q.y << 32) | q.x);
          b = 0;
          e = (int32_t)q.z - 1;
        } else {
          row_ptr = *reinterpret_cast<const int64_t*>(rec);
          const int32_t* te = reinterpret_cast<const int32_t*>(rec + 8);
          b = t == 0 ? 0 : te[t - 1];
          e = te[t] - 1;
        }
        if (e >= b) {                                     // node.cc:133-135
          d.row_ptr = row_ptr;
          d.lo = row_ptr + b;
          d.hi = row_ptr + e;
          d.limit_begin = b == 0 ? 0.f : BlockedPw(a.g, d.lo - 1);
          d.limit_end = BlockedPw(a.g, d.hi);
          d.b_lo = (int32_t)(d.lo / kEdgesPerBlock);
          d.b_hi = (int32_t)(d.hi / kEdgesPerBlock);
          int32_t level = 1, xl = d.b_lo, xh = d.b_hi;
          while (level <= 3 && xh - xl > kStage) {
            xl = xl / kSkipFanout;
            xh = (xh - 1) / kSkipFanout;
            ++level;
          }
          d.level = level <= 3 ? level : 0;
          d.x_lo = xl;
          d.cnt = xh - xl;
        }
      }
      L.desc[lane] = d;
    }
    WaveSync();
    // ---- stage every root's candidate entries (one coalesced load each) -----
    for (int32_t q = 0; q < nr; ++q) {
      const int32_t level = L.desc[q].level;
      const int32_t cnt = L.desc[q].cnt;
      if (level >= 1 && lane < cnt) {
        const float* arr = level == 1 ? a.g.skip1 : level == 2 ? a.g.skip2 : a.g.skip3;
        L.top[q][lane] = arr[L.desc[q].x_lo + lane];
      }
    }
    WaveSync();
    // ---- sample phase ---------------------------------------------------------
    int64_t r = r_first;
    int32_t j = 0, q = 0;
    if (in) {
      if (small) {
        const uint32_t rq = (uint32_t)s / (uint32_t)a.count;
        r = rq;
        j = (int32_t)((uint32_t)s - rq * (uint32_t)a.count);
      } else {
        r = s / a.count;
        j = (int32_t)(s - r * a.count);
      }
      q = (int32_t)(r - r_first);
    }
    int32_t level = in ? L.desc[q].level : -1;
    const bool valid = level >= 0;
    const int32_t b_lo = L.desc[q].b_lo, b_hi = L.desc[q].b_hi;
    double u = 0.0, rr = 0.0;
    bool replay = false;
    int32_t x = 0;
    if (valid) {
      const uint64_t node = L.desc[q].node;
      const Philox4 blk = RngBlock(a.seed, a.call_id, kDomainNeighbor, node,
                                   ((uint32_t)j) >> 1);
      u = (j & 1) ? UnitFromWords(blk.w[2], blk.w[3])
                  : UnitFromWords(blk.w[0], blk.w[1]);
      const float limit_end = L.desc[q].limit_end;
      rr = ScaleDraw(u, L.desc[q].limit_begin, limit_end);
      replay = !((double)limit_end > rr);                 // Q3
      if (!replay && level >= 1)
        x = L.desc[q].x_lo + UpperBound32(L.top[q], 0, L.desc[q].cnt, rr);
    }
    // ---- remaining skip levels: one 128-byte node per sample, via LDS ---------
    for (int32_t lv = 3; lv >= 2; --lv) {
      const bool act = valid && !replay && level == lv;
      if (__ballot(act) == 0) continue;
      const float* arr = lv == 3 ? a.g.skip2 : a.g.skip1;   // level lv - 1
      const int32_t xs = act ? x : -1;
      int32_t c_lo = 0, c_hi = 0;
      if (act) {
        LevelRange(lv - 1, b_lo, b_hi, &c_lo, &c_hi);
        c_lo = max(c_lo, x * kSkipFanout);
        c_hi = min(c_hi, x * kSkipFanout + kSkipFanout);
      }
      for (int32_t half = 0; half < 2; ++half) {
        WaveSync();
#pragma unroll
        for (int32_t k = 0; k < 4; ++k) {
          const int32_t idx = 8 * k + (lane >> 3);          // sample of this half
          const int32_t src = __shfl(xs, 32 * half + idx);
          if (src >= 0) {
            const uint4 v = *reinterpret_cast<const uint4*>(
                arr + (int64_t)src * kSkipFanout + 4 * (lane & 7));
            *reinterpret_cast<uint4*>(&L.buf[idx * kNodeStride + 4 * (lane & 7)]) = v;
          }
        }
        WaveSync();
        if (act && (lane >> 5) == half) {
          const float* nodep = &L.buf[(lane & 31) * kNodeStride];
          const int32_t base = x * kSkipFanout;
          x = base + UpperBound32(nodep, c_lo - base, c_hi - base, rr);
          level = lv - 1;
        }
      }
    }
    // ---- leaf: the 10 running sums of block x, 3 lanes x 16 B per sample ------
    const bool leaf = valid && !replay && level == 1;
    uint64_t id = 0;
    float w = 0.f;
    {
      const int32_t xs = leaf ? x : -1;
      WaveSync();
#pragma unroll
      for (int32_t k = 0; k < 4; ++k) {
        const int32_t idx = 21 * k + lane / 3;
        const int32_t part = lane - (lane / 3) * 3;
        const int32_t src = __shfl(xs, idx < 64 ? idx : 63);
        if (lane < 63 && idx < 64 && src >= 0) {
          const uint4 v = *reinterpret_cast<const uint4*>(
              reinterpret_cast<const char*>(a.g.blk + src) + 16 * part);
          *reinterpret_cast<uint4*>(&L.buf[idx * 12 + 4 * part]) = v;
        }
      }
      WaveSync();
      if (leaf) {
        const float* pw = &L.buf[lane * 12];
        const int64_t base = (int64_t)x * kEdgesPerBlock;
        const int64_t lo = L.desc[q].lo, hi = L.desc[q].hi;
        const int32_t i_lo = lo > base ? (int32_t)(lo - base) : 0;
        const int32_t i_hi = hi - base < kEdgesPerBlock - 1 ? (int32_t)(hi - base)
                                                            : kEdgesPerBlock - 1;
        const int32_t i = UpperBound32(pw, i_lo, i_hi, rr);
        float prev;
        if (base + i == L.desc[q].row_ptr) prev = 0.f;
        else if (i > 0) prev = pw[i - 1];
        else prev = a.g.skip1[x - 1];
        w = __fsub_rn(pw[i], prev);
        id = a.g.blk[x].nbr[i];
      }
    }
    if (valid && !replay && level == 0) {
      // more than 64*32*32 blocks: per-lane search on the index
      float pw_m, pw_prev;
      BlockedSearch(a.g, L.desc[q].row_ptr, L.desc[q].lo, L.desc[q].hi, rr, &pw_m,
                    &pw_prev, &id);
      w = __fsub_rn(pw_m, pw_prev);
    }
    if (valid && replay) {
      // Q3: r rounded up to the end of the segment - replay the reference
      const int64_t row_ptr = L.desc[q].row_ptr;
      const float* nw = a.g.prefix_w + row_ptr;
      const int32_t m = (int32_t)RandomSelect(
          nw, (uint64_t)(L.desc[q].lo - row_ptr), (uint64_t)(L.desc[q].hi - row_ptr), u);
      id = a.g.nbr[row_ptr + m];
      w = __fsub_rn(nw[m], m == 0 ? 0.f : nw[m - 1]);
    }
    int32_t ot = t;
    if (!valid) {
      if (TF_LAYOUT) { id = (uint64_t)a.default_node; w = 0.f; ot = -1; }
      else { id = 0; w = 0.f; ot = 0; }
    }
    if (in) {
      a.out_id[s] = id;
      a.out_w[s] = w;
      a.out_t[s] = ot;
      if (j == 0 && a.out_row_mask != nullptr) a.out_row_mask[r] = valid ? 0 : 1;
    }
    WaveSync();   // descriptors / buffers are rewritten by the next chunk
  }
}

// ------------------------------------------------------------------------
// K1 pivot path (default).
//
// tools/k1_phases.py (s_memtime stamps around every phase of the flat kernel on
// the metric workload) shows that each vector-memory round trip of a wave -
// the coalesced root-id load as much as a divergent probe - takes the same
// ~3000 ticks: the CU's memory pipeline is a queue, and the kernel's time is
// (vector-memory instructions per wave) x (queue service time).  Fewer bytes
// (blocked index), more loads in flight (ILP) or whole-line staging in LDS do
// not change it; fewer memory INSTRUCTIONS do.  A binary search spends
// ceil(log2 deg) of them; this kernel spends ~log5(deg):
//
//   level 1 entry q = nw[4q+3]; level k+1 entry q = level k entry 5q+4.
//   The answer's possible positions at level k are [lo/D_k, hi/D_k] (D_1 = 4,
//   D_k = 4*5^(k-1)).  Start at the first level K where that range has <= 4
//   candidates, and walk down: each step is ONE unaligned 16-byte load of the
//   <= 4 candidate entries below the chosen entry ("found" entries bound their
//   last child, so a step never needs a fifth key).  The leaf window is
//   shifted by one so that it also holds nw[m-1]; nw[m] is a loaded key, the
//   bounding key carried down, or the segment's limit.  The result index is
//   the first m of [b, e] with nw[m] > r, i.e. RandomSelect's answer on
//   non-decreasing rows; Q3 lanes replay the reference loop.
// ------------------------------------------------------------------------
typedef float float4u __attribute__((ext_vector_type(4), aligned(4)));

__device__ __forceinline__ float Pick4(const float4u& v, int32_t i) {
  return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w;
}

struct Segment {
  int64_t row_ptr;      // first edge of the row
  int64_t lo, hi;       // searched edges [lo, hi] (global indices)
  float limit_begin, limit_end;
  int32_t b, e;         // the same segment, row-relative
};

// One draw u on a segment: the neighbour RandomSelect picks and its weight.
__device__ __forceinline__ void PivotSample(const GraphView& g, const Segment& sg,
                                            double u, uint64_t* id, float* w) {
  const float* __restrict__ A0 = g.prefix_w;
  const int64_t lo = sg.lo, hi = sg.hi;
  const double rr = ScaleDraw(u, sg.limit_begin, sg.limit_end);
  if (!((double)sg.limit_end > rr)) {
    // Q3: r rounded up to the end of the segment - replay the reference
    const float* nw = A0 + sg.row_ptr;
    const int32_t m = (int32_t)RandomSelect(nw, (uint64_t)sg.b, (uint64_t)sg.e, u);
    *id = g.nbr[sg.row_ptr + m];
    *w = __fsub_rn(nw[m], m == 0 ? 0.f : nw[m - 1]);
    return;
  }
  // candidate ranges of every level; K = first level with <= 4 of them
  uint32_t l[kPivotLevels + 1], h[kPivotLevels + 1];
  l[1] = (uint32_t)(lo >> 2);
  h[1] = (uint32_t)(hi >> 2);
#pragma unroll
  for (int k = 2; k <= kPivotLevels; ++k) { l[k] = l[k - 1] / 5u; h[k] = h[k - 1] / 5u; }
  int32_t K = 0;
  if (hi - lo > 3) {
    K = kPivotLevels + 1;
#pragma unroll
    for (int k = kPivotLevels; k >= 1; --k)
      if (h[k] - l[k] <= 4u) K = k;
  }
  if (K > kPivotLevels) {
    // rows beyond the pivot levels' reach: plain upper-bound search
    int64_t lo2 = lo, hi2 = hi;
    while (lo2 < hi2) {
      const int64_t mid = (lo2 + hi2) >> 1;
      if ((double)A0[mid] > rr) hi2 = mid; else lo2 = mid + 1;
    }
    *id = g.nbr[lo2];
    *w = __fsub_rn(A0[lo2], lo2 == sg.row_ptr ? 0.f : A0[lo2 - 1]);
    return;
  }
  uint32_t x = 0;        // chosen entry of the level above
  bool found = false;    // its key was compared (> r): it bounds its children
  float kv = 0.f;        // that key
#pragma unroll
  for (int k = kPivotLevels; k >= 1; --k) {
    if (k <= K) {
      uint32_t c_lo, c_hi;
      if (k == K) { c_lo = l[k]; c_hi = h[k]; }
      else {
        c_lo = max(l[k], 5u * x);
        c_hi = found ? 5u * x + 4u : h[k];
      }
      const int32_t cnt = (int32_t)(c_hi - c_lo);
      const float4u kw =
          *reinterpret_cast<const float4u*>(g.pivots + g.piv_off[k] + c_lo);
      int32_t pos = 0;
      pos += (0 < cnt && !((double)kw.x > rr)) ? 1 : 0;
      pos += (1 < cnt && !((double)kw.y > rr)) ? 1 : 0;
      pos += (2 < cnt && !((double)kw.z > rr)) ? 1 : 0;
      pos += (3 < cnt && !((double)kw.w > rr)) ? 1 : 0;
      x = c_lo + (uint32_t)pos;
      if (pos < cnt) { found = true; kv = Pick4(kw, pos); }
    }
  }
  // leaf: candidates among the flat elements below level-1 entry x
  int64_t c_lo = lo, c_hi = hi;
  if (K >= 1) {
    c_lo = max(lo, (int64_t)x * 4);
    c_hi = found ? (int64_t)x * 4 + 3 : hi;
  }
  const int32_t cnt = (int32_t)(c_hi - c_lo);       // <= 3
  int64_t ws = c_lo - 1;                            // window start
  if (ws > g.n_edges - 4) ws = g.n_edges - 4;
  if (ws < 0) ws = 0;
  const int32_t sh = (int32_t)(c_lo - ws);          // key i sits at sh + i
  const float4u wv = *reinterpret_cast<const float4u*>(A0 + ws);
  int32_t pos = 0;
  pos += (0 < cnt && !((double)Pick4(wv, sh) > rr)) ? 1 : 0;
  pos += (1 < cnt && !((double)Pick4(wv, sh + 1) > rr)) ? 1 : 0;
  pos += (2 < cnt && !((double)Pick4(wv, sh + 2) > rr)) ? 1 : 0;
  const int64_t m = c_lo + pos;
  const float nw_m = pos < cnt ? Pick4(wv, sh + pos) : (found ? kv : sg.limit_end);
  // `mid ? nw[mid-1] : 0` is row-relative
  const float prev = m == sg.row_ptr ? 0.f : Pick4(wv, sh + pos - 1);
  *id = g.nbr[m];
  *w = __fsub_rn(nw_m, prev);
}

// ------------------------------------------------------------------------
// Block-pivot search (K1 variant 6).  Over the distinct roots of a dedup'ed
// hop every row is cold and the launch runs at the chip's random-line rate
// (46 of ~54 G L2 misses/s), touching ~3 cold lines per sample: a level-1
// pivot window, the leaf window of prefix_w, and the id in nbr.  Here the
// pivots index 128-byte EdgeBlocks (10 edges: sums + ids + the previous block's
// last sum in ONE line) instead of 4-element groups of the flat array: level 1
// = one entry per block (skip1), level k+1 entry q = level k entry 5q+4.  A
// sample then touches a level-1 window (a row of degree d has d/320 lines of
// them, shared by its samples) and one block line.
// Same contract as PivotSample: first m in [lo, hi] with nw[m] > r.
// ------------------------------------------------------------------------
__device__ __forceinline__ void BlockPivotSample(const GraphView& g, const Segment& sg,
                                                 double u, uint64_t* id, float* w) {
  const int64_t lo = sg.lo, hi = sg.hi;
  const double rr = ScaleDraw(u, sg.limit_begin, sg.limit_end);
  if (!((double)sg.limit_end > rr)) {
    // Q3: r rounded up to the end of the segment - replay the reference
    const float* nw = g.prefix_w + sg.row_ptr;
    const int32_t m = (int32_t)RandomSelect(nw, (uint64_t)sg.b, (uint64_t)sg.e, u);
    *id = g.nbr[sg.row_ptr + m];
    *w = __fsub_rn(nw[m], m == 0 ? 0.f : nw[m - 1]);
    return;
  }
  // ranges of the levels, bottom up, only as far as needed: K = first level
  // with <= 4 candidates (most rows stop at level 1 or 2, and a wave whose
  // lanes have all stopped skips the remaining divisions)
  uint32_t l[kPivotLevels + 1], h[kPivotLevels + 1];
  l[1] = (uint32_t)(lo / kEdgesPerBlock);
  h[1] = (uint32_t)(hi / kEdgesPerBlock);
  int32_t K = 0;                     // 0: the segment lies inside one block
  if (h[1] != l[1]) {
    K = h[1] - l[1] <= 4u ? 1 : kPivotLevels + 1;
#pragma unroll
    for (int k = 2; k <= kPivotLevels; ++k) {
      l[k] = 0; h[k] = 0;
      if (K > kPivotLevels) {
        l[k] = l[k - 1] / 5u;
        h[k] = h[k - 1] / 5u;
        if (h[k] - l[k] <= 4u) K = k;
      }
    }
  }
  uint32_t x = l[1];
  bool found = false;
  if (K > kPivotLevels) {
    // beyond the levels' reach: bisect the block entries
    uint32_t a = l[1], b = h[1];
    while (a < b) {
      const uint32_t mid = (a + b) >> 1;
      if ((double)g.skip1[mid] > rr) b = mid; else a = mid + 1;
    }
    x = a;
    found = a < h[1];
  } else {
#pragma unroll
    for (int k = kPivotLevels; k >= 1; --k) {
      if (k <= K) {
        uint32_t c_lo, c_hi;
        if (k == K) { c_lo = l[k]; c_hi = h[k]; }
        else {
          c_lo = max(l[k], 5u * x);
          c_hi = found ? 5u * x + 4u : h[k];
        }
        const int32_t cnt = (int32_t)(c_hi - c_lo);
        const float* lvl = k == 1 ? g.skip1 : g.bpiv + g.bpiv_off[k];
        const float4u kw = *reinterpret_cast<const float4u*>(lvl + c_lo);
        int32_t pos = 0;
        pos += (0 < cnt && !((double)kw.x > rr)) ? 1 : 0;
        pos += (1 < cnt && !((double)kw.y > rr)) ? 1 : 0;
        pos += (2 < cnt && !((double)kw.z > rr)) ? 1 : 0;
        pos += (3 < cnt && !((double)kw.w > rr)) ? 1 : 0;
        x = c_lo + (uint32_t)pos;
        if (pos < cnt) found = true;
      }
    }
  }
  // leaf: block x holds the answer (its last sum exceeds r when `found`,
  // otherwise it is the block of hi, whose sum exceeds r)
  const EdgeBlock* bk = g.blk + x;
  const int64_t base = (int64_t)x * kEdgesPerBlock;
  const int32_t i_lo = lo > base ? (int32_t)(lo - base) : 0;
  const int32_t i_hi = found ? kEdgesPerBlock - 1 : (int32_t)(hi - base);   // inclusive
  const float4 a0 = *reinterpret_cast<const float4*>(bk->pw);
  const float4 a1 = *reinterpret_cast<const float4*>(bk->pw + 4);
  const float4 a2 = *reinterpret_cast<const float4*>(bk->pw + 8);   // pw[8], pw[9], prev_last, pad
  const float v[10] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w, a2.x, a2.y};
  int32_t i = i_lo;
#pragma unroll
  for (int j = 0; j < kEdgesPerBlock - 1; ++j)
    i += (j >= i_lo && j < i_hi && !((double)v[j] > rr)) ? 1 : 0;
  float nw_m = v[0], prev = a2.z;
#pragma unroll
  for (int j = 0; j < kEdgesPerBlock; ++j) {
    if (j == i) nw_m = v[j];
    if (j + 1 == i) prev = v[j];
  }
  if (base + i == sg.row_ptr) prev = 0.f;          // `mid ? nw[mid-1] : 0`, row-relative
  *id = bk->nbr[i];
  *w = __fsub_rn(nw_m, prev);
}

// Row record -> searched segment of the listed type; false = empty / invalid
// (node.cc:127-136).
template <bool BLOCKED = false>
__device__ __forceinline__ bool LoadSegment(const GraphView& g, int64_t row,
                                            int32_t t, Segment* sg) {
  if (row < 0 || t < 0 || t >= g.T) return false;
  const uint8_t* rec = g.row_meta + row * (int64_t)g.meta_stride;
  if (g.T == 1) {
    const uint4 q = *reinterpret_cast<const uint4*>(rec);
    sg->row_ptr = (int64_t)(((uint64_t)q.y << 32) | q.x);
    sg->b = 0;
    sg->e = (int32_t)q.z - 1;
    if (g.total_in_meta) {
      // the record's type sum IS the row's last running sum (verified at build):
      // one dependent load less per root
      if (sg->e < 0) return false;
      sg->lo = sg->row_ptr;
      sg->hi = sg->row_ptr + sg->e;
      sg->limit_begin = 0.f;
      sg->limit_end = __uint_as_float(q.w);
      return true;
    }
  } else {
    sg->row_ptr = *reinterpret_cast<const int64_t*>(rec);
    const int32_t* te = reinterpret_cast<const int32_t*>(rec + 8);
    sg->b = t == 0 ? 0 : te[t - 1];
    sg->e = te[t] - 1;
  }
  if (sg->e < sg->b) return false;
  sg->lo = sg->row_ptr + sg->b;
  sg->hi = sg->row_ptr + sg->e;
  if (BLOCKED) {     // same values, read from the block lines the search will touch
    sg->limit_end = BlockedPw(g, sg->hi);
    sg->limit_begin = sg->b == 0 ? 0.f : BlockedPw(g, sg->lo - 1);
  } else {
    sg->limit_end = g.prefix_w[sg->hi];
    sg->limit_begin = sg->b == 0 ? 0.f : g.prefix_w[sg->lo - 1];
  }
  return true;
}

// U = 1: one sample per lane.  U = 2 (even `count`): a lane draws the two
// adjacent samples (j, j+1) of one root - one root id / row record / limit
// load, one Philox block and one 16-byte id store per PAIR.
template <bool TF_LAYOUT, int U, bool BLOCKED>
__device__ __forceinline__ void PivotPass(const SampleNbArgs& a, const int64_t n_roots,
                                          const int64_t stride_rows,
                                          const int32_t stride_slots) {
  const int64_t total = n_roots * (int64_t)a.count;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x * U;
  int64_t s = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * U;
  if (s >= total) return;
  int64_t r = s / a.count;
  int32_t j = (int32_t)(s - r * a.count);
  const int32_t t = a.et[0];
  for (; s < total; s += stride) {
    uint64_t node = a.roots[r];
    if (a.root_mask != nullptr && a.root_mask[r / a.root_group]) node = 0;
    Segment sg;
    const bool valid = LoadSegment<BLOCKED>(a.g, FindRow(a.g, node), t, &sg);
    uint64_t id[U];
    float w[U];
    int32_t ot = t;
    if (valid) {
      const Philox4 blk = RngBlock(a.seed, a.call_id, kDomainNeighbor, node,
                                   ((uint32_t)j) >> 1);
      if (U == 1) {
        const double u = (j & 1) ? UnitFromWords(blk.w[2], blk.w[3])
                                 : UnitFromWords(blk.w[0], blk.w[1]);
        if (BLOCKED) BlockPivotSample(a.g, sg, u, &id[0], &w[0]);
        else PivotSample(a.g, sg, u, &id[0], &w[0]);
      } else if (BLOCKED) {
        BlockPivotSample(a.g, sg, UnitFromWords(blk.w[0], blk.w[1]), &id[0], &w[0]);
        BlockPivotSample(a.g, sg, UnitFromWords(blk.w[2], blk.w[3]), &id[U - 1], &w[U - 1]);
      } else {
        PivotSample(a.g, sg, UnitFromWords(blk.w[0], blk.w[1]), &id[0], &w[0]);
        PivotSample(a.g, sg, UnitFromWords(blk.w[2], blk.w[3]), &id[U - 1], &w[U - 1]);
      }
    } else {
#pragma unroll
      for (int x = 0; x < U; ++x) {
        id[x] = TF_LAYOUT ? (uint64_t)a.default_node : 0;
        w[x] = 0.f;
      }
      ot = TF_LAYOUT ? -1 : 0;
    }
    if (a.packed != nullptr) {
      // wire row of root r: ids (2 words each) | weights | types | mask, pad
      int32_t* row = a.packed + r * (int64_t)(4 * a.count + 2);
#pragma unroll
      for (int x = 0; x < U; ++x) {
        *reinterpret_cast<uint64_t*>(row + 2 * (j + x)) = id[x];
        row[2 * a.count + j + x] = __float_as_int(w[x]);
        row[3 * a.count + j + x] = ot;
      }
      if (j == 0) *reinterpret_cast<int2*>(row + 4 * a.count) = make_int2(valid ? 0 : 1, 0);
    } else if (U == 1) {
      a.out_id[s] = id[0];
      a.out_w[s] = w[0];
      a.out_t[s] = ot;
    } else {
      *reinterpret_cast<ulonglong2*>(a.out_id + s) = make_ulonglong2(id[0], id[U - 1]);
      *reinterpret_cast<float2*>(a.out_w + s) = make_float2(w[0], w[U - 1]);
      *reinterpret_cast<int2*>(a.out_t + s) = make_int2(ot, ot);
    }
    if (j == 0 && a.out_row_mask != nullptr) a.out_row_mask[r] = valid ? 0 : 1;
    if (a.mark_owner != nullptr) {
#pragma unroll
      for (int x = 0; x < U; ++x) MarkNextHop(a.g, a.mark_owner, id[x], valid, s + x);
    }
    r += stride_rows;
    j += stride_slots;
    if (j >= a.count) { j -= a.count; ++r; }
  }
}

template <bool TF_LAYOUT, int U, bool BLOCKED = false>
__global__ __launch_bounds__(256) void SampleNeighborPivotKernel(
    const SampleNbArgs a, const int64_t stride_rows, const int32_t stride_slots) {
  int64_t n_roots;
  if (!DedupGate(a, &n_roots)) return;
  PivotPass<TF_LAYOUT, U, BLOCKED>(a, n_roots, stride_rows, stride_slots);
}

// Both passes of a duplicate-root call in one launch: the device-side count
// picks which one runs - `a` (the given roots, U samples per lane) or `b` (the
// distinct roots into scratch rows, one sample per lane).  A gated launch that
// only exits still costs ~8 us for its 32 768 workgroups.
template <bool TF_LAYOUT, int U, bool BLOCKED>
__global__ __launch_bounds__(256) void SampleNeighborPivotDualKernel(
    const SampleNbArgs a, const int64_t a_rows, const int32_t a_slots,
    const SampleNbArgs b, const int64_t b_rows, const int32_t b_slots) {
  if (DedupActive(a.dd_counter, a.dd_n_in)) {
    PivotPass<TF_LAYOUT, 1, BLOCKED>(b, (int64_t)(*a.dd_counter), b_rows, b_slots);
  } else {
    PivotPass<TF_LAYOUT, U, BLOCKED>(a, a.n, a_rows, a_slots);
  }
}

// ------------------------------------------------------------------------
// Group mode of the block-pivot kernel: a lane draws U adjacent samples of one
// root (count % U == 0).  The root id, row record and limits are loaded once per
// U samples, and when the searched segment lies inside ONE EdgeBlock (the usual
// case for the uniformly drawn roots of a first hop: average degree 10) so are
// the three leaf loads - every sample then costs one id load.  Used for odd
// counts that are a multiple of 5 (fanout 25), where the two-sample mode with
// its 16-byte stores does not apply.
// ------------------------------------------------------------------------
template <bool TF_LAYOUT, int U>
__global__ __launch_bounds__(256) void SampleNeighborGroupKernel(
    const SampleNbArgs a, const int64_t stride_rows, const int32_t stride_slots) {
  int64_t n_roots;
  if (!DedupGate(a, &n_roots)) return;
  const int64_t total = n_roots * (int64_t)a.count;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x * U;
  int64_t s = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * U;
  if (s >= total) return;
  int64_t r = s / a.count;
  int32_t j = (int32_t)(s - r * a.count);
  const int32_t t = a.et[0];
  for (; s < total; s += stride) {
    uint64_t node = a.roots[r];
    if (a.root_mask != nullptr && a.root_mask[r / a.root_group]) node = 0;
    Segment sg;
    const bool valid = LoadSegment<true>(a.g, FindRow(a.g, node), t, &sg);
    uint64_t id[U];
    float w[U];
    int32_t ot = t;
    if (valid) {
      double u[U];
#pragma unroll
      for (int x = 0; x < U; x += 2) {
        // draws j+x, j+x+1 (j is a multiple of U; U odd -> the parity of j varies)
        const uint32_t d = (uint32_t)(j + x);
        const Philox4 b0 = RngBlock(a.seed, a.call_id, kDomainNeighbor, node, d >> 1);
        if ((d & 1) == 0) {
          u[x] = UnitFromWords(b0.w[0], b0.w[1]);
          if (x + 1 < U) u[x + 1] = UnitFromWords(b0.w[2], b0.w[3]);
        } else {
          u[x] = UnitFromWords(b0.w[2], b0.w[3]);
          if (x + 1 < U) {
            const Philox4 b1 = RngBlock(a.seed, a.call_id, kDomainNeighbor, node,
                                        (d + 1) >> 1);
            u[x + 1] = UnitFromWords(b1.w[0], b1.w[1]);
          }
        }
      }
      const int64_t blk_lo = sg.lo / kEdgesPerBlock;
      if (blk_lo == sg.hi / kEdgesPerBlock) {
        // the whole segment sits in one block: one set of leaf loads for U draws
        const EdgeBlock* bk = a.g.blk + blk_lo;
        const int64_t base = blk_lo * kEdgesPerBlock;
        const int32_t i_lo = (int32_t)(sg.lo - base), i_hi = (int32_t)(sg.hi - base);
        const float4 a0 = *reinterpret_cast<const float4*>(bk->pw);
        const float4 a1 = *reinterpret_cast<const float4*>(bk->pw + 4);
        const float4 a2 = *reinterpret_cast<const float4*>(bk->pw + 8);
        const float v[10] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w, a2.x, a2.y};
#pragma unroll
        for (int x = 0; x < U; ++x) {
          const double rr = ScaleDraw(u[x], sg.limit_begin, sg.limit_end);
          if (!((double)sg.limit_end > rr)) {          // Q3: replay the reference
            const float* nw = a.g.prefix_w + sg.row_ptr;
            const int32_t m = (int32_t)RandomSelect(nw, (uint64_t)sg.b, (uint64_t)sg.e, u[x]);
            id[x] = a.g.nbr[sg.row_ptr + m];
            w[x] = __fsub_rn(nw[m], m == 0 ? 0.f : nw[m - 1]);
            continue;
          }
          int32_t i = i_lo;
#pragma unroll
          for (int q = 0; q < kEdgesPerBlock - 1; ++q)
            i += (q >= i_lo && q < i_hi && !((double)v[q] > rr)) ? 1 : 0;
          float nw_m = v[0], prev = a2.z;
#pragma unroll
          for (int q = 0; q < kEdgesPerBlock; ++q) {
            if (q == i) nw_m = v[q];
            if (q + 1 == i) prev = v[q];
          }
          if (base + i == sg.row_ptr) prev = 0.f;
          id[x] = bk->nbr[i];
          w[x] = __fsub_rn(nw_m, prev);
        }
      } else {
#pragma unroll
        for (int x = 0; x < U; ++x) BlockPivotSample(a.g, sg, u[x], &id[x], &w[x]);
      }
    } else {
#pragma unroll
      for (int x = 0; x < U; ++x) {
        id[x] = TF_LAYOUT ? (uint64_t)a.default_node : 0;
        w[x] = 0.f;
      }
      ot = TF_LAYOUT ? -1 : 0;
    }
#pragma unroll
    for (int x = 0; x < U; ++x) {
      a.out_id[s + x] = id[x];
      a.out_w[s + x] = w[x];
      a.out_t[s + x] = ot;
    }
    if (j == 0 && a.out_row_mask != nullptr) a.out_row_mask[r] = valid ? 0 : 1;
    r += stride_rows;
    j += stride_slots;
    if (j >= a.count) { j -= a.count; ++r; }
  }
}

template <int U>
static void LaunchIlp(bool tf, int grid, int block, hipStream_t stream,
                      const SampleNbArgs& a) {
  if (tf) {
    hipLaunchKernelGGL((SampleNeighborIlpKernel<U, true>), dim3(grid), dim3(block),
                       0, stream, a);
  } else {
    hipLaunchKernelGGL((SampleNeighborIlpKernel<U, false>), dim3(grid), dim3(block),
                       0, stream, a);
  }
}

template <int U, int V>
static void LaunchExpandUV(bool ct, int grid, int block, hipStream_t stream,
                           const ExpandArgs& x, int64_t stride_rows, int32_t stride_slots) {
  if (ct) {
    hipLaunchKernelGGL((DedupExpandKernel<U, V, true>), dim3(grid), dim3(block), 0, stream,
                       x, stride_rows, stride_slots);
  } else {
    hipLaunchKernelGGL((DedupExpandKernel<U, V, false>), dim3(grid), dim3(block), 0, stream,
                       x, stride_rows, stride_slots);
  }
}

static void LaunchExpand(int U, int V, bool ct, int grid, int block, hipStream_t stream,
                         const ExpandArgs& x, int64_t stride_rows, int32_t stride_slots) {
  if (U == 2) {
    if (V >= 4) LaunchExpandUV<2, 4>(ct, grid, block, stream, x, stride_rows, stride_slots);
    else if (V >= 2) LaunchExpandUV<2, 2>(ct, grid, block, stream, x, stride_rows, stride_slots);
    else LaunchExpandUV<2, 1>(ct, grid, block, stream, x, stride_rows, stride_slots);
  } else {
    if (V >= 4) LaunchExpandUV<1, 4>(ct, grid, block, stream, x, stride_rows, stride_slots);
    else if (V >= 2) LaunchExpandUV<1, 2>(ct, grid, block, stream, x, stride_rows, stride_slots);
    else LaunchExpandUV<1, 1>(ct, grid, block, stream, x, stride_rows, stride_slots);
  }
}

// Kernel selection for one pass over a.n roots (a.dd_role says which pass).
static int LaunchK1(const euler_gpu_graph* g, hipStream_t stream,
                    const SampleNbArgs& a) {
  const int64_t n = a.n;
  const int32_t count = a.count, layout = a.layout, k = a.k;
  uint64_t* out_id = a.out_id;
  float* out_w = a.out_w;
  int32_t* out_t = a.out_t;
  const int block = 256;
  // K1 launches up to 32768 workgroups (128 per CU) rather than GridFor's 16 per
  // CU: hub-heavy and leaf-heavy workgroups finish at very different times and
  // the finer grain lets the dispatcher even that out (measured -7 %).
  int grid;
  {
    int64_t blocks = (n * (int64_t)count + block - 1) / block;
    const int64_t cap = g_k1_grid_cap > 0 ? g_k1_grid_cap : kK1GridCap;
    if (blocks > cap) blocks = cap;
    grid = (int)(blocks < 1 ? 1 : blocks);
  }
  const bool single = k == 1 && g->view.monotone;
  const bool tf_zero = layout == EULER_GPU_LAYOUT_TF && g->view.has_zero_nbr != 0;
  if ((g_k1_variant == 5 || g_k1_variant == 6) && single && !tf_zero) {
    const bool blocked = g_k1_variant == 6;
    // two samples per lane pay when the launch is bound by memory-instruction
    // throughput (millions of roots with hot rows); the pass over the distinct
    // roots (dd_role 2) is small and cold - there the shorter dependent chain of
    // one sample per lane wins (0.139 vs 0.149 ms on the metric workload)
    const bool pair = g_k1_pair != 0 && a.dd_role != 2 && a.cold_roots == 0 &&
                      a.packed == nullptr && count % 2 == 0 &&
                      ((uintptr_t)out_id % 16 == 0) && ((uintptr_t)out_w % 8 == 0) &&
                      ((uintptr_t)out_t % 8 == 0);
    const int U = pair ? 2 : 1;
    int gridp = grid;
    if (pair) {
      int64_t blocks = (n * (int64_t)count / 2 + block - 1) / block;
      const int64_t cap = g_k1_grid_cap > 0 ? g_k1_grid_cap : kK1GridCap;
      if (blocks > cap) blocks = cap;
      gridp = (int)(blocks < 1 ? 1 : blocks);
    }
    const int64_t stride = (int64_t)gridp * block * U;
    const int64_t stride_rows = stride / count;
    const int32_t stride_slots = (int32_t)(stride - stride_rows * count);
    const bool tf = layout == EULER_GPU_LAYOUT_TF;
    if (blocked && !pair && g_k1_group != 0 && count % 5 == 0 && a.dd_role != 2 &&
        a.mark_owner == nullptr && a.packed == nullptr) {
      // odd multiple of 5 (fanout 25): five adjacent samples per lane
      int64_t blocks = (n * (int64_t)count / 5 + block - 1) / block;
      const int64_t cap = g_k1_grid_cap > 0 ? g_k1_grid_cap : kK1GridCap;
      if (blocks > cap) blocks = cap;
      const int gridg = (int)(blocks < 1 ? 1 : blocks);
      const int64_t gstride = (int64_t)gridg * block * 5;
      const int64_t g_rows = gstride / count;
      const int32_t g_slots = (int32_t)(gstride - g_rows * count);
      if (tf) {
        hipLaunchKernelGGL((SampleNeighborGroupKernel<true, 5>), dim3(gridg), dim3(block),
                           0, stream, a, g_rows, g_slots);
      } else {
        hipLaunchKernelGGL((SampleNeighborGroupKernel<false, 5>), dim3(gridg), dim3(block),
                           0, stream, a, g_rows, g_slots);
      }
      EG_HIP(hipGetLastError());
      return EULER_GPU_OK;
    }
    auto kern = blocked
        ? (pair ? (tf ? SampleNeighborPivotKernel<true, 2, true>
                      : SampleNeighborPivotKernel<false, 2, true>)
                : (tf ? SampleNeighborPivotKernel<true, 1, true>
                      : SampleNeighborPivotKernel<false, 1, true>))
        : (pair ? (tf ? SampleNeighborPivotKernel<true, 2>
                      : SampleNeighborPivotKernel<false, 2>)
                : (tf ? SampleNeighborPivotKernel<true, 1>
                      : SampleNeighborPivotKernel<false, 1>));
    hipLaunchKernelGGL(kern, dim3(gridp), dim3(block), 0, stream, a, stride_rows,
                       stride_slots);
  } else if (g_k1_variant == 4 && single && !tf_zero && count >= 8) {
    const int64_t chunks = (n * (int64_t)count + 63) / 64;
    int64_t blocks = (chunks + 3) / 4;
    if (blocks > 256 * 5) blocks = 256 * 5;     // 5 blocks of 4 waves per CU (LDS)
    if (blocks < 1) blocks = 1;
    if (layout == EULER_GPU_LAYOUT_TF) {
      hipLaunchKernelGGL(SampleNeighborWaveKernel<true>, dim3((int)blocks),
                         dim3(block), 0, stream, a);
    } else {
      hipLaunchKernelGGL(SampleNeighborWaveKernel<false>, dim3((int)blocks),
                         dim3(block), 0, stream, a);
    }
  } else if (g_k1_variant >= 3 && single && !tf_zero) {
    const int64_t stride = (int64_t)grid * block;
    const int64_t stride_rows = stride / count;
    const int32_t stride_slots = (int32_t)(stride - stride_rows * count);
    if (layout == EULER_GPU_LAYOUT_TF) {
      hipLaunchKernelGGL(SampleNeighborBlockedKernel<true>, dim3(grid), dim3(block),
                         0, stream, a, stride_rows, stride_slots, g_k1_ablate);
    } else {
      hipLaunchKernelGGL(SampleNeighborBlockedKernel<false>, dim3(grid), dim3(block),
                         0, stream, a, stride_rows, stride_slots, g_k1_ablate);
    }
  } else if (g_k1_variant == 2 && single && !tf_zero) {
    const bool tf = layout == EULER_GPU_LAYOUT_TF;
    const int U = g_k1_ilp;
    const int gridu = GridFor((n * (int64_t)count + U - 1) / U, block);
    if (U == 1) LaunchIlp<1>(tf, gridu, block, stream, a);
    else if (U == 2) LaunchIlp<2>(tf, gridu, block, stream, a);
    else if (U == 8) LaunchIlp<8>(tf, gridu, block, stream, a);
    else LaunchIlp<4>(tf, gridu, block, stream, a);
  } else if (g_k1_variant >= 1 && single) {
    const int64_t stride = (int64_t)grid * block;
    const int64_t stride_rows = stride / count;
    const int32_t stride_slots = (int32_t)(stride - stride_rows * count);
    const bool tf = layout == EULER_GPU_LAYOUT_TF;
    const bool zc = g->view.has_zero_nbr != 0;
    auto kern = tf ? (zc ? SampleNeighborFastKernel<true, true>
                         : SampleNeighborFastKernel<true, false>)
                   : SampleNeighborFastKernel<false, false>;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(block), 0, stream, a, stride_rows,
                       stride_slots);
  } else {
    hipLaunchKernelGGL(SampleNeighborKernel, dim3(grid), dim3(block), 0, stream, a);
  }
  EG_HIP(hipGetLastError());
  return EULER_GPU_OK;
}

// Per-(graph, stream) scratch, grown on demand.
static int GetWorkspace(const euler_gpu_graph* g, hipStream_t stream, size_t bytes,
                        void** out) {
  std::lock_guard<std::mutex> lk(g->ws_mu);
  auto& slot = g->ws[(void*)stream];
  if (slot.second < bytes) {
    if (slot.first != nullptr) {
      EG_HIP(hipStreamSynchronize(stream));   // earlier calls may still use it
      EG_HIP(hipFree(slot.first));
      slot.first = nullptr; slot.second = 0;
    }
    const size_t want = bytes + bytes / 4;
    hipError_t e = hipMalloc(&slot.first, want);
    if (e != hipSuccess) {
      slot.first = nullptr;
      return Fail(EULER_GPU_ENOMEM, std::string("sample_neighbor workspace: ") +
                                        hipGetErrorString(e));
    }
    slot.second = want;
  }
  *out = slot.first;
  return EULER_GPU_OK;
}

constexpr int64_t kDedupMinRoots = 16384;

// Measurement hook (euler_gpu_time_sample_neighbor_phases): when set, the
// launcher records these 4 events on its stream at the phase boundaries
// [0] start, [1] duplicate detection done, [2] sampling done, [3] expand done.
thread_local hipEvent_t* t_phase_events = nullptr;
thread_local int64_t g_last_unique_offset = -1;   // of the unique count in the workspace
static int64_t ReadU32(const uint8_t* dev) {
  uint32_t v = 0;
  if (hipMemcpy(&v, dev, 4, hipMemcpyDeviceToHost) != hipSuccess) return -1;
  return (int64_t)v;
}
static void PhaseMark(hipStream_t stream, int i) {
  if (t_phase_events != nullptr) (void)hipEventRecord(t_phase_events[i], stream);
}

// The two gated passes of a duplicate-root call as one launch, when both would
// run the pivot kernel (see LaunchK1's selection); false = not applicable.
static bool LaunchK1Dual(const euler_gpu_graph* g, hipStream_t stream,
                         const SampleNbArgs& a, const SampleNbArgs& b) {
  const bool single = a.k == 1 && g->view.monotone;
  const bool tf_zero = a.layout == EULER_GPU_LAYOUT_TF && g->view.has_zero_nbr != 0;
  if (g_k1_dual == 0 || !(g_k1_variant == 5 || g_k1_variant == 6) || !single || tf_zero)
    return false;
  const bool blocked = g_k1_variant == 6;
  const int32_t count = a.count;
  const bool pair = g_k1_pair != 0 && count % 2 == 0 &&
                    ((uintptr_t)a.out_id % 16 == 0) && ((uintptr_t)a.out_w % 8 == 0) &&
                    ((uintptr_t)a.out_t % 8 == 0);
  if (blocked && !pair && g_k1_group != 0 && count % 5 == 0 && a.mark_owner == nullptr)
    return false;                              // pass 1 would take the group kernel
  const int block = 256;
  const int U = pair ? 2 : 1;
  int64_t blocks = (a.n * (int64_t)count / U + block - 1) / block;
  const int64_t cap = g_k1_grid_cap > 0 ? g_k1_grid_cap : kK1GridCap;
  if (blocks > cap) blocks = cap;
  const int grid = (int)(blocks < 1 ? 1 : blocks);
  const int64_t a_stride = (int64_t)grid * block * U;
  const int64_t a_rows = a_stride / count;
  const int32_t a_slots = (int32_t)(a_stride - a_rows * count);
  const int64_t b_stride = (int64_t)grid * block;
  const int64_t b_rows = b_stride / count;
  const int32_t b_slots = (int32_t)(b_stride - b_rows * count);
  const bool tf = a.layout == EULER_GPU_LAYOUT_TF;
#define EG_DUAL(TF, UU, BL)                                                            \
  hipLaunchKernelGGL((SampleNeighborPivotDualKernel<TF, UU, BL>), dim3(grid), dim3(block), \
                     0, stream, a, a_rows, a_slots, b, b_rows, b_slots)
  if (tf) {
    if (pair) { if (blocked) EG_DUAL(true, 2, true); else EG_DUAL(true, 2, false); }
    else { if (blocked) EG_DUAL(true, 1, true); else EG_DUAL(true, 1, false); }
  } else {
    if (pair) { if (blocked) EG_DUAL(false, 2, true); else EG_DUAL(false, 2, false); }
    else { if (blocked) EG_DUAL(false, 1, true); else EG_DUAL(false, 1, false); }
  }
#undef EG_DUAL
  return true;
}

// Scratch of one duplicate-root call, carved out of the per-stream workspace.
// `owner` sits at offset 0 whatever n is: the previous hop of a fanout fills it
// for the next one before that hop's layout exists.
struct DedupLayout {
  size_t scan_bytes, o_owner, o_slot, o_pos, o_uidx, o_uniq, o_cnt, o_scan, o_mask, o_tid,
      o_tw, o_tt, bytes;
};

static int MakeDedupLayout(const euler_gpu_graph* g, hipStream_t stream, int64_t n,
                           int32_t count, DedupLayout* L) {
  auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
  const size_t total_out = (size_t)n * (size_t)count;
  L->scan_bytes = 0;
  {
    hipcub::CountingInputIterator<uint32_t> pos_it(0u);
    hipcub::TransformInputIterator<uint32_t, DedupFlagOp,
                                   hipcub::CountingInputIterator<uint32_t>>
        flag_it(pos_it, DedupFlagOp{nullptr, nullptr, n});
    EG_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, L->scan_bytes, flag_it,
                                            (uint32_t*)nullptr, (int)(n + 1), stream));
    size_t pre = 0;
    hipcub::TransformInputIterator<uint32_t, DedupFlagPremarkedOp,
                                   hipcub::CountingInputIterator<uint32_t>>
        pre_it(pos_it, DedupFlagPremarkedOp{});
    EG_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, pre, pre_it, (uint32_t*)nullptr,
                                            (int)(n + 1), stream));
    if (pre > L->scan_bytes) L->scan_bytes = pre;
  }
  L->o_owner = 0;
  L->o_slot = L->o_owner + al(((size_t)g->view.n_rows + 1) * 4);
  L->o_pos = L->o_slot + al((size_t)n * 4);
  L->o_uidx = L->o_pos + al(((size_t)n + 1) * 4);
  L->o_uniq = L->o_uidx + al((size_t)n * 4);
  L->o_cnt = L->o_uniq + al((size_t)n * 8);
  L->o_scan = L->o_cnt + 256;
  L->o_mask = L->o_scan + al(L->scan_bytes);
  L->o_tid = L->o_mask + al((size_t)n);
  L->o_tw = L->o_tid + al(total_out * 8);
  L->o_tt = L->o_tw + al(total_out * 4);
  L->bytes = L->o_tt + al(total_out * 4);
  return EULER_GPU_OK;
}

// will a call with n roots go through the duplicate-root path?
static bool WantsDedup(const euler_gpu_graph* g, int64_t n, int dedup) {
  return dedup > 0 && g_k1_dedup != 0 && (g_k1_variant == 5 || g_k1_variant == 6) &&
         (n >= kDedupMinRoots || g_k1_dedup == 2) && n < (int64_t)0x3fffffff &&
         g->view.n_rows < (int64_t)0xfffffff0;
}

// can LaunchK1 mark the next hop's owner table for this call (pivot kernels,
// one sample or a pair per lane)?
static bool K1CanMark(const euler_gpu_graph* g, int32_t k, int32_t count, int32_t layout) {
  const bool single = k == 1 && g->view.monotone;
  const bool tf_zero = layout == EULER_GPU_LAYOUT_TF && g->view.has_zero_nbr != 0;
  (void)count;
  return g_k1_fuse_mark != 0 && (g_k1_variant == 5 || g_k1_variant == 6) && single &&
         !tf_zero && layout == EULER_GPU_LAYOUT_TF && g->view.map_mode == 0;
}

// will LaunchK1 pick a pivot kernel (the only ones that write wire rows)?
static bool K1WritesPacked(const euler_gpu_graph* g, int32_t k, int32_t layout) {
  const bool single = k == 1 && g->view.monotone;
  const bool tf_zero = layout == EULER_GPU_LAYOUT_TF && g->view.has_zero_nbr != 0;
  return (g_k1_variant == 5 || g_k1_variant == 6) && single && !tf_zero;
}

// Hop chaining of a fanout (euler_gpu_sample_fanout holds launch_mu and has
// sized the stream's workspace for its largest hop, so `owner` does not move).
struct HopFusion {
  bool premarked = false;   // in: the previous hop entered these roots into `owner`
  bool mark_next = false;   // in: the outputs are the next hop's roots, mark them if
                            // the kernels of this call can; out: whether they did
};

// dedup: 0 = never, 1 = automatic (count duplicates on device, decide there),
// -1 = never and the caller knows the roots to be distinct (cold rows: one sample
// per lane).  packed_out: wire rows instead of the four output arrays.
static int LaunchSampleNeighbor(const euler_gpu_graph* g, hipStream_t stream,
                                uint64_t seed, uint32_t call_id,
                                const uint64_t* roots, int64_t n,
                                const uint8_t* root_mask, int32_t root_group,
                                const int32_t* edge_types, int32_t k,
                                int32_t count, int32_t layout,
                                int64_t default_node, uint64_t* out_id,
                                float* out_w, int32_t* out_t,
                                uint8_t* out_row_mask, int dedup = 1,
                                HopFusion* hop = nullptr, int32_t* packed_out = nullptr) {
  const bool premarked = hop != nullptr && hop->premarked;
  const bool want_mark = hop != nullptr && hop->mark_next;
  if (hop != nullptr) hop->mark_next = false;
  if (g == nullptr) return Fail(EULER_GPU_ENOGRAPH, "sample_neighbor: null graph");
  if (n < 0 || count < 0 || k < 0 || k > kMaxListedTypes)
    return Fail(EULER_GPU_EINVAL, "sample_neighbor: bad n/count/k (k <= 32)");
  if (layout != EULER_GPU_LAYOUT_CORE && layout != EULER_GPU_LAYOUT_TF)
    return Fail(EULER_GPU_EINVAL, "sample_neighbor: bad layout");
  if (n == 0 || count == 0) return EULER_GPU_OK;
  if (!roots || (packed_out == nullptr && (!out_id || !out_w || !out_t)))
    return Fail(EULER_GPU_EINVAL, "sample_neighbor: null buffer");
  if (k > 0 && !edge_types)
    return Fail(EULER_GPU_EINVAL, "sample_neighbor: null edge_types");
  if (packed_out != nullptr && (dedup > 0 || !K1WritesPacked(g, k, layout)))
    return Fail(EULER_GPU_EINVAL, "sample_neighbor: packed output needs the pivot kernels");
  if ((g_k1_variant == 3 || g_k1_variant == 4 || g_k1_variant == 6) &&
      g->view.blk == nullptr) {
    const int rc = EnsureBlockedIndex(g);     // A/B variants only: built on first use
    if (rc != EULER_GPU_OK) return rc;
  }
  SampleNbArgs a{};
  a.g = g->view;
  a.seed = seed; a.call_id = call_id;
  a.roots = roots; a.root_mask = root_mask;
  a.root_group = root_group > 0 ? root_group : 1;
  a.out_id = out_id; a.out_w = out_w; a.out_t = out_t;
  a.out_row_mask = out_row_mask;
  a.n = n; a.default_node = default_node;
  a.k = k; a.count = count; a.layout = layout;
  a.packed = packed_out;
  a.cold_roots = dedup < 0 ? 1 : 0;
  for (int i = 0; i < k; ++i) a.et[i] = edge_types[i];
  const bool try_dedup = WantsDedup(g, n, dedup);
  if (premarked && !try_dedup)
    return Fail(EULER_GPU_EINVAL, "sample_neighbor: premarked roots without dedup");
  const bool do_mark = want_mark && K1CanMark(g, k, count, layout);
  void* wsp = nullptr;
  if (do_mark && !try_dedup) {
    // owner table only (the fanout sized the workspace; this is a lookup)
    const int rc = GetWorkspace(g, stream, ((size_t)g->view.n_rows + 1) * 4, &wsp);
    if (rc != EULER_GPU_OK) return rc;
    a.mark_owner = (uint32_t*)wsp;
  }
  if (!try_dedup) {
    PhaseMark(stream, 0);
    PhaseMark(stream, 1);
    const int rc = LaunchK1(g, stream, a);
    PhaseMark(stream, 2);
    PhaseMark(stream, 3);
    if (rc == EULER_GPU_OK && hop != nullptr) hop->mark_next = do_mark;
    return rc;
  }

  std::lock_guard<std::recursive_mutex> launch_lk(g->launch_mu);
  // ---- workspace -----------------------------------------------------------
  DedupLayout L;
  {
    const int rc = MakeDedupLayout(g, stream, n, count, &L);
    if (rc != EULER_GPU_OK) return rc;
  }
  {
    const int rc = GetWorkspace(g, stream, L.bytes, &wsp);
    if (rc != EULER_GPU_OK) return rc;
  }
  uint8_t* ws = (uint8_t*)wsp;
  DedupArgs d{};
  d.g = g->view;
  d.roots = roots; d.root_mask = root_mask; d.n = n; d.root_group = a.root_group;
  d.owner = (uint32_t*)(ws + L.o_owner);
  d.row_slot = (uint32_t*)(ws + L.o_slot);
  d.pos = (uint32_t*)(ws + L.o_pos);
  d.uidx_of = (uint32_t*)(ws + L.o_uidx);
  d.uniq = (uint64_t*)(ws + L.o_uniq);
  d.counter = (uint32_t*)(ws + L.o_cnt);
  g_last_unique_offset = (int64_t)L.o_cnt;
  const IdentityMap idmap{g->view.id_base, g->view.id_stride, g->view.n_rows};
  const int block = 256;
  const int dgrid = GridFor(n + 1, block);
  PhaseMark(stream, 0);
  hipcub::CountingInputIterator<uint32_t> pos_it(0u);
  if (premarked) {
    // the previous hop's kernels stored every position into owner[slot(key)]
    hipcub::TransformInputIterator<uint32_t, DedupFlagPremarkedOp,
                                   hipcub::CountingInputIterator<uint32_t>>
        flag_it(pos_it, DedupFlagPremarkedOp{d.owner, roots, root_mask, idmap, n,
                                             a.root_group});
    size_t sb = L.scan_bytes;
    EG_HIP(hipcub::DeviceScan::ExclusiveSum(ws + L.o_scan, sb, flag_it, d.pos,
                                            (int)(n + 1), stream));
    hipLaunchKernelGGL(DedupIndexPremarkedKernel, dim3(dgrid), dim3(block), 0, stream, d,
                       idmap);
  } else {
    hipLaunchKernelGGL(DedupMarkKernel, dim3(dgrid), dim3(block), 0, stream, d);
    hipcub::TransformInputIterator<uint32_t, DedupFlagOp,
                                   hipcub::CountingInputIterator<uint32_t>>
        flag_it(pos_it, DedupFlagOp{d.owner, d.row_slot, n});
    size_t sb = L.scan_bytes;
    EG_HIP(hipcub::DeviceScan::ExclusiveSum(ws + L.o_scan, sb, flag_it, d.pos,
                                            (int)(n + 1), stream));
    hipLaunchKernelGGL(DedupIndexKernel, dim3(dgrid), dim3(block), 0, stream, d);
  }
  PhaseMark(stream, 1);
  // ---- pass 1: the given roots, straight to the outputs (few duplicates) -----
  // (the owner table is free again: Flag and Index are done with it, so the
  // kernels that write the final ids may fill it for the next hop)
  a.dd_counter = d.counter;
  a.dd_n_in = n;
  a.dd_role = 1;
  a.mark_owner = do_mark ? d.owner : nullptr;
  // ---- pass 2: the unique roots into scratch rows, then expand ---------------
  SampleNbArgs b = a;
  b.dd_role = 2;
  b.mark_owner = nullptr;
  b.roots = d.uniq;
  b.root_mask = nullptr;          // masked roots were entered as node id 0
  b.root_group = 1;
  b.out_id = (uint64_t*)(ws + L.o_tid);
  b.out_w = (float*)(ws + L.o_tw);
  b.out_t = (int32_t*)(ws + L.o_tt);
  b.out_row_mask = ws + L.o_mask;
  if (!LaunchK1Dual(g, stream, a, b)) {
    int rc = LaunchK1(g, stream, a);
    if (rc == EULER_GPU_OK) rc = LaunchK1(g, stream, b);
    if (rc != EULER_GPU_OK) return rc;
  }
  PhaseMark(stream, 2);
  ExpandArgs x{};
  x.counter = d.counter; x.uidx_of = d.uidx_of;
  x.t_id = b.out_id; x.t_w = b.out_w; x.t_t = b.out_t; x.t_mask = b.out_row_mask;
  x.out_id = out_id; x.out_w = out_w; x.out_t = out_t; x.out_mask = out_row_mask;
  x.n = n; x.count = count;
  x.mark_owner = do_mark ? d.owner : nullptr;
  x.map = idmap;
  const size_t total_out = (size_t)n * (size_t)count;
  const bool pair = count % 2 == 0 && ((uintptr_t)out_id % 16 == 0) &&
                    ((uintptr_t)out_w % 8 == 0) && ((uintptr_t)out_t % 8 == 0);
  const int U = pair ? 2 : 1;
  int64_t blocks = ((int64_t)total_out / U + block - 1) / block;
  const int64_t xcap = g_expand_grid_cap > 0 ? g_expand_grid_cap : kK1GridCap;
  if (blocks > xcap) blocks = xcap;
  if (blocks < 1) blocks = 1;
  const int64_t stride = blocks * block * U;
  const int64_t stride_rows = stride / count;
  const int32_t stride_slots = (int32_t)(stride - stride_rows * count);
  // single-type calls served by the pivot kernels: the type column is a
  // function of the row mask (SampleNeighborPivotKernel: t or -1 / 0)
  const bool ct = g_expand_const_type != 0 && k == 1 && g->view.monotone &&
                  !(layout == EULER_GPU_LAYOUT_TF && g->view.has_zero_nbr != 0);
  x.type0 = k == 1 ? edge_types[0] : 0;
  x.masked_type = layout == EULER_GPU_LAYOUT_TF ? -1 : 0;
  LaunchExpand(pair ? 2 : 1, g_expand_steps, ct, (int)blocks, block, stream, x,
               stride_rows, stride_slots);
  PhaseMark(stream, 3);
  EG_HIP(hipGetLastError());
  if (hop != nullptr) hop->mark_next = do_mark;
  return EULER_GPU_OK;
}

// ------------------------------------------------------------------------
// K2  sample_node: Graph::SampleNode (graph.cc:221-275) over alias tables.
// One lane per sample; draw indices follow the reference's program order
// inside one call (domain NODE, stream 0).
// ------------------------------------------------------------------------
struct SampleNodeArgs {
  NodeSamplerView s;
  uint64_t seed;
  uint64_t* out;
  uint32_t call_id;
  int32_t count;
  int32_t mode;          // 0 fixed type, 1 all types (-1), 2 type list
  int32_t type;          // mode 0
  int32_t n_sub;         // mode 2
  int32_t sub_type[kMaxNodeTypes];
  float sub_sum[kMaxNodeTypes];
};

__device__ __forceinline__ uint64_t AliasNext(const AliasEntry* tab, int64_t n,
                                              double u_col, double u_coin) {
  // AliasMethod::Next (alias_method.cc:66-78)
  const int64_t column = (int64_t)floor(__dmul_rn((double)n, u_col));
  const AliasEntry e = tab[column];
  return u_coin < (double)e.prob ? e.id_self : e.id_alias;
}

__global__ __launch_bounds__(256) void SampleNodeKernel(const SampleNodeArgs a) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.count;
       i += stride) {
    int32_t t = a.type;
    uint64_t d = 0;   // index of the next draw of this sample
    if (a.mode == 0) {
      d = 2 * (uint64_t)i;
    } else if (a.mode == 1) {
      d = 4 * (uint64_t)i;
      const Philox4 b = RngBlock(a.seed, a.call_id, kDomainNode, 0,
                                 (uint32_t)(d >> 1));
      const int64_t col = (int64_t)floor(__dmul_rn(
          (double)a.s.n_types, UnitFromWords(b.w[0], b.w[1])));
      t = UnitFromWords(b.w[2], b.w[3]) < (double)a.s.tc_prob[col]
              ? (int32_t)col : a.s.tc_alias[col];
      d += 2;
    } else {
      d = 3 * (uint64_t)i;
      const double u = RngDraw(a.seed, a.call_id, kDomainNode, 0, d);
      t = a.sub_type[RandomSelect(a.sub_sum, 0, (uint64_t)(a.n_sub - 1), u)];
      d += 1;
    }
    const double u_col = RngDraw(a.seed, a.call_id, kDomainNode, 0, d);
    const double u_coin = RngDraw(a.seed, a.call_id, kDomainNode, 0, d + 1);
    const int64_t b = a.s.type_off[t];
    a.out[i] = AliasNext(a.s.entries + b, a.s.type_off[t + 1] - b, u_col, u_coin);
  }
}

// ------------------------------------------------------------------------
// GetFullNeighbor (node.cc:175-197): count pass + fill pass.
// ------------------------------------------------------------------------
struct FullNbArgs {
  GraphView g;
  const uint64_t* ids;
  int64_t n;
  int32_t k;
  int32_t pad;
  int32_t et[kMaxListedTypes];
};

__device__ __forceinline__ int64_t FullNbCount(const FullNbArgs& a, int64_t row) {
  if (row < 0) return 0;
  const RowMeta m = LoadRowMeta(a.g, row);
  int64_t c = 0;
  for (int32_t x = 0; x < a.k; ++x) {
    const int32_t t = a.et[x];
    if (t >= 0 && t < a.g.T)
      c += m.type_end[t] - (t == 0 ? 0 : m.type_end[t - 1]);
  }
  return c;
}

__global__ void FullNbCountKernel(const FullNbArgs a, int64_t* counts) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < a.n) counts[i] = FullNbCount(a, FindRow(a.g, a.ids[i]));
}

// idx[i] = (offset[i], offset[i+1]) as int32 pairs (FillNeighbor layout).
__global__ void OffsetsToIdxKernel(const int64_t* counts, const int64_t* offsets,
                                   int64_t n, int32_t* idx) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    idx[2 * i] = (int32_t)offsets[i];
    idx[2 * i + 1] = (int32_t)(offsets[i] + counts[i]);
  }
}

// One wave per queried node: lanes stride over the row's listed segments.
__global__ __launch_bounds__(256) void FullNbFillKernel(
    const FullNbArgs a, const int32_t* idx, uint64_t* out_id, float* out_w,
    int32_t* out_t) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t i = wave; i < a.n; i += n_waves) {
    const int64_t row = FindRow(a.g, a.ids[i]);
    if (row < 0) continue;
    const RowMeta m = LoadRowMeta(a.g, row);
    const float* nw = a.g.prefix_w + m.row_ptr;
    const uint64_t* nbr = a.g.nbr + m.row_ptr;
    int64_t o = idx[2 * i];
    for (int32_t x = 0; x < a.k; ++x) {
      const int32_t t = a.et[x];
      if (t < 0 || t >= a.g.T) continue;
      const int32_t b = t == 0 ? 0 : m.type_end[t - 1];
      const int32_t e = m.type_end[t];
      for (int32_t p = b + lane; p < e; p += 64) {
        const float pre = p == 0 ? 0.f : nw[p - 1];
        out_id[o + (p - b)] = nbr[p];
        out_w[o + (p - b)] = __fsub_rn(nw[p], pre);
        out_t[o + (p - b)] = t;
      }
      o += e - b;
    }
  }
}

// ------------------------------------------------------------------------
// K4  random walk.
// p = q = 1 (tf_euler/kernels/random_walk_op.cc:207-247): walk_len dependent
// count=1 hops per walker, chained on the CORE id (a missing row continues
// from the sentinel id 0); output 0 -> default_node.
// ------------------------------------------------------------------------
struct WalkArgs {
  GraphView g;
  uint64_t seed;
  const int64_t* nodes;
  const int32_t* edge_types;   // device [walk_len, k]
  int64_t* out;
  int64_t n;
  int64_t default_node;
  uint32_t call_id;
  int32_t k;
  int32_t walk_len;
  float p;
  float q;
};

// FAST: one listed edge type per step on a graph with non-decreasing running
// sums - every step is the block-pivot search of K1 (draw 0 of the current
// node, call_id + step), i.e. ~log5(deg / 10) + 4 dependent loads instead of
// the reference loop's 2 * ceil(log2 deg).
template <bool FAST>
__global__ __launch_bounds__(256) void RandomWalkKernel(const WalkArgs a) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t L = a.walk_len + 1;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n;
       i += stride) {
    uint64_t cur = (uint64_t)a.nodes[i];
    a.out[i * L] = (int64_t)cur;
    for (int32_t s = 0; s < a.walk_len; ++s) {
      uint64_t id = 0; float w; int32_t t;
      if (FAST) {
        Segment sg;
        if (LoadSegment<true>(a.g, FindRow(a.g, cur), a.edge_types[s], &sg)) {
          const Philox4 blk = RngBlock(a.seed, a.call_id + (uint32_t)s, kDomainNeighbor,
                                       cur, 0);
          BlockPivotSample(a.g, sg, UnitFromWords(blk.w[0], blk.w[1]), &id, &w);
        }
      } else {
        RowSampler rs;
        InitRowSampler(rs, a.g, FindRow(a.g, cur), a.edge_types + s * a.k, a.k);
        if (rs.valid) SampleAt(rs, a.seed, a.call_id + (uint32_t)s, cur, 0, &id, &w, &t);
      }
      a.out[i * L + s + 1] = id == 0 ? a.default_node : (int64_t)id;
      cur = id;
    }
  }
}

// Iterator over GetFullNeighbor(node, listed types) in the reference order
// (listed-type order, storage order inside a type) without materialising it.
struct NbIter {
  const uint64_t* nbr;
  const float* nw;
  const int32_t* type_end;
  const int32_t* et;
  int32_t k, T;
  int32_t x;       // current listed-type slot
  int32_t p, e;    // current position / end inside the row
  __device__ __forceinline__ void Seek() {
    while (x < k) {
      const int32_t t = et[x];
      if (t >= 0 && t < T) {
        p = t == 0 ? 0 : type_end[t - 1];
        e = type_end[t];
        if (p < e) return;
      }
      ++x;
    }
  }
  __device__ __forceinline__ void Init(const GraphView& g, int64_t row,
                                       const int32_t* et_, int32_t k_) {
    et = et_; k = k_; T = g.T; x = 0; p = 0; e = 0;
    if (row < 0) { x = k; return; }
    const RowMeta m = LoadRowMeta(g, row);
    nbr = g.nbr + m.row_ptr; nw = g.prefix_w + m.row_ptr; type_end = m.type_end;
    Seek();
  }
  __device__ __forceinline__ bool Done() const { return x >= k; }
  __device__ __forceinline__ int64_t Id() const { return (int64_t)nbr[p]; }
  __device__ __forceinline__ float Weight() const {
    return __fsub_rn(nw[p], p == 0 ? 0.f : nw[p - 1]);
  }
  __device__ __forceinline__ void Next() {
    if (++p >= e) { ++x; Seek(); }
  }
};

// node2vec step weights (BuildWeights, random_walk_op.cc:140-168) streamed:
// the child list is merged against the parent's list with two cursors and the
// biased weight of each child is produced in order.
struct BiasedStream {
  NbIter c, pn;
  int64_t parent_id;
  float p, q;
  __device__ __forceinline__ bool Done() const { return c.Done(); }
  // weight of the current child (advances the parent cursor as the reference)
  __device__ __forceinline__ float Take(int64_t* id) {
    const int64_t cid = c.Id();
    float w = c.Weight();
    for (;;) {
      if (pn.Done()) {
        w = cid != parent_id ? __fdiv_rn(w, q) : __fdiv_rn(w, p);
        break;
      }
      const int64_t pid = pn.Id();
      if (cid < pid) {
        w = cid != parent_id ? __fdiv_rn(w, q) : __fdiv_rn(w, p);
        break;
      } else if (cid == pid) {
        pn.Next();
        break;
      } else {
        pn.Next();
      }
    }
    *id = cid;
    c.Next();
    return w;
  }
};

// node2vec (RWCallback, random_walk_op.cc:83-138).  One lane per walker.  The
// reference materialises w[], builds f32 running sums and binary-searches
// them; with non-negative weights the hit interval is unique, so the same
// index is found by one sequential pass for the total and a second pass that
// stops at the first running sum > r.  The running sums are the same
// sequential f32 adds, hence bit-identical.  (All-zero totals follow the
// reference's fall-through: every probe moves `low` up, ending on the last
// element.)
__global__ __launch_bounds__(256) void Node2VecKernel(const WalkArgs a) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t L = a.walk_len + 1;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n;
       i += stride) {
    int64_t cur = a.nodes[i];
    int64_t parent = cur;         // parent_ids_ starts as the start nodes
    bool have_parent_nb = false;  // parent_neighbors_ starts empty
    a.out[i * L] = cur;
    for (int32_t s = 0; s < a.walk_len; ++s) {
      const int32_t* et = a.edge_types + s * a.k;
      const int64_t row = FindRow(a.g, (uint64_t)cur);
      const int64_t prow = have_parent_nb ? FindRow(a.g, (uint64_t)parent) : -1;
      const int32_t* pet = s > 0 ? a.edge_types + (s - 1) * a.k : et;
      BiasedStream bs;
      bs.parent_id = parent; bs.p = a.p; bs.q = a.q;
      bs.c.Init(a.g, row, et, a.k);
      bs.pn.Init(a.g, prow, pet, a.k);
      int64_t sample_id = a.default_node;
      if (!bs.Done()) {
        float total = 0.f;
        int64_t nc = 0, id;
        while (!bs.Done()) { total = __fadd_rn(total, bs.Take(&id)); ++nc; }
        const double u = RngDraw(a.seed, a.call_id + (uint32_t)s, kDomainWalk,
                                 (uint64_t)i, 0);
        const double r = ScaleDraw(u, 0.f, total);
        bs.c.Init(a.g, row, et, a.k);
        bs.pn.Init(a.g, prow, pet, a.k);
        float acc = 0.f;
        bool found = false;
        while (!bs.Done()) {
          const float w = bs.Take(&id);
          const float prev = acc;
          acc = __fadd_rn(acc, w);
          if ((double)prev <= r && r < (double)acc) { found = true; break; }
        }
        if (!found) {
          // fall-through of RandomSelect: no interval holds r (total == 0).
          // Every probe then takes `interval_end <= r`: low = mid + 1, so the
          // search ends on mid = nc - 1; `id` already is that last element.
        }
        sample_id = id;
      }
      a.out[i * L + s + 1] = sample_id;
      parent = cur;
      have_parent_nb = true;
      cur = sample_id;
    }
  }
}

// ------------------------------------------------------------------------
// node2vec, one WAVE per walker (default).  The step's weights come out of a
// two-cursor walk over the child's and the parent's neighbour lists in storage
// order (BuildWeights, random_walk_op.cc:140-168) - a sequential recurrence
// that cannot be split across lanes without changing which parent entry each
// child is compared with.  What can be shared is the memory traffic: the 64
// lanes copy both lists into LDS in coalesced chunks (ids, and the weights as
// differences of the running sums), and lane 0 runs the recurrence out of LDS
// (tens of cycles per step instead of a dependent HBM round trip per lane and
// step, and no lane waits for a neighbour's hub row).  Pass 1 accumulates the
// total with the reference's sequential f32 adds, pass 2 stops at the first
// running sum > r - the same index the reference's bisection of those sums
// returns (and its last element when the total is 0).  Measured on the metric
// graph (100 K walkers x 10 steps, walkers sit on hubs of 1e5+ neighbours):
// 1.69 s -> 1.15 s; prefetching the next entries by hand made it slower (1.34 s).
// ------------------------------------------------------------------------
constexpr int kN2vChunk = 256;
constexpr int kN2vMaxSeg = kMaxListedTypes;

struct N2vList {           // one neighbour list = listed type segments of a row
  int64_t row_ptr;         // row start in nbr / prefix_w
  int32_t n_seg;
  int32_t total;           // entries
  int32_t seg_b[kN2vMaxSeg];
  int32_t seg_len[kN2vMaxSeg];
};

struct alignas(16) N2vLds {
  uint64_t c_id[kN2vChunk];
  uint64_t p_id[kN2vChunk];
  float c_w[kN2vChunk];
  N2vList child, parent;
};

// Built by lane 0, read by all lanes after a wave sync.
__device__ __forceinline__ void N2vBuildList(N2vList* L, const GraphView& g, int64_t row,
                                             const int32_t* et, int32_t k) {
  L->n_seg = 0; L->total = 0; L->row_ptr = 0;
  if (row < 0) return;
  const RowMeta m = LoadRowMeta(g, row);
  L->row_ptr = m.row_ptr;
  for (int32_t x = 0; x < k; ++x) {
    const int32_t t = et[x];
    if (t < 0 || t >= g.T) continue;
    const int32_t b = t == 0 ? 0 : m.type_end[t - 1];
    const int32_t len = m.type_end[t] - b;
    if (len <= 0) continue;
    L->seg_b[L->n_seg] = b;
    L->seg_len[L->n_seg] = len;
    ++L->n_seg;
    L->total += len;
  }
}

// row-relative position of logical entry j
__device__ __forceinline__ int32_t N2vPhys(const N2vList& L, int32_t j) {
  for (int32_t x = 0; x < L.n_seg; ++x) {
    if (j < L.seg_len[x]) return L.seg_b[x] + j;
    j -= L.seg_len[x];
  }
  return 0;
}

__global__ __launch_bounds__(256) void Node2VecWaveKernel(const WalkArgs a) {
  __shared__ N2vLds lds_all[4];
  N2vLds& S = lds_all[threadIdx.x >> 6];
  const int lane = threadIdx.x & 63;
  const int64_t waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const int64_t L = a.walk_len + 1;
  for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; i < a.n;
       i += waves) {
    int64_t cur = a.nodes[i];
    int64_t parent = cur;          // parent_ids_ starts as the start nodes
    bool have_parent_nb = false;   // parent_neighbors_ starts empty
    if (lane == 0) a.out[i * L] = cur;
    for (int32_t s = 0; s < a.walk_len; ++s) {
      const int32_t* et = a.edge_types + s * a.k;
      const int32_t* pet = s > 0 ? a.edge_types + (s - 1) * a.k : et;
      WaveSync();
      if (lane == 0) {
        N2vBuildList(&S.child, a.g, FindRow(a.g, (uint64_t)cur), et, a.k);
        N2vBuildList(&S.parent, a.g,
                     have_parent_nb ? FindRow(a.g, (uint64_t)parent) : -1, pet, a.k);
      }
      WaveSync();
      const int32_t nc = S.child.total, np = S.parent.total;
      int64_t sample_id = a.default_node;
      if (nc > 0) {
        const float* c_nw = a.g.prefix_w + S.child.row_ptr;
        const uint64_t* c_nbr = a.g.nbr + S.child.row_ptr;
        const uint64_t* p_nbr = a.g.nbr + S.parent.row_ptr;
        float total = 0.f;
        double r = 0.0;
        uint64_t last_id = 0;
        for (int pass = 0; pass < 2; ++pass) {
          int32_t j = 0, k = 0;           // cursors (logical entries)
          int32_t cj0 = 0, pk0 = 0;       // chunk bases
          int32_t c_have = 0, p_have = 0; // entries loaded in each chunk
          float acc = 0.f;
          bool found = false;
          bool need_c = true, need_p = np > 0;
          while (j < nc && !found) {
            if (need_c) {
              WaveSync();
              cj0 = j;
              c_have = min(kN2vChunk, nc - cj0);
              for (int32_t t = lane; t < c_have; t += 64) {
                const int32_t ph = N2vPhys(S.child, cj0 + t);
                S.c_id[t] = c_nbr[ph];
                S.c_w[t] = __fsub_rn(c_nw[ph], ph == 0 ? 0.f : c_nw[ph - 1]);
              }
              need_c = false;
            }
            if (need_p) {
              WaveSync();
              pk0 = k;
              p_have = min(kN2vChunk, np - pk0);
              for (int32_t t = lane; t < p_have; t += 64)
                S.p_id[t] = p_nbr[N2vPhys(S.parent, pk0 + t)];
              need_p = false;
            }
            WaveSync();
            if (lane == 0) {
              const int32_t c_end = cj0 + c_have;
              const int32_t p_end = pk0 + p_have;
              while (j < c_end) {
                const int64_t cid = (int64_t)S.c_id[j - cj0];
                float w = S.c_w[j - cj0];
                if (k < np) {
                  if (k >= p_end) break;               // next parent chunk
                  const int64_t pid = (int64_t)S.p_id[k - pk0];
                  if (cid > pid) { ++k; continue; }    // parent cursor only
                  if (cid == pid) ++k;                 // common neighbour: weight kept
                  else w = cid != parent ? __fdiv_rn(w, a.q) : __fdiv_rn(w, a.p);
                } else {
                  w = cid != parent ? __fdiv_rn(w, a.q) : __fdiv_rn(w, a.p);
                }
                const float prev = acc;
                acc = __fadd_rn(acc, w);
                last_id = (uint64_t)cid;
                ++j;
                if (pass == 1 && (double)prev <= r && r < (double)acc) { found = true; break; }
              }
            }
            j = __shfl(j, 0);
            k = __shfl(k, 0);
            found = __shfl((int)found, 0) != 0;
            need_c = j >= cj0 + c_have;
            need_p = k < np && k >= pk0 + p_have;
          }
          if (pass == 0) {
            total = __shfl(acc, 0);
            const double u = RngDraw(a.seed, a.call_id + (uint32_t)s, kDomainWalk,
                                     (uint64_t)i, 0);
            r = ScaleDraw(u, 0.f, total);
          }
        }
        // found: last_id is the hit; not found (total == 0): RandomSelect's
        // fall-through ends on the last element, which is last_id as well
        const uint32_t lo32 = __shfl((uint32_t)last_id, 0);
        const uint32_t hi32 = __shfl((uint32_t)(last_id >> 32), 0);
        sample_id = (int64_t)(((uint64_t)hi32 << 32) | lo32);
      }
      if (lane == 0) a.out[i * L + s + 1] = sample_id;
      parent = cur;
      have_parent_nb = true;
      cur = sample_id;
    }
  }
}

struct GenPairArgs {
  const int64_t* paths;
  int64_t* out;
  int64_t batch, path_len, pair_count;
  int32_t left, right;
};

// GenPair (tf_euler/kernels/gen_pair_op.cc:66-84): one lane per (path, j).
__global__ void GenPairKernel(const GenPairArgs a) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= a.batch * a.path_len) return;
  const int64_t i = idx / a.path_len, j = idx - i * a.path_len;
  // pairs emitted before position j: sum over j' < j of (min(j',L) + min(len-1-j',R))
  int64_t before = 0;
  for (int64_t x = 0; x < j; ++x) {
    const int64_t l = x < a.left ? x : a.left;
    const int64_t r0 = a.path_len - 1 - x;
    before += l + (r0 < a.right ? r0 : a.right);
  }
  const int64_t* path = a.paths + i * a.path_len;
  int64_t* o = a.out + (i * a.pair_count + before) * 2;
  int k = 0;
  while ((j - k - 1) >= 0 && k < a.left) { *o++ = path[j]; *o++ = path[j - k - 1]; ++k; }
  k = 0;
  while ((j + k + 1) < a.path_len && k < a.right) { *o++ = path[j]; *o++ = path[j + k + 1]; ++k; }
}

// Algorithmic bytes of one sample_neighbor launch: summed per
// root from its actual degree.
__global__ void AlgoBytesKernel(const FullNbArgs a, int32_t count, double* acc) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  double b = 0.0;
  if (i < a.n) {
    const int64_t row = FindRow(a.g, a.ids[i]);
    const int32_t mode = TypeModeOf(a.k, a.g.T);
    // per root: id in (8) + row_ptr pair (16) + type offsets (4k') + idx out (8)
    b = 8.0 + 16.0 + 4.0 * (mode == kTypeSingle ? 1 : a.g.T) + 8.0;
    double per = 16.0;  // id + weight + type out
    if (row >= 0) {
      const RowMeta m = LoadRowMeta(a.g, row);
      int32_t deg;
      if (mode == kTypeSingle) {
        const int32_t t = a.et[0];
        deg = (t >= 0 && t < a.g.T)
                  ? m.type_end[t] - (t == 0 ? 0 : m.type_end[t - 1]) : 0;
      } else {
        deg = m.type_end[a.g.T - 1];
      }
      if (deg > 0) {
        const int32_t d2 = deg < 2 ? 2 : deg;
        per += 8.0 + 8.0 + 4.0 * (double)(32 - __clz(d2 - 1));
        if (mode != kTypeSingle) {
          const int32_t t2 = a.g.T < 2 ? 2 : a.g.T;
          per += 4.0 * (double)(32 - __clz(t2 - 1)) + 8.0;
        }
      }
    }
    b += per * count;
  }
  // wave reduction then one atomic per wave
  for (int off = 32; off > 0; off >>= 1) b += __shfl_down(b, off, 64);
  if ((threadIdx.x & 63) == 0 && b != 0.0) atomicAdd(acc, b);
}

}  // namespace euler_gpu

using namespace euler_gpu;

// The hops of a fanout under one lock: hop h's kernels that write the final ids
// also enter them into hop h+1's owner table (MarkNextHop), so the duplicate
// detection of hop h+1 starts at its scan.  events: 4 per hop (PhaseMark) or
// null; uniq_off: per hop, the workspace offset of the hop's unique count (-1
// when the hop did not look for duplicates) or null.
static int RunFanout(const euler_gpu_graph* g, hipStream_t stream, uint64_t seed,
                     uint32_t call_id, const uint64_t* roots_dev, int64_t n,
                     const int32_t* edge_types_host, int32_t k,
                     const int32_t* counts_host, int32_t layers, int64_t default_node,
                     uint64_t* const* out_id_dev, float* const* out_w_dev,
                     int32_t* const* out_t_dev, void* workspace_dev, hipEvent_t* events,
                     int64_t* uniq_off) {
  std::lock_guard<std::recursive_mutex> launch_lk(g->launch_mu);
  // size the stream's scratch for the largest hop up front: the owner table a
  // hop fills for its successor must not move in between
  {
    size_t need = 0;
    int64_t m = n;
    for (int32_t h = 0; h < layers; ++h) {
      if (m > 0 && counts_host[h] > 0 && WantsDedup(g, m, h == 0 ? 0 : 1)) {
        DedupLayout L;
        const int rc = MakeDedupLayout(g, stream, m, counts_host[h], &L);
        if (rc != EULER_GPU_OK) return rc;
        if (L.bytes > need) need = L.bytes;
      }
      m *= counts_host[h];
    }
    if (need > 0) {
      void* p = nullptr;
      const int rc = GetWorkspace(g, stream, need, &p);
      if (rc != EULER_GPU_OK) return rc;
    }
  }
  const uint64_t* roots = roots_dev;
  const uint8_t* mask = nullptr;
  int32_t group = 1;
  int64_t m = n;
  uint8_t* ws = (uint8_t*)workspace_dev;
  HopFusion hop;
  int rc = EULER_GPU_OK;
  for (int32_t h = 0; h < layers && rc == EULER_GPU_OK; ++h) {
    // hop h: roots are the previous hop's TF-layout ids; rows the previous hop
    // marked as missing sample as the sentinel id 0, which is what the
    // reference's chained GQL feeds on (sample_fanout_op.cc:37-42).
    uint8_t* row_mask = ws;
    ws += ((size_t)m + 15) & ~(size_t)15;
    const int64_t m_next = m * counts_host[h];
    // hop 0 samples the caller's batch; later hops sample sampled neighbours,
    // which repeat
    const int dedup = h == 0 ? 0 : 1;
    hop.mark_next = h + 1 < layers && m_next > 0 && WantsDedup(g, m_next, 1);
    if (events != nullptr) t_phase_events = events + (size_t)h * 4;
    g_last_unique_offset = -1;
    rc = LaunchSampleNeighbor(g, stream, seed, call_id + (uint32_t)h, roots, m, mask,
                              group, edge_types_host + (size_t)h * k, k, counts_host[h],
                              EULER_GPU_LAYOUT_TF, default_node, out_id_dev[h],
                              out_w_dev[h], out_t_dev[h], row_mask, dedup, &hop);
    if (uniq_off != nullptr) uniq_off[h] = g_last_unique_offset;
    hop.premarked = hop.mark_next;     // what this hop did is the next hop's input
    roots = out_id_dev[h];
    mask = row_mask;
    group = counts_host[h];
    m = m_next;
  }
  t_phase_events = nullptr;
  return rc;
}

extern "C" {

int euler_gpu_set_tuning(int32_t key, int32_t value) {
  if (key == 0) { g_k1_variant = value; return EULER_GPU_OK; }
  if (key == 2) { g_k1_ablate = value; return EULER_GPU_OK; }
  if (key == 3) { g_k1_grid_cap = value; return EULER_GPU_OK; }
  if (key == 9) { g_k1_fuse_mark = value != 0; return EULER_GPU_OK; }
  if (key == 10 && (value == 1 || value == 2 || value == 4)) {
    g_expand_steps = value; return EULER_GPU_OK;
  }
  if (key == 11) { g_expand_const_type = value != 0; return EULER_GPU_OK; }
  if (key == 13) { g_k1_dual = value != 0; return EULER_GPU_OK; }
  if (key == 12 && value >= 0) { g_expand_grid_cap = value; return EULER_GPU_OK; }
  if (key == 4) { g_k1_pair = value; return EULER_GPU_OK; }
  if (key == 5) { g_k1_dedup = value; return EULER_GPU_OK; }
  if (key == 6) { g_k1_group = value; return EULER_GPU_OK; }
  if (key == 7) { g_n2v_wave = value; return EULER_GPU_OK; }
  if (key == 8) { g_feature_vec4 = value; return EULER_GPU_OK; }
  if (key == 1 && (value == 1 || value == 2 || value == 4 || value == 8)) {
    g_k1_ilp = value;
    return EULER_GPU_OK;
  }
  return Fail(EULER_GPU_EINVAL, "set_tuning: unknown key");
}

int euler_gpu_sample_neighbor(const euler_gpu_graph* g, void* stream,
                              uint64_t seed, uint32_t call_id,
                              const uint64_t* roots_dev, int64_t n,
                              const uint8_t* root_mask_dev, int32_t root_group,
                              const int32_t* edge_types_host, int32_t k,
                              int32_t count, int32_t layout,
                              int64_t default_node, uint64_t* out_id_dev,
                              float* out_w_dev, int32_t* out_t_dev,
                              uint8_t* out_row_mask_dev) {
  return LaunchSampleNeighbor(g, (hipStream_t)stream, seed, call_id, roots_dev, n,
                              root_mask_dev, root_group, edge_types_host, k,
                              count, layout, default_node, out_id_dev, out_w_dev,
                              out_t_dev, out_row_mask_dev);
}

int euler_gpu_sample_neighbor_distinct(const euler_gpu_graph* g, void* stream,
                                       uint64_t seed, uint32_t call_id,
                                       const uint64_t* roots_dev, int64_t n,
                                       const int32_t* edge_types_host, int32_t k,
                                       int32_t count, int32_t layout,
                                       int64_t default_node, uint64_t* out_id_dev,
                                       float* out_w_dev, int32_t* out_t_dev,
                                       uint8_t* out_row_mask_dev) {
  return LaunchSampleNeighbor(g, (hipStream_t)stream, seed, call_id, roots_dev, n,
                              nullptr, 1, edge_types_host, k, count, layout,
                              default_node, out_id_dev, out_w_dev, out_t_dev,
                              out_row_mask_dev, /*dedup=*/-1);
}

int euler_gpu_sample_neighbor_packed(const euler_gpu_graph* g, void* stream,
                                     uint64_t seed, uint32_t call_id,
                                     const uint64_t* roots_dev, int64_t n,
                                     const int32_t* edge_types_host, int32_t k,
                                     int32_t count, int64_t default_node,
                                     int32_t* packed_dev) {
  if (g == nullptr) return Fail(EULER_GPU_ENOGRAPH, "sample_neighbor_packed: null graph");
  if (n < 0 || count <= 0) return Fail(EULER_GPU_EINVAL, "sample_neighbor_packed: bad n/count");
  if (n == 0) return EULER_GPU_OK;
  if (!packed_dev) return Fail(EULER_GPU_EINVAL, "sample_neighbor_packed: null buffer");
  hipStream_t st = (hipStream_t)stream;
  if (K1WritesPacked(g, k, EULER_GPU_LAYOUT_TF))
    return LaunchSampleNeighbor(g, st, seed, call_id, roots_dev, n, nullptr, 1,
                                edge_types_host, k, count, EULER_GPU_LAYOUT_TF, default_node,
                                nullptr, nullptr, nullptr, nullptr, /*dedup=*/-1, nullptr,
                                packed_dev);
  // type draws, non-monotone rows, the id-0 sentinel rule: the reference loop
  // into scratch arrays, then the pack kernel
  auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
  const size_t total = (size_t)n * (size_t)count;
  const size_t o_w = al(total * 8), o_t = o_w + al(total * 4), o_m = o_t + al(total * 4);
  uint8_t* buf = nullptr;
  EG_HIP(hipMallocAsync((void**)&buf, o_m + al((size_t)n), st));
  int rc = LaunchSampleNeighbor(g, st, seed, call_id, roots_dev, n, nullptr, 1,
                                edge_types_host, k, count, EULER_GPU_LAYOUT_TF, default_node,
                                (uint64_t*)buf, (float*)(buf + o_w), (int32_t*)(buf + o_t),
                                buf + o_m, /*dedup=*/-1);
  if (rc == EULER_GPU_OK)
    rc = euler_gpu_pack_rows(stream, (const uint64_t*)buf, (const float*)(buf + o_w),
                             (const int32_t*)(buf + o_t), buf + o_m, n, count, packed_dev);
  (void)hipFreeAsync(buf, st);
  return rc;
}

size_t euler_gpu_sample_fanout_workspace(int64_t n, const int32_t* counts_host,
                                         int32_t layers) {
  // one mask byte per root of every hop (16-byte aligned slices)
  size_t total = 0;
  int64_t m = n;
  for (int32_t h = 0; h < layers; ++h) {
    total += ((size_t)m + 15) & ~(size_t)15;
    m *= counts_host[h];
  }
  return total + 16;
}

int euler_gpu_sample_fanout(const euler_gpu_graph* g, void* stream,
                            uint64_t seed, uint32_t call_id,
                            const uint64_t* roots_dev, int64_t n,
                            const int32_t* edge_types_host, int32_t k,
                            const int32_t* counts_host, int32_t layers,
                            int64_t default_node, uint64_t* const* out_id_dev,
                            float* const* out_w_dev, int32_t* const* out_t_dev,
                            void* workspace_dev) {
  if (layers < 0 || (layers > 0 && (!counts_host || !out_id_dev || !out_w_dev ||
                                    !out_t_dev)))
    return Fail(EULER_GPU_EINVAL, "sample_fanout: bad arguments");
  if (layers > 0 && n > 0 && !workspace_dev)
    return Fail(EULER_GPU_EINVAL, "sample_fanout: workspace required");
  if (g == nullptr) return Fail(EULER_GPU_ENOGRAPH, "sample_fanout: null graph");
  return RunFanout(g, (hipStream_t)stream, seed, call_id, roots_dev, n, edge_types_host, k,
                   counts_host, layers, default_node, out_id_dev, out_w_dev, out_t_dev,
                   workspace_dev, nullptr, nullptr);
}

int euler_gpu_sample_node(const euler_gpu_graph* g, void* stream, uint64_t seed,
                          uint32_t call_id, const int32_t* node_types_host,
                          int32_t k, int32_t count, uint64_t* out_dev) {
  if (!g) return Fail(EULER_GPU_ENOGRAPH, "sample_node: null graph");
  if (!g->has_sampler)
    return Fail(EULER_GPU_ENOGRAPH, "sample_node: graph has no global sampler");
  if (count < 0 || k < 0 || (k > 0 && !node_types_host))
    return Fail(EULER_GPU_EINVAL, "sample_node: bad arguments");
  if (count == 0) return EULER_GPU_OK;
  if (!out_dev) return Fail(EULER_GPU_EINVAL, "sample_node: null output");
  SampleNodeArgs a{};
  a.s = g->sampler;
  a.seed = seed; a.call_id = call_id; a.count = count; a.out = out_dev;
  const int32_t T = g->sampler.n_types;
  if (k == 1) {                                     // api.cc:33-35
    const int32_t type = node_types_host[0];
    if (type == -1) {                               // graph.cc:229-236
      if (g->sampler.tc_sum == 0.f)
        return Fail(EULER_GPU_EEMPTY, "sample_node: total node weight is 0");
      a.mode = 1;
    } else {
      if (type < 0 || type >= T)
        return Fail(EULER_GPU_EINVAL, "sample_node: node type out of range");
      if (g->sampler.sampler_sum[type] == 0.f ||
          g->sampler.type_off[type + 1] == g->sampler.type_off[type])
        return Fail(EULER_GPU_EEMPTY, "sample_node: type weight is 0");
      a.mode = 0; a.type = type;
    }
  } else {                                          // graph.cc:247-275
    a.mode = 2;
    float acc = 0.f;
    int32_t m = 0;
    for (int32_t t = 0; t < T; ++t) {
      bool in = false;
      for (int32_t j = 0; j < k; ++j) in |= node_types_host[j] == t;
      if (in) {
        acc += g->sampler.type_sum[t];
        a.sub_type[m] = t; a.sub_sum[m] = acc; ++m;
      }
    }
    a.n_sub = m;
    if (m == 0 || !(a.sub_sum[m - 1] > 0.f))
      return Fail(EULER_GPU_EEMPTY, "sample_node: listed types have zero weight");
  }
  const int block = 256;
  hipLaunchKernelGGL(SampleNodeKernel, dim3(GridFor(count, block)), dim3(block),
                     0, (hipStream_t)stream, a);
  EG_HIP(hipGetLastError());
  return EULER_GPU_OK;
}

}  // extern "C"

// exclusive scan helper (mp_kernels.hip)
namespace euler_gpu {
int ExclusiveScanI64(hipStream_t stream, const int64_t* in, int64_t* out,
                     int64_t n);
}

extern "C" {

int euler_gpu_get_full_neighbor(const euler_gpu_graph* g, void* stream,
                                const uint64_t* ids_dev, int64_t n,
                                const int32_t* edge_types_host, int32_t k,
                                int32_t* idx_dev, int64_t* total_host,
                                uint64_t* out_id_dev, float* out_w_dev,
                                int32_t* out_t_dev) {
  if (!g) return Fail(EULER_GPU_ENOGRAPH, "get_full_neighbor: null graph");
  if (n < 0 || k < 0 || k > kMaxListedTypes)
    return Fail(EULER_GPU_EINVAL, "get_full_neighbor: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  if (n == 0) { if (total_host) *total_host = 0; return EULER_GPU_OK; }
  if (!idx_dev || !ids_dev)
    return Fail(EULER_GPU_EINVAL, "get_full_neighbor: null buffer");
  FullNbArgs a{};
  a.g = g->view; a.ids = ids_dev; a.n = n; a.k = k;
  for (int i = 0; i < k; ++i) a.et[i] = edge_types_host[i];
  const int block = 256;
  if (out_id_dev == nullptr) {
    int64_t* counts = nullptr;
    EG_HIP(hipMallocAsync((void**)&counts, (2 * n + 2) * sizeof(int64_t), st));
    int64_t* offsets = counts + n + 1;
    hipLaunchKernelGGL(FullNbCountKernel, dim3((n + block - 1) / block),
                       dim3(block), 0, st, a, counts);
    int rc = ExclusiveScanI64(st, counts, offsets, n);
    if (rc != EULER_GPU_OK) return rc;
    hipLaunchKernelGGL(OffsetsToIdxKernel, dim3((n + block - 1) / block),
                       dim3(block), 0, st, counts, offsets, n, idx_dev);
    int32_t last[2];
    EG_HIP(hipMemcpyAsync(last, idx_dev + 2 * (n - 1), 8, hipMemcpyDeviceToHost, st));
    EG_HIP(hipStreamSynchronize(st));
    EG_HIP(hipFreeAsync(counts, st));
    if (total_host) *total_host = last[1];
    return EULER_GPU_OK;
  }
  const int64_t waves_needed = n;
  const int grid = GridFor(waves_needed * 64, block);
  hipLaunchKernelGGL(FullNbFillKernel, dim3(grid), dim3(block), 0, st, a, idx_dev,
                     out_id_dev, out_w_dev, out_t_dev);
  EG_HIP(hipGetLastError());
  return EULER_GPU_OK;
}

int euler_gpu_random_walk(const euler_gpu_graph* g, void* stream, uint64_t seed,
                          uint32_t call_id, const int64_t* nodes_dev, int64_t n,
                          const int32_t* edge_types_host, int32_t k,
                          int32_t walk_len, float p, float q,
                          int64_t default_node, int64_t* out_dev) {
  if (!g) return Fail(EULER_GPU_ENOGRAPH, "random_walk: null graph");
  if (n < 0 || walk_len < 0 || k < 0 || k > kMaxListedTypes)
    return Fail(EULER_GPU_EINVAL, "random_walk: bad arguments");
  if (n == 0) return EULER_GPU_OK;
  if (!nodes_dev || !out_dev || (k > 0 && walk_len > 0 && !edge_types_host))
    return Fail(EULER_GPU_EINVAL, "random_walk: null buffer");
  hipStream_t st = (hipStream_t)stream;
  int32_t* et_dev = nullptr;
  const size_t et_bytes = (size_t)walk_len * (k > 0 ? k : 1) * sizeof(int32_t) + 16;
  EG_HIP(hipMallocAsync((void**)&et_dev, et_bytes, st));
  if (k > 0 && walk_len > 0)
    EG_HIP(hipMemcpyAsync(et_dev, edge_types_host,
                          (size_t)walk_len * k * sizeof(int32_t),
                          hipMemcpyHostToDevice, st));
  WalkArgs a{};
  a.g = g->view; a.seed = seed; a.call_id = call_id; a.nodes = nodes_dev;
  a.edge_types = et_dev; a.out = out_dev; a.n = n; a.default_node = default_node;
  a.k = k; a.walk_len = walk_len; a.p = p; a.q = q;
  const int block = 256;
  const float kEps = 1.0e-6;
  // random_walk_op.cc:281: fabs(p_ - 1.0) <= kEps && fabs(q_ - 1.0) <= kEps
  if (std::fabs((double)p - 1.0) <= kEps && std::fabs((double)q - 1.0) <= kEps) {
    if (k == 1 && g->view.monotone && g->view.blk != nullptr && g_k1_variant >= 5) {
      hipLaunchKernelGGL(RandomWalkKernel<true>, dim3(GridFor(n, block)), dim3(block), 0,
                         st, a);
    } else {
      hipLaunchKernelGGL(RandomWalkKernel<false>, dim3(GridFor(n, block)), dim3(block), 0,
                         st, a);
    }
  } else {
    if (g_n2v_wave != 0) {
      hipLaunchKernelGGL(Node2VecWaveKernel, dim3(GridFor(n * 64, block)), dim3(block), 0,
                         st, a);
    } else {
      hipLaunchKernelGGL(Node2VecKernel, dim3(GridFor(n, block)), dim3(block), 0, st, a);
    }
  }
  EG_HIP(hipGetLastError());
  // the edge-type table must outlive the kernel: stream-ordered free
  EG_HIP(hipFreeAsync(et_dev, st));
  return EULER_GPU_OK;
}

int64_t euler_gpu_gen_pair_count(int64_t path_len, int32_t left_win,
                                 int32_t right_win) {
  // gen_pair_op.cc:48-54
  int64_t pair_count = path_len * (left_win + right_win);
  for (int i = left_win, j = 0; i > 0 && j < path_len; --i, ++j) pair_count -= i;
  for (int i = right_win, j = 0; i > 0 && j < path_len; --i, ++j) pair_count -= i;
  return pair_count;
}

int euler_gpu_gen_pair(void* stream, const int64_t* paths_dev, int64_t batch,
                       int64_t path_len, int32_t left_win, int32_t right_win,
                       int64_t* out_dev) {
  if (batch < 0 || path_len < 0 || left_win < 0 || right_win < 0)
    return Fail(EULER_GPU_EINVAL, "gen_pair: bad arguments");
  if (batch == 0 || path_len == 0) return EULER_GPU_OK;
  GenPairArgs a{paths_dev, out_dev, batch, path_len,
                euler_gpu_gen_pair_count(path_len, left_win, right_win),
                left_win, right_win};
  const int block = 256;
  const int64_t items = batch * path_len;
  hipLaunchKernelGGL(GenPairKernel, dim3((items + block - 1) / block), dim3(block),
                     0, (hipStream_t)stream, a);
  EG_HIP(hipGetLastError());
  return EULER_GPU_OK;
}

int euler_gpu_time_sample_neighbor(const euler_gpu_graph* g, void* stream,
                                   uint64_t seed, const uint64_t* roots_dev,
                                   int64_t n, const int32_t* edge_types_host,
                                   int32_t k, int32_t count, int32_t layout,
                                   uint64_t* out_id_dev, float* out_w_dev,
                                   int32_t* out_t_dev, int32_t iters,
                                   float* mean_ms_host) {
  if (iters <= 0 || !mean_ms_host)
    return Fail(EULER_GPU_EINVAL, "time_sample_neighbor: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  hipEvent_t e0, e1;
  EG_HIP(hipEventCreate(&e0));
  EG_HIP(hipEventCreate(&e1));
  EG_HIP(hipEventRecord(e0, st));
  for (int32_t it = 0; it < iters; ++it) {
    int rc = LaunchSampleNeighbor(g, st, seed, (uint32_t)it, roots_dev, n, nullptr,
                                  1, edge_types_host, k, count, layout, -1,
                                  out_id_dev, out_w_dev, out_t_dev, nullptr);
    if (rc != EULER_GPU_OK) return rc;
  }
  EG_HIP(hipEventRecord(e1, st));
  EG_HIP(hipEventSynchronize(e1));
  float ms = 0.f;
  EG_HIP(hipEventElapsedTime(&ms, e0, e1));
  EG_HIP(hipEventDestroy(e0));
  EG_HIP(hipEventDestroy(e1));
  *mean_ms_host = ms / (float)iters;
  return EULER_GPU_OK;
}

// ------------------------------------------------------------------------
// Wire format of the multi-GPU result exchange: one row of 4*count + 2 int32
// words per root = [count ids (2 words each) | count weights | count types |
// mask | pad], so that one all-to-all moves everything a hop returns and every
// row starts 8-byte aligned.  PackRows writes it from the sampler's outputs;
// ExpandPacked reads it back per POSITION through `pos` (merge + gather +
// unpack in one pass).
// ------------------------------------------------------------------------
__global__ __launch_bounds__(256) void PackRowsKernel(
    const uint64_t* __restrict__ ids, const float* __restrict__ w,
    const int32_t* __restrict__ t, const uint8_t* __restrict__ mask, int64_t m,
    int32_t count, int32_t* __restrict__ packed) {
  const int32_t words = 4 * count + 2;
  const int64_t total = m * (int64_t)words;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < total;
       s += stride) {
    const int64_t r = s / words;
    const int32_t c = (int32_t)(s - r * words);
    int32_t v;
    if (c < 2 * count) v = reinterpret_cast<const int32_t*>(ids)[r * 2 * count + c];
    else if (c < 3 * count) v = __float_as_int(w[r * count + (c - 2 * count)]);
    else if (c < 4 * count) v = t[r * count + (c - 3 * count)];
    else v = c == 4 * count ? (int32_t)mask[r] : 0;
    packed[s] = v;
  }
}

int euler_gpu_pack_rows(void* stream, const uint64_t* id_dev, const float* w_dev,
                        const int32_t* t_dev, const uint8_t* mask_dev, int64_t m,
                        int32_t count, int32_t* packed_dev) {
  if (m < 0 || count <= 0) return Fail(EULER_GPU_EINVAL, "pack_rows: bad m/count");
  if (m == 0) return EULER_GPU_OK;
  if (!id_dev || !w_dev || !t_dev || !mask_dev || !packed_dev)
    return Fail(EULER_GPU_EINVAL, "pack_rows: null buffer");
  const int block = 256;
  hipLaunchKernelGGL(PackRowsKernel, dim3(GridFor(m * (4LL * count + 2), block)),
                     dim3(block), 0, (hipStream_t)stream, id_dev, w_dev, t_dev, mask_dev,
                     m, count, packed_dev);
  EG_HIP(hipGetLastError());
  return EULER_GPU_OK;
}

int euler_gpu_expand_packed(void* stream, const int32_t* pos_dev, int64_t n,
                            int32_t count, const int32_t* packed_dev,
                            uint64_t* out_id_dev, float* out_w_dev, int32_t* out_t_dev,
                            uint8_t* out_mask_dev) {
  if (n < 0 || count <= 0) return Fail(EULER_GPU_EINVAL, "expand_packed: bad n/count");
  if (n == 0) return EULER_GPU_OK;
  if (!pos_dev || !packed_dev || !out_id_dev || !out_w_dev || !out_t_dev)
    return Fail(EULER_GPU_EINVAL, "expand_packed: null buffer");
  const int block = 256;
  const bool pair = count % 2 == 0 && ((uintptr_t)out_id_dev % 16 == 0) &&
                    ((uintptr_t)out_w_dev % 8 == 0) && ((uintptr_t)out_t_dev % 8 == 0) &&
                    ((uintptr_t)packed_dev % 8 == 0);
  const int U = pair ? 2 : 1;
  int64_t blocks = ((int64_t)n * count / U + block - 1) / block;
  if (blocks > kK1GridCap) blocks = kK1GridCap;
  if (blocks < 1) blocks = 1;
  const int64_t stride = blocks * block * U;
  const int64_t stride_rows = stride / count;
  const int32_t stride_slots = (int32_t)(stride - stride_rows * count);
  if (pair) {
    hipLaunchKernelGGL(ExpandPackedKernel<2>, dim3((int)blocks), dim3(block), 0,
                       (hipStream_t)stream, pos_dev, packed_dev, n, count, out_id_dev,
                       out_w_dev, out_t_dev, out_mask_dev, stride_rows, stride_slots);
  } else {
    hipLaunchKernelGGL(ExpandPackedKernel<1>, dim3((int)blocks), dim3(block), 0,
                       (hipStream_t)stream, pos_dev, packed_dev, n, count, out_id_dev,
                       out_w_dev, out_t_dev, out_mask_dev, stride_rows, stride_slots);
  }
  EG_HIP(hipGetLastError());
  return EULER_GPU_OK;
}

int euler_gpu_expand_rows(void* stream, const int32_t* pos_dev, int64_t n,
                          int32_t count, const uint64_t* row_id_dev,
                          const float* row_w_dev, const int32_t* row_t_dev,
                          const uint8_t* row_mask_dev, uint64_t* out_id_dev,
                          float* out_w_dev, int32_t* out_t_dev, uint8_t* out_mask_dev) {
  if (n < 0 || count < 0) return Fail(EULER_GPU_EINVAL, "expand_rows: bad n/count");
  if (n == 0 || count == 0) return EULER_GPU_OK;
  if (!pos_dev || !row_id_dev || !row_w_dev || !row_t_dev || !out_id_dev ||
      !out_w_dev || !out_t_dev || (out_mask_dev && !row_mask_dev))
    return Fail(EULER_GPU_EINVAL, "expand_rows: null buffer");
  hipStream_t st = (hipStream_t)stream;
  // DedupExpandKernel gates on a device-side counter: give it one that says
  // "expand" (0 distinct roots * 4 <= n * 3)
  uint32_t* zero = nullptr;
  EG_HIP(hipMallocAsync((void**)&zero, 16, st));
  EG_HIP(hipMemsetAsync(zero, 0, 16, st));
  ExpandArgs x{};
  x.counter = zero;
  x.uidx_of = reinterpret_cast<const uint32_t*>(pos_dev);
  x.t_id = row_id_dev; x.t_w = row_w_dev; x.t_t = row_t_dev; x.t_mask = row_mask_dev;
  x.out_id = out_id_dev; x.out_w = out_w_dev; x.out_t = out_t_dev;
  x.out_mask = out_mask_dev;
  x.n = n; x.count = count;
  const int block = 256;
  const bool pair = count % 2 == 0 && ((uintptr_t)out_id_dev % 16 == 0) &&
                    ((uintptr_t)out_w_dev % 8 == 0) && ((uintptr_t)out_t_dev % 8 == 0) &&
                    ((uintptr_t)row_id_dev % 16 == 0) && ((uintptr_t)row_w_dev % 8 == 0) &&
                    ((uintptr_t)row_t_dev % 8 == 0);
  const int U = pair ? 2 : 1;
  int64_t blocks = ((int64_t)n * count / U + block - 1) / block;
  if (blocks > kK1GridCap) blocks = kK1GridCap;
  if (blocks < 1) blocks = 1;
  const int64_t stride = blocks * block * U;
  const int64_t stride_rows = stride / count;
  const int32_t stride_slots = (int32_t)(stride - stride_rows * count);
  LaunchExpand(U, g_expand_steps, false, (int)blocks, block, st, x, stride_rows,
               stride_slots);
  EG_HIP(hipGetLastError());
  EG_HIP(hipFreeAsync(zero, st));
  return EULER_GPU_OK;
}

int euler_gpu_time_sample_neighbor_phases(const euler_gpu_graph* g, void* stream,
                                          uint64_t seed, const uint64_t* roots_dev,
                                          int64_t n, const int32_t* edge_types_host,
                                          int32_t k, int32_t count, int32_t layout,
                                          int32_t dedup, uint64_t* out_id_dev,
                                          float* out_w_dev, int32_t* out_t_dev,
                                          int32_t iters, float* mean_ms3_host,
                                          int64_t* n_unique_host) {
  if (iters <= 0 || iters > 64 || !mean_ms3_host)
    return Fail(EULER_GPU_EINVAL, "time_sample_neighbor_phases: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  std::vector<hipEvent_t> ev((size_t)iters * 4);
  for (auto& e : ev) EG_HIP(hipEventCreate(&e));
  int rc = EULER_GPU_OK;
  for (int32_t it = 0; it < iters && rc == EULER_GPU_OK; ++it) {
    t_phase_events = ev.data() + (size_t)it * 4;
    rc = LaunchSampleNeighbor(g, st, seed, (uint32_t)it, roots_dev, n, nullptr, 1,
                              edge_types_host, k, count, layout, -1, out_id_dev,
                              out_w_dev, out_t_dev, nullptr, dedup);
  }
  t_phase_events = nullptr;
  if (rc == EULER_GPU_OK) {
    EG_HIP(hipStreamSynchronize(st));
    float sum[3] = {0.f, 0.f, 0.f};
    for (int32_t it = 0; it < iters; ++it)
      for (int p = 0; p < 3; ++p) {
        float ms = 0.f;
        EG_HIP(hipEventElapsedTime(&ms, ev[(size_t)it * 4 + p], ev[(size_t)it * 4 + p + 1]));
        sum[p] += ms;
      }
    for (int p = 0; p < 3; ++p) mean_ms3_host[p] = sum[p] / (float)iters;
    if (n_unique_host != nullptr) {
      *n_unique_host = -1;
      std::lock_guard<std::mutex> lk(g->ws_mu);
      auto it = g->ws.find((void*)st);
      // the unique count of the last launch sits at a fixed offset only the
      // launcher knows; report it through the side channel it left behind
      *n_unique_host = g_last_unique_offset >= 0 && it != g->ws.end()
                           ? ReadU32((const uint8_t*)it->second.first + g_last_unique_offset)
                           : -1;
    }
  }
  for (auto& e : ev) (void)hipEventDestroy(e);
  return rc;
}

int euler_gpu_time_sample_fanout_phases(const euler_gpu_graph* g, void* stream,
                                        uint64_t seed, const uint64_t* roots_dev,
                                        int64_t n, const int32_t* edge_types_host,
                                        int32_t k, const int32_t* counts_host,
                                        int32_t layers, int64_t default_node,
                                        uint64_t* const* out_id_dev,
                                        float* const* out_w_dev,
                                        int32_t* const* out_t_dev, void* workspace_dev,
                                        int32_t iters, float* mean_ms_host,
                                        int64_t* n_unique_host) {
  if (!g) return Fail(EULER_GPU_ENOGRAPH, "time_sample_fanout_phases: null graph");
  if (iters <= 0 || iters > 64 || layers <= 0 || layers > 8 || !mean_ms_host ||
      !counts_host || !out_id_dev || !out_w_dev || !out_t_dev || !workspace_dev)
    return Fail(EULER_GPU_EINVAL, "time_sample_fanout_phases: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const size_t per_it = (size_t)layers * 4;
  std::vector<hipEvent_t> ev((size_t)iters * per_it);
  for (auto& e : ev) EG_HIP(hipEventCreate(&e));
  std::vector<int64_t> off((size_t)layers, -1);
  int rc = EULER_GPU_OK;
  for (int32_t it = 0; it < iters && rc == EULER_GPU_OK; ++it)
    rc = RunFanout(g, st, seed, (uint32_t)(it * layers), roots_dev, n, edge_types_host, k,
                   counts_host, layers, default_node, out_id_dev, out_w_dev, out_t_dev,
                   workspace_dev, ev.data() + (size_t)it * per_it, off.data());
  if (rc == EULER_GPU_OK) {
    EG_HIP(hipStreamSynchronize(st));
    for (int32_t h = 0; h < layers; ++h)
      for (int p = 0; p < 3; ++p) {
        float sum = 0.f;
        for (int32_t it = 0; it < iters; ++it) {
          float ms = 0.f;
          const size_t b = (size_t)it * per_it + (size_t)h * 4 + p;
          EG_HIP(hipEventElapsedTime(&ms, ev[b], ev[b + 1]));
          sum += ms;
        }
        mean_ms_host[h * 3 + p] = sum / (float)iters;
      }
    if (n_unique_host != nullptr) {
      // a later hop reuses the scratch of an earlier one, so only the last
      // hop's count is still there after the call; earlier hops report -1
      std::lock_guard<std::mutex> lk(g->ws_mu);
      auto it = g->ws.find((void*)st);
      for (int32_t h = 0; h < layers; ++h) n_unique_host[h] = -1;
      const int32_t last = layers - 1;
      if (off[last] >= 0 && it != g->ws.end())
        n_unique_host[last] = ReadU32((const uint8_t*)it->second.first + off[last]);
    }
  }
  for (auto& e : ev) (void)hipEventDestroy(e);
  return rc;
}

int euler_gpu_sample_neighbor_algo_bytes(const euler_gpu_graph* g, void* stream,
                                         const uint64_t* roots_dev, int64_t n,
                                         const int32_t* edge_types_host,
                                         int32_t k, int32_t count,
                                         double* bytes_host) {
  if (!g) return Fail(EULER_GPU_ENOGRAPH, "algo_bytes: null graph");
  if (n < 0 || k < 0 || k > kMaxListedTypes || !bytes_host)
    return Fail(EULER_GPU_EINVAL, "algo_bytes: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  double* acc = nullptr;
  EG_HIP(hipMallocAsync((void**)&acc, sizeof(double), st));
  EG_HIP(hipMemsetAsync(acc, 0, sizeof(double), st));
  FullNbArgs a{};
  a.g = g->view; a.ids = roots_dev; a.n = n; a.k = k;
  for (int i = 0; i < k; ++i) a.et[i] = edge_types_host[i];
  const int block = 256;
  if (n > 0)
    hipLaunchKernelGGL(AlgoBytesKernel, dim3((n + block - 1) / block), dim3(block),
                       0, st, a, count, acc);
  EG_HIP(hipMemcpyAsync(bytes_host, acc, sizeof(double), hipMemcpyDeviceToHost, st));
  EG_HIP(hipStreamSynchronize(st));
  EG_HIP(hipFreeAsync(acc, st));
  return EULER_GPU_OK;
}

}  // extern "C"
