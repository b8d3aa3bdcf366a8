// sample_neighbor / sample_fanout for gfx950: the default pivot kernels, the
// on-device duplicate-root path (mark / scan / index / expand), hop chaining,
// the launchers and their C-ABI entry points, the multi-GPU wire-row kernels and
// the timing entries of bench.py.  The reference loop and the earlier search
// variants live in k1_variants.hip, SampleNode / GetFullNeighbor / walks in
// walk_kernels.hip, the shared device code in k1_args.h / k1_search.h.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <cmath>
#include <vector>

#include "k1_args.h"
#include "k1_search.h"
#include "fanout_local.h"
#include "fanout_plain.h"

extern thread_local int g_feature_vec4;   // mp_kernels.hip
namespace euler_gpu { extern thread_local int g_walk_collapse, g_walk_grid, g_walk_tail, g_walk_lean; extern std::atomic<int> g_walk_path_ch, g_n2v_list_big, g_n2v_list_big_parent, g_n2v_list_merged, g_n2v_walk_tickets; }   // walk_kernels.hip
namespace euler_gpu { extern std::atomic<int> g_sharded_self_exchange, g_sharded_walk_enqueued, g_sharded_walk_tail, g_sharded_walk_split; }   // sharded.cc (process-wide)
namespace euler_gpu { extern std::atomic<int> g_flow_fused; extern std::atomic<int> g_flow_rowpos; }              // dataflow_kernels.hip (key 60)
namespace euler_gpu { extern std::atomic<int> g_blk_fail_next; }           // graph_build.hip (test hook)
namespace euler_gpu { extern thread_local int g_root_host_batch, g_adj_scan, g_adj_long_row, g_sum_scalar; }   // layer_kernels.hip

namespace euler_gpu {

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

// Numbering the representatives without a device-wide scan over the positions
// (tuning key 14).  The rocPRIM scan evaluated the flags - a random read of
// `owner` each - inside its load phase and took 39 + 5 us for the metric's 3.28 M
// hop-2 roots, the index kernel another 19.  Here: one position per thread reads
// its representative once (full occupancy, nothing else in flight) and the
// workgroup counts its representatives; ONE workgroup scans the 3 200 counts;
// the representatives take block offset + rank among the block's
// representatives; every position then looks up its representative's number.
struct DedupBlockArgs {
  DedupArgs d;
  IdentityMap map;
  int32_t premarked;
  int32_t pad;
  uint32_t* rep;      // [n] representative of every position
  uint32_t* posrep;   // [n] index into uniq, valid at representatives
  uint32_t* bcnt;     // [n_blocks] representatives per workgroup
  uint32_t* boff;     // [n_blocks] exclusive scan of bcnt
};

constexpr int kDedupBlock = 1024;   // positions per workgroup: 3 200 counts to scan for 3.28 M roots

__global__ __launch_bounds__(kDedupBlock) void DedupRepCountKernel(const DedupBlockArgs a) {
  const int64_t i = (int64_t)blockIdx.x * kDedupBlock + threadIdx.x;
  bool is_rep = false;
  if (i < a.d.n) {
    const uint32_t slot = a.premarked ? a.map.Slot(DedupKey(a.d, i)) : a.d.row_slot[i];
    const uint32_t r = a.d.owner[slot];
    a.rep[i] = r;
    is_rep = r == (uint32_t)i;
  }
  const int c = __syncthreads_count(is_rep ? 1 : 0);
  if (threadIdx.x == 0) a.bcnt[blockIdx.x] = (uint32_t)c;
}

__global__ __launch_bounds__(1024) void DedupBlockScanKernel(const uint32_t* cnt, int64_t m,
                                                             uint32_t* off, uint32_t* counter) {
  __shared__ uint32_t part[1024];
  const int tid = threadIdx.x;
  const int64_t per = (m + 1023) / 1024;
  const int64_t b = (int64_t)tid * per;
  const int64_t e = b + per < m ? b + per : m;
  uint32_t sum = 0;
  for (int64_t x = b; x < e; ++x) sum += cnt[x];
  part[tid] = sum;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {
    const uint32_t v = tid >= d ? part[tid - d] : 0u;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  uint32_t run = part[tid] - sum;
  for (int64_t x = b; x < e; ++x) { off[x] = run; run += cnt[x]; }
  if (tid == 1023) counter[0] = part[1023];
}

__global__ __launch_bounds__(kDedupBlock) void DedupAssignKernel(const DedupBlockArgs a) {
  __shared__ uint32_t wave_cnt[kDedupBlock / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t i = (int64_t)blockIdx.x * kDedupBlock + threadIdx.x;
  const bool is_rep = i < a.d.n && a.rep[i] == (uint32_t)i;
  const unsigned long long m = __ballot(is_rep);
  if (lane == 0) wave_cnt[wave] = (uint32_t)__popcll(m);
  __syncthreads();
  if (is_rep) {
    uint32_t before = 0;
    for (int w = 0; w < wave; ++w) before += wave_cnt[w];
    const uint32_t q = a.boff[blockIdx.x] + before + (uint32_t)__popcll(m & ((1ULL << lane) - 1));
    a.d.uniq[q] = DedupKey(a.d, i);
    a.posrep[i] = q;
  }
}

__global__ __launch_bounds__(256) void DedupResolveKernel(const DedupBlockArgs a) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.d.n; i += stride)
    a.d.uidx_of[i] = a.posrep[a.rep[i]];
}

// Numbering in ONE pass (tuning key 14 = 2, default).  The three kernels above
// exist to give the representatives consecutive numbers in position order - an
// order nothing depends on: a number only names the scratch row the distinct
// root is sampled into.  So a workgroup counts the representatives among its
// 4 096 positions and takes that many numbers from the call's counter with ONE
// returning atomic (800 atomics for the metric's 3.28 M hop-2 roots; a single
// word sustains ~88 of them per microsecond, spread over the kernel's run).  The
// representative then leaves its number IN the owner table, flagged with bit 31
// (positions are < 2^30): a reader of owner[slot] sees either the representative's
// position or a flagged number - both say "not me" to everybody else - and
// DedupResolveKernel (or the expansion itself, last hop) finds every position's
// number with one lookup.  The counter ends up holding the number of distinct
// roots; `next_counter` (the other counter of the stream's pair) is cleared for
// the stream's next call.
constexpr uint32_t kOwnerNumbered = 0x80000000u;
constexpr int kNumberTile = 4096, kNumberThreads = 1024;

__global__ __launch_bounds__(kNumberThreads) void DedupNumberKernel(const DedupBlockArgs a,
                                                                    uint32_t* next_counter) {
  __shared__ uint32_t wave_cnt[kNumberThreads / 64];
  __shared__ uint32_t base_s;
  constexpr int kPer = kNumberTile / kNumberThreads;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t tile0 = (int64_t)blockIdx.x * kNumberTile;
  if (blockIdx.x == 0 && threadIdx.x == 0 && next_counter != nullptr) next_counter[0] = 0;
  uint64_t key[kPer];
  uint32_t slot[kPer];
  bool is_rep[kPer];
#pragma unroll
  for (int x = 0; x < kPer; ++x) {
    const int64_t i = tile0 + (int64_t)x * kNumberThreads + threadIdx.x;
    is_rep[x] = false;
    key[x] = 0; slot[x] = 0;
    if (i < a.d.n) {
      key[x] = DedupKey(a.d, i);
      slot[x] = a.premarked ? a.map.Slot(key[x]) : a.d.row_slot[i];
    }
  }
#pragma unroll
  for (int x = 0; x < kPer; ++x) {
    const int64_t i = tile0 + (int64_t)x * kNumberThreads + threadIdx.x;
    if (i < a.d.n) is_rep[x] = a.d.owner[slot[x]] == (uint32_t)i;
  }
  const uint64_t lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  uint32_t rank[kPer];
  uint32_t mine = 0;
#pragma unroll
  for (int x = 0; x < kPer; ++x) {
    const unsigned long long m = __ballot(is_rep[x]);
    rank[x] = mine + (uint32_t)__popcll(m & lt);
    mine += (uint32_t)__popcll(m);
  }
  if (lane == 0) wave_cnt[wave] = mine;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t total = 0;
    for (int w = 0; w < kNumberThreads / 64; ++w) { const uint32_t c = wave_cnt[w]; wave_cnt[w] = total; total += c; }
    base_s = total != 0 ? atomicAdd(a.d.counter, total) : 0u;
  }
  __syncthreads();
  const uint32_t base = base_s + wave_cnt[wave];
#pragma unroll
  for (int x = 0; x < kPer; ++x) {
    if (is_rep[x]) {
      const uint32_t q = base + rank[x];
      a.d.uniq[q] = key[x];
      a.d.owner[slot[x]] = q | kOwnerNumbered;
    }
  }
}

// uidx_of[i] = the number DedupNumberKernel left at position i's slot
__global__ __launch_bounds__(256) void DedupResolveNumberedKernel(const DedupBlockArgs a) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.d.n; i += stride) {
    const uint32_t slot = a.premarked ? a.map.Slot(DedupKey(a.d, i)) : a.d.row_slot[i];
    a.d.uidx_of[i] = a.d.owner[slot] & ~kOwnerNumbered;
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
  // resolve_owner != null (last hop of a numbered call, nothing marks `owner`
  // any more): the row number of position i is owner[slot(i)] & ~kOwnerNumbered -
  // no DedupResolve kernel, no uidx array
  const uint32_t* resolve_owner;
  const uint32_t* resolve_row_slot;   // slot of every position, or null = map.Slot(key)
  const uint64_t* roots;
  const uint8_t* root_mask;
  int32_t root_group;
  int32_t pad;
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
__global__ __launch_bounds__(256, kWavesPerSimd) void DedupExpandKernel(const ExpandArgs a,
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
    if (a.resolve_owner != nullptr) {
      uint32_t sl[V];
#pragma unroll
      for (int v = 0; v < V; ++v) {
        const int64_t p = live[v] ? iv[v] : 0;
        if (a.resolve_row_slot != nullptr) {
          sl[v] = a.resolve_row_slot[p];
        } else {
          uint64_t key = a.roots[p];
          if (a.root_mask != nullptr && a.root_mask[p / a.root_group]) key = 0;
          sl[v] = a.map.Slot(key);
        }
      }
#pragma unroll
      for (int v = 0; v < V; ++v) uv[v] = (int64_t)(a.resolve_owner[sl[v]] & 0x7fffffffu);
    } else {
#pragma unroll
      for (int v = 0; v < V; ++v) uv[v] = (int64_t)a.uidx_of[live[v] ? iv[v] : 0];
    }
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

// The same gather-copy with as few instructions as it takes.  The kernel above
// compiles to ~1 800 instructions (64-bit index arithmetic, a division by `count`,
// the grid-stride bookkeeping of V steps, the mark / resolve options), and with
// 16 M lane-iterations on the metric's second hop its ISSUE time - not the 0.9 GB
// it moves - set its 125-135 us.  Here a workgroup owns 256 / P whole positions
// (P = count / 2 pairs per row), a lane one pair: tid -> (position, pair) is one
// multiply and a shift, all offsets are 32-bit, the body is 4 loads and 3 stores.
// Even counts, no next-hop marking (the last hop of a fanout), n * count < 2^31.
template <bool CT>
__global__ __launch_bounds__(256, kWavesPerSimd) void DedupExpandLeanKernel(
    const uint32_t* __restrict__ counter, const uint32_t* __restrict__ uidx_of,
    const uint64_t* __restrict__ t_id, const float* __restrict__ t_w,
    const int32_t* __restrict__ t_t, const uint8_t* __restrict__ t_mask,
    uint64_t* __restrict__ out_id, float* __restrict__ out_w, int32_t* __restrict__ out_t,
    uint8_t* __restrict__ out_mask, const uint32_t n, const uint32_t count, const uint32_t P,
    const uint32_t inv_p, const uint32_t rows_per_block, const int32_t type0,
    const int32_t masked_type) {
  typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  typedef int i32x2 __attribute__((ext_vector_type(2)));
  if (!DedupActive(counter, (int64_t)n)) return;
  const uint32_t r_in_block = (threadIdx.x * inv_p) >> 16;          // tid / P
  const uint32_t pr = threadIdx.x - r_in_block * P;                 // tid % P
  if (r_in_block >= rows_per_block) return;
  for (uint32_t i = blockIdx.x * rows_per_block + r_in_block; i < n;
       i += gridDim.x * rows_per_block) {
    const uint32_t u = uidx_of[i];
    const uint32_t src = u * count + 2u * pr;
    const uint32_t dst = i * count + 2u * pr;
    const u64x2 i2 = *reinterpret_cast<const u64x2*>(t_id + src);
    const f32x2 w2 = *reinterpret_cast<const f32x2*>(t_w + src);
    i32x2 t2;
    uint8_t m = 0;
    if (CT) {
      m = t_mask[u];
      const int32_t tt = m ? masked_type : type0;
      t2 = i32x2{tt, tt};
    } else {
      t2 = *reinterpret_cast<const i32x2*>(t_t + src);
      if (pr == 0 && out_mask != nullptr) m = t_mask[u];
    }
    __builtin_nontemporal_store(i2, reinterpret_cast<u64x2*>(out_id + dst));
    __builtin_nontemporal_store(w2, reinterpret_cast<f32x2*>(out_w + dst));
    __builtin_nontemporal_store(t2, reinterpret_cast<i32x2*>(out_t + dst));
    if (pr == 0 && out_mask != nullptr) out_mask[i] = m;
  }
}

// Back end of a multi-GPU hop: position i takes row pos[i] of the packed
// answers ((3 + TCOL) * count + 2 int32 words per row: ids, weights, [types],
// mask, pad; without the type column the types are rebuilt from the mask).
// U = 2 (even count, aligned outputs): a lane moves two adjacent samples -
// rows are 8-byte aligned, so the ids are two 8-byte loads and one 16-byte
// store; weights and types one 8-byte load and store each.
template <int U, bool TCOL>
__global__ __launch_bounds__(256, kWavesPerSimd) void ExpandPackedKernel(
    const int32_t* __restrict__ pos, const int32_t* __restrict__ packed, int64_t n,
    int32_t count, int32_t single_type, uint64_t* __restrict__ out_id,
    float* __restrict__ out_w, int32_t* __restrict__ out_t, uint8_t* __restrict__ out_mask,
    const int64_t stride_rows, const int32_t stride_slots) {
  typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  typedef int i32x2 __attribute__((ext_vector_type(2)));
  const int32_t cols = TCOL ? 4 : 3;
  const int32_t words = PackedWords(count, TCOL ? 1 : 0);
  const int64_t total = n * (int64_t)count;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x * U;
  int64_t s = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * U;
  if (s >= total) return;
  int64_t i = s / count;
  int32_t j = (int32_t)(s - i * count);
  for (; s < total; s += stride) {
    const int32_t* row = packed + (int64_t)pos[i] * words;
    const int32_t masked = (!TCOL || (j == 0 && out_mask != nullptr)) ? row[cols * count] : 0;
    const int32_t ct = masked ? -1 : single_type;       // used when !TCOL
    if (U == 1) {
      const uint64_t id = *reinterpret_cast<const uint64_t*>(row + 2 * j);
      __builtin_nontemporal_store(id, out_id + s);
      __builtin_nontemporal_store(__int_as_float(row[2 * count + j]), out_w + s);
      __builtin_nontemporal_store(TCOL ? row[3 * count + j] : ct, out_t + s);
    } else {
      const uint64_t* idp = reinterpret_cast<const uint64_t*>(row + 2 * j);
      const u64x2 id2 = {idp[0], idp[1]};
      const f32x2 w2 = *reinterpret_cast<const f32x2*>(row + 2 * count + j);
      i32x2 t2 = {ct, ct};
      if (TCOL) t2 = *reinterpret_cast<const i32x2*>(row + 3 * count + j);
      __builtin_nontemporal_store(id2, reinterpret_cast<u64x2*>(out_id + s));
      __builtin_nontemporal_store(w2, reinterpret_cast<f32x2*>(out_w + s));
      __builtin_nontemporal_store(t2, reinterpret_cast<i32x2*>(out_t + s));
    }
    if (j == 0 && out_mask != nullptr) out_mask[i] = (uint8_t)masked;
    i += stride_rows;
    j += stride_slots;
    if (j >= count) { j -= count; ++i; }
  }
}
// ---- tuning switches (euler_gpu_set_tuning; declared in k1_args.h) ----
thread_local int g_k1_variant = 6;   // key 0: 6 = block pivots, 5 = pivot levels over the flat arrays,
                                     // 0 = the reference loop for every call
thread_local int g_k1_ablate = 0;    // key 2: measurement only (walk kernels)
thread_local int g_k1_grid_cap = -1; // key 3: workgroup cap of the K1 launches: -1 = by concurrency (see
                        // ConcurrentCall: 4096 = 16 waves per CU when the caller alternates streams, so
                        // that the kernels of another minibatch fit beside them; else kK1GridCap =
                        // 32 768), 0 = always 32 768, > 0 = that many
thread_local int g_k1_pair = 1;      // key 4: pivot kernel: two adjacent samples per lane when count is even
thread_local int g_k1_dedup = 1;     // key 5: 0 = never, 1 = automatic for >= 100 000 roots, 2 = always try
thread_local int g_n2v_wave = 2;     // key 7: node2vec: 3 = launched per step, long lists by a workgroup
                        // (n2v_kernels.h: N2vBigStepKernel), 2 = one launch, one wave per walker, the
                        // two-cursor walk by the whole wave, 1 = lane 0 walks LDS-staged lists, 0 = one
                        // lane per walker
thread_local int g_k1_fuse_mark = 1; // key 9: fanout: a hop's kernels fill the next hop's owner table
thread_local int g_expand_steps = 2;        // key 10: DedupExpandKernel: grid-stride steps in flight per lane (1, 2, 4)
thread_local int g_expand_const_type = 1;   // key 11: ... rebuild the type column of single-type calls from the mask
thread_local int g_expand_grid_cap = 0;     // key 12: ... workgroup cap (0 = kK1GridCap)
thread_local int g_k1_dual = 1;      // key 13: duplicate-root call: both gated passes in one launch
thread_local int g_dedup_block_numbering = 2;   // key 14: 2 = one pass (workgroups take numbers from the call's
                                   // counter), 1 = per-workgroup counts + one small scan, 0 = device-wide scan
thread_local int g_dedup_resolve_in_expand = 0;   // key 20: 1 = last hop: the expansion reads its row number from
                                     // the owner table itself (no resolve kernel, no uidx array) -
                                     // measured: dedup 43 -> 28 us, expansion 134 -> 157 us; off
thread_local int g_fanout_fused = 1;       // key 23: small 2-hop single-type fanouts as one launch, a workgroup per root
thread_local int g_full_nb_balanced = 1;   // key 24: get_full_neighbor fill: a lane owns 4 output entries
thread_local int g_n2v_big = 8192;   // key 25: child lists of this many entries go to the workgroup kernel (0 = none)
// 2-hop single-type fanouts as ONE kernel with the duplicate children found inside the wave
// (fanout_local.h).  key 27: 0 = off (hop by hop, global duplicate path), 1 = on for weighted
// graphs, 2 = on for every graph.
thread_local int g_fanout_local = 1;
thread_local int g_fl_roots = 0;      // key 28: roots per wave (1 .. 16), 0 = the launcher chooses (4, or 8 for a
                                      // caller that alternates streams on a graph with the weight-bucket index)
thread_local int g_fl_cap = 0;        // key 29: hop-2 slots per pass, 0 = the launcher chooses (64 with the
                                      // weight-bucket index, else 8 x roots per wave)
thread_local int g_fl_block = 0;      // key 30: threads per workgroup (64, 128, 256), 0 = the launcher chooses
                                      // (128 with the weight-bucket index, else 64)
thread_local int g_fl_wide = 1;       // key 31: weights / types as 16-byte stores
thread_local int g_fl_grid_cap = -1;  // key 32: waves of the launch: 0 = one tile per wave (no loop), -1 = that for a
                                      // caller on one stream and 16 384 looping waves for one that alternates
                                      // streams (two launches share the chip), > 0 = that many
thread_local int g_fl_min_roots = 32768;  // key 33: smaller batches keep the workgroup-per-root kernel / the hop-by-hop
                                          // path (tools/fl_crossover.py: 4 096 roots 51 vs 36 us, 16 384: 76 vs 63, 32 768:
                                          // 104 vs 102, 65 536: 173 vs 186)
thread_local int g_fl_plain = 2;      // key 34: plain graphs: 2 = the lean kernel, 1 = the general kernel
                                      // constant-folded, 0 = the general kernel
thread_local int g_fl_wps = 5;        // key 35: register budget, waves per SIMD (8, 6 or 5; 5: nothing spilled)
thread_local int g_fp_on = 1;         // key 53: plain graphs with the weight-bucket index take the kernel of
                                      // fanout_plain.h (1); 0 = the lean build of fanout_local.h (round 5)
thread_local int g_fp_coop = 0;       // key 54: ... a block's keys fetched by three lanes as ONE request per line and
                                      // staged in LDS (1); 0 = three 16-byte loads per lane and line (round 5's pattern)
thread_local int g_fp_lite2 = 0;      // key 57: ... hop 2 asks for two key chunks per draw and for the third only at
                                      // the ends of its block (1); 0 = all three
thread_local int g_fp_wps = 5;        // key 55: ... its register budget, waves per SIMD (4 .. 8)
thread_local int g_fl_wb = 1;         // key 45: the lean kernel draws through the weight-bucket index (wb_index.h:
                                      // one line per draw); 0 = the pivot-level search of rounds 2-3
thread_local int g_fl_typed_regs = 1;  // key 48: typed hops on graphs of <= 4 edge-type groups keep the row record in registers (1)
thread_local int g_k1_sets_lds = 1;   // key 47: euler_gpu_sample_neighbor_sets stages the roots' records in LDS (1)
thread_local int g_fl_fat = 1;        // key 49: hashed graphs of <= 2 edge-type groups find a root's record in its
                                      // 64-byte hash slot (1); 0 = 16-byte slot, then the record
thread_local int g_blk_policy = 0;    // key 51: 1 = a graph with a weight-bucket index is served by it ALONE even when many
                                      // of its buckets overflow (wb_lean_ok = 0): no EdgeBlocks, every missed draw bisects
                                      // the flat sums - tests of that fallback; 0 = such graphs get the EdgeBlocks' levels
thread_local int g_fl_ablate = 0;     // key 36: measurement only (FanoutLocalArgs::ablate)
thread_local int g_k1_typed_pivot = 1;   // key 37: calls with type draws search with the block pivots (0 = reference
                                         // loop)
thread_local uint32_t* t_fl_row_index = nullptr;   // set by euler_gpu_sample_fanout_unique around its call
thread_local int t_fl_took_lean = 0;                // ... and whether the lean kernel served it
thread_local const char* t_fl_last_kernel = "";      // euler_gpu_last_fanout_kernel: what the calling thread's last 2-hop fanout launched
thread_local void* g_fl_debug = nullptr;   // euler_gpu_set_debug_buffer: phase stamps of the lean kernel

int SamplingView(const euler_gpu_graph* g, GraphView* out) {
  const bool wb = g_fl_wb != 0 && g_k1_variant == 6;
  if (wb) {
    const int rc = EnsureWbIndex(g);
    if (rc != EULER_GPU_OK) return rc;
  }
  // EdgeBlocks + block pivots (13 bytes per edge), built on first use - and only for a caller the
  // weight-bucket index does not serve: no index (uniform weights, declined, key 45 = 0), or one
  // with so many overflowing buckets that its misses should walk the levels (wb_lean_ok = 0).
  // One copy of the adjacency per search structure actually used.
  const bool wb_serves = wb && g->view.wbg != nullptr && (g->view.wb_lean_ok != 0 || g_blk_policy == 1);
  if (g_k1_variant == 6 && !wb_serves && g->blk_ready.load(std::memory_order_acquire) == 0) {
    const int rc = EnsureBlockedIndex(g);
    if (rc != EULER_GPU_OK) return rc;
  }
  *out = g->view;
  if (!wb) { out->wb = nullptr; out->wbg = nullptr; out->wrec = nullptr; out->n_wb = 0; out->wb_lean_ok = 0;
             out->trec = nullptr; out->trec_stride = 0; out->fat = nullptr; }
  if (g_fl_fat == 0) out->fat = nullptr;
  return EULER_GPU_OK;
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
      // wire row of root r: ids (2 words each) | weights | [types] | mask, pad
      const int32_t cols = 3 + a.packed_tcol;
      int32_t* row = a.packed + r * (int64_t)PackedWords(a.count, a.packed_tcol);
#pragma unroll
      for (int x = 0; x < U; ++x) {
        *reinterpret_cast<uint64_t*>(row + 2 * (j + x)) = id[x];
        row[2 * a.count + j + x] = __float_as_int(w[x]);
        if (a.packed_tcol) row[3 * a.count + j + x] = ot;
      }
      if (j == 0) *reinterpret_cast<int2*>(row + cols * a.count) = make_int2(valid ? 0 : 1, 0);
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
__global__ __launch_bounds__(256, kWavesPerSimd) void SampleNeighborPivotKernel(
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
__global__ __launch_bounds__(256, kWavesPerSimd) void SampleNeighborPivotDualKernel(
    const SampleNbArgs a, const int64_t a_rows, const int32_t a_slots,
    const SampleNbArgs b, const int64_t b_rows, const int32_t b_slots) {
  if (DedupActive(a.dd_counter, a.dd_n_in)) {
    PivotPass<TF_LAYOUT, 1, BLOCKED>(b, (int64_t)(*a.dd_counter), b_rows, b_slots);
  } else {
    PivotPass<TF_LAYOUT, U, BLOCKED>(a, a.n, a_rows, a_slots);
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
// A caller that alternates streams between calls keeps several minibatches in
// flight (bench.py --streams, the reference's 8 query threads): the K1 launches of
// such a call take 16 waves per CU instead of all 32, which overlaps the latency-bound
// phases of one minibatch with the bandwidth-bound expansion of the other.  One stream:
// full grids.
thread_local int t_concurrent = -1;      // -1 = not inside a call
static bool ConcurrentCall(const euler_gpu_graph* g, hipStream_t stream) {
  void* prev = g->last_stream.exchange((void*)stream);
  return prev != nullptr && prev != (void*)stream;
}
static int64_t K1GridCap() {
  if (g_k1_grid_cap > 0) return g_k1_grid_cap;
  if (g_k1_grid_cap == 0) return kK1GridCap;
  return t_concurrent == 1 ? 4096 : kK1GridCap;
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
    const int64_t cap = K1GridCap();
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
      const int64_t cap = K1GridCap();
      if (blocks > cap) blocks = cap;
      gridp = (int)(blocks < 1 ? 1 : blocks);
    }
    const int64_t stride = (int64_t)gridp * block * U;
    const int64_t stride_rows = stride / count;
    const int32_t stride_slots = (int32_t)(stride - stride_rows * count);
    const bool tf = layout == EULER_GPU_LAYOUT_TF;
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
  } else {
    return LaunchK1Variant(g, stream, a, grid);
  }
  EG_HIP(hipGetLastError());
  return EULER_GPU_OK;
}

static size_t CounterOffset(const euler_gpu_graph* g) {
  return (((size_t)g->view.n_rows + 1) * 4 + 255) & ~(size_t)255;
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
    // fresh scratch: both counters of the one-pass numbering start at zero
    if (want >= CounterOffset(g) + 256) {
      EG_HIP(hipMemsetAsync((uint8_t*)slot.first + CounterOffset(g), 0, 256, stream));
    }
    g->ws_parity[(void*)stream] = 0;
  }
  *out = slot.first;
  return EULER_GPU_OK;
}

// Below ~100 K roots the six dependent kernels of the duplicate path (~38 us end to
// end, whatever the size) lose to sampling the given roots directly, even with 90 %
// duplicates (tools/ab_dedup_threshold.py on the metric graph: 25 600 roots 26 vs
// 38 us, 102 400: 42 vs 42, 204 800: 57 vs 48, 819 200: 157 vs 92).  A batch of DISTINCT
// roots loses those ~38 us at any size - 131 072 typed roots x 10: 58-61 us with the
// detection, ~40 without (profiles/r3_v3_hetero_kernel_stats.csv: mark 6 + number 14 +
// resolve 5 + two gated exits) - so the automatic policy starts at 200 K roots: a tie at
// ~60 % duplicates, and past the batch sizes of a first hop.
constexpr int64_t kDedupMinRoots = 200000;
// the workgroup-per-root fanout (SampleFanout2Kernel, key 23) against a launch per hop, metric
// graph, [25, 10]: 1 024 roots 14.7 vs 21.8 us, 2 048: 23.4 vs 26.0, 4 096: 36.6 vs 36.7,
// 6 144: 47.7 vs 45.4, 7 900: 60.5 vs 53.6
constexpr int64_t kFanout2MaxRoots = 4096;

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
  const int block = 256;
  const int U = pair ? 2 : 1;
  int64_t blocks = (a.n * (int64_t)count / U + block - 1) / block;
  const int64_t cap = K1GridCap();
  if (blocks > cap) blocks = cap;
  const int grid = (int)(blocks < 1 ? 1 : blocks);
  const int64_t a_stride = (int64_t)grid * block * U;
  const int64_t a_rows = a_stride / count;
  const int32_t a_slots = (int32_t)(a_stride - a_rows * count);
  // the pass over the distinct roots: one sample per lane (the shorter dependent chain
  // wins on cold rows: 0.139 vs 0.149 ms with two per lane on the metric workload)
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
  size_t scan_bytes, o_owner, o_slot, o_pos, o_uidx, o_uniq, o_cnt, o_scan, o_bcnt, o_boff,
      o_mask, o_tid, o_tw, o_tt, bytes;
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
  // owner table, then the stream's counters at an offset that depends on the graph
  // only (DedupNumberKernel of one call clears the counter of the next): words 0
  // and 32 = the pair of the one-pass numbering, word 16 = the other modes' counter
  L->o_owner = 0;
  L->o_cnt = CounterOffset(g);
  L->o_slot = L->o_cnt + 256;
  L->o_pos = L->o_slot + al((size_t)n * 4);
  L->o_uidx = L->o_pos + al(((size_t)n + 1) * 4);
  L->o_uniq = L->o_uidx + al((size_t)n * 4);
  L->o_scan = L->o_uniq + al((size_t)n * 8);
  const size_t n_blocks = ((size_t)n + 255) / 256 + 1;
  L->o_bcnt = L->o_scan + al(L->scan_bytes);
  L->o_boff = L->o_bcnt + al(n_blocks * 4);
  L->o_mask = L->o_boff + al(n_blocks * 4);
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

// will LaunchK1 pick a pivot kernel (the only ones that write wire rows)?  Single-type calls: the
// pivot kernels; calls that draw a type per sample: SampleNeighborTypedPivotKernel (k1_variants.hip:
// LaunchK1Variant takes it under exactly this condition) - the owners' pass of a typed sharded hop
// wrote dense arrays and packed them in a second kernel (19 us of a 36-40 us pass) before.
static bool K1WritesPacked(const euler_gpu_graph* g, int32_t k, int32_t layout) {
  const bool tf_zero = layout == EULER_GPU_LAYOUT_TF && g->view.has_zero_nbr != 0;
  if (!g->view.monotone || tf_zero) return false;
  if (k == 1) return g_k1_variant == 5 || g_k1_variant == 6;
  if (g_k1_variant != 6 || g_k1_typed_pivot == 0) return false;
  GraphView v;
  return SamplingView(g, &v) == EULER_GPU_OK && HasBlockSearch(v);
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
  struct ConcurrencyScope {      // the outermost call of this thread decides
    bool own;
    ConcurrencyScope(const euler_gpu_graph* g_, hipStream_t st_) : own(t_concurrent < 0) {
      if (own) t_concurrent = (g_ != nullptr && ConcurrentCall(g_, st_)) ? 1 : 0;
    }
    ~ConcurrencyScope() { if (own) t_concurrent = -1; }
  } concurrency_scope(g, stream);
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
  SampleNbArgs a{};
  {
    const int rc = SamplingView(g, &a.g);
    if (rc != EULER_GPU_OK) return rc;
  }
  a.seed = seed; a.call_id = call_id;
  a.roots = roots; a.root_mask = root_mask;
  a.root_group = root_group > 0 ? root_group : 1;
  a.out_id = out_id; a.out_w = out_w; a.out_t = out_t;
  a.out_row_mask = out_row_mask;
  a.n = n; a.default_node = default_node;
  a.k = k; a.count = count; a.layout = layout;
  a.packed = packed_out;
  a.packed_tcol = k == 1 ? 0 : 1;
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
  d.counter = (uint32_t*)(ws + L.o_cnt) + 16;
  g_last_unique_offset = (int64_t)L.o_cnt + 64;
  const IdentityMap idmap{g->view.id_base, g->view.id_stride, g->view.n_rows};
  const int block = 256;
  const int dgrid = GridFor(n + 1, block);
  PhaseMark(stream, 0);
  hipcub::CountingInputIterator<uint32_t> pos_it(0u);
  bool resolve_in_expand = false;
  if (g_dedup_block_numbering == 2) {
    // one pass: workgroups take their numbers from the call's counter; the stream
    // owns a PAIR of counters - this call's was cleared by the previous call's
    // numbering kernel (or at allocation), and clears the other one
    uint32_t* pair = (uint32_t*)(ws + L.o_cnt);
    int* parity = nullptr;
    {
      std::lock_guard<std::mutex> lk(g->ws_mu);
      parity = &g->ws_parity[(void*)stream];
    }
    d.counter = pair + (*parity ? 32 : 0);
    uint32_t* next_counter = pair + (*parity ? 0 : 32);
    *parity ^= 1;
    g_last_unique_offset = (int64_t)((uint8_t*)d.counter - ws);
    if (!premarked) hipLaunchKernelGGL(DedupMarkKernel, dim3(dgrid), dim3(block), 0, stream, d);
    DedupBlockArgs ba{};
    ba.d = d; ba.map = idmap; ba.premarked = premarked ? 1 : 0;
    const int64_t nb = (n + kNumberTile - 1) / kNumberTile;
    hipLaunchKernelGGL(DedupNumberKernel, dim3((unsigned)nb), dim3(kNumberThreads), 0, stream, ba,
                       next_counter);
    resolve_in_expand = g_dedup_resolve_in_expand != 0 && !do_mark;
    if (!resolve_in_expand)
      hipLaunchKernelGGL(DedupResolveNumberedKernel, dim3(dgrid), dim3(block), 0, stream, ba);
  } else if (g_dedup_block_numbering != 0) {
    if (!premarked) hipLaunchKernelGGL(DedupMarkKernel, dim3(dgrid), dim3(block), 0, stream, d);
    DedupBlockArgs ba{};
    ba.d = d; ba.map = idmap; ba.premarked = premarked ? 1 : 0;
    ba.rep = d.pos;                      // the scan's output array is free in this mode
    // scratch-row ids: >= 4 n bytes, idle until the sampling pass (after Resolve)
    ba.posrep = (uint32_t*)(ws + L.o_tid);
    ba.bcnt = (uint32_t*)(ws + L.o_bcnt);
    ba.boff = (uint32_t*)(ws + L.o_boff);
    const int64_t nb = (n + kDedupBlock - 1) / kDedupBlock;
    hipLaunchKernelGGL(DedupRepCountKernel, dim3((unsigned)nb), dim3(kDedupBlock), 0, stream, ba);
    hipLaunchKernelGGL(DedupBlockScanKernel, dim3(1), dim3(1024), 0, stream, ba.bcnt, nb,
                       ba.boff, d.counter);
    hipLaunchKernelGGL(DedupAssignKernel, dim3((unsigned)nb), dim3(kDedupBlock), 0, stream, ba);
    // (letting the expand look posrep[rep[i]] up itself saves this kernel's 8 us
    // and costs the expand 10: measured, not kept)
    hipLaunchKernelGGL(DedupResolveKernel, dim3(dgrid), dim3(block), 0, stream, ba);
  } else if (premarked) {
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
  if (resolve_in_expand) {
    x.resolve_owner = d.owner;
    x.resolve_row_slot = premarked ? nullptr : d.row_slot;
    x.roots = roots; x.root_mask = root_mask; x.root_group = a.root_group;
  }
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
  const bool lean = pair && !do_mark && !resolve_in_expand &&
                    count / 2 <= 128 && (int64_t)n * count < ((int64_t)1 << 31);
  if (lean) {
    const uint32_t P = (uint32_t)count / 2;
    const uint32_t rows_per_block = 256u / P;
    int64_t lb = ((int64_t)n + rows_per_block - 1) / rows_per_block;
    if (lb > xcap) lb = xcap;
    if (lb < 1) lb = 1;
    auto kern = ct ? DedupExpandLeanKernel<true> : DedupExpandLeanKernel<false>;
    hipLaunchKernelGGL(kern, dim3((unsigned)lb), dim3(256), 0, stream, x.counter, x.uidx_of, x.t_id,
                       x.t_w, x.t_t, x.t_mask, out_id, out_w, out_t, out_row_mask, (uint32_t)n,
                       (uint32_t)count, P, 65536u / P + 1u, rows_per_block, x.type0,
                       x.masked_type);
  } else {
    LaunchExpand(pair ? 2 : 1, g_expand_steps, ct, (int)blocks, block, stream, x,
                 stride_rows, stride_slots);
  }
  PhaseMark(stream, 3);
  EG_HIP(hipGetLastError());
  if (hop != nullptr) hop->mark_next = do_mark;
  return EULER_GPU_OK;
}

// ------------------------------------------------------------------------
// Small 2-hop fanouts in ONE launch (tuning key 23).  The latency batch of SURVEY 8 (the reference's
// examples use less) is 1 024 roots: hop 1 is 25 600 samples, hop 2 256 000 - too few for
// the duplicate-root path, so the step was two latency-bound launches and a
// boundary, 26 us.  Hop 2's roots of batch root r are r's own hop-1 samples, so a
// workgroup that owns root r needs nothing from any other workgroup: it draws
// r's c1 samples (lanes 0 .. c1-1), keeps them in LDS, and its 256 lanes then
// draw the c1 * c2 second-hop samples - one launch, no global round trip between
// the hops.  Same draws as the chained kernels: hop h uses call_id + h, a hop-1
// row without samples hands node id 0 to hop 2 (sample_fanout_op.cc:37-42 over
// the core tensors' sentinel).  Single listed type per hop, block-pivot graphs.
// ------------------------------------------------------------------------
struct Fanout2Args {
  GraphView g;
  uint64_t seed;
  uint32_t call_id;
  const uint64_t* roots;
  int64_t n;
  int64_t default_node;
  int32_t c1, c2, t1, t2;
  uint64_t* id1; float* w1; int32_t* ty1; uint8_t* mask0;
  uint64_t* id2; float* w2; int32_t* ty2; uint8_t* mask1;
  // several minibatches in one launch (FanoutLocalArgs has the same three fields)
  int64_t mb_n;
  const uint32_t* call_ids;
  uint32_t call_stride;
};

__global__ __launch_bounds__(256) void SampleFanout2Kernel(const Fanout2Args a) {
  __shared__ uint64_t s_id[256];
  __shared__ int32_t s_valid;
  for (int64_t r = blockIdx.x; r < a.n; r += gridDim.x) {
    uint32_t call0 = a.call_id;
    if (a.mb_n > 0) {
      const uint32_t b = (uint32_t)((uint64_t)r / (uint64_t)a.mb_n);
      call0 = a.call_ids != nullptr ? a.call_ids[b] : a.call_id + b * a.call_stride;
    }
    const int32_t j = threadIdx.x;
    if (j < a.c1) {
      const uint64_t node = a.roots[r];
      Segment sg;
      const bool valid = LoadSegment<true>(a.g, FindRow(a.g, node), a.t1, &sg);
      uint64_t id = (uint64_t)a.default_node;
      float w = 0.f;
      if (valid) {
        const Philox4 pb = RngBlock(a.seed, call0, kDomainNeighbor, node, ((uint32_t)j) >> 1);
        const double u = (j & 1) ? UnitFromWords(pb.w[2], pb.w[3]) : UnitFromWords(pb.w[0], pb.w[1]);
        BlockPivotSample(a.g, sg, u, &id, &w);
      }
      const int64_t d = r * a.c1 + j;
      a.id1[d] = id;
      a.w1[d] = w;
      a.ty1[d] = valid ? a.t1 : -1;
      s_id[j] = valid ? id : 0;             // a missing row samples as node id 0 downstream
      if (j == 0) { s_valid = valid ? 1 : 0; a.mask0[r] = valid ? 0 : 1; }
    }
    __syncthreads();
    const int32_t tasks = a.c1 * a.c2;
    for (int32_t tk = threadIdx.x; tk < tasks; tk += 256) {
      const int32_t q = tk / a.c2;
      const int32_t x = tk - q * a.c2;
      const uint64_t node = s_id[q];
      Segment sg;
      const bool valid = LoadSegment<true>(a.g, FindRow(a.g, node), a.t2, &sg);
      uint64_t id = (uint64_t)a.default_node;
      float w = 0.f;
      if (valid) {
        const Philox4 pb = RngBlock(a.seed, call0 + 1u, kDomainNeighbor, node, ((uint32_t)x) >> 1);
        const double u = (x & 1) ? UnitFromWords(pb.w[2], pb.w[3]) : UnitFromWords(pb.w[0], pb.w[1]);
        BlockPivotSample(a.g, sg, u, &id, &w);
      }
      const int64_t row = r * a.c1 + q;
      const int64_t d = row * a.c2 + x;
      a.id2[d] = id;
      a.w2[d] = w;
      a.ty2[d] = valid ? a.t2 : -1;
      if (x == 0) a.mask1[row] = valid ? 0 : 1;
    }
    __syncthreads();            // s_id is rewritten by the next root
  }
}

// TF-layout sampling of the first *n_dev roots of a worst-case-sized list (the
// count lives on the device: block construction chains hops without telling the
// host how many distinct nodes a hop produced).  The pass-2 gate of the
// duplicate-root path does exactly this: a launch sized for `cap` roots whose
// lanes beyond the device-side count exit.
int LaunchSampleNeighborCounted(const euler_gpu_graph* g, hipStream_t stream, uint64_t seed,
                                uint32_t call_id, const uint64_t* roots, int64_t cap,
                                const uint32_t* n_dev, const int32_t* edge_types, int32_t k,
                                int32_t count, int64_t default_node, uint64_t* out_id,
                                float* out_w, int32_t* out_t) {
  if (g == nullptr) return Fail(EULER_GPU_ENOGRAPH, "sample_neighbor: null graph");
  if (cap < 0 || count <= 0 || k < 0 || k > kMaxListedTypes)
    return Fail(EULER_GPU_EINVAL, "sample_neighbor (counted): bad cap/count/k");
  if (cap == 0) return EULER_GPU_OK;
  if (!(g_k1_variant == 5 || g_k1_variant == 6 || g_k1_variant == 0))
    return Fail(EULER_GPU_EINVAL, "sample_neighbor (counted): needs the default kernels");
  SampleNbArgs a{};
  {
    const int rc = SamplingView(g, &a.g);
    if (rc != EULER_GPU_OK) return rc;
  }
  a.seed = seed; a.call_id = call_id;
  a.roots = roots; a.root_mask = nullptr; a.root_group = 1;
  a.out_id = out_id; a.out_w = out_w; a.out_t = out_t; a.out_row_mask = nullptr;
  a.n = cap; a.default_node = default_node;
  a.k = k; a.count = count; a.layout = EULER_GPU_LAYOUT_TF;
  a.cold_roots = 1;
  a.dd_role = 2;
  a.dd_counter = n_dev;
  a.dd_n_in = (int64_t)1 << 60;          // the gate's "duplicates pay" test always holds
  for (int i = 0; i < k; ++i) a.et[i] = edge_types[i];
  return LaunchK1(g, stream, a);
}

}  // namespace euler_gpu

using namespace euler_gpu;

// TF dense repack of a hop sampled in the CORE layout (tf_euler/kernels/
// sample_fanout_op.cc:124-137, sample_neighbor_op.cc:110-122): a row whose FIRST id is the
// sentinel 0 - a row without samples, or a row with samples that happens to start with
// node id 0 (Q1) - becomes default_node / 0.0 / -1.  Only graphs that contain the id 0
// as a neighbour take this path: their fanouts are chained on the core ids, as the
// reference's single GQL does, and repacked afterwards.
__global__ __launch_bounds__(256) void TfRepackKernel(uint64_t* id, float* w, int32_t* t,
                                                      int64_t n_rows, int32_t count,
                                                      int64_t default_node, uint8_t* row_mask) {
  const int64_t total = n_rows * (int64_t)count;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < total; s += stride) {
    const int64_t r = s / count;
    const bool drop = id[r * count] == 0;
    // the row's first id is what every lane of the row tests, so it is not rewritten
    // here (no ordering between the lanes of a row): TfRepackFirstKernel does it
    if (drop && s != r * count) { id[s] = (uint64_t)default_node; w[s] = 0.f; t[s] = -1; }
    if (s == r * count && row_mask != nullptr) row_mask[r] = drop ? 1 : 0;
  }
}
__global__ __launch_bounds__(256) void TfRepackFirstKernel(uint64_t* id, float* w, int32_t* t,
                                                           int64_t n_rows, int32_t count,
                                                           int64_t default_node) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_rows; r += stride) {
    const int64_t s = r * count;
    if (id[s] == 0) { id[s] = (uint64_t)default_node; w[s] = 0.f; t[s] = -1; }
  }
}

// M minibatches in one launch (euler_gpu_sample_fanout_multi): `n` of RunFanout is then the
// total number of roots, minibatch b = root / n_per.
struct FanoutMulti {
  int64_t n_per;
  int32_t m;
  uint32_t call_stride;
  const uint32_t* call_ids_dev;
};

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
                     int64_t* uniq_off, const FanoutMulti* multi = nullptr) {
  struct FanoutConcurrency {
    bool own;
    FanoutConcurrency(const euler_gpu_graph* g_, hipStream_t st_) : own(t_concurrent < 0) {
      if (own) t_concurrent = ConcurrentCall(g_, st_) ? 1 : 0;
    }
    ~FanoutConcurrency() { if (own) t_concurrent = -1; }
  } fanout_concurrency(g, stream);
  // a 2-hop fanout of single listed types: ONE kernel, duplicates found inside the wave
  // (hops that list several edge types - a type draw per sample - take the same kernel on the
  // graphs the weight-bucket index serves: fanout_local.h, WB == 3)
  const bool typed_hops = k > 1 && k <= kMaxListedTypes;
  // (graphs of uniform weights: the register form only - at most 4 type groups)
  const bool typed_ok = typed_hops && g_fl_plain == 2 && g_fl_wb != 0 && g_k1_typed_pivot != 0 &&
                        g->view.T > 1 && g->view.T <= 127 && g->view.has_zero_nbr == 0 &&
                        (g->view.uniform_w == 0 || g->view.T <= 4) && g->view.n_edges < ((int64_t)1 << 31) &&
                        t_fl_row_index == nullptr && counts_host[1] % 2 == 0;
  // graphs of uniform weights that are not "plain" (several type groups, hashed ids - the shape of
  // the reference's datasets), one listed type per hop: the lean build with the index
  // computation as its draw (WB == 6), or hop by hop as before
  const bool uni_general = k == 1 && g->view.uniform_w != 0 && (g->view.T > 1 || g->view.map_mode != 0) &&
                           g_fanout_local == 1 && g_fl_plain == 2 && g_fl_wb != 0 && g->view.has_zero_nbr == 0 &&
                           g->view.n_edges < ((int64_t)1 << 31) && t_fl_row_index == nullptr &&
                           counts_host[1] % 2 == 0;
  const bool lean_only = typed_hops || uni_general;       // no general-build fallback inside
  // (plain graphs of uniform weights - products-shaped - take the lean build too since it is built
  // for 5 waves per SIMD: 156 G edges/s against 144 hop by hop and 138 with the spilling 8-wave build)
  const bool uni_plain = k == 1 && g->view.uniform_w != 0 && g->view.T == 1 && g->view.map_mode == 0 &&
                         g->view.total_in_meta != 0 && g->view.has_zero_nbr == 0 && g_fl_plain == 2 &&
                         counts_host[1] % 2 == 0 && g->view.n_edges < ((int64_t)1 << 31);
  if (g_fanout_local != 0 &&
      (g->view.uniform_w == 0 || g_fanout_local == 2 || typed_ok || uni_general || uni_plain) &&
      events == nullptr &&
      layers == 2 && (k == 1 || typed_ok) &&
      n >= (multi != nullptr && g_fl_min_roots > 8192 ? 8192 : g_fl_min_roots) &&
      g_k1_variant == 6 && g->view.monotone && counts_host[0] > 0 && counts_host[1] > 0) {
    const int32_t c1 = counts_host[0], c2 = counts_host[1];
    // Geometry.  Keys 28 / 29 / 30 = 0 (the defaults) let the launcher choose: with the
    // weight-bucket index (plain graphs) a launch that has the chip to itself runs best as
    // 4 roots per wave, 64 slots per pass, 2 waves per workgroup (0.219-0.221 ms against
    // 0.218-0.224 as 4 / 32 / 1 and 0.235-0.239 as 8 / 64 / 2); a caller that alternates
    // streams - launches share the chip - as 8 roots per wave, 48 slots per pass, 1 wave per
    // workgroup (profiles/r4_ab_wb_geom*.txt).  Everything else keeps round 3's 4 / 32 / 1.
    const GraphView& gv = g->view;
    const bool wb_plain = g_fl_wb != 0 && g_fl_plain == 2 && gv.has_zero_nbr == 0 && gv.uniform_w == 0 &&
                          gv.monotone != 0;        // (the lean builds over the weight-bucket index)
    int32_t gr = g_fl_roots > 16 ? 16 : g_fl_roots;
    if (gr < 1) gr = wb_plain && t_concurrent == 1 ? 8 : 4;
    while (gr > 1 && (int64_t)gr * c1 > 0x7FFF) gr >>= 1;
    // several minibatches: a tile (gr roots) must not straddle two of them
    while (multi != nullptr && gr > 1 && multi->n_per % gr != 0) gr >>= 1;
    const bool wb_conc = wb_plain && t_concurrent == 1;
    int32_t cap = g_fl_cap > 0 ? g_fl_cap : wb_conc ? 48 : wb_plain ? 64 : 8 * gr;
    if (cap > gr * c1) cap = gr * c1;
    // (three callers' streams, sustained over 3 000 steps on two boxes: 8 / 48 / 1 wave per
    // workgroup 0.1975-0.2008 ms per step, 8 / 64 / 2 waves 0.2079-0.2094)
    const int block = (g_fl_block == 64 || g_fl_block == 128 || g_fl_block == 256) ? g_fl_block
                      : (wb_plain && !wb_conc) ? 128 : 64;
    FanoutLocalLds lay = FanoutLocalLayout(gr, c1, c2, cap);
    // shrink the pass until a workgroup's LDS fits
    while (cap > 1 && (size_t)lay.bytes * (block / 64) > 64 * 1024) {
      cap >>= 1;
      lay = FanoutLocalLayout(gr, c1, c2, cap);
    }
    const uint64_t tile_pos = (uint64_t)gr * c1 * c2;
    const bool fits = (size_t)lay.bytes * (block / 64) <= 64 * 1024 && (int64_t)gr * c1 <= 0x7FFF &&
                      tile_pos * (uint64_t)(c1 > c2 ? c1 : c2) < 0xFFFFFFFFull;
    if (fits) {
      FanoutLocalArgs f{};
      {
        const int rc0 = SamplingView(g, &f.g);
        if (rc0 != EULER_GPU_OK) return rc0;
      }
      f.seed = seed; f.call_id = call_id; f.roots = roots_dev; f.n = n;
      f.default_node = default_node;
      f.c1 = c1; f.c2 = c2;
      f.t1 = edge_types_host[0]; f.t2 = edge_types_host[typed_hops ? k : 1];
      if (typed_hops) {
        f.k = k; f.type_mode = TypeModeOf(k, g->view.T);
        for (int32_t i = 0; i < k; ++i) { f.et1[i] = edge_types_host[i]; f.et2[i] = edge_types_host[k + i]; }
      }
      f.gr = gr; f.cap = cap; f.wave_lds = (int32_t)lay.bytes;
      if (multi != nullptr) {
        f.mb_n = multi->n_per; f.call_ids = multi->call_ids_dev; f.call_stride = multi->call_stride;
      }
      f.div_c1.Set((uint32_t)c1); f.div_c2.Set((uint32_t)c2);
      uint8_t* wsb = (uint8_t*)workspace_dev;
      f.id1 = out_id_dev[0]; f.w1 = out_w_dev[0]; f.ty1 = out_t_dev[0]; f.mask0 = wsb;
      f.id2 = out_id_dev[1]; f.w2 = out_w_dev[1]; f.ty2 = out_t_dev[1];
      f.mask1 = wsb + (((size_t)n + 15) & ~(size_t)15);
      f.vec = (c2 % 2 == 0 && (uintptr_t)f.id2 % 16 == 0 && (uintptr_t)f.w2 % 16 == 0 &&
               (uintptr_t)f.ty2 % 16 == 0) ? 1 : 0;
      f.wide = (f.vec && g_fl_wide != 0 && tile_pos % 4 == 0) ? 1 : 0;
      const int64_t tiles = (n + gr - 1) / gr;
      const int wpb = block / 64;
      int64_t blocks = (tiles + wpb - 1) / wpb;
      const int64_t wave_cap = g_fl_grid_cap > 0 ? g_fl_grid_cap
                               : (g_fl_grid_cap < 0 && t_concurrent == 1) ? 16384 : 0;
      if (wave_cap > 0 && blocks > (wave_cap + wpb - 1) / wpb) blocks = (wave_cap + wpb - 1) / wpb;
      const size_t lds = (size_t)lay.bytes * wpb;
      const GraphView& v = f.g;
      const bool plain_u = g_fl_plain != 0 && v.T == 1 && v.total_in_meta != 0 &&
                           v.map_mode == 0 && v.has_zero_nbr == 0 && f.t1 == 0 && f.t2 == 0;
      const bool plain = plain_u && v.uniform_w == 0;
      // the lean build's general form: any graph the weight-bucket index serves (several
      // edge-type groups, hashed ids), valid listed types, no neighbour id 0
      const bool lean_tu = typed_hops && v.uniform_w != 0 && v.trec != nullptr && v.T <= 4;
      const bool lean_t = lean_tu || (typed_hops && v.uniform_w == 0 && v.trec != nullptr && v.wb != nullptr &&
                                      v.wb_lean_ok != 0);
      const bool lean_g = !typed_hops &&
                          !plain_u && g_fl_plain == 2 && g_fl_wb != 0 && v.wbg != nullptr && v.wb != nullptr &&
                          v.wb_lean_ok != 0 &&
                          v.has_zero_nbr == 0 && v.uniform_w == 0 && f.t1 >= 0 && f.t1 < v.T && f.t2 >= 0 &&
                          f.t2 < v.T && t_fl_row_index == nullptr;
      const bool lean_gu = !typed_hops && !plain_u && g_fl_plain == 2 && g_fl_wb != 0 && v.trec != nullptr &&
                           v.uniform_w != 0 && v.has_zero_nbr == 0 && f.t1 >= 0 && f.t1 < v.T && f.t2 >= 0 &&
                           f.t2 < v.T && t_fl_row_index == nullptr;
      // (the plain build without the index walks the EdgeBlocks' levels: SamplingView builds them
      // for every graph the index does not serve)
      const bool lean_search_ok = lean_g || lean_gu || lean_t || v.uniform_w != 0 || f.g.blk != nullptr ||
                                  (f.g.wrec != nullptr && f.g.wb != nullptr && v.wb_lean_ok != 0);
      // graphs of 2^31 .. 2^32 edges: the kernel of fanout_plain.h alone (unsigned 32-bit edge
      // numbers throughout); the lean / local builds of fanout_local.h keep their 2^31
      const bool big = v.n_edges >= ((int64_t)1 << 31);
      if (((plain_u && !typed_hops) || lean_g || lean_gu || lean_t) && g_fl_plain == 2 && f.vec &&
          lean_search_ok && (!big || (plain && !typed_hops))) {
        // the lean build (pairs of samples per lane, f32 compares, duplicates by edge)
        int32_t lcap = cap;
        FanoutLeanLds ll = FanoutLeanLayout(gr, c1, c2, lcap, lean_t);
        while (lcap > 1 && (size_t)ll.bytes * (block / 64) > 64 * 1024) {
          lcap >>= 1;
          ll = FanoutLeanLayout(gr, c1, c2, lcap, lean_t);
        }
        // plain graph + weight-bucket index (the metric's shape): the kernel of fanout_plain.h
        const bool use_wb0 = v.uniform_w == 0 && f.g.wrec != nullptr && f.g.wb != nullptr && v.wb_lean_ok != 0;
        // (a caller that alternates streams keeps round 5's build: its 8-roots-per-wave geometry is the faster one
        //  when launches share the chip - 0.199 against 0.205-0.212 ms per step, profiles/r6_sweep*.txt)
        if (g_fp_on != 0 && (t_concurrent != 1 || g_fp_on == 2 || big) && plain && !typed_hops && use_wb0 && f.wide) {
          int32_t pgr = g_fl_roots > 0 ? gr : 4;
          while (multi != nullptr && pgr > 1 && multi->n_per % pgr != 0) pgr >>= 1;
          int32_t pcap = g_fl_cap > 0 ? g_fl_cap : (g_fp_coop != 0 ? 64 / (c2 / 2 > 0 ? c2 / 2 : 1) : 32);
          if (pcap > pgr * c1) pcap = pgr * c1;
          const int pblock = (g_fl_block == 64 || g_fl_block == 128 || g_fl_block == 256) ? g_fl_block : 128;
          const bool coop = g_fp_coop != 0;
          const FanoutPlainLds pl = FanoutPlainLayout(pgr, c1, c2, pcap, coop);
          const int64_t tp = (int64_t)pgr * c1 * c2;
          if (pgr * ((c1 + 1) / 2) <= 64 && pgr * c1 <= 255 && pgr * c1 % 4 == 0 && tp % 4 == 0 && c2 <= 64 &&
              c1 <= 128 && (int64_t)pcap * c2 < 4096 && (size_t)pl.bytes * (pblock / 64) <= 64 * 1024) {
            FanoutPlainArgs pa{};
            pa.wrec = f.g.wrec; pa.wb = f.g.wb; pa.prefix_w = f.g.prefix_w; pa.nbr = f.g.nbr;
            pa.roots = roots_dev;
            pa.id1 = f.id1; pa.w1 = f.w1; pa.ty1 = f.ty1; pa.id2 = f.id2; pa.w2 = f.w2; pa.ty2 = f.ty2;
            pa.row_index = t_fl_row_index;
            pa.call_ids = f.call_ids; pa.seed = seed; pa.id_base = f.g.id_base; pa.id_stride = f.g.id_stride;
            pa.n = n; pa.n_rows = f.g.n_rows; pa.default_node = default_node; pa.mb_n = f.mb_n;
            pa.call_id = call_id; pa.call_stride = f.call_stride;
            pa.c1 = c1; pa.c2 = c2; pa.gr = pgr; pa.cap = pcap; pa.wave_lds = (int32_t)pl.bytes;
            const int64_t ptiles = (n + pgr - 1) / pgr;
            const int pwpb = pblock / 64;
            int64_t pblocks = (ptiles + pwpb - 1) / pwpb;
            int64_t pwaves = g_fl_grid_cap > 0 ? g_fl_grid_cap : 0;
            if (pwaves > 0 && pblocks > (pwaves + pwpb - 1) / pwpb) pblocks = (pwaves + pwpb - 1) / pwpb;
            void (*pk)(const FanoutPlainArgs) = nullptr;
#define EG_FP(W) (coop ? SampleFanoutPlainKernel<W, true, false> : g_fp_lite2 != 0 ? SampleFanoutPlainKernel<W, false, true> \
                       : SampleFanoutPlainKernel<W, false, false>)
            pk = g_fp_wps >= 8 ? EG_FP(8) : g_fp_wps == 7 ? EG_FP(7) : g_fp_wps == 6 ? EG_FP(6)
                 : g_fp_wps == 5 ? EG_FP(5) : EG_FP(4);
#undef EG_FP
            t_fl_took_lean = 1;
            t_fl_last_kernel = "SampleFanoutPlainKernel";
            hipLaunchKernelGGL(pk, dim3((unsigned)pblocks), dim3(pblock), (size_t)pl.bytes * pwpb, stream, pa);
            EG_HIP(hipGetLastError());
            if (uniq_off != nullptr) { uniq_off[0] = -1; uniq_off[1] = -1; }
            return EULER_GPU_OK;
          }
        }
        if (!big && (size_t)ll.bytes * (block / 64) <= 64 * 1024) {
          f.cap = lcap; f.wave_lds = (int32_t)ll.bytes;
          f.div_h1.Set((uint32_t)(c1 + 1) / 2); f.div_h2.Set((uint32_t)c2 / 2);
          const size_t llds = (size_t)ll.bytes * wpb;
#ifdef EULER_GPU_MEASURE
          f.dbg = (unsigned long long*)g_fl_debug;
          f.ablate = g_fl_ablate;
#endif
          f.row_index = t_fl_row_index;
          t_fl_took_lean = 1;
          void (*lk)(const FanoutLocalArgs) = nullptr;
          const bool use_wb = v.uniform_w == 0 && f.g.wrec != nullptr && f.g.wb != nullptr && v.wb_lean_ok != 0;
          if (lean_gu) {
            lk = f.wide ? SampleFanoutLeanKernel<true, 5, false, 6> : SampleFanoutLeanKernel<false, 5, false, 6>;
          } else if (lean_tu) {                                  // ... and the draw an index computation
            lk = f.wide ? SampleFanoutLeanKernel<true, 5, false, 5> : SampleFanoutLeanKernel<false, 5, false, 5>;
          } else if (lean_t && v.T <= 4 && g_fl_typed_regs != 0) {      // the row record in registers
            lk = f.wide ? SampleFanoutLeanKernel<true, 5, false, 4> : SampleFanoutLeanKernel<false, 5, false, 4>;
          } else if (lean_t) {
            lk = f.wide ? SampleFanoutLeanKernel<true, 5, false, 3> : SampleFanoutLeanKernel<false, 5, false, 3>;
          } else if (lean_g) {
            lk = f.wide ? (g_fl_wps == 6 ? SampleFanoutLeanKernel<true, 6, false, 2> : SampleFanoutLeanKernel<true, 5, false, 2>)
                        : (g_fl_wps == 6 ? SampleFanoutLeanKernel<false, 6, false, 2> : SampleFanoutLeanKernel<false, 5, false, 2>);
          } else if (v.uniform_w != 0) {
            lk = g_fl_wps == 8 ? (f.wide ? SampleFanoutLeanKernel<true, 8, true> : SampleFanoutLeanKernel<false, 8, true>)
                               : (f.wide ? SampleFanoutLeanKernel<true, 5, true> : SampleFanoutLeanKernel<false, 5, true>);
          } else if (use_wb) {
            lk = f.wide ? (g_fl_wps == 8 ? SampleFanoutLeanKernel<true, 8, false, 1>
                                         : g_fl_wps == 5 ? SampleFanoutLeanKernel<true, 5, false, 1>
                                                         : SampleFanoutLeanKernel<true, 6, false, 1>)
                        : (g_fl_wps == 8 ? SampleFanoutLeanKernel<false, 8, false, 1>
                                         : g_fl_wps == 5 ? SampleFanoutLeanKernel<false, 5, false, 1>
                                                         : SampleFanoutLeanKernel<false, 6, false, 1>);
          } else {
            lk = f.wide ? (g_fl_wps == 8 ? SampleFanoutLeanKernel<true, 8>
                                         : g_fl_wps == 5 ? SampleFanoutLeanKernel<true, 5>
                                                         : SampleFanoutLeanKernel<true, 6>)
                        : (g_fl_wps == 8 ? SampleFanoutLeanKernel<false, 8>
                                         : g_fl_wps == 5 ? SampleFanoutLeanKernel<false, 5>
                                                         : SampleFanoutLeanKernel<false, 6>);
          }
          t_fl_last_kernel = "SampleFanoutLeanKernel";
          hipLaunchKernelGGL(lk, dim3((unsigned)blocks), dim3(block), llds, stream, f);
          EG_HIP(hipGetLastError());
          if (uniq_off != nullptr) { uniq_off[0] = -1; uniq_off[1] = -1; }
          return EULER_GPU_OK;
        }
      }
      if (!lean_only && !big) {  // (typed hops / uniform general graphs the lean build does not take go hop by hop, below)
      void (*kern)(const FanoutLocalArgs) = nullptr;
#define EG_FL(W, P) (g_fl_wps != 8 ? SampleFanoutLocalKernel<W, P, 5> : SampleFanoutLocalKernel<W, P, 8>)
      kern = f.wide ? (plain ? EG_FL(true, true) : EG_FL(true, false))
                    : (plain ? EG_FL(false, true) : EG_FL(false, false));
#undef EG_FL
      t_fl_last_kernel = "SampleFanoutLocalKernel";
      hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(block), lds, stream, f);
      EG_HIP(hipGetLastError());
      if (uniq_off != nullptr) { uniq_off[0] = -1; uniq_off[1] = -1; }
      return EULER_GPU_OK;
      }
    }
  }
  t_fl_last_kernel = "hop by hop";
  std::lock_guard<std::recursive_mutex> launch_lk(g->launch_mu);
  const bool fanout2_ok = g_fanout_fused != 0 && events == nullptr && layers == 2 && k == 1 && n > 0 &&
      g_k1_variant == 6 && g->view.monotone && g->view.has_zero_nbr == 0 &&
      counts_host[0] > 0 && counts_host[0] <= 256 && counts_host[1] > 0;
  if (multi != nullptr && !fanout2_ok) {
    // shapes neither one-launch kernel serves: the minibatches one after the other
    int64_t per[8];
    int64_t mrows = multi->n_per;
    for (int32_t h = 0; h < layers && h < 8; ++h) { mrows *= counts_host[h]; per[h] = mrows; }
    if (layers > 8) return Fail(EULER_GPU_EINVAL, "sample_fanout_multi: more than 8 layers");
    if (multi->call_ids_dev != nullptr)
      return Fail(EULER_GPU_EINVAL, "sample_fanout_multi: device call ids need a 2-hop fanout of "
                                    "single listed types (pass NULL: call_id + b * stride)");
    for (int32_t b = 0; b < multi->m; ++b) {
      uint64_t* oi[8]; float* ow[8]; int32_t* ot[8];
      for (int32_t h = 0; h < layers; ++h) {
        oi[h] = out_id_dev[h] + (size_t)b * per[h];
        ow[h] = out_w_dev[h] + (size_t)b * per[h];
        ot[h] = out_t_dev[h] + (size_t)b * per[h];
      }
      const int rc = RunFanout(g, stream, seed, call_id + (uint32_t)b * multi->call_stride,
                               roots_dev + (size_t)b * multi->n_per, multi->n_per, edge_types_host, k,
                               counts_host, layers, default_node, oi, ow, ot, workspace_dev, nullptr,
                               nullptr, nullptr);
      if (rc != EULER_GPU_OK) return rc;
    }
    return EULER_GPU_OK;
  }
  // size the stream's scratch for the largest hop up front: the owner table a
  // hop fills for its successor must not move in between
  if (multi == nullptr) {
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
  // a small 2-hop fanout of single listed types: one launch (SampleFanout2Kernel)
  if (fanout2_ok && (multi != nullptr || (n <= kFanout2MaxRoots &&
      !WantsDedup(g, n * counts_host[0], 1) && !WantsDedup(g, n, 1)))) {
    Fanout2Args f{};
    {
      const int rc0 = SamplingView(g, &f.g);
      if (rc0 != EULER_GPU_OK) return rc0;
    }
    f.seed = seed; f.call_id = call_id; f.roots = roots_dev; f.n = n;
    f.default_node = default_node;
    f.c1 = counts_host[0]; f.c2 = counts_host[1];
    f.t1 = edge_types_host[0]; f.t2 = edge_types_host[1];
    uint8_t* wsb = (uint8_t*)workspace_dev;
    f.id1 = out_id_dev[0]; f.w1 = out_w_dev[0]; f.ty1 = out_t_dev[0]; f.mask0 = wsb;
    f.id2 = out_id_dev[1]; f.w2 = out_w_dev[1]; f.ty2 = out_t_dev[1];
    f.mask1 = wsb + (((size_t)n + 15) & ~(size_t)15);
    if (multi != nullptr) {
      f.mb_n = multi->n_per; f.call_ids = multi->call_ids_dev; f.call_stride = multi->call_stride;
    }
    int64_t blocks = n < 256 * 16 ? n : 256 * 16;
    hipLaunchKernelGGL(SampleFanout2Kernel, dim3((unsigned)blocks), dim3(256), 0, stream, f);
    EG_HIP(hipGetLastError());
    if (uniq_off != nullptr) { uniq_off[0] = -1; uniq_off[1] = -1; }
    return EULER_GPU_OK;
  }
  const uint64_t* roots = roots_dev;
  const uint8_t* mask = nullptr;
  int32_t group = 1;
  int64_t m = n;
  uint8_t* ws = (uint8_t*)workspace_dev;
  HopFusion hop;
  int rc = EULER_GPU_OK;
  // A graph with a neighbour id 0: the reference chains the hops on the CORE tensors
  // (one GQL, sample_fanout_op.cc:37-42) - a row the TF repack drops because it starts
  // with id 0 still hands its real samples to the next hop.  Sample in the core layout,
  // let the next hop read it, repack afterwards.
  const bool core_chain = g->view.has_zero_nbr != 0;
  auto repack = [&](int32_t h, int64_t rows, uint8_t* row_mask) {
    const int64_t total = rows * counts_host[h];
    if (total <= 0) return;
    hipLaunchKernelGGL(TfRepackKernel, dim3(GridFor(total, 256)), dim3(256), 0, stream,
                       out_id_dev[h], out_w_dev[h], out_t_dev[h], rows, counts_host[h],
                       default_node, row_mask);
    hipLaunchKernelGGL(TfRepackFirstKernel, dim3(GridFor(rows, 256)), dim3(256), 0, stream,
                       out_id_dev[h], out_w_dev[h], out_t_dev[h], rows, counts_host[h],
                       default_node);
  };
  if (core_chain) {
    int64_t rows_prev = 0;
    uint8_t* mask_prev = nullptr;
    for (int32_t h = 0; h < layers && rc == EULER_GPU_OK; ++h) {
      uint8_t* row_mask = ws;
      ws += ((size_t)m + 15) & ~(size_t)15;
      if (events != nullptr) t_phase_events = events + (size_t)h * 4;
      g_last_unique_offset = -1;
      rc = LaunchSampleNeighbor(g, stream, seed, call_id + (uint32_t)h, roots, m, nullptr, 1,
                                edge_types_host + (size_t)h * k, k, counts_host[h],
                                EULER_GPU_LAYOUT_CORE, default_node, out_id_dev[h],
                                out_w_dev[h], out_t_dev[h], nullptr, h == 0 ? 0 : 1, nullptr);
      if (uniq_off != nullptr) uniq_off[h] = g_last_unique_offset;
      if (h > 0) repack(h - 1, rows_prev, mask_prev);     // its ids have been read
      rows_prev = m; mask_prev = row_mask;
      roots = out_id_dev[h];
      m *= counts_host[h];
    }
    if (rc == EULER_GPU_OK && layers > 0) repack(layers - 1, rows_prev, mask_prev);
    t_phase_events = nullptr;
    if (rc == EULER_GPU_OK) EG_HIP(hipGetLastError());
    return rc;
  }
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

int euler_gpu_set_debug_buffer(void* dev) {
#ifdef EULER_GPU_MEASURE
  g_fl_debug = dev;
  return EULER_GPU_OK;
#else
  (void)dev;
  return Fail(EULER_GPU_EINVAL, "set_debug_buffer: this library was built without EULER_GPU_MEASURE "
                                "(make -C euler_amd/csrc MEASURE=1)");
#endif
}

int euler_gpu_set_tuning(int32_t key, int32_t value) {
  if (key == 0 && (value == 0 || value == 5 || value == 6)) { g_k1_variant = value; return EULER_GPU_OK; }
  if (key == 3) { g_k1_grid_cap = value; return EULER_GPU_OK; }
  if (key == 9) { g_k1_fuse_mark = value != 0; return EULER_GPU_OK; }
  if (key == 10 && (value == 1 || value == 2 || value == 4)) {
    g_expand_steps = value; return EULER_GPU_OK;
  }
  if (key == 11) { g_expand_const_type = value != 0; return EULER_GPU_OK; }
  if (key == 13) { g_k1_dual = value != 0; return EULER_GPU_OK; }
  if (key == 14 && value >= 0 && value <= 2) { g_dedup_block_numbering = value; return EULER_GPU_OK; }
  if (key == 12 && value >= 0) { g_expand_grid_cap = value; return EULER_GPU_OK; }
  if (key == 4) { g_k1_pair = value; return EULER_GPU_OK; }
  if (key == 5) { g_k1_dedup = value; return EULER_GPU_OK; }
  if (key == 7) { g_n2v_wave = value; return EULER_GPU_OK; }
  if (key == 8) { g_feature_vec4 = value; return EULER_GPU_OK; }
  if (key == 15 && value >= 0 && value <= 2) { g_root_host_batch = value; return EULER_GPU_OK; }
  if (key == 16) { g_adj_scan = value != 0; return EULER_GPU_OK; }
  if (key == 17 && value >= 0) { g_adj_long_row = value; return EULER_GPU_OK; }
  if (key == 18) { g_sum_scalar = value != 0; return EULER_GPU_OK; }
  if (key == 20) { g_dedup_resolve_in_expand = value != 0; return EULER_GPU_OK; }
  if (key == 23) { g_fanout_fused = value != 0; return EULER_GPU_OK; }
  if (key == 24) { g_full_nb_balanced = value != 0; return EULER_GPU_OK; }
  if (key == 25 && value >= 0) { g_n2v_big = value; return EULER_GPU_OK; }
  if (key == 27 && value >= 0 && value <= 2) { g_fanout_local = value; return EULER_GPU_OK; }
  if (key == 28 && value >= 0 && value <= 16) { g_fl_roots = value; return EULER_GPU_OK; }
  if (key == 29 && value >= 0) { g_fl_cap = value; return EULER_GPU_OK; }
  if (key == 30 && (value == 0 || value == 64 || value == 128 || value == 256)) { g_fl_block = value; return EULER_GPU_OK; }
  if (key == 31) { g_fl_wide = value != 0; return EULER_GPU_OK; }
  if (key == 32 && value >= -1) { g_fl_grid_cap = value; return EULER_GPU_OK; }
  if (key == 33 && value >= 0) { g_fl_min_roots = value; return EULER_GPU_OK; }
  if (key == 34 && value >= 0 && value <= 2) { g_fl_plain = value; return EULER_GPU_OK; }
#ifdef EULER_GPU_MEASURE
  if (key == 2) { g_k1_ablate = value; return EULER_GPU_OK; }
  if (key == 36) { g_fl_ablate = value; return EULER_GPU_OK; }
#endif
  if (key == 37 && value >= 0 && value <= 1) { g_k1_typed_pivot = value; return EULER_GPU_OK; }
  if (key == 38 && value >= 0) { g_walk_collapse = value; return EULER_GPU_OK; }
  if (key == 39 && value >= 0) { g_walk_grid = value; return EULER_GPU_OK; }
  if (key == 43 && value >= 0) { g_walk_tail = value; return EULER_GPU_OK; }
  if (key == 44 && (value == 0 || value == 1)) { g_walk_lean = value; return EULER_GPU_OK; }
  if (key == 35 && (value == 5 || value == 6 || value == 8)) { g_fl_wps = value; return EULER_GPU_OK; }
  if (key == 45 && (value == 0 || value == 1)) { g_fl_wb = value; return EULER_GPU_OK; }
  if (key == 47 && value >= 0 && value <= 2) { g_k1_sets_lds = value; return EULER_GPU_OK; }
  if (key == 48 && (value == 0 || value == 1)) { g_fl_typed_regs = value; return EULER_GPU_OK; }
  if (key == 49 && (value == 0 || value == 1)) { g_fl_fat = value; return EULER_GPU_OK; }
  if (key == 51 && (value == 0 || value == 1)) { g_blk_policy = value; return EULER_GPU_OK; }
  if (key == 52 && (value == 0 || value == 1)) { g_sharded_self_exchange.store(value); return EULER_GPU_OK; }
  if (key == 53 && value >= 0 && value <= 2) { g_fp_on = value; return EULER_GPU_OK; }
  if (key == 54 && (value == 0 || value == 1)) { g_fp_coop = value; return EULER_GPU_OK; }
  if (key == 57 && (value == 0 || value == 1)) { g_fp_lite2 = value; return EULER_GPU_OK; }
  if (key == 60 && (value == 0 || value == 1)) { g_flow_fused.store(value); return EULER_GPU_OK; }
  if (key == 62 && value >= 0 && value <= 2) { g_flow_rowpos.store(value); return EULER_GPU_OK; }
  if (key == 69 && value >= 0) { g_n2v_list_big.store(value); return EULER_GPU_OK; }
  if (key == 71 && value >= 0) { g_n2v_list_big_parent.store(value); return EULER_GPU_OK; }
  if (key == 72 && (value == 0 || value == 1)) { g_n2v_list_merged.store(value); return EULER_GPU_OK; }
  if (key == 73 && (value == 0 || value == 1)) { g_n2v_walk_tickets.store(value); return EULER_GPU_OK; }
  if (key == 67 && value >= 0 && value <= 1024) { g_sharded_walk_split.store(value); return EULER_GPU_OK; }
  if (key == 66 && value >= 0 && value <= 1024) { g_sharded_walk_tail.store(value); return EULER_GPU_OK; }
  if (key == 64 && value >= 1 && value <= 64) { g_walk_path_ch.store(value); return EULER_GPU_OK; }
  if (key == 63 && (value == 0 || value == 1)) { g_sharded_walk_enqueued.store(value); return EULER_GPU_OK; }
  if (key == 55 && value >= 4 && value <= 8) { g_fp_wps = value; return EULER_GPU_OK; }
  if (key == 56 && value >= 0) { g_blk_fail_next.store(value); return EULER_GPU_OK; }
  return Fail(EULER_GPU_EINVAL, "set_tuning: unknown key");
}

const char* euler_gpu_last_fanout_kernel(void) { return t_fl_last_kernel; }

int euler_gpu_sample_fanout_multi(const euler_gpu_graph* g, void* stream, uint64_t seed,
                                  uint32_t call_id, uint32_t call_stride,
                                  const uint32_t* call_ids_dev, int32_t m,
                                  const uint64_t* roots_dev, int64_t n,
                                  const int32_t* edge_types_host, int32_t k,
                                  const int32_t* counts_host, int32_t layers,
                                  int64_t default_node, uint64_t* const* out_id_dev,
                                  float* const* out_w_dev, int32_t* const* out_t_dev,
                                  void* workspace_dev) {
  if (!g) return Fail(EULER_GPU_ENOGRAPH, "sample_fanout_multi: null graph");
  if (m < 0 || n < 0 || layers <= 0 || k <= 0 || !counts_host || !edge_types_host)
    return Fail(EULER_GPU_EINVAL, "sample_fanout_multi: bad arguments");
  if (m == 0 || n == 0) return EULER_GPU_OK;
  int64_t total = (int64_t)m * n;
  for (int32_t h = 0; h < layers; ++h) {
    if (counts_host[h] <= 0) return Fail(EULER_GPU_EINVAL, "sample_fanout_multi: counts must be > 0");
    total *= counts_host[h];
  }
  if (total >= ((int64_t)1 << 31) || (int64_t)m * n >= ((int64_t)1 << 31))
    return Fail(EULER_GPU_EINVAL, "sample_fanout_multi: more than 2^31 output positions");
  FanoutMulti mu{n, m, call_stride, call_ids_dev};
  return RunFanout(g, (hipStream_t)stream, seed, call_id, roots_dev, (int64_t)m * n, edge_types_host, k,
                   counts_host, layers, default_node, out_id_dev, out_w_dev, out_t_dev, workspace_dev,
                   nullptr, nullptr, &mu);
}

int euler_gpu_sample_neighbor_sets(const euler_gpu_graph* g, void* stream, uint64_t seed,
                                   uint32_t call_id, const uint64_t* roots_dev, int64_t n,
                                   const int32_t* edge_types_host, const int32_t* set_k_host,
                                   int32_t n_sets, int32_t count, int64_t default_node,
                                   uint64_t* out_id_dev, float* out_w_dev, int32_t* out_t_dev) {
  if (!g) return Fail(EULER_GPU_ENOGRAPH, "sample_neighbor_sets: null graph");
  if (n < 0 || count <= 0 || n_sets < 0 || (n_sets > 0 && !set_k_host))
    return Fail(EULER_GPU_EINVAL, "sample_neighbor_sets: bad arguments");
  if (n == 0 || n_sets == 0) return EULER_GPU_OK;
  if (!roots_dev || !out_id_dev || !out_w_dev || !out_t_dev)
    return Fail(EULER_GPU_EINVAL, "sample_neighbor_sets: null buffer");
  int32_t total_k = 0;
  for (int32_t s = 0; s < n_sets; ++s) {
    if (set_k_host[s] < 0 || set_k_host[s] > kMaxListedTypes)
      return Fail(EULER_GPU_EINVAL, "sample_neighbor_sets: bad set size");
    total_k += set_k_host[s];
  }
  if (total_k > 0 && !edge_types_host) return Fail(EULER_GPU_EINVAL, "sample_neighbor_sets: null edge types");
  if (n * (int64_t)count * n_sets >= ((int64_t)1 << 40))
    return Fail(EULER_GPU_EINVAL, "sample_neighbor_sets: too many samples");
  int rc = EULER_GPU_OK;
  if (LaunchSampleNeighborSets(g, (hipStream_t)stream, seed, call_id, roots_dev, n, edge_types_host,
                               set_k_host, n_sets, count, default_node, out_id_dev, out_w_dev,
                               out_t_dev, &rc))
    return rc;
  // graphs the one-launch kernel does not serve: the separate calls
  const int64_t per = n * (int64_t)count;
  int32_t off = 0;
  for (int32_t s = 0; s < n_sets && rc == EULER_GPU_OK; ++s) {
    rc = LaunchSampleNeighbor(g, (hipStream_t)stream, seed, call_id + (uint32_t)s, roots_dev, n, nullptr, 1,
                              edge_types_host + off, set_k_host[s], count, EULER_GPU_LAYOUT_TF,
                              default_node, out_id_dev + (size_t)s * per, out_w_dev + (size_t)s * per,
                              out_t_dev + (size_t)s * per, nullptr);
    off += set_k_host[s];
  }
  return rc;
}

int euler_gpu_sample_aggregate_sets(const euler_gpu_graph* g, void* stream, uint64_t seed,
                                    uint32_t call_id, const uint64_t* roots_dev, int64_t n,
                                    const int32_t* edge_types_host, const int32_t* set_k_host,
                                    int32_t n_sets, int32_t count, int64_t default_node,
                                    int32_t mode, const float* feat_dev, int64_t feat_rows, int64_t d,
                                    uint64_t* out_id_dev, float* out_w_dev, int32_t* out_t_dev,
                                    float* out_agg_dev) {
  if (feat_rows < 0 || d < 0 || (d > 0 && (!feat_dev || !out_agg_dev)))
    return Fail(EULER_GPU_EINVAL, "sample_aggregate_sets: bad feature arguments");
  // every sampled id (and the default fill) must name a row of the table
  if (feat_rows >= ((int64_t)1 << 31))
    return Fail(EULER_GPU_EINVAL, "sample_aggregate_sets: the feature table must have fewer than 2^31 rows");
  if (default_node < 0 || default_node >= feat_rows)
    return Fail(EULER_GPU_EINVAL, "sample_aggregate_sets: default_node is not a row of the feature table "
                                  "(pass the id of a row kept for the default fill, e.g. max_id + 1)");
  if (g && g->max_id >= (uint64_t)feat_rows)
    return Fail(EULER_GPU_EINVAL, "sample_aggregate_sets: the feature table has fewer rows than the largest node id");
  const int rc = euler_gpu_sample_neighbor_sets(g, stream, seed, call_id, roots_dev, n, edge_types_host,
                                                set_k_host, n_sets, count, default_node, out_id_dev,
                                                out_w_dev, out_t_dev);
  if (rc != EULER_GPU_OK || n == 0 || n_sets == 0 || d == 0) return rc;
  if ((int64_t)n_sets * n >= ((int64_t)1 << 31))
    return Fail(EULER_GPU_EINVAL, "sample_aggregate_sets: more than 2^31 segments");
  // [sets][n] segments of `count` sampled ids each: one pass over the feature rows
  return euler_gpu_gather_segment_reduce_ids(stream, mode, feat_dev, feat_rows,
                                             reinterpret_cast<const int64_t*>(out_id_dev), nullptr,
                                             count, d, (int32_t)(n_sets * n), out_agg_dev);
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
  // one listed type: the type column stays off the wire (see PackRowsKernel)
  const int32_t single_type = k == 1 ? edge_types_host[0] : -1;
  if (k == 1 && (single_type < 0 || !edge_types_host))
    return Fail(EULER_GPU_EINVAL, "sample_neighbor_packed: bad edge type");
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
                             (const int32_t*)(buf + o_t), buf + o_m, n, count, single_type,
                             packed_dev);
  (void)hipFreeAsync(buf, st);
  return rc;
}

int euler_gpu_sample_neighbor_sets_packed(const euler_gpu_graph* g, void* stream, uint64_t seed,
                                          uint32_t call_id, const uint64_t* roots_dev, int64_t n,
                                          const int32_t* edge_types_host, const int32_t* set_k_host,
                                          int32_t n_sets, int32_t count, int64_t default_node,
                                          int32_t* packed_dev) {
  if (!g) return Fail(EULER_GPU_ENOGRAPH, "sample_neighbor_sets_packed: null graph");
  if (n < 0 || count <= 0 || n_sets < 0 || (n_sets > 0 && !set_k_host))
    return Fail(EULER_GPU_EINVAL, "sample_neighbor_sets_packed: bad arguments");
  if (n == 0 || n_sets == 0) return EULER_GPU_OK;
  if (!roots_dev || !packed_dev) return Fail(EULER_GPU_EINVAL, "sample_neighbor_sets_packed: null buffer");
  int32_t total_k = 0;
  for (int32_t s = 0; s < n_sets; ++s) {
    if (set_k_host[s] < 0 || set_k_host[s] > kMaxListedTypes)
      return Fail(EULER_GPU_EINVAL, "sample_neighbor_sets_packed: bad set size");
    total_k += set_k_host[s];
  }
  if (total_k > 0 && !edge_types_host)
    return Fail(EULER_GPU_EINVAL, "sample_neighbor_sets_packed: null edge types");
  if (n * (int64_t)count * n_sets >= ((int64_t)1 << 40))
    return Fail(EULER_GPU_EINVAL, "sample_neighbor_sets_packed: too many samples");
  // a set of ONE type must name a valid type (its rows travel without the type column)
  {
    int32_t off = 0;
    for (int32_t s = 0; s < n_sets; ++s) {
      if (set_k_host[s] == 1 && edge_types_host[off] < 0)
        return Fail(EULER_GPU_EINVAL, "sample_neighbor_sets_packed: bad edge type");
      off += set_k_host[s];
    }
  }
  int rc = EULER_GPU_OK;
  if (LaunchSampleNeighborSets(g, (hipStream_t)stream, seed, call_id, roots_dev, n, edge_types_host,
                               set_k_host, n_sets, count, default_node, nullptr, nullptr, nullptr, &rc,
                               packed_dev))
    return rc;
  // graphs / settings the one-launch kernel does not serve: the separate calls
  int32_t off = 0;
  int64_t poff = 0;
  for (int32_t s = 0; s < n_sets && rc == EULER_GPU_OK; ++s) {
    rc = euler_gpu_sample_neighbor_packed(g, stream, seed, call_id + (uint32_t)s, roots_dev, n,
                                          edge_types_host + off, set_k_host[s], count, default_node,
                                          packed_dev + poff);
    off += set_k_host[s];
    poff += n * (int64_t)PackedWords(count, set_k_host[s] == 1 ? 0 : 1);
  }
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
// TF SampleFanoutWithFeature (tf_euler/kernels/sample_fanout_with_feature_op.cc:135-233:
// `v(nodes).as(nb_0).sampleNB(..).as(nb_1) ... .v_select(nb_i).values(..).as(fea_i)`): the
// fanout and, for every layer's nodes (layer 0 = the roots), the dense features - one
// enqueue, nothing returns to the host.  dense_out_dev[layer * n_dense + j] = [m_layer,
// dims[j]] float32 (zero rows for default_node / unknown nodes / missing slots, :160-178).
int euler_gpu_sample_fanout_with_feature(const euler_gpu_graph* g, void* stream, uint64_t seed,
                                         uint32_t call_id, const uint64_t* roots_dev, int64_t n,
                                         const int32_t* edge_types_host, int32_t k,
                                         const int32_t* counts_host, int32_t layers,
                                         int64_t default_node, uint64_t* const* out_id_dev,
                                         float* const* out_w_dev, int32_t* const* out_t_dev,
                                         void* workspace_dev, const int32_t* dense_fids_host,
                                         const int32_t* dense_dims_host, int32_t n_dense,
                                         float* const* dense_out_dev) {
  if (n_dense < 0 || (n_dense > 0 && (!dense_fids_host || !dense_dims_host || !dense_out_dev)))
    return Fail(EULER_GPU_EINVAL, "sample_fanout_with_feature: bad feature arguments");
  int rc = euler_gpu_sample_fanout(g, stream, seed, call_id, roots_dev, n, edge_types_host, k,
                                   counts_host, layers, default_node, out_id_dev, out_w_dev,
                                   out_t_dev, workspace_dev);
  int64_t m = n;
  for (int32_t layer = 0; layer <= layers && rc == EULER_GPU_OK; ++layer) {
    const uint64_t* nodes = layer == 0 ? roots_dev : out_id_dev[layer - 1];
    for (int32_t j = 0; j < n_dense && rc == EULER_GPU_OK; ++j)
      rc = euler_gpu_get_dense_feature(g, stream, nodes, m, dense_fids_host[j], dense_dims_host[j],
                                       dense_out_dev[(size_t)layer * n_dense + j]);
    if (layer < layers) m *= counts_host[layer];
  }
  return rc;
}

// The 2-hop fanout in the (unique rows, index) form - what the reference's GQL holds before
// DATA_GATHER expands it (core/kernels/data_gather_op.cc:33-80 over ID_UNIQUE's gather_idx,
// id_unique_op.cc:35-64): hop 1 as euler_gpu_sample_fanout writes it, and for hop 2 the
// DISTINCT rows only - rows_*_dev hold n * counts[0] row slots of counts[1] samples each, of
// which a group of 4 roots fills the first few of ITS 4 * counts[0] (one per distinct child it
// drew); row_index_dev[i] names the row of hop-1 sample i.  The expanded tensor is
// rows[row_index] - 44 MB + 13 MB written per metric step instead of 524 MB.  Plain graphs
// only (the lean kernel of fanout_local.h: one edge-type group, identity ids, no id 0, even
// counts[1]); EULER_GPU_EINVAL otherwise - the caller falls back to euler_gpu_sample_fanout.
int euler_gpu_sample_fanout_unique(const euler_gpu_graph* g, void* stream, uint64_t seed,
                                   uint32_t call_id, const uint64_t* roots_dev, int64_t n,
                                   const int32_t* edge_types_host, const int32_t* counts_host,
                                   int64_t default_node, uint64_t* out_id1_dev, float* out_w1_dev,
                                   int32_t* out_t1_dev, uint32_t* row_index_dev,
                                   uint64_t* rows_id_dev, float* rows_w_dev, int32_t* rows_t_dev,
                                   void* workspace_dev) {
  if (!g) return Fail(EULER_GPU_ENOGRAPH, "sample_fanout_unique: null graph");
  if (n < 0 || !counts_host || !edge_types_host || (n > 0 && (!roots_dev || !out_id1_dev ||
      !out_w1_dev || !out_t1_dev || !row_index_dev || !rows_id_dev || !rows_w_dev || !rows_t_dev ||
      !workspace_dev)))
    return Fail(EULER_GPU_EINVAL, "sample_fanout_unique: bad arguments");
  if (n == 0) return EULER_GPU_OK;
  if (n * (int64_t)counts_host[0] >= ((int64_t)1 << 32))
    return Fail(EULER_GPU_EINVAL, "sample_fanout_unique: more than 2^32 row slots");
  uint64_t* ids[2] = {out_id1_dev, rows_id_dev};
  float* ws[2] = {out_w1_dev, rows_w_dev};
  int32_t* ts[2] = {out_t1_dev, rows_t_dev};
  const int save_min = g_fl_min_roots, save_local = g_fanout_local;
  g_fl_min_roots = 0; g_fanout_local = 2;          // this form exists in the one-kernel path only
  t_fl_row_index = row_index_dev;
  t_fl_took_lean = 0;
  const int rc = RunFanout(g, (hipStream_t)stream, seed, call_id, roots_dev, n, edge_types_host, 1,
                           counts_host, 2, default_node, ids, ws, ts, workspace_dev, nullptr,
                           nullptr);
  t_fl_row_index = nullptr;
  g_fl_min_roots = save_min; g_fanout_local = save_local;
  if (rc != EULER_GPU_OK) return rc;
  if (!t_fl_took_lean)
    return Fail(EULER_GPU_EINVAL, "sample_fanout_unique: this graph / fanout is not served by the "
                                  "one-kernel path (the outputs hold the dense form's first rows)");
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
  {
    // untimed: first-use work of this stream (scratch, indexes built on first use)
    int rc = LaunchSampleNeighbor(g, st, seed, 0u, roots_dev, n, nullptr, 1, edge_types_host, k, count, layout, -1,
                                  out_id_dev, out_w_dev, out_t_dev, nullptr);
    if (rc != EULER_GPU_OK) return rc;
  }
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
// Wire format of the multi-GPU result exchange: one int32 row per root =
// [count ids (2 words each) | count weights | count types | mask | pad] -
// 4 * count + 2 words - so that one all-to-all moves everything a hop returns and
// every row starts 8-byte aligned.  A call with ONE listed edge type leaves the
// type column out (3 * count + 2 words, padded to even: a quarter less on the wire): every
// sample of a valid row has that type, a masked row -1 (TF layout).  PackRows
// writes the rows from the sampler's outputs (the pivot kernels write them
// directly); ExpandPacked reads them back per POSITION through `pos` (merge +
// gather + unpack in one pass).
// ------------------------------------------------------------------------
__global__ __launch_bounds__(256) void PackRowsKernel(
    const uint64_t* __restrict__ ids, const float* __restrict__ w,
    const int32_t* __restrict__ t, const uint8_t* __restrict__ mask, int64_t m,
    int32_t count, int32_t tcol, int32_t* __restrict__ packed) {
  const int32_t words = PackedWords(count, tcol);
  const int64_t total = m * (int64_t)words;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < total;
       s += stride) {
    const int64_t r = s / words;
    const int32_t c = (int32_t)(s - r * words);
    int32_t v;
    if (c < 2 * count) v = reinterpret_cast<const int32_t*>(ids)[r * 2 * count + c];
    else if (c < 3 * count) v = __float_as_int(w[r * count + (c - 2 * count)]);
    else if (c < (3 + tcol) * count) v = t[r * count + (c - 3 * count)];
    else v = c == (3 + tcol) * count ? (int32_t)mask[r] : 0;
    packed[s] = v;
  }
}

int euler_gpu_pack_rows(void* stream, const uint64_t* id_dev, const float* w_dev,
                        const int32_t* t_dev, const uint8_t* mask_dev, int64_t m,
                        int32_t count, int32_t single_type, int32_t* packed_dev) {
  if (m < 0 || count <= 0) return Fail(EULER_GPU_EINVAL, "pack_rows: bad m/count");
  if (m == 0) return EULER_GPU_OK;
  if (!id_dev || !w_dev || !t_dev || !mask_dev || !packed_dev)
    return Fail(EULER_GPU_EINVAL, "pack_rows: null buffer");
  const int block = 256;
  const int32_t tcol = single_type >= 0 ? 0 : 1;
  hipLaunchKernelGGL(PackRowsKernel, dim3(GridFor(m * (int64_t)PackedWords(count, tcol), block)),
                     dim3(block), 0, (hipStream_t)stream, id_dev, w_dev, t_dev, mask_dev,
                     m, count, tcol, packed_dev);
  EG_HIP(hipGetLastError());
  return EULER_GPU_OK;
}

int euler_gpu_expand_packed(void* stream, const int32_t* pos_dev, int64_t n,
                            int32_t count, int32_t single_type, const int32_t* packed_dev,
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
#define EG_XP(UU, TC)                                                                     \
  hipLaunchKernelGGL((ExpandPackedKernel<UU, TC>), dim3((int)blocks), dim3(block), 0,      \
                     (hipStream_t)stream, pos_dev, packed_dev, n, count, single_type,      \
                     out_id_dev, out_w_dev, out_t_dev, out_mask_dev, stride_rows,          \
                     stride_slots)
  if (single_type >= 0) { if (pair) EG_XP(2, false); else EG_XP(1, false); }
  else { if (pair) EG_XP(2, true); else EG_XP(1, true); }
#undef EG_XP
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

int euler_gpu_time_sample_fanout(const euler_gpu_graph* g, void* stream, uint64_t seed,
                                 const uint64_t* roots_dev, int64_t n,
                                 const int32_t* edge_types_host, int32_t k,
                                 const int32_t* counts_host, int32_t layers,
                                 int64_t default_node, uint64_t* const* out_id_dev,
                                 float* const* out_w_dev, int32_t* const* out_t_dev,
                                 void* workspace_dev, int32_t iters, float* mean_ms_host) {
  if (!g) return Fail(EULER_GPU_ENOGRAPH, "time_sample_fanout: null graph");
  if (iters <= 0 || !mean_ms_host) return Fail(EULER_GPU_EINVAL, "time_sample_fanout: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  hipEvent_t e0, e1;
  EG_HIP(hipEventCreate(&e0));
  EG_HIP(hipEventCreate(&e1));
  int rc = euler_gpu_sample_fanout(g, stream, seed, 0, roots_dev, n, edge_types_host, k,
                                   counts_host, layers, default_node, out_id_dev, out_w_dev,
                                   out_t_dev, workspace_dev);       // untimed: first-use work
  EG_HIP(hipEventRecord(e0, st));
  for (int32_t it = 0; it < iters && rc == EULER_GPU_OK; ++it)
    rc = euler_gpu_sample_fanout(g, stream, seed, (uint32_t)(it * layers), roots_dev, n,
                                 edge_types_host, k, counts_host, layers, default_node,
                                 out_id_dev, out_w_dev, out_t_dev, workspace_dev);
  EG_HIP(hipEventRecord(e1, st));
  EG_HIP(hipEventSynchronize(e1));
  float ms = 0.f;
  EG_HIP(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  *mean_ms_host = ms / (float)iters;
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

}  // extern "C"
