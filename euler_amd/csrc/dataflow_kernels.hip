// Block construction on the device: the whole
// SageDataFlow of the reference - tf_euler/python/dataflow/sage_dataflow.py:35-50
// (get_neighbors: sample, then tf.unique of [neighbours | nodes]) followed by
// neighbor_dataflow.py:84-110 (UniqueDataFlow.produce_subgraph: tf.unique of
// [neighbours | nodes] again, res_n_id, edge_index, self loops) - enqueued on one
// stream WITHOUT a host round trip between the hops.
//
// TensorFlow gives every hop a dynamic shape (the number of distinct nodes); a
// host that wants that shape has to wait for the device.  Here every array is
// sized for the worst case (cap_0 = n, cap_{h+1} = cap_h * (count_h + 1)) and the
// true sizes stay in a device-side counts array: hop h+1's sampler and unique
// kernels are launched for cap_{h+1} items and their lanes beyond counts[h+1]
// exit.  The host reads the counts once, after the last hop (or never: padded
// tensors + counts are a valid result for a consumer that masks).
//
// Per hop (`cnt` = counts[h], all on device):
//   1. SampleNeighbor(count) of n_id[0 .. cnt)                 -> nb [cnt * count]
//   2. tf.unique of the virtual concatenation V = [nb | n_id], first-occurrence
//      order (id_unique_op.cc:35-64 semantics): open-addressing claim + atomicMin
//      of the position, first-occurrence flags, ONE exclusive scan = the rank
//      -> n_id' [cnt'], inv [cnt * (count + 1)]
//   3. res_n_id = inv[cnt * count ..], edge_dst = inv (or its first cnt * count
//      entries without self loops), edge_src[e] = e / count for the sampled
//      edges and e - cnt * count for the self loops.
// The reference runs tf.unique twice per hop on the same input (once inside
// get_neighbors, once in produce_subgraph): one pass serves both.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <atomic>
#include <vector>

#include "common.h"
#include "device_fns.h"
#include "k1_search.h"

namespace euler_gpu {

// tuning key 62: fused Sage flows send the ids WITH a graph row through the hop's own {row, smallest
// position} hash table (a few MB that stay near the chip) instead of the stream's row-indexed table
// (8 B x n_rows: every claim a cold line).  2 = flows of at most 32 768 roots (16 384 roots x [25, 10]:
// 0.117 -> 0.111 ms; at 131 072 roots the hop's table is 128 MB itself and the flow 3 % SLOWER:
// profiles/r6_rowpos_ab.txt), 1 = always, 0 = never
std::atomic<int> g_flow_rowpos{2};
std::atomic<int> g_flow_fused{1};     // tuning key 60: Sage flows: sampler + insert in one kernel, tables cleared by the kernels before (0 = op by op)

namespace {

constexpr uint64_t kFlowEmptyKey = ~0ULL;
constexpr int kFlowChunk = 1024;      // positions of V per workgroup of the flag / emit kernels

struct FlowTable {
  unsigned long long* keys;   // [cap + 1]; slot `cap` is the side slot of the all-ones id
  uint32_t* minpos;           // [cap + 1]
  int32_t* rank;              // [cap + 1]
  uint64_t mask;              // cap - 1
  // not null: ids WITH a graph row go through this table instead of the stream's row-indexed one -
  // one word {row, smallest position} per slot, claimed with one compare-and-swap (tuning key 62)
  unsigned long long* rowpos; // [cap + 1]
};

struct FlowHop {
  const uint64_t* nb;         // [cap_n * count] sampled neighbours (valid: cnt * count)
  const uint64_t* n_id;       // [cap_n] nodes of the previous layer (valid: cnt)
  const uint32_t* cnt;        // device-side count of n_id
  uint32_t* cnt_out;          // device-side count of the new layer
  const uint32_t* m_nb_dev;   // not null: the length of nb lives on the device (full-neighbour
                              // flows: rows of different lengths) instead of being cnt * count
  const int32_t* nb_src;      // ... and the source (index into n_id) of every entry of nb
  int32_t count;
  int32_t self_loops;
  int64_t cap_m;              // cap_n * (count + 1): worst-case length of V
  FlowTable t;                // ids without a graph row (unknown ids, the default fill)
  FlowTable t_next;           // not null keys: the NEXT hop's table, cleared by this hop's emit kernel
  int32_t count_next;         // ... for a layer of cnt_out nodes: cnt_out * (count_next + 1) positions
  // ids WITH a row: a table indexed by row (common.h: FlowTableDense) - one atomicMin of
  // {~epoch, position} claims the row for this hop and keeps the smallest position; no
  // probing, nothing to clear
  GraphView g;
  unsigned long long* dense_min;   // [n_rows] or null (hash table for every id)
  uint32_t* dense_rank;            // [n_rows]
  unsigned long long epoch_hi;     // (~epoch) << 32
  uint32_t* slot_of;          // [cap_m]: the row, or 0x80000000 | hash slot
  uint32_t* blk_cnt;          // [n_blk + 1] first occurrences per chunk of kFlowChunk positions
  unsigned long long* first_bits;   // [cap_m / 64 + 17] is-first-occurrence, one bit per position of V
  uint32_t* word_pre;         // [cap_m / 64 + 17] first occurrences of the chunk before the word (behind first_bits)
  int64_t n_blk;              // ceil(cap_m / kFlowChunk)
  // several minibatches per launch (euler_gpu_sage_blocks_multi): minibatch b = blockIdx.y has
  // its own copy of every array, b strides further (FlowOf); 1 = the single flow
  int32_t n_mb;
  int64_t mb_nb, mb_nid, mb_cnt, mb_tab, mb_m, mb_blk;
  uint64_t* new_n_id;         // [cap_m]
  int64_t* inv;               // [cap_m] edge_dst: index of every element of V in new_n_id
  int64_t* edge_src;          // [cap_m]
  int64_t* res_n_id;          // [cap_n]
};

__device__ __forceinline__ int64_t FlowNbLen(const FlowHop& h, int64_t cnt) {
  return h.m_nb_dev != nullptr ? (int64_t)(*h.m_nb_dev) : cnt * h.count;
}

__device__ __forceinline__ uint64_t FlowElem(const FlowHop& h, int64_t i, int64_t m_nb) {
  return i < m_nb ? h.nb[i] : h.n_id[i - m_nb];
}

// the hop as minibatch b of the launch sees it
__device__ __forceinline__ FlowHop FlowOf(const FlowHop& h0, uint32_t b) {
  FlowHop h = h0;
  if (h0.n_mb > 1) {
    h.nb += (int64_t)b * h0.mb_nb;        h.n_id += (int64_t)b * h0.mb_nid;
    h.cnt += (int64_t)b * h0.mb_cnt;      h.cnt_out += (int64_t)b * h0.mb_cnt;
    h.t.keys += (int64_t)b * h0.mb_tab;   h.t.minpos += (int64_t)b * h0.mb_tab;
    h.t.rank += (int64_t)b * h0.mb_tab;   h.slot_of += (int64_t)b * h0.mb_m;
    h.blk_cnt += (int64_t)b * h0.mb_blk;
    h.first_bits = reinterpret_cast<unsigned long long*>(h.blk_cnt + ((h0.n_blk + 2) & ~(int64_t)1));
    h.word_pre = reinterpret_cast<uint32_t*>(h.first_bits + (h0.cap_m / 64 + 18));
    h.new_n_id += (int64_t)b * h0.mb_m;   h.inv += (int64_t)b * h0.mb_m;
    h.edge_src += (int64_t)b * h0.mb_m;   h.res_n_id += (int64_t)b * h0.mb_nid;
  }
  return h;
}

// The hash table is allocated for the worst case, but a minibatch fills a fraction of it
// (16 384 roots x [25, 10]: 0.65 M of 4.7 M positions): every kernel sizes the part it uses
// from the device-side length - a power of two >= 2 m slots - so clearing and probing touch
// megabytes that stay in the L2 instead of the 200 MB a worst-case table takes
// (profiles/r4_sage_blocks_kernel_stats.csv: 87 us of fills + an insert that missed the L2).
__device__ __forceinline__ uint64_t FlowMaskOf(uint64_t table_mask, int64_t m) {
  uint64_t cap = 64;
  while (cap < (uint64_t)m * 2 && cap <= table_mask) cap <<= 1;
  return (cap > table_mask + 1 ? table_mask + 1 : cap) - 1;
}
// Flows with the row-indexed table send only the ids WITHOUT a graph row to the hash table (unknown
// roots, dangling neighbour ids, the default fill - a handful of distinct ids in a real flow):
// 1.25 slots per position (the worst case - every position a distinct id without a row - probes
// longer and still fits) instead of 2, half the bytes to clear per hop.  Hash-only flows keep 2.
__device__ __forceinline__ uint64_t FlowMask(const FlowHop& h, int64_t m) {
  return FlowMaskOf(h.t.mask, h.dense_min != nullptr ? (m * 5 + 7) / 8 : m);
}

// the part of table t a hop of m positions uses, emptied by the whole grid
__device__ __forceinline__ void FlowClearTable(const FlowTable& t, int64_t m, bool dense) {
  const int64_t slots = (int64_t)FlowMaskOf(t.mask, dense ? (m * 5 + 7) / 8 : m) + 1;      // (FlowMask below)
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t first = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (first == 0) {                    // the side slot of the all-ones id
    t.keys[t.mask + 1] = kFlowEmptyKey;
    t.minpos[t.mask + 1] = 0xFFFFFFFFu;
  }
  for (int64_t i = first; i < slots; i += stride) {
    t.keys[i] = kFlowEmptyKey;
    t.minpos[i] = 0xFFFFFFFFu;
  }
  if (t.rowpos != nullptr) {          // ids with a row: up to one slot per position, 2 slots per position
    const int64_t rslots = (int64_t)FlowMaskOf(t.mask, m) + 1;
    for (int64_t i = first; i < rslots; i += stride) t.rowpos[i] = kFlowEmptyKey;
  }
}

__global__ __launch_bounds__(256) void FlowClearKernel(const FlowHop h0) {
  const FlowHop h = FlowOf(h0, blockIdx.y);
  const int64_t cnt = (int64_t)(*h.cnt);
  const int64_t m = FlowNbLen(h, cnt) + cnt;
  FlowClearTable(h.t, m, h.dense_min != nullptr);
}

constexpr uint32_t kFlowHashed = 0x80000000u;


// Ids with a graph row (all of them, normally) are first reduced INSIDE the workgroup: a chunk
// of kFlowChunk positions goes through an LDS table {row, smallest position}, and only the
// table's entries go to the row-indexed table in HBM.  A sampled frontier repeats its hubs
// thousands of times; without this every wave that starts before the first atomic has landed
// sends its own, and the memory side works them off one by one (41 us for the 0.43 M
// positions of the first hop: profiles/r4_sage_blocks_kernel_stats.csv).
constexpr int kFlowLds = 2 * kFlowChunk;          // LDS slots per workgroup (power of two)

constexpr int kFlowInsertThreads = 256;    // (1024 - one position per lane - measured slower: 28.8 vs 26.8 us)

// BY_ID = false: the flow has the row-indexed table (h.dense_min): rows reduced in LDS as above, the
// rare ids without a row straight to the hash table in HBM.  BY_ID = true: a flow WITHOUT that
// table (the multi-minibatch launch - two minibatches of a launch would fight over a row -, a
// stream beyond the table budget): every id goes to the hash table, after the same reduction by
// ID - atomics of many lanes on one slot are worked off one by one by the memory side (round 5,
// the 64-minibatch launch: hop 1's insert, 6.8 copies of every distinct id, 168 -> 31 us; hop 2's,
// 1.5 copies, stays at 230 us - the memory side's atomic rate; taking the loads in front of the
// atomics away instead changed nothing).
template <bool BY_ID>
__global__ __launch_bounds__(kFlowInsertThreads) void FlowInsertKernel(const FlowHop h0) {
  const FlowHop h = FlowOf(h0, blockIdx.y);
  // BY_ID: u64 ids [kFlowLds] + u32 smallest position, then the global slot [kFlowLds]; else u32 rows + u32 positions
  __shared__ unsigned long long s_buf[BY_ID ? kFlowLds + kFlowLds / 2 : kFlowLds];
  __shared__ uint32_t s_side;                         // BY_ID: smallest position of the all-ones id
  uint32_t* const s_row = reinterpret_cast<uint32_t*>(s_buf);
  uint32_t* const s_pos = s_row + kFlowLds;
  unsigned long long* const s_id = s_buf;
  uint32_t* const s_ipos = reinterpret_cast<uint32_t*>(s_buf + kFlowLds);
  const int64_t cnt = (int64_t)(*h.cnt);
  const int64_t m_nb = FlowNbLen(h, cnt), m = m_nb + cnt;
  const uint64_t mask = FlowMask(h, m);
  constexpr uint32_t kNone = 0xFFFFFFFFu, kSide = 0xFFFFFFFEu;
  for (int64_t base = (int64_t)blockIdx.x * kFlowChunk; base < m; base += (int64_t)gridDim.x * kFlowChunk) {
    if (BY_ID) {
      for (int x = threadIdx.x; x < kFlowLds; x += kFlowInsertThreads) { s_id[x] = kFlowEmptyKey; s_ipos[x] = 0xFFFFFFFFu; }
      if (threadIdx.x == 0) s_side = 0xFFFFFFFFu;
    } else {
      for (int x = threadIdx.x; x < kFlowLds; x += kFlowInsertThreads) { s_row[x] = 0xFFFFFFFFu; s_pos[x] = 0xFFFFFFFFu; }
    }
    __syncthreads();
    uint32_t ls[kFlowChunk / kFlowInsertThreads];     // BY_ID: the LDS slot of a position's id (or the side slot)
#pragma unroll
    for (int x = 0; x < kFlowChunk / kFlowInsertThreads; ++x) {
      ls[x] = kNone;
      const int64_t i = base + x * kFlowInsertThreads + threadIdx.x;
      if (i >= m) continue;
      const uint64_t id = FlowElem(h, i, m_nb);
      if (BY_ID) {
        if (id == kFlowEmptyKey) {
          atomicMin(&s_side, (uint32_t)i);
          ls[x] = kSide;
          continue;
        }
        uint32_t s = (uint32_t)(Mix64(id) & (uint64_t)(kFlowLds - 1));
        for (;;) {                                    // at most kFlowChunk of kFlowLds slots fill up
          const unsigned long long old = atomicCAS(&s_id[s], (unsigned long long)kFlowEmptyKey, (unsigned long long)id);
          if (old == kFlowEmptyKey || old == id) break;
          s = (s + 1) & (uint32_t)(kFlowLds - 1);
        }
        atomicMin(&s_ipos[s], (uint32_t)i);
        ls[x] = s;
        continue;
      }
      const int64_t row = FindRow(h.g, id);
      if (row >= 0) {
        h.slot_of[i] = (uint32_t)row;
        uint32_t s = (uint32_t)(Mix64((uint64_t)row) & (uint64_t)(kFlowLds - 1));
        for (;;) {
          const uint32_t old = atomicCAS(&s_row[s], 0xFFFFFFFFu, (uint32_t)row);
          if (old == 0xFFFFFFFFu || old == (uint32_t)row) break;
          s = (s + 1) & (uint32_t)(kFlowLds - 1);
        }
        atomicMin(&s_pos[s], (uint32_t)i);
        continue;
      }
      uint64_t s;
      if (id == kFlowEmptyKey) {
        s = h.t.mask + 1;
      } else {
        s = Mix64(id) & mask;
        for (;;) {
          const unsigned long long old =
              atomicCAS(&h.t.keys[s], (unsigned long long)kFlowEmptyKey, (unsigned long long)id);
          if (old == kFlowEmptyKey || old == id) break;
          s = (s + 1) & mask;
        }
      }
      atomicMin(&h.t.minpos[s], (uint32_t)i);
      h.slot_of[i] = kFlowHashed | (uint32_t)s;
    }
    __syncthreads();
    if (BY_ID) {
      // the chunk's distinct ids: one claim of a hash slot + one minimum each; the slot is left in
      // LDS for the ids' positions
      for (int x = threadIdx.x; x < kFlowLds; x += kFlowInsertThreads) {
        const unsigned long long id = s_id[x];
        if (id == kFlowEmptyKey) continue;
        uint64_t s = Mix64(id) & mask;
        for (;;) {
          const unsigned long long old = atomicCAS(&h.t.keys[s], (unsigned long long)kFlowEmptyKey, id);
          if (old == kFlowEmptyKey || old == id) break;
          s = (s + 1) & mask;
        }
        atomicMin(&h.t.minpos[s], s_ipos[x]);
        s_ipos[x] = (uint32_t)s;
      }
      if (threadIdx.x == 0 && s_side != 0xFFFFFFFFu) atomicMin(&h.t.minpos[h.t.mask + 1], s_side);
      __syncthreads();
#pragma unroll
      for (int x = 0; x < kFlowChunk / kFlowInsertThreads; ++x) {
        if (ls[x] == kNone) continue;
        const int64_t i = base + x * kFlowInsertThreads + threadIdx.x;
        h.slot_of[i] = kFlowHashed | (ls[x] == kSide ? (uint32_t)(h.t.mask + 1) : s_ipos[ls[x]]);
      }
    } else {
      // the chunk's distinct rows, one atomicMin of {~epoch, position} each
      for (int x = threadIdx.x; x < kFlowLds; x += kFlowInsertThreads) {
        const uint32_t row = s_row[x];
        if (row == 0xFFFFFFFFu) continue;
        const unsigned long long mine = h.epoch_hi | (unsigned long long)s_pos[x];
        atomicMin(&h.dense_min[row], mine);
      }
    }
    __syncthreads();
  }
}

// first occurrences per chunk of kFlowChunk positions (workgroup b = chunk b; chunks past
// the valid prefix count zero): one bit per position, per 64-position word the chunk's first
// occurrences BEFORE the word, per chunk their number - the rank of a first occurrence at
// position f is then (chunks before f's) + word_pre[f / 64] + (bits of its word below f): three
// small arrays that stay in the L2, no table of ranks
__global__ __launch_bounds__(256) void FlowFlagKernel(const FlowHop h0) {
  const FlowHop h = FlowOf(h0, blockIdx.y);
  __shared__ uint32_t s_pc[kFlowChunk / 64];
  const int64_t cnt = (int64_t)(*h.cnt);
  const int64_t m = FlowNbLen(h, cnt) + cnt;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int64_t b = blockIdx.x; b <= h.n_blk; b += gridDim.x) {
    const int64_t base = b * kFlowChunk;
    const bool live = base < m && b < h.n_blk;
    if (live) {
      // the chunk's 4 positions of this lane: every load of a kind in flight together
      uint32_t sw[kFlowChunk / 256];
      bool first[kFlowChunk / 256];
#pragma unroll
      for (int x = 0; x < kFlowChunk / 256; ++x) {
        const int64_t i = base + x * 256 + threadIdx.x;
        sw[x] = i < m ? h.slot_of[i] : 0u;
      }
#pragma unroll
      for (int x = 0; x < kFlowChunk / 256; ++x) {
        const int64_t i = base + x * 256 + threadIdx.x;
        first[x] = false;
        if (i < m) {
          // the element's first position takes the slot word's place: the emit kernel reads it
          // in position order instead of going back to the tables
          const uint32_t f = (sw[x] & kFlowHashed) ? h.t.minpos[sw[x] & ~kFlowHashed]
                             : h.t.rowpos != nullptr ? (uint32_t)h.t.rowpos[sw[x]] : (uint32_t)h.dense_min[sw[x]];
          h.slot_of[i] = f;
          first[x] = f == (uint32_t)i;
        }
      }
#pragma unroll
      for (int x = 0; x < kFlowChunk / 256; ++x) {
        const int64_t i = base + x * 256 + threadIdx.x;
        const unsigned long long bal = __ballot(first[x]);
        if (lane == 0) {                                  // (lane 0: i is a multiple of 64)
          h.first_bits[i >> 6] = bal;
          s_pc[x * 4 + wv] = (uint32_t)__popcll(bal);
        }
      }
    }
    __syncthreads();
    if (threadIdx.x < kFlowChunk / 64) {
      uint32_t before = 0, all = 0;
      if (live) {
#pragma unroll
        for (int y = 0; y < kFlowChunk / 64; ++y) {
          const uint32_t v = s_pc[y];
          if (y < (int)threadIdx.x) before += v;
          all += v;
        }
        h.word_pre[(base >> 6) + threadIdx.x] = before;
      }
      if (threadIdx.x == 0) h.blk_cnt[b] = all;
    }
    __syncthreads();
  }
}

// Exclusive sums of the valid chunk counts, in place, + the new layer's size - flows whose chunk
// counts do not fit the emit kernel's LDS (more than kFlowLdsChunks chunks of V).  One workgroup.
constexpr int kFlowLdsChunks = 8192;
__global__ __launch_bounds__(1024) void FlowScanKernel(const FlowHop h0) {
  const FlowHop h = FlowOf(h0, blockIdx.y);
  __shared__ uint32_t s_part[1024];
  const int64_t cnt = (int64_t)(*h.cnt);
  const int64_t m = FlowNbLen(h, cnt) + cnt;
  const int64_t nc = (m + kFlowChunk - 1) / kFlowChunk;
  const int64_t per = (nc + 1023) / 1024;
  const int64_t lo = (int64_t)threadIdx.x * per, hi = lo + per < nc ? lo + per : nc;
  uint32_t sum = 0;
  for (int64_t x = lo; x < hi; ++x) sum += h.blk_cnt[x];
  s_part[threadIdx.x] = sum;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t acc = 0;
    for (int x = 0; x < 1024; ++x) { const uint32_t v = s_part[x]; s_part[x] = acc; acc += v; }
    *h.cnt_out = acc;
  }
  __syncthreads();
  uint32_t run = s_part[threadIdx.x];
  for (int64_t x = lo; x < hi; ++x) { const uint32_t v = h.blk_cnt[x]; h.blk_cnt[x] = run; run += v; }
}

// The new layer and the block's index arrays in one pass over V: position i reads its element's
// first position f (the table entry the flag kernel has just read), f's rank from the three
// small arrays, and writes inv / edge_src / res_n_id; i == f also writes new_n_id[rank].
// SCANNED = false: every workgroup sums the chunk counts into LDS itself (<= kFlowLdsChunks);
// true: FlowScanKernel has left the exclusive sums in blk_cnt.
template <bool SCANNED>
__global__ __launch_bounds__(256) void FlowEmitIndexKernel(const FlowHop h0) {
  const FlowHop h = FlowOf(h0, blockIdx.y);
  __shared__ uint32_t s_pre[SCANNED ? 1 : kFlowLdsChunks];
  __shared__ uint32_t s_part[256];
  __shared__ uint32_t s_total;
  const int64_t cnt = (int64_t)(*h.cnt);
  const int64_t m_nb = FlowNbLen(h, cnt), m = m_nb + cnt;
  if (!SCANNED) {
    const int64_t nc = (m + kFlowChunk - 1) / kFlowChunk;       // <= kFlowLdsChunks (the launcher's choice)
    const int64_t per = (nc + 255) / 256;
    const int64_t lo = (int64_t)threadIdx.x * per, hi = lo + per < nc ? lo + per : nc;
    uint32_t sum = 0;
    for (int64_t x = lo; x < hi; ++x) { const uint32_t v = h.blk_cnt[x]; s_pre[x] = v; sum += v; }
    s_part[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x < 64) {
      // exclusive scan of the 256 partial sums by one wave: 4 per lane
      uint32_t v0 = s_part[4 * threadIdx.x], v1 = s_part[4 * threadIdx.x + 1], v2 = s_part[4 * threadIdx.x + 2],
               v3 = s_part[4 * threadIdx.x + 3];
      uint32_t tot = v0 + v1 + v2 + v3, inc = tot;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = __shfl_up(inc, off);
        if ((int)threadIdx.x >= off) inc += o;
      }
      const uint32_t ex = inc - tot;
      s_part[4 * threadIdx.x] = ex; s_part[4 * threadIdx.x + 1] = ex + v0;
      s_part[4 * threadIdx.x + 2] = ex + v0 + v1; s_part[4 * threadIdx.x + 3] = ex + v0 + v1 + v2;
      if (threadIdx.x == 63) { s_total = inc; if (blockIdx.x == 0) *h.cnt_out = inc; }
    }
    __syncthreads();
    uint32_t run = s_part[threadIdx.x];
    for (int64_t x = lo; x < hi; ++x) { const uint32_t v = s_pre[x]; s_pre[x] = run; run += v; }
    __syncthreads();
  }
  if (h.t_next.keys != nullptr)        // the next hop's table, for the layer this hop has just sized
    FlowClearTable(h.t_next, (int64_t)(SCANNED ? *h.cnt_out : s_total) * ((int64_t)h.count_next + 1), true);
  const uint32_t* pre = SCANNED ? h.blk_cnt : s_pre;
  constexpr int kU = 4;
  const int64_t stride = (int64_t)gridDim.x * 256 * kU;
  for (int64_t i0 = (int64_t)blockIdx.x * 256 * kU + threadIdx.x; i0 < m; i0 += stride) {
    uint32_t fp[kU], wp[kU];
    unsigned long long bits[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const int64_t i = i0 + (int64_t)u * 256;
      fp[u] = i < m ? h.slot_of[i] : 0u;           // (the flag kernel left the first position there)
    }
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const int64_t i = i0 + (int64_t)u * 256;
      bits[u] = 0ull; wp[u] = 0u;
      if (i < m) { bits[u] = h.first_bits[fp[u] >> 6]; wp[u] = h.word_pre[fp[u] >> 6]; }
    }
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const int64_t i = i0 + (int64_t)u * 256;
      if (i >= m) continue;
      const uint32_t f = fp[u];
      const uint32_t below = (uint32_t)__popcll(bits[u] & ((1ull << (f & 63u)) - 1ull));
      const int64_t dst = (int64_t)(pre[f / kFlowChunk] + wp[u] + below);
      if ((uint32_t)i == f) h.new_n_id[dst] = FlowElem(h, i, m_nb);
      if (i < m_nb) {
        h.inv[i] = dst;
        h.edge_src[i] = h.nb_src != nullptr ? (int64_t)h.nb_src[i] : i / h.count;
      } else {
        h.res_n_id[i - m_nb] = dst;
        if (h.self_loops) {
          h.inv[i] = dst;
          h.edge_src[i] = i - m_nb;        // last_idx = arange: node j keeps an edge to itself
        }
      }
    }
  }
}

// flag -> [scan ->] emit + index for one hop (f.n_blk = chunks of the worst case)
void LaunchFlowRanks(const FlowHop& f, dim3 grid_b, int64_t work, int per_mb_cap, hipStream_t st) {
  hipLaunchKernelGGL(FlowFlagKernel, grid_b, dim3(256), 0, st, f);
  int64_t gx = (work + 256 * 4 - 1) / (256 * 4);
  const int64_t cap = per_mb_cap > 0 ? per_mb_cap : 1024;       // every workgroup pays the prologue once
  if (gx > cap) gx = cap;
  if (gx < 1) gx = 1;
  const dim3 grid((unsigned)gx, grid_b.y);
  if (f.n_blk <= kFlowLdsChunks) {
    hipLaunchKernelGGL(FlowEmitIndexKernel<false>, grid, dim3(256), 0, st, f);
  } else {
    hipLaunchKernelGGL(FlowScanKernel, dim3(1, grid_b.y), dim3(1024), 0, st, f);
    hipLaunchKernelGGL(FlowEmitIndexKernel<true>, grid, dim3(256), 0, st, f);
  }
}

// ---- full-neighbour flows (GCNDataFlow, RelationDataFlow) ------------------------------
// tf_euler/python/dataflow/gcn_dataflow.py:33-47, relation_dataflow.py:30-72: every hop takes
// ALL neighbours (of the listed edge types) of the nodes seen so far, then the same
// tf.unique / res_n_id / edge_index as above.  Rows have different lengths, so the length of
// the hop's neighbour list is a second device-side count; the caller gives the capacity of
// the edge arrays (a hop that would overflow it sets the overflow word and produces no
// edges: the host sees it in the one read of the counts and takes the op-by-op path).
struct FlowFull {
  GraphView g;
  const uint64_t* n_id;       // [cap_n]
  const uint32_t* cnt;
  int32_t k;
  int32_t et[kMaxListedTypes];
  uint32_t* lens;             // [cap_n + 1] row lengths (zeros past cnt), then their exclusive scan
  uint32_t* offs;             // [cap_n + 1]
  uint32_t* m_nb;             // out: the hop's neighbour count
  uint32_t* overflow;         // set to hop + 1 by the first hop whose m_nb > cap_e
  int32_t hop;
  int64_t cap_n, cap_e;
  uint64_t* nb; int32_t* nb_src; int32_t* nb_t;
};

__device__ __forceinline__ uint32_t FlowRowLen(const FlowFull& f, int64_t row) {
  if (row < 0) return 0;
  const RowMeta m = LoadRowMeta(f.g, row);
  uint32_t len = 0;
  for (int32_t x = 0; x < f.k; ++x) {
    const int32_t t = f.et[x];
    if (t < 0 || t >= f.g.T) continue;
    len += (uint32_t)(m.type_end[t] - (t == 0 ? 0 : m.type_end[t - 1]));
  }
  return len;
}

// The offsets are 32-bit: a hop whose true neighbour total reaches 2^32 must not wrap
// into a small number that passes the capacity test - the scan saturates instead
// (saturating addition of unsigned numbers is associative).
struct SatAdd {
  __host__ __device__ __forceinline__ uint32_t operator()(uint32_t a, uint32_t b) const {
    const uint32_t s = a + b;
    return s < a ? 0xFFFFFFFFu : s;
  }
};

__global__ __launch_bounds__(256) void FlowFullCountKernel(const FlowFull f) {
  const int64_t cnt = (int64_t)(*f.cnt);
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= f.cap_n; i += stride)
    f.lens[i] = i < cnt ? FlowRowLen(f, FindRow(f.g, f.n_id[i])) : 0u;
}

__global__ void FlowFullTotalKernel(const FlowFull f) {
  const uint32_t total = f.offs[f.cap_n];       // exclusive scan over cap_n + 1 entries
  if ((int64_t)total > f.cap_e) {
    if (*f.overflow == 0u) *f.overflow = (uint32_t)f.hop + 1u;    // hops run in stream order
    *f.m_nb = 0u;
  } else {
    *f.m_nb = total;
  }
}

// one lane per output entry: its row is the last i with offs[i] <= e
__global__ __launch_bounds__(256) void FlowFullFillKernel(const FlowFull f) {
  const int64_t total = (int64_t)(*f.m_nb);
  const int64_t cnt = (int64_t)(*f.cnt);
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
    int64_t lo = 0, hi = cnt;                   // offs[lo] <= e < offs[hi]
    while (hi - lo > 1) {
      const int64_t mid = (lo + hi) >> 1;
      if ((int64_t)f.offs[mid] <= e) lo = mid; else hi = mid;
    }
    int32_t p = (int32_t)(e - (int64_t)f.offs[lo]);      // position in the row's listed order
    const RowMeta m = LoadRowMeta(f.g, FindRow(f.g, f.n_id[lo]));
    uint64_t id = 0;
    int32_t ty = 0;
    for (int32_t x = 0; x < f.k; ++x) {
      const int32_t t = f.et[x];
      if (t < 0 || t >= f.g.T) continue;
      const int32_t b = t == 0 ? 0 : m.type_end[t - 1];
      const int32_t len = m.type_end[t] - b;
      if (p < len) { id = f.g.nbr[m.row_ptr + b + p]; ty = t; break; }
      p -= len;
    }
    f.nb[e] = id;
    f.nb_src[e] = (int32_t)lo;
    if (f.nb_t != nullptr) f.nb_t[e] = ty;
  }
}

__global__ void FlowInitKernel(uint32_t* counts, uint32_t n) { counts[0] = n; }

// ... and the first hop's table emptied (its length is known to the host)
__global__ __launch_bounds__(256) void FlowInitClearKernel(uint32_t* counts, uint32_t n, const FlowTable t, int64_t m0) {
  if (blockIdx.x == 0 && threadIdx.x == 0) counts[0] = n;
  FlowClearTable(t, m0, true);
}

// counts of minibatch b at counts + b * stride
__global__ void FlowInitMultiKernel(uint32_t* counts, uint32_t n, int32_t n_mb, int32_t stride) {
  for (int32_t b = threadIdx.x; b < n_mb; b += blockDim.x) counts[(int64_t)b * stride] = n;
}

// The hop's sampler for SEVERAL minibatches in one launch (euler_gpu_sage_blocks_multi): lane per
// sample of the first cnt[b] nodes of minibatch b = blockIdx.y's layer, neighbour ids only (the
// flow uses nothing else).  One listed edge type on a monotone graph: exactly the draw of
// SampleNeighborPivotKernel<TF, 1, BLOCKED> (sample_kernels.hip: PivotPass) - sample j of a node
// is word pair j & 1 of Philox block (seed, call id, node, j >> 1), a node without a row or without
// edges of the type answers default_node - with minibatch b's own call id, so that the launch
// equals the separate calls.
struct FlowSample {
  GraphView g;
  uint64_t seed;
  uint32_t call_id, call_stride;      // minibatch b draws with call_id + b * call_stride
  const uint64_t* n_id;               // [n_mb][cap_n]
  const uint32_t* cnt;                // minibatch b: cnt[b * mb_cnt]
  uint64_t* nb;                       // [n_mb][cap_n * count]
  int64_t cap_n, mb_cnt, default_node;
  int32_t count, type;
};

__global__ __launch_bounds__(256, kWavesPerSimd) void FlowSampleKernel(const FlowSample s) {
  const uint32_t b = blockIdx.y;
  const int64_t total = (int64_t)s.cnt[(int64_t)b * s.mb_cnt] * s.count;
  const uint64_t* n_id = s.n_id + (int64_t)b * s.cap_n;
  uint64_t* nb = s.nb + (int64_t)b * s.cap_n * s.count;
  const uint32_t call = s.call_id + b * s.call_stride;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int64_t r = i / s.count;
    const int32_t j = (int32_t)(i - r * s.count);
    const uint64_t node = n_id[r];
    uint64_t id = (uint64_t)s.default_node;
    Segment sg;
    if (LoadSegment<true>(s.g, FindRow(s.g, node), s.type, &sg)) {
      const Philox4 blk = RngBlock(s.seed, call, kDomainNeighbor, node, ((uint32_t)j) >> 1);
      const double u = (j & 1) ? UnitFromWords(blk.w[2], blk.w[3]) : UnitFromWords(blk.w[0], blk.w[1]);
      float w;
      BlockPivotSample(s.g, sg, u, &id, &w);
    }
    nb[i] = id;
  }
}

// The hop's sampler and the insert of what it draws in ONE kernel (single flows with the
// row-indexed table; one listed edge type on a monotone graph - FlowSampleKernel's draw): a lane
// draws ONE sample (or takes one node of the previous layer: V = [nb | n_id]), the workgroup's 256
// ids are reduced by row in LDS, the distinct rows claim {~epoch, smallest position} in the table.
// The atomics of a workgroup leave while the others still wait for their samples' cold lines: the
// insert's 42 us (653 K positions, 16 384 roots) hide behind the sampler's 22 instead of following
// them, and the neighbour list is not read back.
struct FlowSI {
  FlowHop h;
  uint64_t seed;
  uint32_t call_id;
  int32_t type;
  int64_t default_node;
};
template <int kThreads>
__global__ __launch_bounds__(kThreads) void FlowSampleInsertKernel(const FlowSI a) {
  constexpr int kFlowSiLds = 2 * kThreads;
  const FlowHop& h = a.h;
  __shared__ uint32_t s_row[kFlowSiLds];
  __shared__ uint32_t s_pos[kFlowSiLds];
  const int64_t cnt = (int64_t)(*h.cnt);
  const int64_t m_nb = cnt * h.count, m = m_nb + cnt;
  const uint64_t mask = FlowMask(h, m);
  for (int64_t base = (int64_t)blockIdx.x * kThreads; base < m; base += (int64_t)gridDim.x * kThreads) {
    for (int x = threadIdx.x; x < kFlowSiLds; x += kThreads) { s_row[x] = 0xFFFFFFFFu; s_pos[x] = 0xFFFFFFFFu; }
    __syncthreads();
    const int64_t i = base + threadIdx.x;
    uint32_t my_ls = 0xFFFFFFFFu;                      // (rowpos tables) the LDS slot of this position's row
    if (i < m) {
      uint64_t id;
      if (i < m_nb) {
        const int64_t r = i / h.count;
        const int32_t j = (int32_t)(i - r * h.count);
        const uint64_t node = h.n_id[r];
        id = (uint64_t)a.default_node;
        Segment sg;
        if (LoadSegment<true>(h.g, FindRow(h.g, node), a.type, &sg)) {
          const Philox4 blk = RngBlock(a.seed, a.call_id, kDomainNeighbor, node, ((uint32_t)j) >> 1);
          const double u = (j & 1) ? UnitFromWords(blk.w[2], blk.w[3]) : UnitFromWords(blk.w[0], blk.w[1]);
          float w;
          BlockPivotSample(h.g, sg, u, &id, &w);
        }
        const_cast<uint64_t*>(h.nb)[i] = id;
      } else {
        id = h.n_id[i - m_nb];
      }
      const int64_t row = FindRow(h.g, id);
      if (row >= 0) {
        h.slot_of[i] = (uint32_t)row;
        uint32_t sl = (uint32_t)(Mix64((uint64_t)row) & (uint64_t)(kFlowSiLds - 1));
        for (;;) {
          const uint32_t old = atomicCAS(&s_row[sl], 0xFFFFFFFFu, (uint32_t)row);
          if (old == 0xFFFFFFFFu || old == (uint32_t)row) break;
          sl = (sl + 1) & (uint32_t)(kFlowSiLds - 1);
        }
        atomicMin(&s_pos[sl], (uint32_t)i);
        my_ls = sl;
      } else {
        uint64_t sl;
        if (id == kFlowEmptyKey) {
          sl = h.t.mask + 1;
        } else {
          sl = Mix64(id) & mask;
          for (;;) {
            const unsigned long long old =
                atomicCAS(&h.t.keys[sl], (unsigned long long)kFlowEmptyKey, (unsigned long long)id);
            if (old == kFlowEmptyKey || old == id) break;
            sl = (sl + 1) & mask;
          }
        }
        atomicMin(&h.t.minpos[sl], (uint32_t)i);
        h.slot_of[i] = kFlowHashed | (uint32_t)sl;
      }
    }
    __syncthreads();
    if (h.t.rowpos != nullptr) {
      // the workgroup's distinct rows claim {row, smallest position} words of the hop's own hash
      // table (a few MB that stay near the chip, against cold lines of the 8 B x n_rows table):
      // one compare-and-swap per first claim, + a minimum when another workgroup was first
      const uint64_t rmask = FlowMaskOf(h.t.mask, m);
      for (int x = threadIdx.x; x < kFlowSiLds; x += kThreads) {
        const uint32_t row = s_row[x];
        if (row == 0xFFFFFFFFu) continue;
        const unsigned long long mine = (unsigned long long)row << 32 | (unsigned long long)s_pos[x];
        uint64_t sl = Mix64((uint64_t)row) & rmask;
        for (;;) {
          const unsigned long long old = atomicCAS(&h.t.rowpos[sl], (unsigned long long)kFlowEmptyKey, mine);
          if (old == kFlowEmptyKey) break;
          if ((uint32_t)(old >> 32) == row) { atomicMin(&h.t.rowpos[sl], mine); break; }
          sl = (sl + 1) & rmask;
        }
        s_pos[x] = (uint32_t)sl;                      // the rows' positions read their slot from here
      }
      __syncthreads();
      if (my_ls != 0xFFFFFFFFu) h.slot_of[base + threadIdx.x] = s_pos[my_ls];
    } else {
      for (int x = threadIdx.x; x < kFlowSiLds; x += kThreads) {
        const uint32_t row = s_row[x];
        if (row == 0xFFFFFFFFu) continue;
        atomicMin(&h.dense_min[row], h.epoch_hi | (unsigned long long)s_pos[x]);
      }
    }
    __syncthreads();
  }
}

// The same fusion for SEVERAL minibatches per launch (euler_gpu_sage_blocks_multi: minibatch
// blockIdx.y, its own call id, its own hash table - every id goes through it, reduced by ID in LDS
// first as FlowInsertKernel<true> does): the draw of FlowSampleKernel, then the insert of the
// workgroup's 256 positions.
struct FlowSIM {
  FlowHop h;
  uint64_t seed;
  uint32_t call_id, call_stride;
  int32_t type;
  int64_t default_node;
};

__global__ __launch_bounds__(256, kWavesPerSimd) void FlowSampleInsertMultiKernel(const FlowSIM a) {
  constexpr int kLds = 512;
  const FlowHop h = FlowOf(a.h, blockIdx.y);
  __shared__ unsigned long long s_id[kLds];
  __shared__ uint32_t s_ipos[kLds];
  __shared__ uint32_t s_side;
  const int64_t cnt = (int64_t)(*h.cnt);
  const int64_t m_nb = cnt * h.count, m = m_nb + cnt;
  const uint64_t mask = FlowMask(h, m);
  const uint32_t call = a.call_id + blockIdx.y * a.call_stride;
  constexpr uint32_t kNone = 0xFFFFFFFFu, kSide = 0xFFFFFFFEu;
  for (int64_t base = (int64_t)blockIdx.x * 256; base < m; base += (int64_t)gridDim.x * 256) {
    for (int x = threadIdx.x; x < kLds; x += 256) { s_id[x] = kFlowEmptyKey; s_ipos[x] = 0xFFFFFFFFu; }
    if (threadIdx.x == 0) s_side = 0xFFFFFFFFu;
    __syncthreads();
    const int64_t i = base + threadIdx.x;
    uint32_t ls = kNone;
    if (i < m) {
      uint64_t id;
      if (i < m_nb) {
        const int64_t r = i / h.count;
        const int32_t j = (int32_t)(i - r * h.count);
        const uint64_t node = h.n_id[r];
        id = (uint64_t)a.default_node;
        Segment sg;
        if (LoadSegment<true>(h.g, FindRow(h.g, node), a.type, &sg)) {
          const Philox4 blk = RngBlock(a.seed, call, kDomainNeighbor, node, ((uint32_t)j) >> 1);
          const double u = (j & 1) ? UnitFromWords(blk.w[2], blk.w[3]) : UnitFromWords(blk.w[0], blk.w[1]);
          float w;
          BlockPivotSample(h.g, sg, u, &id, &w);
        }
        const_cast<uint64_t*>(h.nb)[i] = id;
      } else {
        id = h.n_id[i - m_nb];
      }
      if (id == kFlowEmptyKey) {
        atomicMin(&s_side, (uint32_t)i);
        ls = kSide;
      } else {
        uint32_t sl = (uint32_t)(Mix64(id) & (uint64_t)(kLds - 1));
        for (;;) {                                    // at most 256 of the 512 slots fill up
          const unsigned long long old = atomicCAS(&s_id[sl], (unsigned long long)kFlowEmptyKey, (unsigned long long)id);
          if (old == kFlowEmptyKey || old == id) break;
          sl = (sl + 1) & (uint32_t)(kLds - 1);
        }
        atomicMin(&s_ipos[sl], (uint32_t)i);
        ls = sl;
      }
    }
    __syncthreads();
    for (int x = threadIdx.x; x < kLds; x += 256) {
      const unsigned long long id = s_id[x];
      if (id == kFlowEmptyKey) continue;
      uint64_t sl = Mix64(id) & mask;
      for (;;) {
        const unsigned long long old = atomicCAS(&h.t.keys[sl], (unsigned long long)kFlowEmptyKey, id);
        if (old == kFlowEmptyKey || old == id) break;
        sl = (sl + 1) & mask;
      }
      atomicMin(&h.t.minpos[sl], s_ipos[x]);
      s_ipos[x] = (uint32_t)sl;
    }
    if (threadIdx.x == 0 && s_side != 0xFFFFFFFFu) atomicMin(&h.t.minpos[h.t.mask + 1], s_side);
    __syncthreads();
    if (ls != kNone) h.slot_of[i] = kFlowHashed | (ls == kSide ? (uint32_t)(h.t.mask + 1) : s_ipos[ls]);
    __syncthreads();
  }
}

// The stream's row-indexed table and the epochs of `hops` hops (common.h: FlowTableDense).
// No table (allocation failed, more than 2^31 rows): the hash table serves every id.
int FlowDenseTable(const euler_gpu_graph* g, hipStream_t st, int32_t hops,
                   unsigned long long** dense_min, uint32_t** dense_rank, uint32_t* epoch0) {
  *dense_min = nullptr; *dense_rank = nullptr; *epoch0 = 0;
  const size_t rows = (size_t)g->view.n_rows;
  if (rows == 0 || rows >= ((size_t)1 << 31)) return EULER_GPU_OK;
  std::lock_guard<std::mutex> lk(g->ws_mu);
  // a table is 12 bytes per graph row and lives as long as the graph: at most kMaxFlowTables
  // streams get one (a caller that keeps creating streams falls back to the hash table, which
  // needs no persistent memory)
  constexpr size_t kMaxFlowTables = 4;
  if (g->flow_tables.find((void*)st) == g->flow_tables.end() && g->flow_tables.size() >= kMaxFlowTables)
    return EULER_GPU_OK;
  auto& ft = g->flow_tables[(void*)st];
  if (ft.p == nullptr || ft.rows < rows) {
    if (ft.p != nullptr) { (void)hipStreamSynchronize(st); (void)hipFree(ft.p); ft.p = nullptr; ft.rows = 0; }
    if (hipMalloc(&ft.p, rows * 12 + 64) != hipSuccess) {
      (void)hipGetLastError();
      ft.p = nullptr;
      return EULER_GPU_OK;
    }
    ft.rows = rows; ft.next_epoch = 1;
    EG_HIP(hipMemsetAsync(ft.p, 0xFF, rows * 8, st));       // {~0, ~0}: older than every epoch
  }
  if ((uint64_t)ft.next_epoch + (uint64_t)hops + 2 >= 0xFFFFFFF0ull) {     // the epochs are used up
    EG_HIP(hipMemsetAsync(ft.p, 0xFF, ft.rows * 8, st));
    ft.next_epoch = 1;
  }
  *dense_min = (unsigned long long*)ft.p;
  *dense_rank = (uint32_t*)((uint8_t*)ft.p + ft.rows * 8);
  *epoch0 = ft.next_epoch;
  ft.next_epoch += (uint32_t)hops;
  return EULER_GPU_OK;
}

size_t Al(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace
}  // namespace euler_gpu

using namespace euler_gpu;

extern "C" {

// capacities of layer h (nodes) and of hop h's edge arrays
static int64_t FlowCap(int64_t n, const int32_t* fanouts, int32_t h) {
  int64_t c = n;
  for (int32_t i = 0; i < h; ++i) c *= (int64_t)fanouts[i] + 1;
  return c;
}

// region A: the arrays of one hop (the largest hop's: they are reused hop after hop); behind it
// one hash table per hop - hop h's emit kernel clears hop h + 1's table while hop h's arrays are
// still in use, so the tables cannot share region A (round 6)
static size_t SageTableBytes(int64_t cap_m) {
  uint64_t tcap = 64;
  while (tcap < (uint64_t)cap_m * 2) tcap <<= 1;
  return Al((tcap + 1) * 8) + Al((tcap + 1) * 4) + Al((tcap + 1) * 8);      // keys, minpos, rowpos
}
static size_t SageRegionA(int64_t n, const int32_t* fanouts_host, int32_t layers) {
  size_t best = 0;
  for (int32_t h = 0; h < layers; ++h) {
    const int64_t cap_n = FlowCap(n, fanouts_host, h);
    const int64_t cap_m = cap_n * ((int64_t)fanouts_host[h] + 1);
    uint64_t tcap = 64;
    while (tcap < (uint64_t)cap_m * 2) tcap <<= 1;
    size_t scan_bytes = 0;
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, (uint32_t*)nullptr,
                                           (uint32_t*)nullptr, (int)(cap_m + 1), nullptr);
    const size_t b = Al((size_t)cap_n * fanouts_host[h] * 8)      // nb ids
                     + Al((size_t)cap_n * fanouts_host[h] * 4) * 2   // weights, types (unused outputs)
                     + Al((tcap + 1) * 8) + Al((tcap + 1) * 4) * 2  // table (the op-by-op path's)
                     + Al((size_t)cap_m * 4)                        // slot_of
                     + Al(((size_t)cap_m + 1) * 4) * 2              // chunk counts, first-occurrence bits, word sums
                     + Al(scan_bytes) + 256;
    if (b > best) best = b;
  }
  return Al(best + 256);
}

size_t euler_gpu_sage_blocks_workspace(int64_t n, const int32_t* fanouts_host, int32_t layers) {
  size_t total = SageRegionA(n, fanouts_host, layers);
  for (int32_t h = 0; h < layers; ++h)
    total += SageTableBytes(FlowCap(n, fanouts_host, h) * ((int64_t)fanouts_host[h] + 1));
  return total + 256;
}

int euler_gpu_sage_blocks(const euler_gpu_graph* g, void* stream, uint64_t seed, uint32_t call_id,
                          const uint64_t* roots_dev, int64_t n, const int32_t* edge_types_host,
                          int32_t k, const int32_t* fanouts_host, int32_t layers,
                          int64_t default_node, int32_t add_self_loops, void* workspace_dev,
                          uint64_t* const* n_id_dev, int64_t* const* res_n_id_dev,
                          int64_t* const* edge_src_dev, int64_t* const* edge_dst_dev,
                          uint32_t* counts_dev) {
  if (!g) return Fail(EULER_GPU_ENOGRAPH, "sage_blocks: null graph");
  if (n < 0 || layers <= 0 || layers > 8 || k < 0 || !fanouts_host || !n_id_dev || !res_n_id_dev ||
      !edge_src_dev || !edge_dst_dev || !counts_dev || (n > 0 && (!roots_dev || !workspace_dev)))
    return Fail(EULER_GPU_EINVAL, "sage_blocks: bad arguments");
  for (int32_t h = 0; h < layers; ++h)
    if (fanouts_host[h] <= 0) return Fail(EULER_GPU_EINVAL, "sage_blocks: fanouts must be > 0");
  // (slot_of packs "hashed" into bit 31 and the side slot of the all-ones id is slot tcap:
  // tcap <= 2^30 keeps the two apart, i.e. cap_m <= 2^29)
  if (FlowCap(n, fanouts_host, layers) > ((int64_t)1 << 29))
    return Fail(EULER_GPU_EINVAL, "sage_blocks: worst-case layer size > 2^29");
  hipStream_t st = (hipStream_t)stream;
  if (n == 0) {
    EG_HIP(hipMemsetAsync(counts_dev, 0, sizeof(uint32_t) * (layers + 1), st));
    return EULER_GPU_OK;
  }
  const uint64_t* n_id = roots_dev;
  unsigned long long* dense_min = nullptr;
  uint32_t* dense_rank = nullptr;
  uint32_t epoch0 = 0;
  // The stream's row-indexed table is shared by every block-construction call on that stream:
  // a call's hops (epochs e .. e + layers - 1) must be enqueued as one uninterrupted run, or a
  // second host thread on the same stream would slip its Insert (newer epoch) between this
  // call's Insert and Flag / Emit.  Same rule as the fanout's scratch (RunFanout).
  std::lock_guard<std::recursive_mutex> launch_lk(g->launch_mu);
  {
    const int rcd = FlowDenseTable(g, st, layers, &dense_min, &dense_rank, &epoch0);
    if (rcd != EULER_GPU_OK) return rcd;
  }
  (void)dense_rank;
  // One listed type per hop on a monotone graph with the row-indexed table: three launches per hop
  // (sampler + insert, flag, emit + index) and the tables cleared by the kernels before them.
  GraphView view = g->view;
  bool fused = k == 1 && dense_min != nullptr && g_flow_fused != 0;
  if (fused) {
    if (SamplingView(g, &view) != EULER_GPU_OK) { (void)hipGetLastError(); fused = false; view = g->view; }
  }
  fused = fused && view.monotone != 0 && view.has_zero_nbr == 0 && HasBlockSearch(view);
  for (int32_t h = 0; fused && h < layers; ++h) fused = edge_types_host[h] >= 0;
  // the per-hop tables behind region A (fused flow)
  FlowTable tabs[8];
  {
    uint8_t* tp = (uint8_t*)workspace_dev + SageRegionA(n, fanouts_host, layers);
    for (int32_t h = 0; h < layers; ++h) {
      const int64_t cm = FlowCap(n, fanouts_host, h) * ((int64_t)fanouts_host[h] + 1);
      uint64_t tcap = 64;
      while (tcap < (uint64_t)cm * 2) tcap <<= 1;
      tabs[h].keys = (unsigned long long*)tp;
      tabs[h].minpos = (uint32_t*)(tp + Al((tcap + 1) * 8));
      tabs[h].rank = nullptr;
      tabs[h].mask = tcap - 1;
      const int rp = g_flow_rowpos.load();
      tabs[h].rowpos = rp == 1 || (rp == 2 && n <= 32768)
                           ? (unsigned long long*)(tp + Al((tcap + 1) * 8) + Al((tcap + 1) * 4)) : nullptr;
      tp += SageTableBytes(cm);
    }
  }
  if (fused) {
    const int64_t m0 = n * ((int64_t)fanouts_host[0] + 1);
    int64_t gi = (2 * m0 + 64 + 255) / 256;
    if (gi > 1024) gi = 1024;
    hipLaunchKernelGGL(FlowInitClearKernel, dim3((unsigned)gi), dim3(256), 0, st, counts_dev, (uint32_t)n, tabs[0], m0);
  } else {
    hipLaunchKernelGGL(FlowInitKernel, dim3(1), dim3(1), 0, st, counts_dev, (uint32_t)n);
  }
  for (int32_t h = 0; h < layers; ++h) {
    const int32_t count = fanouts_host[h];
    const int64_t cap_n = FlowCap(n, fanouts_host, h);
    const int64_t cap_m = cap_n * ((int64_t)count + 1);
    uint64_t tcap = 64;
    while (tcap < (uint64_t)cap_m * 2) tcap <<= 1;
    uint8_t* p = (uint8_t*)workspace_dev;
    uint64_t* nb = (uint64_t*)p;        p += Al((size_t)cap_n * count * 8);
    float* nb_w = (float*)p;            p += Al((size_t)cap_n * count * 4);
    int32_t* nb_t = (int32_t*)p;        p += Al((size_t)cap_n * count * 4);
    FlowHop f{};
    f.g = view; f.dense_min = dense_min; f.dense_rank = nullptr;
    f.epoch_hi = (unsigned long long)(uint32_t)~(epoch0 + (uint32_t)h) << 32;
    f.t.keys = (unsigned long long*)p;  p += Al((tcap + 1) * 8);
    f.t.minpos = (uint32_t*)p;          p += Al((tcap + 1) * 4);
    f.t.rank = (int32_t*)p;             p += Al((tcap + 1) * 4);
    f.t.mask = tcap - 1;
    f.slot_of = (uint32_t*)p;           p += Al((size_t)cap_m * 4);
    // (the workspace holds two arrays of cap_m + 1 words here: the chunk counts need far less,
    // and the first-occurrence bits - cap_m / 64 + 18 double words - and the words' sums follow them)
    f.blk_cnt = (uint32_t*)p;           p += Al(((size_t)cap_m + 1) * 4);
    p += Al(((size_t)cap_m + 1) * 4);
    f.n_blk = (cap_m + kFlowChunk - 1) / kFlowChunk;
    f.first_bits = (unsigned long long*)(f.blk_cnt + ((f.n_blk + 2) & ~(int64_t)1));   // same region
    f.word_pre = (uint32_t*)(f.first_bits + (cap_m / 64 + 18));                          // same region
    f.nb = nb; f.n_id = n_id; f.cnt = counts_dev + h; f.cnt_out = counts_dev + h + 1;
    f.count = count; f.self_loops = add_self_loops ? 1 : 0; f.cap_m = cap_m;
    f.new_n_id = n_id_dev[h]; f.inv = edge_dst_dev[h]; f.edge_src = edge_src_dev[h];
    f.res_n_id = res_n_id_dev[h];
    const int block = 256;
    const int grid = GridFor(cap_m + 1, block);
    const int grid_b = (int)(f.n_blk + 1 < 65536 ? f.n_blk + 1 : 65536);
    if (fused) {
      // 1 + 2. the hop's sampler over the first counts[h] nodes of the layer and the insert of
      // [nb | n_id] in one kernel; the table was cleared by the kernel before it
      f.t = tabs[h];
      if (h + 1 < layers) { f.t_next = tabs[h + 1]; f.count_next = fanouts_host[h + 1]; }
      FlowSI si{};
      si.h = f; si.seed = seed; si.call_id = call_id + (uint32_t)h; si.type = edge_types_host[(size_t)h * k];
      si.default_node = default_node;
      // (a workgroup of 256: 0.120 ms per 16 384-root flow against 0.122 / 0.130 with 512 / 1 024)
      // (the grid is sized for the WORST-case layer, which a sampled flow fills to ~1/7: 4 096
      // workgroups that stride over what there is - 16 384 roots 0.112 -> 0.107 ms against 16 384
      // workgroups of which most only read the count and leave)
      int64_t gs = (cap_m + 255) / 256;
      if (gs > 4096) gs = 4096;
      hipLaunchKernelGGL(FlowSampleInsertKernel<256>, dim3((unsigned)gs), dim3(256), 0, st, si);
    } else {
      // 1. the hop's sampler over the first counts[h] nodes of the layer
      int rc = LaunchSampleNeighborCounted(g, st, seed, call_id + (uint32_t)h, n_id, cap_n,
                                           counts_dev + h, edge_types_host + (size_t)h * k, k, count,
                                           default_node, nb, nb_w, nb_t);
      if (rc != EULER_GPU_OK) return rc;
      // 2. first-occurrence unique of [nb | n_id]
      hipLaunchKernelGGL(FlowClearKernel, dim3(grid), dim3(block), 0, st, f);
      if (f.dense_min != nullptr) hipLaunchKernelGGL(FlowInsertKernel<false>, dim3(grid_b), dim3(kFlowInsertThreads), 0, st, f);
      else hipLaunchKernelGGL(FlowInsertKernel<true>, dim3(grid_b), dim3(kFlowInsertThreads), 0, st, f);
    }
    // 3. ranks, the new layer, res_n_id, edge_index
    LaunchFlowRanks(f, dim3(grid_b), cap_m, 0, st);
    EG_HIP(hipGetLastError());
    n_id = n_id_dev[h];
  }
  return EULER_GPU_OK;
}

// ---- several minibatches per enqueue ------------------------------------------------------
// The reference's GraphSAGE callers build one flow per minibatch of a few hundred roots
// (sage_dataflow.py:35-50; run_graphsage.py:35 defaults to 32): a GPU is not filled by 1 024
// roots, and a flow is a dozen dependent launches of microseconds each.  M minibatches of n roots
// in ONE enqueue: every kernel of the flow gets a second grid dimension (the minibatch, FlowOf),
// every array M copies b strides apart, the hops' samplers one launch for all minibatches with
// minibatch b's own call id - the result of M euler_gpu_sage_blocks calls, bit for bit.  The
// first-occurrence tables are the hash tables (the row-indexed table is one per stream: two
// minibatches of a launch would fight over a row).
static size_t MultiHopBytes(int32_t n_mb, int64_t cap_n, int32_t count, int64_t* r_words) {
  const int64_t cap_m = cap_n * ((int64_t)count + 1);
  uint64_t tcap = 64;
  while (tcap < (uint64_t)cap_m * 2) tcap <<= 1;
  const int64_t n_blk = (cap_m + kFlowChunk - 1) / kFlowChunk;
  const int64_t R = ((n_blk + 2) & ~(int64_t)1) + 3 * (cap_m / 64 + 18) + 1;
  if (r_words) *r_words = R;
  return Al((size_t)n_mb * cap_n * count * 8) + Al((size_t)n_mb * (tcap + 1) * 8) +
         Al((size_t)n_mb * (tcap + 1) * 4) * 2 + Al((size_t)n_mb * cap_m * 4) + Al((size_t)n_mb * R * 4);
}

size_t euler_gpu_sage_blocks_multi_workspace(int32_t n_mb, int64_t n, const int32_t* fanouts_host,
                                             int32_t layers) {
  size_t best = euler_gpu_sage_blocks_workspace(n, fanouts_host, layers);     // the fallback's
  for (int32_t h = 0; h < layers; ++h) {
    const size_t b = MultiHopBytes(n_mb, FlowCap(n, fanouts_host, h), fanouts_host[h], nullptr);
    if (b > best) best = b;
  }
  return best + 256;
}

int euler_gpu_sage_blocks_multi(const euler_gpu_graph* g, void* stream, uint64_t seed, uint32_t call_id,
                                uint32_t call_stride, int32_t n_mb, const uint64_t* roots_dev, int64_t n,
                                const int32_t* edge_types_host, int32_t k, const int32_t* fanouts_host,
                                int32_t layers, int64_t default_node, int32_t add_self_loops,
                                void* workspace_dev, uint64_t* const* n_id_dev,
                                int64_t* const* res_n_id_dev, int64_t* const* edge_src_dev,
                                int64_t* const* edge_dst_dev, uint32_t* counts_dev) {
  if (!g) return Fail(EULER_GPU_ENOGRAPH, "sage_blocks_multi: null graph");
  if (n_mb <= 0 || n_mb > 65535 || n < 0 || layers <= 0 || layers > 8 || k < 0 || !fanouts_host || !n_id_dev ||
      !res_n_id_dev || !edge_src_dev || !edge_dst_dev || !counts_dev || (n > 0 && (!roots_dev || !workspace_dev)))
    return Fail(EULER_GPU_EINVAL, "sage_blocks_multi: bad arguments");
  for (int32_t h = 0; h < layers; ++h)
    if (fanouts_host[h] <= 0) return Fail(EULER_GPU_EINVAL, "sage_blocks_multi: fanouts must be > 0");
  if (FlowCap(n, fanouts_host, layers) > ((int64_t)1 << 29))
    return Fail(EULER_GPU_EINVAL, "sage_blocks_multi: worst-case layer size > 2^29");
  hipStream_t st = (hipStream_t)stream;
  if (n == 0) {
    EG_HIP(hipMemsetAsync(counts_dev, 0, sizeof(uint32_t) * (size_t)n_mb * (layers + 1), st));
    return EULER_GPU_OK;
  }
  GraphView view;
  {
    const int rc = SamplingView(g, &view);
    if (rc != EULER_GPU_OK) return rc;
  }
  bool fast = k == 1 && view.monotone != 0 && view.has_zero_nbr == 0 && HasBlockSearch(view);
  for (int32_t h = 0; fast && h < layers; ++h) fast = edge_types_host[h] >= 0;
  if (!fast || n_mb == 1) {
    // hops with type draws, graphs with the id-0 sentinel rule, non-monotone rows: the separate
    // calls, one after the other on the stream (same results, the launches of M flows)
    std::vector<uint64_t*> a_n(layers);
    std::vector<int64_t*> a_r(layers), a_s(layers), a_d(layers);
    for (int32_t b = 0; b < n_mb; ++b) {
      for (int32_t h = 0; h < layers; ++h) {
        const int64_t cap_n = FlowCap(n, fanouts_host, h), cap_m = cap_n * ((int64_t)fanouts_host[h] + 1);
        a_n[h] = n_id_dev[h] + (int64_t)b * cap_m;  a_r[h] = res_n_id_dev[h] + (int64_t)b * cap_n;
        a_s[h] = edge_src_dev[h] + (int64_t)b * cap_m;  a_d[h] = edge_dst_dev[h] + (int64_t)b * cap_m;
      }
      const int rc = euler_gpu_sage_blocks(g, stream, seed, call_id + (uint32_t)b * call_stride,
                                           roots_dev + (int64_t)b * n, n, edge_types_host, k, fanouts_host,
                                           layers, default_node, add_self_loops, workspace_dev, a_n.data(),
                                           a_r.data(), a_s.data(), a_d.data(),
                                           counts_dev + (int64_t)b * (layers + 1));
      if (rc != EULER_GPU_OK) return rc;
    }
    return EULER_GPU_OK;
  }
  hipLaunchKernelGGL(FlowInitMultiKernel, dim3(1), dim3(256), 0, st, counts_dev, (uint32_t)n, n_mb, layers + 1);
  // workgroups per minibatch and kernel: the kernels stride over their work, and the launch as a
  // whole should be a few thousand workgroups, not n_mb times the worst case
  const int per_mb = 4096 / n_mb > 0 ? 4096 / n_mb : 1;
  const uint64_t* n_id = roots_dev;
  for (int32_t h = 0; h < layers; ++h) {
    const int32_t count = fanouts_host[h];
    const int64_t cap_n = FlowCap(n, fanouts_host, h);
    const int64_t cap_m = cap_n * ((int64_t)count + 1);
    uint64_t tcap = 64;
    while (tcap < (uint64_t)cap_m * 2) tcap <<= 1;
    int64_t R = 0;
    (void)MultiHopBytes(n_mb, cap_n, count, &R);
    uint8_t* p = (uint8_t*)workspace_dev;
    uint64_t* nb = (uint64_t*)p;        p += Al((size_t)n_mb * cap_n * count * 8);
    FlowHop f{};
    f.g = view; f.dense_min = nullptr; f.dense_rank = nullptr; f.epoch_hi = 0;
    f.t.keys = (unsigned long long*)p;  p += Al((size_t)n_mb * (tcap + 1) * 8);
    f.t.minpos = (uint32_t*)p;          p += Al((size_t)n_mb * (tcap + 1) * 4);
    f.t.rank = (int32_t*)p;             p += Al((size_t)n_mb * (tcap + 1) * 4);
    f.t.mask = tcap - 1;
    f.slot_of = (uint32_t*)p;           p += Al((size_t)n_mb * cap_m * 4);
    f.blk_cnt = (uint32_t*)p;           p += Al((size_t)n_mb * R * 4);
    f.n_blk = (cap_m + kFlowChunk - 1) / kFlowChunk;
    f.first_bits = (unsigned long long*)(f.blk_cnt + ((f.n_blk + 2) & ~(int64_t)1));
    f.word_pre = (uint32_t*)(f.first_bits + (cap_m / 64 + 18));
    f.n_mb = n_mb;
    f.mb_nb = cap_n * count; f.mb_nid = cap_n; f.mb_cnt = layers + 1; f.mb_tab = (int64_t)tcap + 1;
    f.mb_m = cap_m; f.mb_blk = R;
    // 1 + 2. clear, then the hop's sampler and the insert of [nb | n_id] in one kernel (a minibatch
    // per blockIdx.y); 3. ranks, the new layer, res_n_id / edge_index - the kernels of the single flow
    f.nb = nb; f.n_id = n_id; f.cnt = counts_dev + h; f.cnt_out = counts_dev + h + 1;
    f.count = count; f.self_loops = add_self_loops ? 1 : 0; f.cap_m = cap_m;
    f.new_n_id = n_id_dev[h]; f.inv = edge_dst_dev[h]; f.edge_src = edge_src_dev[h];
    f.res_n_id = res_n_id_dev[h];
    const int block = 256;
    int64_t gx = (cap_m + 1 + block - 1) / block;
    if (gx > 4 * per_mb) gx = 4 * per_mb;
    int64_t gb = f.n_blk + 1;
    if (gb > per_mb) gb = per_mb;
    const dim3 grid((unsigned)gx, (unsigned)n_mb), grid_b((unsigned)gb, (unsigned)n_mb);
    hipLaunchKernelGGL(FlowClearKernel, grid, dim3(block), 0, st, f);
    if (g_flow_fused != 0) {
      FlowSIM sm{};
      sm.h = f; sm.seed = seed; sm.call_id = call_id + (uint32_t)h; sm.call_stride = call_stride;
      sm.type = edge_types_host[(size_t)h * k]; sm.default_node = default_node;
      // (a lane draws ONE sample per trip: as many workgroups as the worst case has positions, up to 8 x per_mb)
      int64_t gs = (cap_m + block - 1) / block;
      if (gs > 8 * per_mb) gs = 8 * per_mb;
      hipLaunchKernelGGL(FlowSampleInsertMultiKernel, dim3((unsigned)gs, (unsigned)n_mb), dim3(block), 0, st, sm);
    } else {
      FlowSample fs{};
      fs.g = view; fs.seed = seed; fs.call_id = call_id + (uint32_t)h; fs.call_stride = call_stride;
      fs.n_id = n_id; fs.cnt = counts_dev + h; fs.nb = nb; fs.cap_n = cap_n; fs.mb_cnt = layers + 1;
      fs.default_node = default_node; fs.count = count; fs.type = edge_types_host[(size_t)h * k];
      int64_t gs = (cap_n * count + block - 1) / block;
      if (gs > 8 * per_mb) gs = 8 * per_mb;
      hipLaunchKernelGGL(FlowSampleKernel, dim3((unsigned)gs, (unsigned)n_mb), dim3(block), 0, st, fs);
      hipLaunchKernelGGL(FlowInsertKernel<true>, grid_b, dim3(kFlowInsertThreads), 0, st, f);
    }
    LaunchFlowRanks(f, grid_b, cap_m, 4 * per_mb, st);
    EG_HIP(hipGetLastError());
    n_id = n_id_dev[h];
  }
  return EULER_GPU_OK;
}

static void FullCaps(int64_t n, const int64_t* edge_caps, int32_t h, int64_t* cap_n, int64_t* cap_e) {
  int64_t c = n;
  for (int32_t i = 0; i < h; ++i) c += edge_caps[i];
  *cap_n = c; *cap_e = edge_caps[h];
}

size_t euler_gpu_full_blocks_workspace(int64_t n, const int64_t* edge_caps_host, int32_t layers) {
  size_t best = 0;
  for (int32_t h = 0; h < layers; ++h) {
    int64_t cap_n, cap_e;
    FullCaps(n, edge_caps_host, h, &cap_n, &cap_e);
    const int64_t cap_m = cap_e + cap_n;
    uint64_t tcap = 64;
    while (tcap < (uint64_t)cap_m * 2) tcap <<= 1;
    size_t scan_bytes = 0, scan2 = 0;
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, (uint32_t*)nullptr,
                                           (uint32_t*)nullptr, (int)(cap_m + 1), nullptr);
    (void)hipcub::DeviceScan::ExclusiveScan(nullptr, scan2, (uint32_t*)nullptr, (uint32_t*)nullptr,
                                            SatAdd(), 0u, (int)(cap_n + 1), nullptr);
    if (scan2 > scan_bytes) scan_bytes = scan2;
    const size_t b = Al(((size_t)cap_n + 1) * 4) * 2 + Al((size_t)cap_e * 8) + Al((size_t)cap_e * 4)
                     + Al((tcap + 1) * 8) + Al((tcap + 1) * 4) * 2 + Al((size_t)cap_m * 4)
                     + Al(((size_t)cap_m + 1) * 4) * 2 + Al(scan_bytes) + 512;
    if (b > best) best = b;
  }
  return best + 256;
}

// counts_dev: [layers + 1] nodes per layer, then [layers] edges per hop (without the self
// loops), then one overflow word: 0, or h + 1 for the first hop h whose edge list did not fit.  Layer h + 1 holds at most cap_n[h] + edge_caps[h] nodes.
int euler_gpu_full_blocks(const euler_gpu_graph* g, void* stream, const uint64_t* roots_dev,
                          int64_t n, const int32_t* edge_types_host, int32_t k, int32_t layers,
                          int32_t add_self_loops, const int64_t* edge_caps_host,
                          void* workspace_dev, uint64_t* const* n_id_dev,
                          int64_t* const* res_n_id_dev, int64_t* const* edge_src_dev,
                          int64_t* const* edge_dst_dev, int32_t* const* e_type_dev,
                          uint32_t* counts_dev) {
  if (!g) return Fail(EULER_GPU_ENOGRAPH, "full_blocks: null graph");
  if (n < 0 || layers <= 0 || layers > 8 || k < 0 || k > kMaxListedTypes || !edge_caps_host ||
      !n_id_dev || !res_n_id_dev || !edge_src_dev || !edge_dst_dev || !counts_dev ||
      (k > 0 && !edge_types_host) || (n > 0 && (!roots_dev || !workspace_dev)))
    return Fail(EULER_GPU_EINVAL, "full_blocks: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  EG_HIP(hipMemsetAsync(counts_dev, 0, sizeof(uint32_t) * (2 * layers + 2), st));
  if (n == 0) return EULER_GPU_OK;
  {
    int64_t cap_n, cap_e;
    FullCaps(n, edge_caps_host, layers - 1, &cap_n, &cap_e);
    if (cap_n + cap_e > ((int64_t)1 << 29))       // see sage_blocks: bit 31 of slot_of is a flag
      return Fail(EULER_GPU_EINVAL, "full_blocks: capacities > 2^29");
  }
  hipLaunchKernelGGL(FlowInitKernel, dim3(1), dim3(1), 0, st, counts_dev, (uint32_t)n);
  const uint64_t* n_id = roots_dev;
  unsigned long long* dense_min = nullptr;
  uint32_t* dense_rank = nullptr;
  uint32_t epoch0 = 0;
  // The stream's row-indexed table is shared by every block-construction call on that stream:
  // a call's hops (epochs e .. e + layers - 1) must be enqueued as one uninterrupted run, or a
  // second host thread on the same stream would slip its Insert (newer epoch) between this
  // call's Insert and Flag / Emit.  Same rule as the fanout's scratch (RunFanout).
  std::lock_guard<std::recursive_mutex> launch_lk(g->launch_mu);
  {
    const int rcd = FlowDenseTable(g, st, layers, &dense_min, &dense_rank, &epoch0);
    if (rcd != EULER_GPU_OK) return rcd;
  }
  uint32_t* overflow = counts_dev + 2 * layers + 1;
  const int block = 256;
  for (int32_t h = 0; h < layers; ++h) {
    int64_t cap_n, cap_e;
    FullCaps(n, edge_caps_host, h, &cap_n, &cap_e);
    if (cap_e < 0) return Fail(EULER_GPU_EINVAL, "full_blocks: negative capacity");
    const int64_t cap_m = cap_e + cap_n;
    uint64_t tcap = 64;
    while (tcap < (uint64_t)cap_m * 2) tcap <<= 1;
    uint8_t* p = (uint8_t*)workspace_dev;
    FlowFull ff{};
    ff.g = g->view; ff.n_id = n_id; ff.cnt = counts_dev + h; ff.k = k;
    for (int32_t x = 0; x < k; ++x) ff.et[x] = edge_types_host[(size_t)h * k + x];
    ff.lens = (uint32_t*)p;             p += Al(((size_t)cap_n + 1) * 4);
    ff.offs = (uint32_t*)p;             p += Al(((size_t)cap_n + 1) * 4);
    ff.nb = (uint64_t*)p;               p += Al((size_t)cap_e * 8);
    ff.nb_src = (int32_t*)p;            p += Al((size_t)cap_e * 4);
    ff.nb_t = e_type_dev != nullptr ? e_type_dev[h] : nullptr;
    ff.m_nb = counts_dev + layers + 1 + h; ff.overflow = overflow;
    ff.cap_n = cap_n; ff.cap_e = cap_e; ff.hop = h;
    FlowHop f{};
    f.g = g->view; f.dense_min = dense_min; f.dense_rank = dense_rank;
    f.epoch_hi = (unsigned long long)(uint32_t)~(epoch0 + (uint32_t)h) << 32;
    f.t.keys = (unsigned long long*)p;  p += Al((tcap + 1) * 8);
    f.t.minpos = (uint32_t*)p;          p += Al((tcap + 1) * 4);
    f.t.rank = (int32_t*)p;             p += Al((tcap + 1) * 4);
    f.t.mask = tcap - 1;
    f.slot_of = (uint32_t*)p;           p += Al((size_t)cap_m * 4);
    f.blk_cnt = (uint32_t*)p;           p += Al(((size_t)cap_m + 1) * 4);
    p += Al(((size_t)cap_m + 1) * 4);   // (was the counts' scan; the workspace formula keeps it)
    f.n_blk = (cap_m + kFlowChunk - 1) / kFlowChunk;
    f.first_bits = (unsigned long long*)(f.blk_cnt + ((f.n_blk + 2) & ~(int64_t)1));   // same region
    f.word_pre = (uint32_t*)(f.first_bits + (cap_m / 64 + 18));                          // same region
    void* scan_tmp = p;
    size_t scan2 = 0;
    EG_HIP(hipcub::DeviceScan::ExclusiveScan(nullptr, scan2, ff.lens, ff.offs, SatAdd(), 0u, (int)(cap_n + 1), st));
    // 1. the rows of the layer's nodes: lengths -> offsets -> the hop's neighbour list
    hipLaunchKernelGGL(FlowFullCountKernel, dim3(GridFor(cap_n + 1, block)), dim3(block), 0, st, ff);
    EG_HIP(hipcub::DeviceScan::ExclusiveScan(scan_tmp, scan2, ff.lens, ff.offs, SatAdd(), 0u, (int)(cap_n + 1), st));
    hipLaunchKernelGGL(FlowFullTotalKernel, dim3(1), dim3(1), 0, st, ff);
    if (cap_e > 0)
      hipLaunchKernelGGL(FlowFullFillKernel, dim3(GridFor(cap_e, block)), dim3(block), 0, st, ff);
    // 2. first-occurrence unique of [nb | n_id], 3. res_n_id / edge_index - the kernels of
    // the Sage flow with the list length and the edge sources read from the device
    f.nb = ff.nb; f.n_id = n_id; f.cnt = counts_dev + h; f.cnt_out = counts_dev + h + 1;
    f.m_nb_dev = ff.m_nb; f.nb_src = ff.nb_src;
    f.count = 0; f.self_loops = add_self_loops ? 1 : 0; f.cap_m = cap_m;
    f.new_n_id = n_id_dev[h]; f.inv = edge_dst_dev[h]; f.edge_src = edge_src_dev[h];
    f.res_n_id = res_n_id_dev[h];
    const int grid = GridFor(cap_m + 1, block);
    const int grid_b = (int)(f.n_blk + 1 < 65536 ? f.n_blk + 1 : 65536);
    hipLaunchKernelGGL(FlowClearKernel, dim3(grid), dim3(block), 0, st, f);
    if (f.dense_min != nullptr) hipLaunchKernelGGL(FlowInsertKernel<false>, dim3(grid_b), dim3(kFlowInsertThreads), 0, st, f);
    else hipLaunchKernelGGL(FlowInsertKernel<true>, dim3(grid_b), dim3(kFlowInsertThreads), 0, st, f);
    LaunchFlowRanks(f, dim3(grid_b), cap_m, 0, st);
    EG_HIP(hipGetLastError());
    n_id = n_id_dev[h];
  }
  return EULER_GPU_OK;
}

}  // extern "C"
