// SampleNode, GetFullNeighbor, RandomWalk / node2vec and gen_pair kernels for
// gfx950 with their C-ABI entry points.
#include <hip/hip_runtime.h>
#include <atomic>
#include <hipcub/hipcub.hpp>

#include <cmath>
#include <vector>

#include "k1_args.h"
#include "k1_search.h"
#include "fanout_local.h"     // LeanSamplePair (the walks over merged walkers)
#include "wave_sums.h"

namespace euler_gpu {

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
        if (out_t != nullptr) out_t[o + (p - b)] = t;
      }
      o += e - b;
    }
  }
}

// Balanced fill: a lane owns 4 consecutive OUTPUT entries, whatever rows they
// belong to (the wave-per-node kernel above left 60 lanes idle on the 2-edge rows
// most nodes have and one wave alone on a hub's 10^5 edges).  The row of an entry
// e is the first i with idx[2i+1] > e (row ends are non-decreasing; rows without
// neighbours have begin == end and are skipped by the search).
constexpr int kFullNbPerLane = 4;
constexpr int kFullNbWindow = 64 * kFullNbPerLane;
constexpr int kFullNbSuper = 8;      // windows per pair of row searches when the call has many entries

// first row i of [lo, hi] with idx[2 i + 1] > e (hi if none)
__device__ __forceinline__ int64_t FullNbRowOf(const int32_t* __restrict__ idx, int64_t lo, int64_t hi, int64_t e) {
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if ((int64_t)idx[2 * mid + 1] > e) hi = mid; else lo = mid + 1;
  }
  return lo;
}

// entry q of a row's listed segments, in listed order: its position in the row and its type
__device__ __forceinline__ void FullNbEntry(const FullNbArgs& a, const RowMeta& m, int32_t q, int32_t* p_out,
                                            int32_t* t_out) {
  int32_t t = 0, p = 0;
  for (int32_t x = 0; x < a.k; ++x) {
    t = a.et[x];
    if (t < 0 || t >= a.g.T) continue;
    const int32_t b = t == 0 ? 0 : m.type_end[t - 1];
    const int32_t len = m.type_end[t] - b;
    if (q < len) { p = b + q; break; }
    q -= len;
  }
  *p_out = p; *t_out = t;
}

__global__ __launch_bounds__(256) void FullNbFillBalancedKernel(
    const FullNbArgs a, const int32_t* __restrict__ idx, uint64_t* __restrict__ out_id,
    float* __restrict__ out_w, int32_t* __restrict__ out_t) {
  const int64_t total = (int64_t)idx[2 * (a.n - 1) + 1];
  const int lane = threadIdx.x & 63;
  const int wave_in_block = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  // A wave takes SUPER windows of 256 entries at a time and finds the rows of the first and last
  // entry by the scalar unit - two bisections of ~15 dependent loads each.  Calls with many
  // entries (a node2vec walk fetches 140 M a step, rows of 1 500 entries on average) take 8
  // windows per pair of searches: the searches, not the copies, were the kernel's time (1.5 ms a
  // step at 2.2 TB/s).  Small calls keep one window per wave-step (more waves than CUs).
  const int super = total >= ((int64_t)32 << 20) ? kFullNbSuper : 1;
  const int64_t span = (int64_t)kFullNbWindow * super;
  const int64_t stride = (int64_t)gridDim.x * (blockDim.x >> 6) * span;
  for (int64_t W0 = ((int64_t)blockIdx.x * (blockDim.x >> 6) + wave_in_block) * span; W0 < total; W0 += stride) {
    const int64_t W_n = W0 + span < total ? span : total - W0;
    const int64_t R0 = FullNbRowOf(idx, 0, a.n - 1, W0);
    const int64_t R1 = FullNbRowOf(idx, R0, a.n - 1, W0 + W_n - 1);
    for (int64_t w0 = W0; w0 < W0 + W_n; w0 += kFullNbWindow) {
      const int64_t w_n = w0 + kFullNbWindow < W0 + W_n ? kFullNbWindow : W0 + W_n - w0;
      int64_t r0 = R0, r1 = R1;
      if (R0 != R1 && super > 1) {
        r0 = FullNbRowOf(idx, R0, R1, w0);
        r1 = FullNbRowOf(idx, r0, R1, w0 + w_n - 1);
      }
      if (r0 == r1) {
        // (round 6) the window - or the whole super window - lies inside ONE row (a hub's): its
        // entries lane by lane, entry w0 + 64 j + lane: every load and store of the wave one
        // contiguous run, the row's record read once
        const int64_t row = FindRow(a.g, a.ids[r0]);
        const RowMeta m = LoadRowMeta(a.g, row < 0 ? 0 : row);
        const int64_t begin = idx[2 * r0];
        const float* nw = a.g.prefix_w + m.row_ptr;
        const uint64_t* nbr = a.g.nbr + m.row_ptr;
        const int64_t e_end = R0 == R1 ? W0 + W_n : w0 + w_n;
        for (int64_t e = w0 + lane; e < e_end; e += 64) {
          int32_t p, t;
          FullNbEntry(a, m, (int32_t)(e - begin), &p, &t);
          out_id[e] = nbr[p];
          out_w[e] = __fsub_rn(nw[p], p == 0 ? 0.f : nw[p - 1]);
          if (out_t != nullptr) out_t[e] = t;
        }
        if (R0 == R1) break;            // the super window is done
        continue;
      }
      const int64_t e0 = w0 + (int64_t)lane * kFullNbPerLane;
      if (e0 >= w0 + w_n) continue;
      int64_t i = FullNbRowOf(idx, r0, r1, e0);          // first row whose end exceeds e0
      int64_t row = FindRow(a.g, a.ids[i]);
      RowMeta m = LoadRowMeta(a.g, row < 0 ? 0 : row);
      int64_t begin = idx[2 * i], end = idx[2 * i + 1];
      const int64_t e1 = e0 + kFullNbPerLane < w0 + w_n ? e0 + kFullNbPerLane : w0 + w_n;
      for (int64_t e = e0; e < e1; ++e) {
        while (e >= end) {                           // next row that has entries
          ++i;
          begin = idx[2 * i]; end = idx[2 * i + 1];
          if (end > begin) { row = FindRow(a.g, a.ids[i]); m = LoadRowMeta(a.g, row); }
        }
        int32_t p, t;
        FullNbEntry(a, m, (int32_t)(e - begin), &p, &t);
        const float* nw = a.g.prefix_w + m.row_ptr;
        out_id[e] = a.g.nbr[m.row_ptr + p];
        out_w[e] = __fsub_rn(nw[p], p == 0 ? 0.f : nw[p - 1]);
        if (out_t != nullptr) out_t[e] = t;
      }
    }
  }
}

// ------------------------------------------------------------------------
// get_top_k_neighbor in one kernel (tf_euler/kernels/get_top_k_neighbor_op.cc:
// `v(nodes).outV(edge_types).order_by(weight, desc).limit(k)` filled into dense
// [n, k] tensors with default_node / 0.0 / -1).  One wave per queried node over
// the row's listed type segments in listed order - the order of
// Node::GetFullNeighbor (node.cc:175-197), whose positions break weight ties
// (the sort is stable: euler_gpu_neighbor_post_process).  Rows of up to 64
// entries are ranked with shuffles; longer rows keep 8 candidates per lane in one
// pass and merge them (k <= 8), or run k rounds of "heaviest entry after the
// previous pick".  Replaces count + fill + rank / segmented sort +
// limit + repack + to_dense (8 kernels, 3 host syncs: 2.3 ms for 131 072 roots).
// ------------------------------------------------------------------------
constexpr int kTopKLocal = 8;     // long rows: per-lane candidates kept in registers (k <= 8)

struct TopKArgs {
  GraphView g;
  const uint64_t* ids;
  int64_t n;
  int64_t default_node;
  uint64_t* out_id;
  float* out_w;
  int32_t* out_t;
  int32_t k_types;
  int32_t k;
  int32_t et[kMaxListedTypes];
};

struct TopKList {
  int64_t row_ptr;
  int32_t n_seg;
  int32_t total;
  int32_t seg_b[kMaxListedTypes];
  int32_t seg_len[kMaxListedTypes];
  int32_t seg_t[kMaxListedTypes];
};

// (row-relative position, type) of logical entry j
__device__ __forceinline__ int32_t TopKPhys(const TopKList& L, int32_t j, int32_t* t) {
  for (int32_t x = 0; x < L.n_seg; ++x) {
    if (j < L.seg_len[x]) { *t = L.seg_t[x]; return L.seg_b[x] + j; }
    j -= L.seg_len[x];
  }
  *t = -1;
  return 0;
}

__global__ __launch_bounds__(256) void TopKNeighborKernel(const TopKArgs a) {
  __shared__ TopKList lists[4];
  const int lane = threadIdx.x & 63;
  TopKList& L = lists[threadIdx.x >> 6];
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t i = wave; i < a.n; i += n_waves) {
    if (lane == 0) {
      L.n_seg = 0; L.total = 0; L.row_ptr = 0;
      const int64_t row = FindRow(a.g, a.ids[i]);
      if (row >= 0) {
        const RowMeta m = LoadRowMeta(a.g, row);
        L.row_ptr = m.row_ptr;
        for (int32_t x = 0; x < a.k_types; ++x) {
          const int32_t t = a.et[x];
          if (t < 0 || t >= a.g.T) continue;
          const int32_t b = t == 0 ? 0 : m.type_end[t - 1];
          const int32_t len = m.type_end[t] - b;
          if (len <= 0) continue;
          L.seg_b[L.n_seg] = b; L.seg_len[L.n_seg] = len; L.seg_t[L.n_seg] = t;
          ++L.n_seg;
          L.total += len;
        }
      }
    }
    WaveSync();
    const int32_t total = L.total;
    const float* nw = a.g.prefix_w + L.row_ptr;
    const uint64_t* nbr = a.g.nbr + L.row_ptr;
    const int64_t o = i * (int64_t)a.k;
    for (int32_t r = (total < a.k ? total : a.k) + lane; r < a.k; r += 64) {
      a.out_id[o + r] = (uint64_t)a.default_node;
      a.out_w[o + r] = 0.f;
      a.out_t[o + r] = -1;
    }
    if (total > 0 && total <= 64) {
      const bool live = lane < total;
      int32_t t = -1;
      const int32_t p = live ? TopKPhys(L, lane, &t) : 0;
      const float mine = live ? __fsub_rn(nw[p], p == 0 ? 0.f : nw[p - 1]) : 0.f;
      int32_t rank = 0;
      for (int32_t q = 0; q < total; ++q) {
        const float other = __shfl(mine, q);
        rank += (other > mine || (other == mine && q < lane)) ? 1 : 0;
      }
      if (live && rank < a.k) {
        a.out_id[o + rank] = nbr[p];
        a.out_w[o + rank] = mine;
        a.out_t[o + rank] = t;
      }
    } else if (total > 64 && a.k <= kTopKLocal) {
      // one pass: every lane keeps the kTopKLocal heaviest entries of its strided
      // share, sorted (weight descending, position ascending); the global top k
      // is then a 64-way merge of those lists, one wave reduction per output
      float bw[kTopKLocal];
      int32_t bj[kTopKLocal], bp[kTopKLocal], bt[kTopKLocal];
#pragma unroll
      for (int x = 0; x < kTopKLocal; ++x) { bw[x] = 0.f; bj[x] = 0x7fffffff; bp[x] = 0; bt[x] = -1; }
      for (int32_t j0 = lane; j0 < total; j0 += 256) {
        int32_t t[4], p[4];
        float w[4];
#pragma unroll
        for (int x = 0; x < 4; ++x) {
          const int32_t j = j0 + 64 * x;
          p[x] = j < total ? TopKPhys(L, j, &t[x]) : 0;
          if (j >= total) t[x] = -1;
        }
#pragma unroll
        for (int x = 0; x < 4; ++x)
          w[x] = __fsub_rn(nw[p[x]], p[x] == 0 ? 0.f : nw[p[x] - 1]);
#pragma unroll
        for (int x = 0; x < 4; ++x) {
          const int32_t j = j0 + 64 * x;
          if (j >= total) break;
          // positions ascend within a lane: an equal weight sorts AFTER the kept ones
          if (bj[kTopKLocal - 1] != 0x7fffffff && !(w[x] > bw[kTopKLocal - 1])) continue;
          float cw = w[x];
          int32_t cj = j, cp = p[x], ct = t[x];
#pragma unroll
          for (int y = 0; y < kTopKLocal; ++y) {
            const bool before = bj[y] == 0x7fffffff || cw > bw[y] || (cw == bw[y] && cj < bj[y]);
            if (before) {          // insert here, carry the displaced entry down
              const float tw = bw[y]; const int32_t tj = bj[y], tp = bp[y], tt = bt[y];
              bw[y] = cw; bj[y] = cj; bp[y] = cp; bt[y] = ct;
              cw = tw; cj = tj; cp = tp; ct = tt;
              if (cj == 0x7fffffff) break;
            }
          }
        }
      }
      int32_t head = 0;
      const int32_t rounds = total < a.k ? total : a.k;
      for (int32_t r = 0; r < rounds; ++r) {
        float best_w = 0.f;
        int32_t best_j = 0x7fffffff, best_p = 0, best_t = -1;
#pragma unroll
        for (int y = 0; y < kTopKLocal; ++y)
          if (y == head) { best_w = bw[y]; best_j = bj[y]; best_p = bp[y]; best_t = bt[y]; }
        const int32_t my_j = best_j;
        for (int off = 32; off > 0; off >>= 1) {
          const float ow = __shfl_xor(best_w, off);
          const int32_t oj = __shfl_xor(best_j, off);
          const int32_t op = __shfl_xor(best_p, off);
          const int32_t ot = __shfl_xor(best_t, off);
          const bool take = oj != 0x7fffffff &&
                            (best_j == 0x7fffffff || ow > best_w || (ow == best_w && oj < best_j));
          if (take) { best_w = ow; best_j = oj; best_p = op; best_t = ot; }
        }
        if (my_j == best_j && my_j != 0x7fffffff) ++head;      // the winner moves on
        if (lane == 0) {
          a.out_id[o + r] = nbr[best_p];
          a.out_w[o + r] = best_w;
          a.out_t[o + r] = best_t;
        }
      }
    } else if (total > 64) {
      float prev_w = 0.f;
      int32_t prev_j = -1;
      const int32_t rounds = total < a.k ? total : a.k;
      for (int32_t r = 0; r < rounds; ++r) {
        float best_w = 0.f;
        int32_t best_j = 0x7fffffff, best_p = 0, best_t = -1;
        // four entries per lane in flight (the loads of a step do not depend on
        // each other); candidates are then taken in ascending j, so ties keep the
        // first
        for (int32_t j0 = lane; j0 < total; j0 += 256) {
          int32_t t[4], p[4];
          float w[4];
#pragma unroll
          for (int x = 0; x < 4; ++x) {
            const int32_t j = j0 + 64 * x;
            p[x] = j < total ? TopKPhys(L, j, &t[x]) : 0;
            if (j >= total) t[x] = -1;
          }
#pragma unroll
          for (int x = 0; x < 4; ++x)
            w[x] = __fsub_rn(nw[p[x]], p[x] == 0 ? 0.f : nw[p[x] - 1]);
#pragma unroll
          for (int x = 0; x < 4; ++x) {
            const int32_t j = j0 + 64 * x;
            if (j >= total) break;
            const bool after_prev = r == 0 || w[x] < prev_w || (w[x] == prev_w && j > prev_j);
            const bool better = best_j == 0x7fffffff || w[x] > best_w;
            if (after_prev && better) { best_w = w[x]; best_j = j; best_p = p[x]; best_t = t[x]; }
          }
        }
        for (int off = 32; off > 0; off >>= 1) {
          const float ow = __shfl_xor(best_w, off);
          const int32_t oj = __shfl_xor(best_j, off);
          const int32_t op = __shfl_xor(best_p, off);
          const int32_t ot = __shfl_xor(best_t, off);
          const bool take = oj != 0x7fffffff &&
                            (best_j == 0x7fffffff || ow > best_w || (ow == best_w && oj < best_j));
          if (take) { best_w = ow; best_j = oj; best_p = op; best_t = ot; }
        }
        if (lane == 0) {
          a.out_id[o + r] = nbr[best_p];
          a.out_w[o + r] = best_w;
          a.out_t[o + r] = best_t;
        }
        prev_w = best_w;
        prev_j = best_j;
      }
    }
    WaveSync();          // the list is rebuilt by the next iteration
  }
}

// ------------------------------------------------------------------------
// K4  random walk.
// p = q = 1 (tf_euler/kernels/random_walk_op.cc:207-247): walk_len dependent
// count=1 hops per walker, chained on the CORE id (a missing row continues
// from the sentinel id 0); output 0 -> default_node.
// ------------------------------------------------------------------------
// key 38: DeepWalk (p = q = 1) of at least this many walkers runs over groups of merged
// walkers (CwSampleKernel ...); 0 = never
thread_local int g_walk_collapse = 262144;
thread_local int g_walk_lean = 1;         // key 44: plain graphs draw with the lean search of the one-kernel fanout
thread_local int g_walk_tail = 9;         // key 43: first step of the merged walk that stops looking for mergers
                                          // (the rest of the walk is one launch; 0 = never)
// key 69: node2vec steps on fetched rows (the sharded walk): child rows of at least this many entries
// go to a workgroup each (N2vBigStepListKernel); 0 = none.  100 000 x 10 on the metric graph, one
// rank: 122 ms without, 123 / 111 / 114 / 99 / 99 / 104 ms with 4 096 / 8 192 / 32 768 / 65 536 /
// 131 072 / 262 144.
std::atomic<int> g_n2v_list_big{65536};
// key 71: ... and so do walkers whose PARENT's row has at least this many entries: a wave moves the parent cursor
// 64 entries a (dependent) step - 9 000 steps = 6 ms on the 578 088-entry hub - a workgroup 1 024 (0 = by the child row alone)
std::atomic<int> g_n2v_list_big_parent{65536};
std::atomic<int> g_n2v_walk_tickets{1};    // key 73: the single-launch node2vec walk hands its walkers out by ticket
std::atomic<int> g_n2v_list_merged{1};     // key 72: both queues in one launch (N2vListMergedKernel); 0 = two launches
std::atomic<int> g_walk_path_ch{16};      // key 64: columns the sharded walk's path kernel parks in LDS at a time
thread_local int g_walk_grid = 1024;      // key 39: workgroups of its per-step launches (0 = one per 256 walkers)

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
  // node2vec launched step by step (tuning key 7 = 3): this launch does steps
  // [step_begin, step_end) from the paths written so far; a step whose child list
  // has big_threshold entries or more is left to N2vBigStepKernel (queued by
  // N2vClassifyKernel).  step_end == 0: all steps, nothing queued.
  int32_t step_begin;
  int32_t step_end;
  int32_t big_threshold;
  int32_t* big_queue;     // walker indices
  int32_t* big_count;     // [0] entries queued, [1] next entry to hand out; the list step also: [3] next ticket of
                          // the wave kernel
  int32_t big_parent;     // list step: parent rows of at least this many entries also go to a workgroup (0 = none)
  unsigned long long* walk_ticket;   // single-launch node2vec: the walkers' ticket counter (NULL = static assignment)
  int32_t ticket_batch;   // list step: walkers a wave takes per ticket (1 on long rows, 8 on rows of a few entries)
  // p (q) a power of two: w / p == w * inv_p in every bit (both are the correctly
  // rounded w / p); 0 = divide
  float inv_p;
  float inv_q;
  const int32_t* nonneg_flag;   // not null (node2vec step on fetched lists): device word, 1 = no fetched
                                // weight is negative or NaN
#ifdef EULER_GPU_MEASURE
  int32_t ablate;         // measurement builds only (tuning key 2): 8 = random_walk keeps its path to itself
#endif
};

// FAST: one listed edge type per step on a graph with non-decreasing running
// sums - every step is the block-pivot search of K1 (draw 0 of the current
// node, call_id + step), i.e. ~log5(deg / 10) + 4 dependent loads instead of
// the reference loop's 2 * ceil(log2 deg).
// The path of a walker is walk_len + 1 consecutive int64: a lane that stored every step
// itself issued 8-byte writes 8 * (walk_len + 1) bytes apart - a partial sector each, 14 %
// of the kernel (profiles/r2_walk_ab.txt).  The steps are staged in LDS instead
// (kWalkStage per walker, one padded row per lane) and written by the whole workgroup,
// eight lanes per walker: 64 contiguous bytes.
constexpr int kWalkStage = 8;

template <bool FAST>
__global__ __launch_bounds__(256, kWavesPerSimd) void RandomWalkKernel(const WalkArgs a) {
  __shared__ int64_t stage[256 * (kWalkStage + 1)];
  const int64_t L = a.walk_len + 1;
  const int64_t tiles = (a.n + 255) / 256;
  for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const int64_t i = tile * 256 + threadIdx.x;
    const bool live = i < a.n;
    uint64_t cur = live ? (uint64_t)a.nodes[i] : 0;
    if (live) a.out[i * L] = (int64_t)cur;
    for (int32_t s0 = 0; s0 < a.walk_len; s0 += kWalkStage) {
      const int32_t ns = min(kWalkStage, a.walk_len - s0);
      for (int32_t k = 0; k < ns; ++k) {
        const int32_t s = s0 + k;
        uint64_t id = 0; float w; int32_t t;
        if (live) {
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
        }
        stage[threadIdx.x * (kWalkStage + 1) + k] = id == 0 ? a.default_node : (int64_t)id;
        cur = id;
      }
      __syncthreads();
      // entry e = (walker e / ns of the tile, step e % ns): consecutive lanes, consecutive words
      for (int32_t e = threadIdx.x; e < 256 * ns; e += 256) {
        const int32_t wl = e / ns, k = e - wl * ns;
        const int64_t wi = tile * 256 + wl;
#ifdef EULER_GPU_MEASURE
        if (a.ablate & 8) continue;
#endif
        if (wi < a.n)
          a.out[wi * L + s0 + k + 1] = stage[wl * (kWalkStage + 1) + k];
      }
      __syncthreads();
    }
  }
}

#include "n2v_kernels.h"      // NbIter ... N2vBigStepKernel
// ------------------------------------------------------------------------
// DeepWalk with the walkers that have MERGED walked once.
//
// The draw of a step is keyed by (seed, call_id + step, node id): two walkers standing on
// the same node in the same step take the same next node - what the reference's ID_UNIQUE
// -> sample -> GATHER rewrite gives (parser/compiler.cc:76-90) - and therefore stay
// together for the rest of the walk.  On the metric graph 1M walkers are 63 % distinct
// nodes after one step, 11 % after ten, 2 % after forty: a tenth of the walker-steps
// of a walk are distinct (tools/walk_coincidence.py, profiles/r3_walk_coincidence.json).
// So the walk is run over GROUPS of merged walkers:
//   level s holds the nodes of the n[s] groups alive at step s (level 0 = the walkers) as
//                    16-byte records {node, group at level s + 1};
//   CwSampleKernel   draws every group's next node and enters the group into the owner
//                    table at that node's row (plain stores, one survivor per row - the
//                    trick of the duplicate-root path, sample_kernels.hip);
//   CwNumberKernel   the survivor of a row is its representative: representatives take
//                    the numbers of level s + 1 (one atomic per workgroup) and leave
//                    them in the table;
//   CwSampleKernel   (next step, same launch) first reads every group's number back into
//                    its record;
//   CwTailKernel     from step `tail` on (tuning key 43, default 9) the groups walk on WITHOUT
//                    looking for further mergers, all remaining steps in one launch: by then
//                    a step is ~100 K groups - two launches of ~20 + ~13 us that are mostly
//                    latency - and the mergers still to come save less than they cost.  A
//                    group's tail is left as ONE row of ids (8 bytes per step);
//   CwPathKernel     a wave takes 64 walkers: lane = walker follows the records of the merging
//                    levels (`tail` dependent 16-byte loads), then the wave copies the 64 tail
//                    rows and the heads into the op's [walker][step] layout - every path byte
//                    written once, in runs of a row (1M x 40: 225 us, sweep of `tail` in
//                    profiles/r4_walk_tail_sweep.txt);
//   CwChainKernel + CwTransposeKernel  the same through records for every level and a
//                    transposed copy (282 + 122 us) - kept for walks whose head tile does not
//                    fit in LDS (tail >= 127 steps).
//   (One kernel that staged 8 steps per walker and wrote 64-byte pieces of rows 328 bytes
//   apart took 0.65 ms for 1 M walkers x 40 steps - a third of the walk - at 0.5 TB/s.)
// Counts stay on the device; every launch is sized for the walkers and exits past n[s].
// ------------------------------------------------------------------------
struct CwArgs {
  GraphView g;
  uint64_t seed;
  const int32_t* edge_types;    // device [walk_len, k]
  uint32_t* counts;             // [walk_len + 2] groups per level
  struct Rec { uint64_t id; uint32_t next; uint32_t pad; };
  Rec* rec;                     // [walk_len + 1][cap]: the node of group g of level s and its
                                // group at level s + 1 - one 16-byte load per step for the
                                // walker that follows the chain (CwChainKernel)
  uint64_t* tmp_id[2];          // [cap] next node of every group (before numbering)
  uint32_t* tmp_slot[2];        // [cap] its owner-table slot
  uint32_t* owner[2];           // [n_rows + 1], alternating between steps
  int64_t cap;                  // walkers
  int64_t default_node;
  uint32_t call_id;
  int32_t k;
  int32_t walk_len;
  int32_t step;                 // the step this launch samples (walk_len: none)
  int32_t fast;                 // one listed type per step on a monotone graph
  uint64_t* tail_rows;          // not null: CwTailKernel leaves a group's steps as ONE row of
                                // walk_len - step ids (CwPathKernel copies it), not as records
};

constexpr uint32_t kCwFlag = 0x80000000u;

__global__ __launch_bounds__(256) void CwInitKernel(const CwArgs a, const int64_t* starts) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.cap; i += stride)
    a.rec[i].id = (uint64_t)starts[i];
}

// One step's draw for the node `cur` (draw 0 of its Philox block, call_id + step); 0 = none.
// MODE 0: the reference loop (several listed types, or running sums that are not
// monotone); 1: block-pivot search of one listed type; 2: plain graphs (one edge-type
// group, the row total in its record, identity id map, fewer than 2^31 edges) - the lean
// search of the one-kernel fanout with its interpolation start: the groups of the later
// steps stand on hubs, where it skips the upper pivot levels (fanout_local.h).
template <int MODE>
__device__ __forceinline__ uint64_t CwDraw(const CwArgs& a, const uint64_t cur, const int32_t s) {
  uint64_t id = 0;
  float w;
  int32_t t;
  if (MODE == 3) {            // plain graphs with the weight-bucket index: ONE line per draw
    const GraphView& g = a.g;
    const int64_t row = LeanFindRow(g, cur);
    WbRec wr{0u, 0u, 0u, 0.f};
    if (row >= 0 && a.edge_types[s] == 0) wr = g.wrec[row];
    const bool live = wr.deg > 0u;
    const Philox4 blk = RngBlock(a.seed, a.call_id + (uint32_t)s, kDomainNeighbor, cur, 0);
    uint64_t id2[2]; float w2[2]; uint32_t m2[2];
    WbSamplePair<false>(g, wr, live, UnitFromWords(blk.w[0], blk.w[1]), 0.0, id2, w2, m2);
    return live ? id2[0] : 0;
  }
  if (MODE == 2) {
    const GraphView& g = a.g;
    const int64_t row = LeanFindRow(g, cur);
    uint32_t lo = 0;
    int32_t deg = 0;
    float total = 0.f;
    if (row >= 0 && a.edge_types[s] == 0) {
      const uint4 rec = *reinterpret_cast<const uint4*>(g.row_meta + row * 16);
      lo = rec.x; deg = (int32_t)rec.z; total = __uint_as_float(rec.w);
    }
    const bool live = deg > 0;
    const Philox4 blk = RngBlock(a.seed, a.call_id + (uint32_t)s, kDomainNeighbor, cur, 0);
    uint64_t id2[2]; float w2[2]; uint32_t m2[2];
    LeanSamplePair<false>(g, lo, deg, total, live, UnitFromWords(blk.w[0], blk.w[1]), 0.0, id2, w2, m2);
    return live ? id2[0] : 0;
  }
  const int64_t row = FindRow(a.g, cur);
  if (MODE == 1) {
    Segment sg;
    if (LoadSegment<true>(a.g, row, a.edge_types[s], &sg)) {
      const Philox4 blk = RngBlock(a.seed, a.call_id + (uint32_t)s, kDomainNeighbor, cur, 0);
      BlockPivotSample(a.g, sg, UnitFromWords(blk.w[0], blk.w[1]), &id, &w);
    }
  } else {
    RowSampler rs;
    InitRowSampler(rs, a.g, row, a.edge_types + s * a.k, a.k);
    if (rs.valid) SampleAt(rs, a.seed, a.call_id + (uint32_t)s, cur, 0, &id, &w, &t);
  }
  return id;
}

template <int MODE>
__global__ __launch_bounds__(256, kWavesPerSimd) void CwSampleKernel(const CwArgs a) {
  const int32_t s = a.step;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t n_prev = s > 0 ? (int64_t)a.counts[s - 1] : 0;
  const int64_t n_cur = s < a.walk_len ? (int64_t)a.counts[s] : 0;
  // (the grid is a fraction of the walkers: the groups of the later steps are few, and a
  // launch of 4 096 workgroups that mostly exit costs as much as the work)
  for (int64_t gidx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
       gidx < n_prev || gidx < n_cur; gidx += stride) {
    // (1) the numbers the previous step's representatives left in its table
    if (gidx < n_prev) {
      const int par = (s - 1) & 1;
      const uint32_t sl = a.tmp_slot[par][gidx];
      // (a group whose next id has no row was numbered on its own: CwNumberKernel wrote its map)
      if (sl != (uint32_t)a.g.n_rows)
        a.rec[(int64_t)(s - 1) * a.cap + gidx].next = a.owner[par][sl] & ~kCwFlag;
    }
    // (2) this step's draw
    if (gidx >= n_cur) continue;
    const uint64_t cur = a.rec[(int64_t)s * a.cap + gidx].id;
    const uint64_t id = CwDraw<MODE>(a, cur, s);
    const int par = s & 1;
    const int64_t nrow = FindRow(a.g, id);
    const uint32_t slot = nrow < 0 ? (uint32_t)a.g.n_rows : (uint32_t)nrow;
    a.tmp_id[par][gidx] = id;
    a.tmp_slot[par][gidx] = slot;
    a.owner[par][slot] = (uint32_t)gidx;       // benign race: one group naming the row survives
  }
}

// The rest of the walk for the groups of level a.step, no more merging: group g of every later
// level is group g.  (First the numbers step a.step - 1 left in its table, as CwSampleKernel.)
template <int MODE>
__global__ __launch_bounds__(256, kWavesPerSimd) void CwTailKernel(const CwArgs a) {
  const int32_t s0 = a.step;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t n_prev = (int64_t)a.counts[s0 - 1];
  const int64_t n_cur = (int64_t)a.counts[s0];
  for (int64_t gidx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
       gidx < n_prev || gidx < n_cur; gidx += stride) {
    if (gidx < n_prev) {
      const int par = (s0 - 1) & 1;
      const uint32_t sl = a.tmp_slot[par][gidx];
      if (sl != (uint32_t)a.g.n_rows)
        a.rec[(int64_t)(s0 - 1) * a.cap + gidx].next = a.owner[par][sl] & ~kCwFlag;
    }
    if (gidx >= n_cur) continue;
    uint64_t cur = a.rec[(int64_t)s0 * a.cap + gidx].id;
    a.rec[(int64_t)s0 * a.cap + gidx].next = (uint32_t)gidx;
    const int32_t tl = a.walk_len - s0;
    for (int32_t s = s0; s < a.walk_len; ++s) {
      const uint64_t id = CwDraw<MODE>(a, cur, s);
      if (a.tail_rows != nullptr)
        a.tail_rows[gidx * tl + (s - s0)] = id;
      else
        *reinterpret_cast<uint4*>(&a.rec[(int64_t)(s + 1) * a.cap + gidx]) =
            make_uint4((uint32_t)id, (uint32_t)(id >> 32), (uint32_t)gidx, 0u);
      cur = id;
    }
  }
}

// (unknown ids share the slot n_rows: they are "no such node" for every later step, and a
// walker there keeps returning default_node - but they may be DIFFERENT ids, which the path
// must show for this step: the level keeps the id of the representative only.  Ids without
// a row are therefore numbered one by one, never merged.)
__global__ __launch_bounds__(256) void CwNumberKernel(const CwArgs a) {
  __shared__ uint32_t s_base;
  __shared__ uint32_t s_wave[4];
  const int32_t s = a.step;
  const int par = s & 1;
  const int64_t n = (int64_t)a.counts[s];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int64_t base = (int64_t)blockIdx.x * blockDim.x; base < n;
       base += (int64_t)gridDim.x * blockDim.x) {        // block-uniform trip count
    const int64_t gidx = base + threadIdx.x;
    bool rep = false;
    uint32_t slot = 0;
    if (gidx < n) {
      slot = a.tmp_slot[par][gidx];
      rep = slot == (uint32_t)a.g.n_rows || a.owner[par][slot] == (uint32_t)gidx;
    }
    const uint64_t bal = __ballot(rep);
    if (lane == 0) s_wave[wv] = (uint32_t)__popcll(bal);
    __syncthreads();
    if (threadIdx.x == 0)
      s_base = atomicAdd(&a.counts[s + 1], s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3]);
    __syncthreads();
    if (rep) {
      uint32_t nid = s_base + (uint32_t)__popcll(bal & (lane == 0 ? 0ull : (~0ull >> (64 - lane))));
      for (int x = 0; x < wv; ++x) nid += s_wave[x];
      a.rec[(int64_t)(s + 1) * a.cap + nid].id = a.tmp_id[par][gidx];
      if (slot != (uint32_t)a.g.n_rows) a.owner[par][slot] = nid | kCwFlag;
      else a.rec[(int64_t)s * a.cap + gidx].next = nid;
    }
    __syncthreads();             // s_base / s_wave are rewritten by the next round
  }
}

constexpr int kCwChains = 4;
__global__ __launch_bounds__(256, kWavesPerSimd) void CwChainKernel(const CwArgs a,
                                                                    const int64_t* starts,
                                                                    int64_t* tr) {
  const int64_t span = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < a.cap; i0 += span * kCwChains) {
    uint32_t grp[kCwChains];
    bool live[kCwChains];
#pragma unroll
    for (int u = 0; u < kCwChains; ++u) {
      const int64_t i = i0 + u * span;
      live[u] = i < a.cap;
      // the walker's group at level 1: level 0 is the walkers themselves
      grp[u] = live[u] ? a.rec[i].next : 0u;
      if (live[u]) tr[i] = starts[i];
    }
    for (int32_t s = 0; s < a.walk_len; ++s) {
      uint4 q[kCwChains];
#pragma unroll
      for (int u = 0; u < kCwChains; ++u) {
        q[u] = make_uint4(0, 0, 0, 0);
        // grp = this walker's group at level s + 1; its record gives the node and the way on
        if (live[u]) q[u] = *reinterpret_cast<const uint4*>(&a.rec[(int64_t)(s + 1) * a.cap + grp[u]]);
      }
#pragma unroll
      for (int u = 0; u < kCwChains; ++u) {
        const uint64_t id = ((uint64_t)q[u].y << 32) | q[u].x;
        if (live[u]) tr[(int64_t)(s + 1) * a.cap + i0 + u * span] = id == 0 ? a.default_node : (int64_t)id;
        grp[u] = q[u].z;
      }
    }
  }
}

// tr [L][cap] -> out [cap][L].  A workgroup takes 64 walkers and `ch` steps at a time (all
// of them when the rows fit in LDS: the 64 rows then leave as ONE contiguous run).
__global__ __launch_bounds__(256) void CwTransposeKernel(const int64_t* tr, int64_t* out, int64_t cap,
                                                         int32_t L, int32_t ch) {
  extern __shared__ __align__(16) int64_t cw_stage[];       // [64][ch | 1]
  const int32_t ls = ch | 1;
  const int lane = threadIdx.x & 63, part = threadIdx.x >> 6;
  const int64_t tiles = (cap + 63) / 64;
  for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const int64_t w0 = tile * 64;
    const int32_t nw = (int32_t)(cap - w0 < 64 ? cap - w0 : 64);
    for (int32_t c0 = 0; c0 < L; c0 += ch) {
      const int32_t ns = L - c0 < ch ? L - c0 : ch;
      for (int32_t x = part; x < ns; x += 4)
        if (lane < nw) cw_stage[lane * ls + x] = tr[(int64_t)(c0 + x) * cap + w0 + lane];
      __syncthreads();
      for (int32_t e = threadIdx.x; e < nw * ns; e += 256) {
        const int32_t wl = e / ns, x = e - wl * ns;
        out[(w0 + wl) * L + c0 + x] = cw_stage[wl * ls + x];
      }
      __syncthreads();
    }
  }
}


// The walkers' paths, written where the op wants them ([walker][step]) by ONE kernel: a wave
// takes 64 walkers.  (i) Lane = walker follows the records of the levels that still merged -
// s0 dependent 16-byte loads - and parks those s0 + 1 ids in the wave's LDS tile; it ends on
// its group of level s0, whose remaining steps CwTailKernel left as one row of ids.  (ii)
// The wave copies the 64 tail rows, consecutive lanes = consecutive ids of a row: loads and
// stores that are independent of each other and cover whole runs of a row.  (iii) The heads
// leave through the LDS tile the same way.  Against CwChainKernel + CwTransposeKernel
// (1M walkers x 40 steps: 282 + 122 us, a chain of 40 loads per walker and the paths written
// twice) the chain is s0 = 12 long and every path byte is written once.
__global__ __launch_bounds__(256) void CwPathKernel(const CwArgs a, const int64_t* __restrict__ starts,
                                                    const int32_t s0, const SmallDiv div_tl,
                                                    const SmallDiv div_h, int64_t* __restrict__ out) {
  extern __shared__ __align__(16) uint64_t cw_head[];       // per wave: [64][hs] ids, then [64] groups
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, waves = blockDim.x >> 6;
  const int32_t H = s0 + 1, hs = H | 1, TL = a.walk_len - s0, L = a.walk_len + 1;
  uint64_t* head = cw_head + (size_t)wv * (64 * hs + 32);
  uint32_t* s_grp = reinterpret_cast<uint32_t*>(head + 64 * hs);
  const uint64_t* __restrict__ rows = a.tail_rows;
  const int64_t tiles = (a.cap + 63) / 64;
  constexpr int kBatch = 8;                      // independent loads in flight per lane
  for (int64_t tile = (int64_t)blockIdx.x * waves + wv; tile < tiles; tile += (int64_t)gridDim.x * waves) {
    const int64_t w0 = tile * 64, w = w0 + lane;
    const int32_t nw = (int32_t)(a.cap - w0 < 64 ? a.cap - w0 : 64);
    const bool live = lane < nw;
    uint32_t grp = 0;
    if (live) {
      head[lane * hs] = (uint64_t)starts[w];
      grp = a.rec[w].next;                       // level 0 is the walkers themselves
    }
    for (int32_t s = 0; s < s0; ++s) {
      if (live) {
        const uint4 q = *reinterpret_cast<const uint4*>(&a.rec[(int64_t)(s + 1) * a.cap + grp]);
        head[lane * hs + s + 1] = ((uint64_t)q.y << 32) | q.x;
        grp = q.z;                               // (level s0: the group itself, CwTailKernel)
      }
    }
    s_grp[lane] = grp;
    WaveSync();
    // the tail rows: kBatch loads issued, then their stores (a load-store-load-store loop
    // would pay a full round trip per element)
    const int32_t n_tail = nw * TL;
    for (int32_t e0 = lane; e0 < n_tail; e0 += 64 * kBatch) {
      uint64_t v[kBatch];
      int64_t at[kBatch];
#pragma unroll
      for (int u = 0; u < kBatch; ++u) {
        const int32_t e = e0 + 64 * u;
        v[u] = 0; at[u] = -1;
        if (e < n_tail) {
          const int32_t wl = (int32_t)div_tl((uint32_t)e), j = e - wl * TL;
          v[u] = rows[(int64_t)s_grp[wl] * TL + j];
          at[u] = (w0 + wl) * L + H + j;
        }
      }
#pragma unroll
      for (int u = 0; u < kBatch; ++u)
        if (at[u] >= 0) out[at[u]] = v[u] == 0 ? a.default_node : (int64_t)v[u];
    }
    const int32_t n_head = nw * H;
    for (int32_t e = lane; e < n_head; e += 64) {
      const int32_t wl = (int32_t)div_h((uint32_t)e), x = e - wl * H;
      const uint64_t id = head[wl * hs + x];
      out[(w0 + wl) * L + x] = (id == 0 && x != 0) ? a.default_node : (int64_t)id;
    }
    WaveSync();                                  // the tile is rewritten by the next round
  }
}


// ---- sharded DeepWalk (csrc/sharded.cc: euler_gpu_sharded_random_walk) -----------------
// The walk over groups of merged walkers with the groups' nodes OWNED BY OTHER RANKS: level s
// holds the distinct nodes the walkers stand on at step s; a step sends them to their owners
// (front end of a sharded hop: distinct ids bucketed by owner + every entry's place among the
// answers), the OWNER draws - WalkOwnedKernel, the draw of CwSampleKernel - and the answers,
// in the order asked, are level s + 1.  The walkers themselves are touched twice: as level 0
// and by ShWalkPathKernel, which follows every walker through the levels' `next` indices.
template <int MODE>
__global__ __launch_bounds__(256, kWavesPerSimd) void WalkOwnedKernel(const CwArgs a,
                                                                      const uint64_t* __restrict__ ids,
                                                                      const int64_t n,
                                                                      uint64_t* __restrict__ out) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    out[i] = CwDraw<MODE>(a, ids[i], a.step);
}

// the same draw over SLABS of ids (the walk enqueued without host waits): word 0 of a slab = the
// number of ids behind it, the rest of the slab is padding nobody reads
template <int MODE>
__global__ __launch_bounds__(256, kWavesPerSimd) void WalkOwnedSlabKernel(const CwArgs a,
                                                                          const uint64_t* __restrict__ ids,
                                                                          const uint32_t* __restrict__ lens,
                                                                          const uint32_t stride,
                                                                          const int64_t n_pos,
                                                                          uint64_t* __restrict__ out) {
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_pos; i += step) {
    const uint32_t p = (uint32_t)i / stride, j = (uint32_t)i - p * stride;
    if (j == 0u || (uint64_t)j > (lens != nullptr ? (uint64_t)lens[p] : ids[(int64_t)p * stride])) continue;
    out[i] = CwDraw<MODE>(a, ids[i], a.step);
  }
}

constexpr int kShPathLevels = 120;  // levels whose pointers travel as kernel arguments
struct ShPathArgs {
  const int64_t* starts;            // [n] level 0
  const uint64_t* const* ids;       // device [walk_len + 1]: ids[s] = the nodes of level s (ids[0] unused)
  const int32_t* const* next;       // device [walk_len]: next[s][e] = entry e of level s at level s + 1
  int64_t* out;                     // [n, walk_len + 1]
  int64_t n, default_node;
  int32_t walk_len, ch;             // steps per LDS tile
  SmallDiv div_ch, div_last;        // by ch and by the last chunk's length
  // The walk in TWO passes (WalkPathsFromLevels): walkers that have merged share the rest of their
  // path, so the levels from T on are walked once per ENTRY of level T (n = its slab positions,
  // live_lens / live_stride say which hold an entry, column 0 = the entry's own node: map0) into
  // rows of `row_stride` words, and the walkers walk levels 0 .. T - 1 into their first columns
  // and leave the entry they reach in p_out - ShSuffixCopyKernel appends that entry's row.
  int64_t row_stride;               // words between two rows of `out` (walk_len + 1 when one pass)
  const uint32_t* live_lens;        // not null: position p = slab * live_stride + j holds a walker iff 1 <= j <= live_lens[slab]
  uint32_t live_stride;
  int32_t map0;                     // column 0: 0 -> default_node as in the other columns
  uint32_t* p_out;                  // not null: [n] the entry of level walk_len + 1 each walker reaches (next_arg[walk_len] valid)
  // walks of up to kShPathLevels steps: the tables themselves (ids / next above are null)
  const uint64_t* ids_arg[kShPathLevels + 1];
  const int32_t* next_arg[kShPathLevels];
};

// A wave takes 64 walkers: lane = walker follows its chain (one dependent 4-byte load per
// level, the level's id beside it) and parks `ch` columns at a time in the wave's LDS tile;
// the wave then writes the tile as runs of the op's [walker][step] rows.
__global__ __launch_bounds__(256) void ShWalkPathKernel(const ShPathArgs a) {
  extern __shared__ __align__(16) uint64_t sh_tile[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, waves = blockDim.x >> 6;
  const int32_t L1 = a.walk_len + 1, ls = a.ch | 1;
  uint64_t* t = sh_tile + (size_t)wv * 64 * ls;
  const int64_t tiles = (a.n + 63) / 64;
  const uint64_t* const* lvl_ids = a.ids != nullptr ? a.ids : a.ids_arg;
  const int32_t* const* lvl_next = a.next != nullptr ? a.next : a.next_arg;
  for (int64_t tile = (int64_t)blockIdx.x * waves + wv; tile < tiles; tile += (int64_t)gridDim.x * waves) {
    const int64_t w0 = tile * 64;
    const int32_t nw = (int32_t)(a.n - w0 < 64 ? a.n - w0 : 64);
    bool live = lane < nw;
    if (a.live_lens != nullptr) {
      // (a tile lies inside one slab or straddles two: whole tiles of padding leave at once)
      const uint32_t pos = (uint32_t)(w0 + lane), sl = pos / a.live_stride, j = pos - sl * a.live_stride;
      live = live && j != 0u && j <= a.live_lens[sl];
      if (__ballot(live) == 0ull) continue;
    }
    uint32_t p = (uint32_t)(w0 + lane);
    for (int32_t c0 = 0; c0 < L1; c0 += a.ch) {
      const int32_t ns = L1 - c0 < a.ch ? L1 - c0 : a.ch;
      if (live) {
        for (int32_t x = 0; x < ns; ++x) {
          const int32_t col = c0 + x;
          uint64_t v;
          if (col == 0) {
            v = (uint64_t)a.starts[w0 + lane];
            if (a.map0 != 0 && v == 0) v = (uint64_t)a.default_node;
          } else {
            p = (uint32_t)lvl_next[col - 1][p];
            if ((int32_t)p < 0) p = (uint32_t)lvl_next[col - 1][~p];    // (slab levels: ~representative)
            v = lvl_ids[col][p];
            if (v == 0) v = (uint64_t)a.default_node;
          }
          t[lane * ls + x] = v;
        }
      }
      WaveSync();
      const int32_t total = nw * ns;
      for (int32_t e = lane; e < total; e += 64) {
        const int32_t wl = (int32_t)(ns == a.ch ? a.div_ch((uint32_t)e) : a.div_last((uint32_t)e));
        const int32_t x = e - wl * ns;
        a.out[(w0 + wl) * a.row_stride + c0 + x] = (int64_t)t[wl * ls + x];
      }
      WaveSync();
    }
    if (a.p_out != nullptr && live) {
      p = (uint32_t)lvl_next[L1 - 1][p];
      if ((int32_t)p < 0) p = (uint32_t)lvl_next[L1 - 1][~p];
      a.p_out[w0 + lane] = p;
    }
  }
}

// columns first_col .. first_col + cols - 1 of walker w = the row of the level-T entry it reached:
// 2^shift lanes a walker (consecutive lanes = consecutive columns: runs of a row on both sides)
__global__ __launch_bounds__(256) void ShSuffixCopyKernel(const uint32_t* __restrict__ p_of,
                                                          const int64_t* __restrict__ suffix,
                                                          const int64_t n, const int32_t cols, const int32_t shift,
                                                          const int64_t out_stride, const int32_t first_col,
                                                          int64_t* __restrict__ out) {
  const int32_t per = 256 >> shift;                 // walkers a workgroup takes at a time
  const int32_t wl = (int32_t)threadIdx.x >> shift, c0 = (int32_t)threadIdx.x & ((1 << shift) - 1);
  for (int64_t w = (int64_t)blockIdx.x * per + wl; w < n; w += (int64_t)gridDim.x * per) {
    const int64_t row = (int64_t)p_of[w] * cols;
    for (int32_t c = c0; c < cols; c += 1 << shift) out[w * out_stride + first_col + c] = suffix[row + c];
  }
}

int WalkEdgeTypes(hipStream_t st, const int32_t* edge_types_host, int32_t k, int32_t walk_len,
                  int32_t** et_dev) {
  *et_dev = nullptr;
  const size_t et_bytes = (size_t)walk_len * (k > 0 ? k : 1) * sizeof(int32_t) + 16;
  EG_HIP(hipMallocAsync((void**)et_dev, et_bytes, st));
  if (k > 0 && walk_len > 0)
    EG_HIP(hipMemcpyAsync(*et_dev, edge_types_host, (size_t)walk_len * k * sizeof(int32_t),
                          hipMemcpyHostToDevice, st));
  return EULER_GPU_OK;
}

static int WalkOwnedLaunch(const euler_gpu_graph* g, hipStream_t st, uint64_t seed, uint32_t call_id,
                           const int32_t* et_dev, int32_t k, int32_t walk_len, int32_t step,
                           const uint64_t* ids_dev, int64_t n, const uint32_t* slab_lens, uint32_t slab_stride,
                           uint64_t* out_dev) {
  if (n <= 0) return EULER_GPU_OK;
  CwArgs c{};
  {
    const int rcv = SamplingView(g, &c.g);
    if (rcv != EULER_GPU_OK) return rcv;
  }
  c.seed = seed; c.call_id = call_id; c.edge_types = et_dev; c.k = k; c.walk_len = walk_len;
  c.step = step;
  const GraphView& v = g->view;
  const bool fast = k == 1 && v.monotone && HasBlockSearch(c.g) && g_k1_variant >= 5;
  int mode = !fast ? 0
             : (g_walk_lean != 0 && v.T == 1 && v.total_in_meta != 0 && v.map_mode == 0 &&
                v.uniform_w == 0 && v.n_edges < ((int64_t)1 << 31)) ? 2 : 1;
  if (mode == 2 && c.g.wrec != nullptr && c.g.wb != nullptr && c.g.wb_lean_ok != 0) mode = 3;
  if (mode == 2 && c.g.blk == nullptr) mode = 1;      // (the lean search of mode 2 walks the EdgeBlocks' levels)
  const int block = 256;
  unsigned grid = (unsigned)((n + block - 1) / block);
  if (grid > 4096u) grid = 4096u;
  if (slab_stride != 0u) {
    auto kern = mode == 3 ? WalkOwnedSlabKernel<3> : mode == 2 ? WalkOwnedSlabKernel<2>
                : mode == 1 ? WalkOwnedSlabKernel<1> : WalkOwnedSlabKernel<0>;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(block), 0, st, c, ids_dev, slab_lens, slab_stride, n, out_dev);
  } else {
    auto kern = mode == 3 ? WalkOwnedKernel<3> : mode == 2 ? WalkOwnedKernel<2>
                : mode == 1 ? WalkOwnedKernel<1> : WalkOwnedKernel<0>;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(block), 0, st, c, ids_dev, n, out_dev);
  }
  EG_HIP(hipGetLastError());
  return EULER_GPU_OK;
}

int WalkOwnedStep(const euler_gpu_graph* g, hipStream_t st, uint64_t seed, uint32_t call_id,
                  const int32_t* et_dev, int32_t k, int32_t walk_len, int32_t step,
                  const uint64_t* ids_dev, int64_t n, uint64_t* out_dev) {
  return WalkOwnedLaunch(g, st, seed, call_id, et_dev, k, walk_len, step, ids_dev, n, nullptr, 0u, out_dev);
}

int WalkOwnedSlabs(const euler_gpu_graph* g, hipStream_t st, uint64_t seed, uint32_t call_id,
                   const int32_t* et_dev, int32_t k, int32_t walk_len, int32_t step,
                   const uint64_t* slabs_dev, const uint32_t* lens_dev, int32_t n_slabs, uint32_t stride,
                   uint64_t* out_dev) {
  if (n_slabs <= 0 || stride == 0u || (int64_t)n_slabs * stride >= ((int64_t)1 << 32))
    return Fail(EULER_GPU_EINVAL, "walk_owned_slabs: bad arguments");
  return WalkOwnedLaunch(g, st, seed, call_id, et_dev, k, walk_len, step, slabs_dev,
                         (int64_t)n_slabs * stride, lens_dev, stride, out_dev);
}

// one launch of ShWalkPathKernel: `n` chains from `starts_dev` through levels first .. first + len
static int LaunchShPath(hipStream_t st, const int64_t* starts_dev, int64_t n, int32_t len,
                        const uint64_t* const* level_ids_host, const int32_t* const* level_next_host,
                        int64_t default_node, int64_t* out_dev, int64_t row_stride, const uint32_t* live_lens,
                        uint32_t live_stride, bool map0, uint32_t* p_out) {
  ShPathArgs a{};
  a.starts = starts_dev; a.out = out_dev;
  a.n = n; a.default_node = default_node; a.walk_len = len;
  a.row_stride = row_stride; a.live_lens = live_lens; a.live_stride = live_stride;
  a.map0 = map0 ? 1 : 0; a.p_out = p_out;
  const int32_t n_next = len + (p_out != nullptr ? 1 : 0);      // next tables the kernel reads
  void* tab = nullptr;
  if (n_next <= kShPathLevels) {
    for (int32_t s = 0; s <= len; ++s) a.ids_arg[s] = level_ids_host[s];
    for (int32_t s = 0; s < n_next; ++s) a.next_arg[s] = level_next_host[s];
  } else {
    // a long walk: the tables through device memory (the copies are waited for - their
    // sources are the caller's host arrays)
    const size_t tb = ((size_t)len + 1) * 8, tn = (size_t)n_next * 8;
    EG_HIP(hipMallocAsync(&tab, tb + tn, st));
    EG_HIP(hipMemcpyAsync(tab, level_ids_host, tb, hipMemcpyHostToDevice, st));
    EG_HIP(hipMemcpyAsync((uint8_t*)tab + tb, level_next_host, tn, hipMemcpyHostToDevice, st));
    EG_HIP(hipStreamSynchronize(st));
    a.ids = (const uint64_t* const*)tab;
    a.next = (const int32_t* const*)((uint8_t*)tab + tb);
  }
  // The kernel is a chain of walk_len dependent 4-byte loads per walker: it wants WAVES in flight,
  // not a wide tile.  16 columns at a time (128-byte runs of a walker's row, 8.7 KB of LDS a wave:
  // 16+ waves a CU) against the whole row (41 columns = 21 KB: 6 waves a CU, 585 us for 1M x 40).
  const int32_t L1 = len + 1;
  const int32_t ch_max = g_walk_path_ch.load();
  a.ch = L1 < ch_max ? L1 : ch_max;
  a.div_ch.Set((uint32_t)a.ch);
  a.div_last.Set((uint32_t)(L1 % a.ch == 0 ? a.ch : L1 % a.ch));
  const size_t wave_bytes = (size_t)64 * (a.ch | 1) * 8;
  const int waves = wave_bytes * 4 <= 64 * 1024 ? 4 : wave_bytes * 2 <= 64 * 1024 ? 2 : 1;
  const int64_t tiles = (n + 63) / 64, wgs = (tiles + waves - 1) / waves;
  hipLaunchKernelGGL(ShWalkPathKernel, dim3((unsigned)(wgs < 65536 ? wgs : 65536)), dim3(64 * waves),
                     wave_bytes * waves, st, a);
  if (tab != nullptr) (void)hipFreeAsync(tab, st);
  EG_HIP(hipGetLastError());
  return EULER_GPU_OK;
}

int WalkPathsFromLevels(hipStream_t st, const int64_t* starts_dev, int64_t n, int32_t walk_len,
                        const uint64_t* const* level_ids_host, const int32_t* const* level_next_host,
                        int64_t default_node, int64_t* out_dev) {
  if (n <= 0) return EULER_GPU_OK;
  return LaunchShPath(st, starts_dev, n, walk_len, level_ids_host, level_next_host, default_node, out_dev,
                      (int64_t)walk_len + 1, nullptr, 0u, false, nullptr);
}

// The same paths in two passes (slab levels): walkers that have merged share the rest of their
// path, and by level T a 1M-walker walk stands on ~110 K nodes - so levels T .. walk_len are walked
// once per ENTRY of level T (rows of walk_len - T + 1 nodes in scratch_dev), the walkers walk
// levels 0 .. T - 1 and append the row of the entry they reach.  The one-pass kernel makes ~83
// scattered loads per walker, 65 % of the L2's request rate for 470-490 us (1M x 40); this form
// ~37 + a 200-byte run.  scratch_dev: slab_positions * (walk_len - T + 1) * 8 + n * 4 bytes.
int WalkPathsFromLevelsTail(hipStream_t st, const int64_t* starts_dev, int64_t n, int32_t walk_len,
                            const uint64_t* const* level_ids_host, const int32_t* const* level_next_host,
                            int64_t default_node, int64_t* out_dev, int32_t T, const uint32_t* lens_T_dev,
                            uint32_t stride, int32_t n_slabs, void* scratch_dev) {
  if (n <= 0) return EULER_GPU_OK;
  if (T < 1 || T >= walk_len || scratch_dev == nullptr)
    return Fail(EULER_GPU_EINVAL, "walk_paths_tail: bad arguments");
  const int64_t slab = (int64_t)stride * n_slabs;
  const int32_t cols = walk_len - T + 1;
  int64_t* suffix = (int64_t*)scratch_dev;
  uint32_t* p_of = (uint32_t*)(suffix + slab * cols);
  // 1. the entries of level T through levels T + 1 .. walk_len (column 0 = the entry's own node)
  int rc = LaunchShPath(st, (const int64_t*)level_ids_host[T], slab, walk_len - T, level_ids_host + T,
                        level_next_host + T, default_node, suffix, cols, lens_T_dev, stride, true, nullptr);
  if (rc != EULER_GPU_OK) return rc;
  // 2. the walkers through levels 0 .. T - 1, the entry of level T they reach into p_of
  rc = LaunchShPath(st, starts_dev, n, T - 1, level_ids_host, level_next_host, default_node, out_dev,
                    (int64_t)walk_len + 1, nullptr, 0u, false, p_of);
  if (rc != EULER_GPU_OK) return rc;
  // 3. columns T .. walk_len
  int32_t shift = 0;
  while ((1 << shift) < cols && shift < 8) ++shift;
  const int64_t per = 256 >> shift;
  const int64_t wgs = (n + per - 1) / per;
  hipLaunchKernelGGL(ShSuffixCopyKernel, dim3((unsigned)(wgs < 65536 ? wgs : 65536)), dim3(256), 0, st,
                     (const uint32_t*)p_of, (const int64_t*)suffix, n, cols, shift, (int64_t)walk_len + 1, T, out_dev);
  EG_HIP(hipGetLastError());
  return EULER_GPU_OK;
}

// ---- pieces of the sharded node2vec walk (csrc/sharded.cc: euler_gpu_sharded_node2vec_walk) ----
__global__ void N2vLensKernel(const int32_t* idx, int64_t m, int32_t* lens) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < m) lens[i] = idx[2 * i + 1] - idx[2 * i];
}
__global__ void N2vIdxKernel(const int32_t* lens, const int32_t* ends, int64_t m, int32_t* idx) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < m) { idx[2 * i] = ends[i] - lens[i]; idx[2 * i + 1] = ends[i]; }
}
__global__ void N2vColumnKernel(const int64_t* src, int64_t n, int64_t stride, int64_t col, int64_t* out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i * stride + col] = src[i];
}
// the (begin, end) offsets of the FillNeighbor layout at rows bounds[0 .. w] (bounds[p] = first
// row of peer p's ids): bound_off[p] = idx[bounds[p] - 1].end, 0 for row 0
__global__ void N2vBoundsKernel(const int32_t* idx, const int64_t* bounds, int32_t w, int64_t* bound_off) {
  const int32_t p = (int32_t)threadIdx.x;
  if (p <= w) bound_off[p] = bounds[p] <= 0 ? 0 : (int64_t)idx[2 * (bounds[p] - 1) + 1];
}

// row lengths of an owner's answer (idx [m, 2] as euler_gpu_get_full_neighbor writes it)
int N2vRowLens(hipStream_t st, const int32_t* idx_dev, int64_t m, int32_t* lens_dev) {
  if (m <= 0) return EULER_GPU_OK;
  hipLaunchKernelGGL(N2vLensKernel, dim3(GridFor(m, 256)), dim3(256), 0, st, idx_dev, m, lens_dev);
  EG_HIP(hipGetLastError());
  return EULER_GPU_OK;
}

// ... and back: idx [m, 2] of the concatenated rows from their lengths (tmp_ends: m int32)
int N2vIdxFromLens(hipStream_t st, const int32_t* lens_dev, int64_t m, int32_t* tmp_ends_dev, int32_t* idx_dev) {
  if (m <= 0) return EULER_GPU_OK;
  size_t bytes = 0;
  EG_HIP(hipcub::DeviceScan::InclusiveSum(nullptr, bytes, lens_dev, tmp_ends_dev, (int)m, st));
  void* tmp = nullptr;
  EG_HIP(hipMallocAsync(&tmp, bytes + 16, st));
  const hipError_t e = hipcub::DeviceScan::InclusiveSum(tmp, bytes, lens_dev, tmp_ends_dev, (int)m, st);
  (void)hipFreeAsync(tmp, st);
  EG_HIP(e);
  hipLaunchKernelGGL(N2vIdxKernel, dim3(GridFor(m, 256)), dim3(256), 0, st, lens_dev, tmp_ends_dev, m, idx_dev);
  EG_HIP(hipGetLastError());
  return EULER_GPU_OK;
}

int N2vStoreColumn(hipStream_t st, const int64_t* src_dev, int64_t n, int64_t stride, int64_t col, int64_t* out_dev) {
  if (n <= 0) return EULER_GPU_OK;
  hipLaunchKernelGGL(N2vColumnKernel, dim3(GridFor(n, 256)), dim3(256), 0, st, src_dev, n, stride, col, out_dev);
  EG_HIP(hipGetLastError());
  return EULER_GPU_OK;
}

int N2vBoundOffsets(hipStream_t st, const int32_t* idx_dev, const int64_t* bounds_dev, int32_t w, int64_t* out_dev) {
  hipLaunchKernelGGL(N2vBoundsKernel, dim3(1), dim3(((w + 1 + 63) / 64) * 64), 0, st, idx_dev, bounds_dev, w, out_dev);
  EG_HIP(hipGetLastError());
  return EULER_GPU_OK;
}

// node2vec step over explicit lists (euler_gpu_node2vec_step): one lane per walker, the
// reference's own two passes - BuildWeights while summing, then the first running sum > r
// (Node2VecKernel above says why that is RandomSelect's index; same sequential f32 adds).
struct N2vListArgs {
  uint64_t seed;
  uint32_t call_id;
  int64_t n;
  const int32_t* c_row; const int32_t* c_idx; const uint64_t* c_ids; const float* c_w;
  const int32_t* p_row; const int32_t* p_idx; const uint64_t* p_ids;
  const int64_t* parent_ids;
  float p, q;
  int64_t default_node;
  int64_t* out;
};

// BuildWeights (random_walk_op.cc:140-168), one child at a time: the weight of child j
// given where the parent cursor stands (advanced as the reference advances it)
__device__ __forceinline__ float N2vListTake(const N2vListArgs& a, int32_t j, const float w,
                                             const int64_t cid, int32_t* pk, const int32_t pe,
                                             const int64_t parent_id) {
  (void)j;
  for (;;) {
    if (*pk >= pe) return cid != parent_id ? __fdiv_rn(w, a.q) : __fdiv_rn(w, a.p);
    const int64_t pid = (int64_t)a.p_ids[*pk];
    if (cid < pid) return cid != parent_id ? __fdiv_rn(w, a.q) : __fdiv_rn(w, a.p);
    ++*pk;
    if (cid == pid) return w;
  }
}

__global__ __launch_bounds__(256) void Node2VecListStepKernel(const N2vListArgs a) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += stride) {
    const int32_t cr = a.c_row[i];
    const int32_t cb = a.c_idx[2 * cr], ce = a.c_idx[2 * cr + 1];
    int32_t pb = 0, pe = 0;
    if (a.p_row != nullptr) { const int32_t pr = a.p_row[i]; pb = a.p_idx[2 * pr]; pe = a.p_idx[2 * pr + 1]; }
    const int64_t parent_id = a.parent_ids[i];
    int64_t sample_id = a.default_node;
    if (ce > cb) {
      float total = 0.f;
      int32_t pk = pb;
      for (int32_t j = cb; j < ce; ++j)
        total = __fadd_rn(total, N2vListTake(a, j, a.c_w[j], (int64_t)a.c_ids[j], &pk, pe, parent_id));
      const double u = RngDraw(a.seed, a.call_id, kDomainWalk, (uint64_t)i, 0);
      const double r = ScaleDraw(u, 0.f, total);
      float acc = 0.f;
      pk = pb;
      int64_t id = 0;
      for (int32_t j = cb; j < ce; ++j) {
        id = (int64_t)a.c_ids[j];
        const float w = N2vListTake(a, j, a.c_w[j], id, &pk, pe, parent_id);
        const float prev = acc;
        acc = __fadd_rn(acc, w);
        if ((double)prev <= r && r < (double)acc) break;
        // no interval holds r (total == 0): RandomSelect ends on the last element - `id`
      }
      sample_id = id;
    }
    a.out[i] = sample_id;
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

// Algorithmic bytes of a finished walk: p = q = 1 - every step
// is one SampleNeighbor(count = 1) of the walker's node: the K1 per-root and
// per-edge terms; node2vec - (deg(cur) + deg(prev)) * (8 + 4) per step, the two
// neighbour lists BuildWeights merges (random_walk_op.cc:140-168).  One lane
// per (walker, step); degrees over the listed types of that step.
struct WalkBytesArgs {
  GraphView g;
  const int64_t* walks;      // [n, walk_len + 1]
  const int32_t* edge_types; // [walk_len, k]
  int64_t n;
  int32_t walk_len, k, node2vec;
};

__device__ __forceinline__ int32_t ListedDegree(const GraphView& g, int64_t row,
                                                const int32_t* et, int32_t k) {
  if (row < 0) return 0;
  const RowMeta m = LoadRowMeta(g, row);
  const int32_t mode = TypeModeOf(k, g.T);
  if (mode == kTypeAll) return m.type_end[g.T - 1];
  int32_t deg = 0;
  for (int32_t x = 0; x < k; ++x) {
    const int32_t t = et[x];
    if (t >= 0 && t < g.T) deg += m.type_end[t] - (t == 0 ? 0 : m.type_end[t - 1]);
  }
  return deg;
}

__global__ void WalkAlgoBytesKernel(const WalkBytesArgs a, double* acc) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  double b = 0.0;
  if (idx < a.n * a.walk_len) {
    const int64_t i = idx / a.walk_len;
    const int32_t s = (int32_t)(idx - i * a.walk_len);
    const int64_t L = a.walk_len + 1;
    const uint64_t cur = (uint64_t)a.walks[i * L + s];
    const int32_t* et = a.edge_types + (size_t)s * a.k;
    const int32_t deg = ListedDegree(a.g, FindRow(a.g, cur), et, a.k);
    if (a.node2vec) {
      int32_t pdeg = 0;
      if (s > 0)
        pdeg = ListedDegree(a.g, FindRow(a.g, (uint64_t)a.walks[i * L + s - 1]),
                            a.edge_types + (size_t)(s - 1) * a.k, a.k);
      b = 12.0 * ((double)deg + (double)pdeg) + 8.0;       // + the step's output id
    } else {
      const int32_t mode = TypeModeOf(a.k, a.g.T);
      b = 8.0 + 16.0 + 4.0 * (mode == kTypeSingle ? 1 : a.g.T) + 8.0;
      double per = 16.0;
      if (deg > 0) {
        const int32_t d2 = deg < 2 ? 2 : deg;
        per += 8.0 + 8.0 + 4.0 * (double)(32 - __clz(d2 - 1));
        if (mode != kTypeSingle) {
          const int32_t t2 = a.g.T < 2 ? 2 : a.g.T;
          per += 4.0 * (double)(32 - __clz(t2 - 1)) + 8.0;
        }
      }
      b += per;
    }
  }
  for (int off = 32; off > 0; off >>= 1) b += __shfl_down(b, off, 64);
  if ((threadIdx.x & 63) == 0 && b != 0.0) atomicAdd(acc, b);
}

// The same step by the whole wave (n2v_kernels.h: the two-cursor walk with integer running
// sums), one wave per walker, the lists read from the fetched rows: what the single-GPU
// walk runs per walker and step, so the sharded walk's step costs what a step costs there
// (the lane-per-walker loop above: 20 K walkers x 10 steps on hub rows 1.1 s).
__global__ __launch_bounds__(256) void N2vNonNegKernel(const float* w, int64_t n, int32_t* flag) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  bool bad = false;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) bad = bad || !(w[i] >= 0.f);
  if (__ballot(bad) != 0ull && (threadIdx.x & 63) == 0) *flag = 0;
}

// entries of walker i's fetched child row and of its parent's row
__device__ __forceinline__ void N2vListSizes(const N2vListArgs& l, int64_t i, int32_t* nc, int32_t* np) {
  const int32_t r = l.c_row[i];
  *nc = r >= 0 ? l.c_idx[2 * (int64_t)r + 1] - l.c_idx[2 * (int64_t)r] : 0;
  const int32_t pr = l.p_row != nullptr ? l.p_row[i] : -1;
  *np = pr >= 0 && l.p_idx != nullptr ? l.p_idx[2 * (int64_t)pr + 1] - l.p_idx[2 * (int64_t)pr] : 0;
}
// does a workgroup take walker i (N2vBigStepListKernel / the first phase of N2vListMergedKernel)?
__device__ __forceinline__ bool N2vListIsBig(const WalkArgs& a, const N2vListArgs& l, int64_t i) {
  int32_t nc, np;
  N2vListSizes(l, i, &nc, &np);
  return a.big_threshold > 0 && (nc >= a.big_threshold || (a.big_parent > 0 && nc > 0 && np >= a.big_parent));
}

// The walkers are HANDED OUT (a.big_count != NULL): a wave takes a ticket per walker, in index order,
// instead of every 16 384th walker.  Which wave draws for a walker does not matter: the draw is keyed
// by the walker's index.  (Long rows first - a second queue in front of the index order - measured
// slower: 75.5 against 71.1 ms, profiles/r6_sharded_n2v_ab2.txt; removed.)
template <bool PAR>
__device__ __forceinline__ void N2vListWaveLoop(const WalkArgs& a, const N2vListArgs& l, N2vLds& S) {
  const int lane = threadIdx.x & 63;
  const int64_t waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const bool tickets = a.big_count != nullptr;
  int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  // (a.ticket_batch walkers per atomic: 100 000 single tickets on one address made the FIRST step -
  // rows of ten entries - 1.47 ms long where the static assignment took 0.31, 0.54 with 8 per ticket;
  // on rows of 1 400 entries 8 per ticket cost 3.0 ms a step against 2.5 with one)
  const int32_t batch = a.ticket_batch > 0 ? a.ticket_batch : 1;
  int32_t t_next = 0, t_end = 0;
  for (;;) {
    if (tickets) {
      if (t_next == t_end) {
        int32_t t0 = 0;
        if (lane == 0) t0 = atomicAdd(a.big_count + 3, batch);
        t_next = __builtin_amdgcn_readfirstlane(t0);
        t_end = t_next + batch;
      }
      i = t_next++;
      if (i >= l.n) break;
      if (N2vListIsBig(a, l, i)) continue;                // queued: a workgroup takes it
    } else if (i >= l.n) {
      break;
    }
    const int64_t parent = l.parent_ids[i];
    WaveSync();
    if (lane == 0) {
      N2vBuildListFetched(&S.child, l.c_idx, l.c_ids, l.c_w, l.c_row[i]);
      N2vBuildListFetched(&S.parent, l.p_idx, l.p_ids, nullptr, l.p_row != nullptr ? l.p_row[i] : -1);
    }
    WaveSync();
    const int32_t nc = S.child.total;
    const bool is_big = a.big_threshold > 0 &&
                        (nc >= a.big_threshold || (a.big_parent > 0 && nc > 0 && S.parent.total >= a.big_parent));
    if (!is_big) {                                                 // (else N2vBigStepListKernel's)
      int64_t sample_id = l.default_node;
      bool done = false;
      if (PAR && nc > 0)
        done = N2vStepParallel(a, S, lane, parent, i, 0, &sample_id, N2vSameFetched(S.child, S.parent, lane) ? 1 : 0);
      if (lane == 0 && nc > 0) { N2vCount(done ? 0 : 2, 1); N2vCount(done ? 1 : 3, (unsigned long long)nc); }
      if (nc > 0 && !done) sample_id = N2vStepSequential(a, S, lane, parent, i, 0);
      if (lane == 0) l.out[i] = sample_id;
    }
    if (!tickets) i += waves;
  }
}

template <bool PAR>
__global__ __launch_bounds__(256, kWavesPerSimd) void Node2VecListWaveKernel(const WalkArgs a,
                                                                             const N2vListArgs l) {
  __shared__ N2vLds lds_all[4];
  N2vListWaveLoop<PAR>(a, l, lds_all[threadIdx.x >> 6]);
}

// A step of the sharded walk is ONE launch over fetched rows, so its duration is its slowest
// wave's - the walker that stands on the 578 088-neighbour hub (9.6 ms a step on the metric
// graph; the single-GPU walk hides that wave behind the other walkers' remaining steps).  As
// N2vClassifyKernel / N2vBigStepKernel do for the step-by-step single-GPU walk (key 7 = 3):
// walkers whose fetched child row has big_threshold entries or more are queued, and a workgroup
// of 16 waves takes each of them (N2vBigStepBody).
__global__ __launch_bounds__(256) void N2vListClassifyKernel(const WalkArgs a, const N2vListArgs l) {
  const int lane = threadIdx.x & 63;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t n64 = (l.n + 63) & ~(int64_t)63;          // (whole waves run the loop: ballot, shuffle)
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n64; i += stride) {
    const bool big = i < l.n && N2vListIsBig(a, l, i);
    const unsigned long long m = __ballot(big);
    if (m == 0ull) continue;
    int32_t base = 0;
    if (lane == 0) base = atomicAdd(a.big_count, (int32_t)__popcll(m));
    base = __shfl(base, 0);
    if (big) a.big_queue[base + __popcll(m & ((1ull << lane) - 1ull))] = (int32_t)i;
  }
}

__global__ __launch_bounds__(64 * kN2vBigWaves) void N2vBigStepListKernel(const WalkArgs a, const N2vListArgs l) {
  __shared__ N2vBigLds S;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int32_t queued = a.big_count[0];
  int phase = 0;
  for (;;) {
    __syncthreads();
    if (threadIdx.x == 0) S.next = atomicAdd(a.big_count + 1, 1);
    __syncthreads();
    const int64_t qe = S.next;
    if (qe >= queued) break;
    const int64_t i = a.big_queue[qe];
    const int64_t parent = l.parent_ids[i];
    if (threadIdx.x == 0) {
      N2vBuildListFetched(&S.seq.child, l.c_idx, l.c_ids, l.c_w, l.c_row[i]);
      N2vBuildListFetched(&S.seq.parent, l.p_idx, l.p_ids, nullptr, l.p_row != nullptr ? l.p_row[i] : -1);
    }
    __syncthreads();
    const int same = N2vSameFetchedBlock(S.seq.child, S.seq.parent) ? 1 : 0;
    const int64_t result = N2vBigStepBody(a, S, &phase, wv, lane, parent, i, 0, same);
    if (threadIdx.x == 0) l.out[i] = result;
  }
}

// Both queues in ONE launch (key 72, the default): workgroups of 16 waves first take the long rows
// (the queue N2vListClassifyKernel filled), then their waves go on as 16 single waves over the
// tickets of the other walkers - the long rows start first, and neither launch waits for the
// other's last wave (one after the other: waves 2.1-2.4 ms + workgroups 0.95 ms a step).
__global__ __launch_bounds__(64 * kN2vBigWaves, kWavesPerSimd) void N2vListMergedKernel(const WalkArgs a,
                                                                                       const N2vListArgs l) {
  constexpr size_t kWaveBytes = sizeof(N2vLds) * kN2vBigWaves;
  constexpr size_t kBytes = sizeof(N2vBigLds) > kWaveBytes ? sizeof(N2vBigLds) : kWaveBytes;
  __shared__ __align__(16) uint8_t smem[kBytes];
  N2vBigLds& S = *reinterpret_cast<N2vBigLds*>(smem);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int32_t queued = a.big_count[0];
  int phase = 0;
  for (;;) {
    __syncthreads();
    if (threadIdx.x == 0) S.next = atomicAdd(a.big_count + 1, 1);
    __syncthreads();
    const int64_t qe = S.next;
    if (qe >= queued) break;
    const int64_t i = a.big_queue[qe];
    const int64_t parent = l.parent_ids[i];
    if (threadIdx.x == 0) {
      N2vBuildListFetched(&S.seq.child, l.c_idx, l.c_ids, l.c_w, l.c_row[i]);
      N2vBuildListFetched(&S.seq.parent, l.p_idx, l.p_ids, nullptr, l.p_row != nullptr ? l.p_row[i] : -1);
    }
    __syncthreads();
    const int same = N2vSameFetchedBlock(S.seq.child, S.seq.parent) ? 1 : 0;
    const int64_t result = N2vBigStepBody(a, S, &phase, wv, lane, parent, i, 0, same);
    if (threadIdx.x == 0) l.out[i] = result;
  }
  __syncthreads();              // (the waves' lists lie over the workgroup's)
  N2vListWaveLoop<true>(a, l, reinterpret_cast<N2vLds*>(smem)[wv]);
}

}  // namespace euler_gpu

using namespace euler_gpu;

extern "C" {

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
  if (g_full_nb_balanced != 0) {
    // the grid covers the worst case the caller's arrays can hold is unknown here: size it
    // for 16 entries per queried node and let the grid-stride loop do the rest
    const int grid = GridFor((n * 16 + kFullNbPerLane - 1) / kFullNbPerLane, block);
    hipLaunchKernelGGL(FullNbFillBalancedKernel, dim3(grid), dim3(block), 0, st, a, idx_dev,
                       out_id_dev, out_w_dev, out_t_dev);
  } else {
    const int64_t waves_needed = n;
    const int grid = GridFor(waves_needed * 64, block);
    hipLaunchKernelGGL(FullNbFillKernel, dim3(grid), dim3(block), 0, st, a, idx_dev,
                       out_id_dev, out_w_dev, out_t_dev);
  }
  EG_HIP(hipGetLastError());
  return EULER_GPU_OK;
}

int euler_gpu_get_top_k_neighbor(const euler_gpu_graph* g, void* stream,
                                 const uint64_t* ids_dev, int64_t n,
                                 const int32_t* edge_types_host, int32_t k_types, int32_t k,
                                 int64_t default_node, uint64_t* out_id_dev,
                                 float* out_w_dev, int32_t* out_t_dev) {
  if (!g) return Fail(EULER_GPU_ENOGRAPH, "get_top_k_neighbor: null graph");
  if (n < 0 || k < 0 || k_types < 0 || k_types > kMaxListedTypes)
    return Fail(EULER_GPU_EINVAL, "get_top_k_neighbor: bad n / k / edge types (<= 32)");
  if (n == 0 || k == 0) return EULER_GPU_OK;
  if (!ids_dev || !out_id_dev || !out_w_dev || !out_t_dev || (k_types > 0 && !edge_types_host))
    return Fail(EULER_GPU_EINVAL, "get_top_k_neighbor: null buffer");
  TopKArgs a{};
  a.g = g->view; a.ids = ids_dev; a.n = n; a.default_node = default_node;
  a.out_id = out_id_dev; a.out_w = out_w_dev; a.out_t = out_t_dev;
  a.k_types = k_types; a.k = k;
  for (int32_t i = 0; i < k_types; ++i) a.et[i] = edge_types_host[i];
  const int block = 256;
  hipLaunchKernelGGL(TopKNeighborKernel, dim3(GridFor(n * 64, block)), dim3(block), 0,
                     (hipStream_t)stream, a);
  EG_HIP(hipGetLastError());
  return EULER_GPU_OK;
}

// One node2vec step over EXPLICIT neighbour lists - what the reference's client does
// (tf_euler/kernels/random_walk_op.cc:83-138, RWCallback::operator()): the lists of the
// walkers' current nodes came back from the (remote) `v(nodes).outV(edge_types)` query, the
// parents' lists are those of the previous step; BuildWeights (:140-168) merges the two and
// CompactWeightedCollection draws.  No graph argument: on a sharded graph the requester runs
// this on the rows it fetched (euler_amd/distributed.py: ShardedSampler.random_walk).
// Lists are rows of a packed set: walker i's child list is row c_row[i] of (c_idx, c_ids,
// c_w), its parent's list row p_row[i] of (p_idx, p_ids) - several walkers on one node share
// a row; p_row == NULL: no parent lists yet (the first step).  Draw: domain WALK, stream =
// walker index, call_id - the single-GPU kernels' (Node2VecKernel above).
int euler_gpu_node2vec_step(void* stream, uint64_t seed, uint32_t call_id, int64_t n,
                            const int32_t* c_row_dev, const int32_t* c_idx_dev,
                            const uint64_t* c_ids_dev, const float* c_w_dev, int64_t c_entries,
                            const int32_t* p_row_dev, const int32_t* p_idx_dev,
                            const uint64_t* p_ids_dev, const int64_t* parent_ids_dev,
                            float p, float q, int64_t default_node, int64_t* out_dev) {
  if (n < 0) return Fail(EULER_GPU_EINVAL, "node2vec_step: bad n");
  if (n == 0) return EULER_GPU_OK;
  if (!c_row_dev || !c_idx_dev || !parent_ids_dev || !out_dev ||
      (p_row_dev != nullptr && !p_idx_dev))
    return Fail(EULER_GPU_EINVAL, "node2vec_step: null buffer");
  N2vListArgs a{};
  a.seed = seed; a.call_id = call_id; a.n = n;
  a.c_row = c_row_dev; a.c_idx = c_idx_dev; a.c_ids = c_ids_dev; a.c_w = c_w_dev;
  a.p_row = p_row_dev; a.p_idx = p_idx_dev; a.p_ids = p_ids_dev;
  a.parent_ids = parent_ids_dev; a.p = p; a.q = q; a.default_node = default_node;
  a.out = out_dev;
  if (c_entries < 0) return Fail(EULER_GPU_EINVAL, "node2vec_step: bad c_entries");
  if (g_n2v_wave >= 2 && c_w_dev != nullptr && c_ids_dev != nullptr) {
    // the wave kernels (tuning key 7: 2 or 3 = the whole wave walks, 1 = lane 0 walks LDS-staged
    // lists, 0 = the lane-per-walker reference loop below)
    WalkArgs w{};
    w.seed = seed; w.call_id = call_id; w.n = n; w.default_node = default_node; w.p = p; w.q = q;
    {
      int ep = 0, eq = 0;
      const bool p2 = p > 0.f && std::isfinite(p) && std::frexp(p, &ep) == 0.5f && ep > -100 && ep < 100;
      const bool q2 = q > 0.f && std::isfinite(q) && std::frexp(q, &eq) == 0.5f && eq > -100 && eq < 100;
      w.inv_p = p2 && q2 ? 1.0f / p : 0.f;
      w.inv_q = p2 && q2 ? 1.0f / q : 0.f;
    }
    // are the running sums of a step monotone (no negative weight among the fetched ones)?
    // The whole-wave path then re-runs only the chunks after a checkpoint (n2v_kernels.h).
    int32_t* flag = nullptr;
    EG_HIP(hipMallocAsync((void**)&flag, 16, (hipStream_t)stream));
    EG_HIP(hipMemsetD32Async((hipDeviceptr_t)flag, 1, 1, (hipStream_t)stream));
    if (c_entries > 0)
      hipLaunchKernelGGL(N2vNonNegKernel, dim3(GridFor(c_entries, 256)), dim3(256), 0, (hipStream_t)stream,
                         c_w_dev, c_entries, flag);
    w.nonneg_flag = flag;
    // long rows by a workgroup each (key 69: the threshold, 0 = none)
    int32_t* q = nullptr;
    const int32_t big_at = g_n2v_list_big.load();
    const bool big = big_at > 0 && n <= (1ll << 30);
    if (big) {
      const size_t q_bytes = ((size_t)n * 4 + 15) & ~(size_t)15;
      EG_HIP(hipMallocAsync((void**)&q, q_bytes + 16, (hipStream_t)stream));
      w.big_queue = q;
      w.big_count = (int32_t*)((uint8_t*)q + q_bytes);
      w.big_threshold = big_at;
      w.big_parent = g_n2v_list_big_parent.load();
      w.ticket_batch = c_entries < 32 * n ? 8 : c_entries < 256 * n ? 2 : 1;
      EG_HIP(hipMemsetAsync(w.big_count, 0, 16, (hipStream_t)stream));
      hipLaunchKernelGGL(N2vListClassifyKernel, dim3(GridFor(n, 256)), dim3(256), 0, (hipStream_t)stream, w, a);
    }
    if (big && g_n2v_list_merged.load() != 0) {
      // 2 workgroups of 1 024 threads a CU; no more workgroups than the walkers need waves
      int64_t wgs = (n + kN2vBigWaves - 1) / kN2vBigWaves;
      if (wgs > 512) wgs = 512;
      hipLaunchKernelGGL(N2vListMergedKernel, dim3((unsigned)wgs), dim3(64 * kN2vBigWaves), 0, (hipStream_t)stream, w, a);
      EG_HIP(hipGetLastError());
      EG_HIP(hipFreeAsync(flag, (hipStream_t)stream));
      EG_HIP(hipFreeAsync(q, (hipStream_t)stream));
      return EULER_GPU_OK;
    }
    hipLaunchKernelGGL(Node2VecListWaveKernel<true>, dim3(GridFor(n * 64, 256)), dim3(256), 0,
                       (hipStream_t)stream, w, a);
    // (the long rows' workgroups on a side stream, launched before or after the waves: slower at
    // every size tried - 512 workgroups of 1 024 threads hold every CU, fewer become the step's
    // tail: 96 / 192 workgroups 78 / 72 ms against 70.5, profiles/r6_sharded_n2v_ab2.txt)
    if (big)
      hipLaunchKernelGGL(N2vBigStepListKernel, dim3(512), dim3(64 * kN2vBigWaves), 0, (hipStream_t)stream, w, a);
    EG_HIP(hipGetLastError());
    EG_HIP(hipFreeAsync(flag, (hipStream_t)stream));
    if (q != nullptr) EG_HIP(hipFreeAsync(q, (hipStream_t)stream));
    return EULER_GPU_OK;
  } else {
    hipLaunchKernelGGL(Node2VecListStepKernel, dim3(GridFor(n, 256)), dim3(256), 0,
                       (hipStream_t)stream, a);
  }
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
  {
    const int rcv = SamplingView(g, &a.g);
    if (rcv != EULER_GPU_OK) { (void)hipFreeAsync(et_dev, st); return rcv; }
  }
  a.seed = seed; a.call_id = call_id; a.nodes = nodes_dev;
  a.edge_types = et_dev; a.out = out_dev; a.n = n; a.default_node = default_node;
  a.k = k; a.walk_len = walk_len; a.p = p; a.q = q;
#ifdef EULER_GPU_MEASURE
  a.ablate = g_k1_ablate;
#endif
  {
    // w / p by an exact reciprocal when p and q are powers of two (both then are the
    // correctly rounded quotient)
    int ep = 0, eq = 0;
    const bool p2 = p > 0.f && std::isfinite(p) && std::frexp(p, &ep) == 0.5f && ep > -100 && ep < 100;
    const bool q2 = q > 0.f && std::isfinite(q) && std::frexp(q, &eq) == 0.5f && eq > -100 && eq < 100;
    a.inv_p = p2 && q2 ? 1.0f / p : 0.f;
    a.inv_q = p2 && q2 ? 1.0f / q : 0.f;
  }
  const int block = 256;
  const float kEps = 1.0e-6;
  // random_walk_op.cc:281: fabs(p_ - 1.0) <= kEps && fabs(q_ - 1.0) <= kEps
  if (std::fabs((double)p - 1.0) <= kEps && std::fabs((double)q - 1.0) <= kEps) {
    const bool fast = k == 1 && g->view.monotone && HasBlockSearch(a.g) && g_k1_variant >= 5;
    // The walk over groups of merged walkers needs (walk_len + 1) * n * 16 bytes (24 when the paths go through a transposed copy) + 8 bytes per
    // graph row of stream-ordered scratch (0.7 GB for 1M walkers x 40 steps; tens of GB when
    // every node of a large graph walks).  It is an optimisation: when the scratch is not
    // to be had - more than a third of the free HBM, or the allocation fails - the call falls
    // through to the per-walker kernel, which needs none.
    bool merged = g_walk_collapse != 0 && walk_len >= 4 && n >= g_walk_collapse && n < ((int64_t)1 << 31) &&
                  g->view.n_rows < ((int64_t)1 << 31) - 2 && k > 0;
    uint8_t* buf = nullptr;
    size_t o_rec = 0, o_tid = 0, o_tsl = 0, o_own = 0, o_tr = 0, total = 0;
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    // first step of the tail (walk_len: none)
    const int32_t tail = g_walk_tail > 0 && g_walk_tail < walk_len ? g_walk_tail : walk_len;
    // the paths by CwPathKernel when a wave's tile of head ids fits in LDS (else records for
    // every level, CwChainKernel + CwTransposeKernel through a transposed copy)
    const size_t tile_bytes = ((size_t)64 * ((tail + 1) | 1) + 32) * 8;
    const bool by_path = tile_bytes <= 64 * 1024 && walk_len - tail < 8192;
    if (merged) {
      const size_t cap = (size_t)n, rows = (size_t)g->view.n_rows + 1;
      o_rec = al(((size_t)walk_len + 2) * 4);
      o_tid = o_rec + al(((size_t)walk_len + 1) * cap * 16); o_tsl = o_tid + al(2 * cap * 8);
      o_own = o_tsl + al(2 * cap * 4); o_tr = o_own + al(2 * rows * 4);
      total = o_tr + (by_path ? 0 : al(((size_t)walk_len + 1) * cap * 8));
      if (total > ((size_t)2 << 30)) {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || total > free_b / 3) merged = false;
      }
      if (merged && hipMallocAsync((void**)&buf, total, st) != hipSuccess) {
        (void)hipGetLastError();          // clear the sticky allocation error
        buf = nullptr;
        merged = false;
      }
    }
    if (merged) {
      // (CwSampleKernel ...; every error return below releases the scratch first)
      struct Release {
        uint8_t* p; hipStream_t s;
        ~Release() { if (p != nullptr) (void)hipFreeAsync(p, s); }
      } release{buf, st};
      struct ReleaseEt {
        int32_t* p; hipStream_t s; bool armed;
        ~ReleaseEt() { if (armed) (void)hipFreeAsync(p, s); }
      } release_et{et_dev, st, true};
      CwArgs c{};
      c.g = a.g; c.seed = seed; c.call_id = call_id; c.edge_types = et_dev; c.k = k;
      c.walk_len = walk_len; c.cap = n; c.default_node = default_node; c.fast = fast ? 1 : 0;
      const size_t cap = (size_t)n, rows = (size_t)g->view.n_rows + 1;
      c.counts = (uint32_t*)buf;
      c.rec = (CwArgs::Rec*)(buf + o_rec);
      c.tmp_id[0] = (uint64_t*)(buf + o_tid); c.tmp_id[1] = c.tmp_id[0] + cap;
      c.tmp_slot[0] = (uint32_t*)(buf + o_tsl); c.tmp_slot[1] = c.tmp_slot[0] + cap;
      c.owner[0] = (uint32_t*)(buf + o_own); c.owner[1] = c.owner[0] + rows;
      EG_HIP(hipMemsetAsync(c.counts, 0, ((size_t)walk_len + 2) * 4, st));
      EG_HIP(hipMemsetD32Async((hipDeviceptr_t)c.counts, (int)n, 1, st));
      hipLaunchKernelGGL(CwInitKernel, dim3(GridFor(n, block)), dim3(block), 0, st, c, nodes_dev);
      unsigned grid = (unsigned)((n + block - 1) / block);
      if (g_walk_grid > 0 && grid > (unsigned)g_walk_grid) grid = (unsigned)g_walk_grid;
      c.step = 0;
      const GraphView& v = g->view;
      int mode = !fast ? 0
                 : (g_walk_lean != 0 && v.T == 1 && v.total_in_meta != 0 && v.map_mode == 0 &&
                    v.uniform_w == 0 && v.n_edges < ((int64_t)1 << 31)) ? 2 : 1;
      if (mode == 2 && c.g.wrec != nullptr && c.g.wb != nullptr && c.g.wb_lean_ok != 0) mode = 3;
      if (mode == 2 && c.g.blk == nullptr) mode = 1;    // (the lean search of mode 2 walks the EdgeBlocks' levels)
      auto sample_kernel = mode == 3 ? CwSampleKernel<3> : mode == 2 ? CwSampleKernel<2>
                           : mode == 1 ? CwSampleKernel<1> : CwSampleKernel<0>;
      hipLaunchKernelGGL(sample_kernel, dim3(grid), dim3(block), 0, st, c);
      for (int32_t s2 = 0; s2 < tail; ++s2) {
        c.step = s2;
        hipLaunchKernelGGL(CwNumberKernel, dim3(grid), dim3(block), 0, st, c);
        c.step = s2 + 1;
        if (s2 + 1 < tail || tail == walk_len)
          hipLaunchKernelGGL(sample_kernel, dim3(grid), dim3(block), 0, st, c);
      }
      if (tail < walk_len) {
        c.step = tail;
        // (the rows take the place of the records of the levels past `tail`: 8 bytes per group
        // and step where those had 16)
        if (by_path) c.tail_rows = (uint64_t*)(c.rec + (size_t)(tail + 1) * cap);
        auto tail_kernel = mode == 3 ? CwTailKernel<3> : mode == 2 ? CwTailKernel<2>
                           : mode == 1 ? CwTailKernel<1> : CwTailKernel<0>;
        hipLaunchKernelGGL(tail_kernel, dim3(grid), dim3(block), 0, st, c);
      }
      if (by_path) {
        const int waves = tile_bytes * 4 <= 64 * 1024 ? 4 : tile_bytes * 2 <= 64 * 1024 ? 2 : 1;
        const int64_t tiles = (n + 63) / 64, wgs = (tiles + waves - 1) / waves;
        SmallDiv div_tl, div_h;                   // (exact for e * d < 2^32: e < 64 * d)
        div_tl.Set((uint32_t)(walk_len - tail)); div_h.Set((uint32_t)(tail + 1));
        hipLaunchKernelGGL(CwPathKernel, dim3((unsigned)(wgs < 65536 ? wgs : 65536)), dim3(64 * waves),
                           tile_bytes * waves, st, c, nodes_dev, tail, div_tl, div_h, out_dev);
      } else {
        int64_t* tr = (int64_t*)(buf + o_tr);
        const int64_t chain_blocks = (n + (int64_t)block * kCwChains - 1) / ((int64_t)block * kCwChains);
        hipLaunchKernelGGL(CwChainKernel, dim3((unsigned)chain_blocks), dim3(block), 0, st, c, nodes_dev, tr);
        const int32_t L = walk_len + 1;
        const int32_t ch = L <= 96 ? L : 64;                 // 64 x (ch | 1) x 8 bytes of LDS
        const size_t lds = (size_t)64 * (ch | 1) * 8;
        const int64_t tiles = (n + 63) / 64;
        hipLaunchKernelGGL(CwTransposeKernel, dim3((unsigned)(tiles < 65536 ? tiles : 65536)), dim3(block),
                           lds, st, tr, out_dev, n, L, ch);
      }
      EG_HIP(hipGetLastError());
      release_et.armed = false;       // the common exit below frees the edge-type table
    } else if (fast) {
      hipLaunchKernelGGL(RandomWalkKernel<true>, dim3(GridFor(n, block)), dim3(block), 0,
                         st, a);
    } else {
      hipLaunchKernelGGL(RandomWalkKernel<false>, dim3(GridFor(n, block)), dim3(block), 0,
                         st, a);
    }
  } else {
    if (g_n2v_wave >= 3 && walk_len > 0 && n < (1ll << 31)) {
      // step by step: classify, the short lists one wave per walker, the long ones one
      // workgroup per walker from a queue handed out by an atomic counter
      int32_t* q = nullptr;
      const size_t q_bytes = ((size_t)n * 4 + 15) & ~(size_t)15;
      EG_HIP(hipMallocAsync((void**)&q, q_bytes + (size_t)walk_len * 8, st));
      int32_t* counters = (int32_t*)((uint8_t*)q + q_bytes);
      EG_HIP(hipMemsetAsync(counters, 0, (size_t)walk_len * 8, st));
      a.big_threshold = g_n2v_big > 0 ? g_n2v_big : (1 << 30);
      a.big_queue = q;
      for (int32_t s = 0; s < walk_len; ++s) {
        a.step_begin = s; a.step_end = s + 1;
        a.big_count = counters + 2 * s;
        if (g_n2v_big > 0)
          hipLaunchKernelGGL(N2vClassifyKernel, dim3((unsigned)((n + block - 1) / block)), dim3(block),
                             0, st, a);
        hipLaunchKernelGGL(Node2VecWaveKernel<true>, dim3(GridFor(n * 64, block)), dim3(block), 0,
                           st, a);
        if (g_n2v_big > 0)
          hipLaunchKernelGGL(N2vBigStepKernel, dim3(512), dim3(64 * kN2vBigWaves), 0, st, a);
      }
      EG_HIP(hipGetLastError());
      EG_HIP(hipFreeAsync(q, st));
    } else if (g_n2v_wave >= 2) {
      unsigned long long* ticket = nullptr;
      if (g_n2v_walk_tickets.load() != 0 && n > 16384) {      // (fewer walkers than waves: one each)
        EG_HIP(hipMallocAsync((void**)&ticket, 8, st));
        EG_HIP(hipMemsetAsync(ticket, 0, 8, st));
        a.walk_ticket = ticket;
      }
      hipLaunchKernelGGL(Node2VecWaveKernel<true>, dim3(GridFor(n * 64, block)), dim3(block), 0,
                         st, a);
      if (ticket != nullptr) EG_HIP(hipFreeAsync(ticket, st));
    } else if (g_n2v_wave != 0) {
      hipLaunchKernelGGL(Node2VecWaveKernel<false>, dim3(GridFor(n * 64, block)), dim3(block), 0,
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

int euler_gpu_random_walk_stats(uint64_t* out8_host, int32_t reset) {
  if (out8_host) {
    EG_HIP(hipDeviceSynchronize());
    EG_HIP(hipMemcpyFromSymbol(out8_host, HIP_SYMBOL(g_n2v_stats), 8 * sizeof(uint64_t)));
  }
  if (reset) {
    const unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    EG_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_n2v_stats), z, sizeof(z)));
    const int on = reset == 2 ? 1 : 0;      // 2 = clear and count from now on, 1 = clear and stop
    EG_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_n2v_stats_on), &on, sizeof(on)));
  }
  return EULER_GPU_OK;
}

int euler_gpu_random_walk_algo_bytes(const euler_gpu_graph* g, void* stream,
                                     const int64_t* walks_dev, int64_t n,
                                     const int32_t* edge_types_host, int32_t k,
                                     int32_t walk_len, float p, float q, double* bytes_host) {
  if (!g) return Fail(EULER_GPU_ENOGRAPH, "random_walk_algo_bytes: null graph");
  if (n < 0 || walk_len < 0 || k < 0 || k > kMaxListedTypes || !bytes_host)
    return Fail(EULER_GPU_EINVAL, "random_walk_algo_bytes: bad arguments");
  *bytes_host = 0.0;
  if (n == 0 || walk_len == 0) return EULER_GPU_OK;
  if (!walks_dev || (k > 0 && !edge_types_host))
    return Fail(EULER_GPU_EINVAL, "random_walk_algo_bytes: null buffer");
  hipStream_t st = (hipStream_t)stream;
  uint8_t* buf = nullptr;
  const size_t et_bytes = ((size_t)walk_len * (k > 0 ? k : 1) * sizeof(int32_t) + 15) & ~(size_t)15;
  EG_HIP(hipMallocAsync((void**)&buf, et_bytes + 16, st));
  double* acc = (double*)(buf + et_bytes);
  EG_HIP(hipMemsetAsync(acc, 0, 8, st));
  if (k > 0)
    EG_HIP(hipMemcpyAsync(buf, edge_types_host, (size_t)walk_len * k * sizeof(int32_t),
                          hipMemcpyHostToDevice, st));
  WalkBytesArgs a{};
  a.g = g->view; a.walks = walks_dev; a.edge_types = (const int32_t*)buf; a.n = n;
  a.walk_len = walk_len; a.k = k;
  const float kEps = 1.0e-6;
  a.node2vec = (std::fabs((double)p - 1.0) <= kEps && std::fabs((double)q - 1.0) <= kEps) ? 0 : 1;
  const int block = 256;
  const int64_t items = n * walk_len;
  hipLaunchKernelGGL(WalkAlgoBytesKernel, dim3((unsigned)((items + block - 1) / block)), dim3(block),
                     0, st, a, acc);
  EG_HIP(hipGetLastError());
  EG_HIP(hipMemcpyAsync(bytes_host, acc, 8, hipMemcpyDeviceToHost, st));
  EG_HIP(hipStreamSynchronize(st));
  EG_HIP(hipFreeAsync(buf, st));
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
