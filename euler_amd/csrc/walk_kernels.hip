// SampleNode, GetFullNeighbor, RandomWalk / node2vec and gen_pair kernels for
// gfx950 with their C-ABI entry points.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <cmath>
#include <vector>

#include "k1_args.h"
#include "k1_search.h"
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
        out_t[o + (p - b)] = t;
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
__global__ __launch_bounds__(256) void FullNbFillBalancedKernel(
    const FullNbArgs a, const int32_t* __restrict__ idx, uint64_t* __restrict__ out_id,
    float* __restrict__ out_w, int32_t* __restrict__ out_t) {
  const int64_t total = (int64_t)idx[2 * (a.n - 1) + 1];
  const int64_t stride = (int64_t)gridDim.x * blockDim.x * kFullNbPerLane;
  for (int64_t e0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * kFullNbPerLane; e0 < total;
       e0 += stride) {
    int64_t lo = 0, hi = a.n - 1;                 // first row whose end exceeds e0
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if ((int64_t)idx[2 * mid + 1] > e0) hi = mid; else lo = mid + 1;
    }
    int64_t i = lo;
    int64_t row = FindRow(a.g, a.ids[i]);
    RowMeta m = LoadRowMeta(a.g, row < 0 ? 0 : row);
    int64_t begin = idx[2 * i], end = idx[2 * i + 1];
    const int64_t e1 = e0 + kFullNbPerLane < total ? e0 + kFullNbPerLane : total;
    for (int64_t e = e0; e < e1; ++e) {
      while (e >= end) {                           // next row that has entries
        ++i;
        begin = idx[2 * i]; end = idx[2 * i + 1];
        if (end > begin) { row = FindRow(a.g, a.ids[i]); m = LoadRowMeta(a.g, row); }
      }
      // entry q of the row's listed segments, in listed order
      int32_t q = (int32_t)(e - begin);
      int32_t t = 0, p = 0;
      for (int32_t x = 0; x < a.k; ++x) {
        t = a.et[x];
        if (t < 0 || t >= a.g.T) continue;
        const int32_t b = t == 0 ? 0 : m.type_end[t - 1];
        const int32_t len = m.type_end[t] - b;
        if (q < len) { p = b + q; break; }
        q -= len;
      }
      const float* nw = a.g.prefix_w + m.row_ptr;
      out_id[e] = a.g.nbr[m.row_ptr + p];
      out_w[e] = __fsub_rn(nw[p], p == 0 ? 0.f : nw[p - 1]);
      out_t[e] = t;
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
  int32_t* big_count;     // [0] entries queued, [1] next entry to hand out
  // p (q) a power of two: w / p == w * inv_p in every bit (both are the correctly
  // rounded w / p); 0 = divide
  float inv_p;
  float inv_q;
  int32_t ablate;         // measurement only (tuning key 2): 8 = random_walk keeps its path to itself
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

template <bool FAST, bool BLOCKED = false>
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
              if (BLOCKED) BlockedSample(a.g, sg, UnitFromWords(blk.w[0], blk.w[1]), &id, &w);
              else BlockPivotSample(a.g, sg, UnitFromWords(blk.w[0], blk.w[1]), &id, &w);
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
        if (wi < a.n && !(a.ablate & 8))
          a.out[wi * L + s0 + k + 1] = stage[wl * (kWalkStage + 1) + k];
      }
      __syncthreads();
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
__global__ __launch_bounds__(256, kWavesPerSimd) void Node2VecKernel(const WalkArgs a) {
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
constexpr int kN2vChunk = 128;
constexpr int kN2vMaxSeg = kMaxListedTypes;

struct N2vList {           // one neighbour list = listed type segments of a row
  int64_t row_ptr;         // row start in nbr / prefix_w
  int32_t n_seg;
  int32_t total;           // entries
  int32_t seg_b[kN2vMaxSeg];
  int32_t seg_len[kN2vMaxSeg];
};

constexpr int kN2vCk = 2 * kN2vChunk;   // checkpoints of the whole-wave path (they reuse the staging arrays)

struct alignas(16) N2vLds {
  union { uint64_t c_id[kN2vChunk]; float ck_acc[kN2vCk]; };
  union { uint64_t p_id[kN2vChunk]; int32_t ck_k[kN2vCk]; };
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

// The child list IS the parent list (the walker took a self loop and the steps list the
// same types): the two cursors then move in lockstep and every child is common.
__device__ __forceinline__ bool N2vSameLists(const N2vList& c, const N2vList& p) {
  if (c.total == 0 || c.total != p.total || c.row_ptr != p.row_ptr || c.n_seg != p.n_seg)
    return false;
  for (int32_t x = 0; x < c.n_seg; ++x)
    if (c.seg_b[x] != p.seg_b[x] || c.seg_len[x] != p.seg_len[x]) return false;
  return true;
}

// row-relative position of logical entry j
__device__ __forceinline__ int32_t N2vPhys(const N2vList& L, int32_t j) {
  for (int32_t x = 0; x < L.n_seg; ++x) {
    if (j < L.seg_len[x]) return L.seg_b[x] + j;
    j -= L.seg_len[x];
  }
  return 0;
}

// ------------------------------------------------------------------------
// node2vec, the two-cursor walk done by the WHOLE wave (tuning key 7 >= 2).
// BuildWeights (random_walk_op.cc:140-168) moves a parent cursor k forward only:
// child j is resolved against pn[k] -
//   cn[j] <  pn[k] : "not a common neighbour", weight / q (or / p), j++
//   cn[j] == pn[k] : common neighbour, weight kept,            j++, k++
//   cn[j] >  pn[k] : k++ and look again
// - so between two moves of k every child compares with the SAME pn[k], and a run
// of children below it is resolved by all lanes at once (one ballot).  Only a
// child that is >= pn[k] is an event: the wave then scans pn from k in 64-entry
// steps for the first entry >= that child (another ballot) and goes on.  On lists
// in storage order - what the reference's `outV` returns and what this backend's
// synthetic graphs hold - pn[k] soon sits on a large id and events are rare (3 per
// step on the metric graph, after which the parent list is used up).  Lists that
// ARE ascending make every child an event; a step whose first chunk has many is
// handed to the lane-0 automaton (same results, different speed).
//
// A lane holds kN2vR CONSECUTIVE child entries (a chunk is 64 * kN2vR entries): the
// per-chunk work - ballots, the scan across lanes, the loop - is paid once per
// four entries, and the kernel is bound by instruction issue (a 64-entry chunk
// cost ~200 wave instructions).
//
// The running sums stay the reference's sequential f32 adds, mostly without doing
// them one by one.  For a carry m * ulp in [2^e, 2^(e+1)) and carry + d below
// 2^(e+1), fl(carry + d) = (m + n) * ulp with n = d / ulp rounded to nearest - the
// same n for every m unless d / ulp ends in exactly .5 (then the tie goes to the
// even m + n and depends on m).  So with no such tie and no sum reaching 2^(e+1),
// the f32 chain IS an integer running sum of the n's over the mantissa.  n comes
// out of the adder itself: fl(2^e + d) has mantissa offset n (2^e is an even m, and
// without a tie the parity does not matter); d - (fl(2^e + d) - 2^e) is exact and
// equals +-ulp/2 exactly on a tie.  A lane the integer sum cannot pass - the sum
// leaves the binade there, a tie, a negative entry, a carry of 0 - does its entries
// by real f32 adds from its predecessor's sum and the lanes after it start over
// from there (binade crossings: ~20 per list); a chunk with more than four such
// lanes runs the add chain lane after lane (a row_shr DPP chain).  On the metric
// graph's hub rows 92 % of the 64-entry chunks need no real add at all
// (tools/n2v_binade_model.py restates the scheme in numpy against the sequential
// sums).
//
// Pass 1 runs the chunks once for the total and leaves (running sum, parent cursor)
// checkpoints in LDS - one per 2^sh chunks; the draw r then names the first
// checkpoint whose sum exceeds it, and only the chunks after the previous
// checkpoint are run again to find the entry (the sums never decrease when the
// weights, p and q are non-negative; otherwise the second pass starts from the
// first entry as the reference's scan would).  That entry is the index
// RandomSelect's bisection of the sums returns (its last element when the total
// is 0).
// ------------------------------------------------------------------------
constexpr int kN2vR = 4;
constexpr int kN2vChunkR = 64 * kN2vR;

// Diagnostic counters of the node2vec kernels (euler_gpu_random_walk_stats): 0 steps by
// the whole-wave path, 1 their child entries, 2 steps handed to the sequential
// automaton, 3 their child entries, 4 moves of the parent cursor, 5 chunks / wave-chunks
// whose running sums needed the add chain, 6 steps by the workgroup kernel, 7 their entries.
__device__ unsigned long long g_n2v_stats[8];
__device__ int g_n2v_stats_on;          // set by euler_gpu_random_walk_stats(.., reset = 2)
__device__ __forceinline__ void N2vCount(int slot, unsigned long long v) {
  if (g_n2v_stats_on) atomicAdd(&g_n2v_stats[slot], v);
}

__device__ __forceinline__ int64_t ReadLane64(int64_t v, int src) {
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, src);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)v >> 32), src);
  return (int64_t)(((uint64_t)hi << 32) | lo);
}

// The add chain: lane after lane, a lane's kN2vR entries one after the other.
// *cin = the sum before this lane's first entry; returns the sum after the last lane.
__device__ __forceinline__ float ChunkChainVec(float carry, const float (&d)[kN2vR], int lane,
                                               float* cin_out) {
  float s = 0.f, cin = 0.f, my_rin = 0.f;
#pragma unroll
  for (int row = 0; row < 4; ++row) {
    const float rin = row == 0 ? carry : ReadLaneF(s, 16 * row - 1);
    if ((lane >> 4) == row) my_rin = rin;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      // after iteration t the first t + 1 lanes of this row hold their sums; the
      // rows before it recompute theirs unchanged
      float in = __int_as_float(__builtin_amdgcn_update_dpp(
          0, __float_as_int(s), 0x111 /* row_shr:1 */, 0xf, 0xf, true));
      if ((lane & 15) == 0) in = my_rin;
      cin = in;
      float x = in;
#pragma unroll
      for (int r = 0; r < kN2vR; ++r) x = __fadd_rn(x, d[r]);
      s = x;
    }
  }
  *cin_out = cin;
  return ReadLaneF(s, 63);
}

// Running sums of one chunk: *cin = the sum before this lane's first entry (its
// entries' sums follow by kN2vR adds); returns the sum after the chunk.
__device__ __forceinline__ float WaveSumsVec(float carry, const float (&d)[kN2vR], int lane,
                                             float* cin_out) {
  const float carry0 = carry;
  float cin = 0.f, lout = 0.f;
  int start = 0;
  for (int iter = 0; iter < 4; ++iter) {
    const uint32_t cb = __float_as_uint(carry);
    const uint32_t e = cb >> 23;                     // sign bit set => e >= 256
    const bool range_ok = e >= 30u && e < 254u;
    const uint32_t bb = cb & 0xFF800000u;
    const float B = __uint_as_float(bb);
    const float twoB = __fadd_rn(B, B);
    const float half_ulp = __uint_as_float(bb - (24u << 23));
    uint32_t N = 0;
    bool okl = true;
#pragma unroll
    for (int r = 0; r < kN2vR; ++r) {
      const float t = __fadd_rn(B, d[r]);
      const float err = __fsub_rn(d[r], __fsub_rn(t, B));
      const bool ok = d[r] >= 0.f && fabsf(err) != half_ulp && t < twoB;
      okl = okl && ok;
      N += ok ? __float_as_uint(t) - bb : 0u;        // each < 2^23
    }
    const bool active = lane >= start;
    if (!active) N = 0;
    // <= (2^23 - 1) * (64 * 4 + 1): fits 32 bits unsigned
    const uint32_t off_out = (cb - bb) + WaveInclusiveAdd(N, lane);
    const unsigned long long prob =
        __ballot(active && (!range_ok || !okl || off_out >= (1u << 23)));
    const int c = prob != 0 ? __ffsll((long long)prob) - 1 : 64;
    if (active && lane < c) {
      cin = __uint_as_float(bb + off_out - N);
      lout = __uint_as_float(bb + off_out);
    }
    if (c == 64) { *cin_out = cin; return ReadLaneF(lout, 63); }
    const float cin_c = c == start ? carry : ReadLaneF(lout, c - 1);
    float x = cin_c;
#pragma unroll
    for (int r = 0; r < kN2vR; ++r) x = __fadd_rn(x, ReadLaneF(d[r], c));
    if (lane == c) { cin = cin_c; lout = x; }
    carry = x;
    start = c + 1;
    if (start == 64) { *cin_out = cin; return carry; }
  }
  if (lane == 0) N2vCount(5, 1);
  return ChunkChainVec(carry0, d, lane, cin_out);
}

// BuildWeights' weight of a child that is not a common neighbour (random_walk_op.cc:
// 152-160): w / p for the parent itself, w / q otherwise.
__device__ __forceinline__ float N2vScaled(const WalkArgs& a, float w, bool is_parent) {
  const float inv = is_parent ? a.inv_p : a.inv_q;
  if (a.inv_p != 0.f && a.inv_q != 0.f) return __fmul_rn(w, inv);
  return is_parent ? __fdiv_rn(w, a.p) : __fdiv_rn(w, a.q);
}

// A lane's kN2vR consecutive entries of the child list from logical entry jl (ids,
// and the weights as differences of the row's running sums - what `outV` hands the
// reference).
struct N2vVec {
  int64_t cid[kN2vR];
  float w[kN2vR];
  uint32_t live;                  // bit r: the entry exists
};

__device__ __forceinline__ N2vVec N2vLoadVec(const WalkArgs& a, const N2vList& L, int32_t nc,
                                             int32_t jl) {
  N2vVec v;
  v.live = 0;
  const float* c_nw = a.g.prefix_w + L.row_ptr;
  const uint64_t* c_nbr = a.g.nbr + L.row_ptr;
  if (L.n_seg == 1 && jl + kN2vR <= nc) {
    // one listed type (or one non-empty): the entries are adjacent in the row
    const int32_t ph = L.seg_b[0] + jl;
    float prev = ph == 0 ? 0.f : c_nw[ph - 1];
#pragma unroll
    for (int r = 0; r < kN2vR; ++r) {
      v.cid[r] = (int64_t)c_nbr[ph + r];
      const float x = c_nw[ph + r];
      v.w[r] = __fsub_rn(x, prev);
      prev = x;
    }
    v.live = (1u << kN2vR) - 1;
  } else {
#pragma unroll
    for (int r = 0; r < kN2vR; ++r) {
      v.cid[r] = 0;
      v.w[r] = 0.f;
      if (jl + r < nc) {
        const int32_t ph = N2vPhys(L, jl + r);
        v.cid[r] = (int64_t)c_nbr[ph];
        v.w[r] = __fsub_rn(c_nw[ph], ph == 0 ? 0.f : c_nw[ph - 1]);
        v.live |= 1u << r;
      }
    }
  }
  return v;
}

// lane-local pick of entry r.  Written as a masked OR so that it stays a select
// network: an if-chain over e.cid[q] is folded into a dynamically indexed load,
// which puts the whole struct into scratch / LDS.
__device__ __forceinline__ int64_t N2vPick(const N2vVec& e, int r) {
  uint64_t x = 0;
#pragma unroll
  for (int q = 0; q < kN2vR; ++q) x |= (uint64_t)e.cid[q] & (r == q ? ~0ull : 0ull);
  return (int64_t)x;
}

// The parent cursor: k, and pn[k] once it has been read (a run of chunks whose
// children all sit below pn[k] reads it once).
struct N2vCursor {
  int32_t k;
  int32_t m_k;      // the k that M belongs to, -1 = none
  int64_t M;
};

// BuildWeights' comparisons for one chunk: returns the mask of this lane's entries
// that are common neighbours; *events = moves of the parent cursor.
__device__ __forceinline__ uint32_t N2vEventsWave(const WalkArgs& a, const N2vList& P, int lane,
                                                  int32_t np, const N2vVec& e, N2vCursor* c,
                                                  int32_t* events_out) {
  const uint64_t* p_nbr = a.g.nbr + P.row_ptr;
  int32_t k = c->k;
  uint32_t keep = 0;
  int res_lane = -1, res_r = -1;    // entries up to (res_lane, res_r) are resolved
  int32_t events = 0;
  while (k < np) {
    if (c->m_k != k) { c->M = (int64_t)p_nbr[N2vPhys(P, k)]; c->m_k = k; }
    uint32_t evm = 0;
#pragma unroll
    for (int r = 0; r < kN2vR; ++r)
      if (((e.live >> r) & 1u) && (lane > res_lane || (lane == res_lane && r > res_r)) &&
          e.cid[r] >= c->M)
        evm |= 1u << r;
    const unsigned long long ev = __ballot(evm != 0);
    if (ev == 0) break;                       // every remaining child is below pn[k]
    const int f = __ffsll((long long)ev) - 1;
    const int rsel = evm != 0 ? __ffs((int)evm) - 1 : 0;
    const int64_t cf = ReadLane64(N2vPick(e, rsel), f);
    // first k' >= k with pn[k'] >= cf (the cursor skips the smaller entries)
    bool eq = false;
    for (;;) {
      const int32_t kk = k + lane;
      int64_t pv = 0;
      if (kk < np) pv = (int64_t)p_nbr[N2vPhys(P, kk)];
      const unsigned long long ge = __ballot(kk < np && pv >= cf);
      if (ge != 0) {
        const int g = __ffsll((long long)ge) - 1;
        k += g;
        c->M = ReadLane64(pv, g);
        c->m_k = k;
        eq = c->M == cf;
        break;
      }
      k += 64;
      if (k >= np) { k = np; break; }
    }
    if (eq) { if (lane == f) keep |= 1u << rsel; ++k; }
    res_lane = f;
    if (lane == f) res_r = rsel;
    ++events;
  }
  c->k = k;
  *events_out = events;
  if (lane == 0 && events > 0) N2vCount(4, (unsigned long long)events);
  return keep;
}

__device__ __forceinline__ void N2vWeights(const WalkArgs& a, const N2vVec& e, uint32_t keep,
                                           int64_t parent, float (&d)[kN2vR]) {
#pragma unroll
  for (int r = 0; r < kN2vR; ++r)
    d[r] = !((e.live >> r) & 1u) ? 0.f
           : ((keep >> r) & 1u)  ? e.w[r]
                                 : N2vScaled(a, e.w[r], e.cid[r] == parent);
}

// The entry whose interval [sum before, sum after) holds r, if this lane has one:
// bit r of the result.
__device__ __forceinline__ uint32_t N2vHits(const N2vVec& e, const float (&d)[kN2vR], float cin,
                                            double r) {
  uint32_t hit = 0;
  float x = cin;
#pragma unroll
  for (int q = 0; q < kN2vR; ++q) {
    const float prev = x;
    x = __fadd_rn(x, d[q]);
    if (((e.live >> q) & 1u) && (double)prev <= r && r < (double)x) hit |= 1u << q;
  }
  return hit;
}

// One step of one walker by the whole wave; returns false when the lists look
// ascending (the caller then runs the sequential automaton).  *out = sampled id.
__device__ __forceinline__ bool N2vStepParallel(const WalkArgs& a, N2vLds& S, int lane,
                                                int64_t parent, int64_t walker, int32_t step,
                                                int64_t* out) {
  const int32_t nc = S.child.total, np = S.parent.total;
  const int32_t nchunks = (nc + kN2vChunkR - 1) / kN2vChunkR;
  int32_t sh = 0;
  while ((nchunks >> sh) > kN2vCk) ++sh;
  const int32_t n_slots = nchunks >> sh;
  const bool same = N2vSameLists(S.child, S.parent);
  int32_t events = 0;
  float acc = 0.f, cin;
  float d[kN2vR];
  N2vCursor cur{0, -1, 0};
  // the next chunk's entries are requested before this chunk is worked on: a wave
  // has one dependent round trip per chunk, not two.  (Two chunks ahead costs 20
  // more registers: 23.7 ms against 18.8 on the hub workload, and the slowest
  // walker is no faster.)
  N2vVec e = N2vLoadVec(a, S.child, nc, lane * kN2vR), nx = e;
  for (int32_t ci = 0; ci < nchunks; ++ci) {
    if (ci + 1 < nchunks) nx = N2vLoadVec(a, S.child, nc, (ci + 1) * kN2vChunkR + lane * kN2vR);
    const uint32_t keep = same ? e.live : N2vEventsWave(a, S.parent, lane, np, e, &cur, &events);
    if (ci == 0 && nc >= 64 && events > 16) return false;   // ascending lists
    N2vWeights(a, e, keep, parent, d);
    acc = WaveSumsVec(acc, d, lane, &cin);
    if (((ci + 1) & ((1 << sh) - 1)) == 0 && lane == 0) {
      S.ck_acc[((ci + 1) >> sh) - 1] = acc;
      S.ck_k[((ci + 1) >> sh) - 1] = cur.k;
    }
    e = nx;
  }
  const float total = acc;
  const double u = RngDraw(a.seed, a.call_id + (uint32_t)step, kDomainWalk, (uint64_t)walker, 0);
  const double r = ScaleDraw(u, 0.f, total);
  WaveSync();
  int32_t first = 0;
  if (a.g.monotone && a.p > 0.f && a.q > 0.f) {
    first = n_slots;
    for (int32_t base = 0; base < n_slots; base += 64) {
      const int32_t idx = base + lane;
      const unsigned long long gt = __ballot(idx < n_slots && (double)S.ck_acc[idx] > r);
      if (gt != 0) { first = base + __ffsll((long long)gt) - 1; break; }
    }
  }
  acc = first == 0 ? 0.f : S.ck_acc[first - 1];
  cur.k = first == 0 ? 0 : S.ck_k[first - 1];
  cur.m_k = -1;
  bool found = false;
  int64_t result = a.default_node;
  for (int32_t ci = first << sh; ci < nchunks && !found; ++ci) {
    e = N2vLoadVec(a, S.child, nc, ci * kN2vChunkR + lane * kN2vR);
    const uint32_t keep = same ? e.live : N2vEventsWave(a, S.parent, lane, np, e, &cur, &events);
    N2vWeights(a, e, keep, parent, d);
    acc = WaveSumsVec(acc, d, lane, &cin);
    const uint32_t hm = N2vHits(e, d, cin, r);
    const unsigned long long hit = __ballot(hm != 0);
    if (hit != 0) {
      found = true;
      result = ReadLane64(N2vPick(e, hm != 0 ? __ffs((int)hm) - 1 : 0), __ffsll((long long)hit) - 1);
    }
  }
  // no interval holds r (total == 0): RandomSelect's fall-through ends on the last element
  if (!found) result = (int64_t)(a.g.nbr + S.child.row_ptr)[N2vPhys(S.child, nc - 1)];
  WaveSync();          // the checkpoints share LDS with the next step's lists
  *out = result;
  return true;
}

// One step of one walker, lists in LDS chunk by chunk, lane 0 runs the two-cursor
// recurrence (see the header comment above); every lane returns the sampled id.
__device__ __forceinline__ int64_t N2vStepSequential(const WalkArgs& a, N2vLds& S, int lane,
                                                     int64_t parent, int64_t walker,
                                                     int32_t step) {
  const int32_t nc = S.child.total, np = S.parent.total;
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
      const double u = RngDraw(a.seed, a.call_id + (uint32_t)step, kDomainWalk,
                               (uint64_t)walker, 0);
      r = ScaleDraw(u, 0.f, total);
    }
  }
  // found: last_id is the hit; not found (total == 0): RandomSelect's
  // fall-through ends on the last element, which is last_id as well
  const uint32_t lo32 = __shfl((uint32_t)last_id, 0);
  const uint32_t hi32 = __shfl((uint32_t)(last_id >> 32), 0);
  return (int64_t)(((uint64_t)hi32 << 32) | lo32);
}

// 64 registers (8 waves per SIMD) and 12 KB of LDS per workgroup: 18.8 ms against
// 20.4 with 82 registers on the hub workload.
template <bool PAR>
__global__ __launch_bounds__(256, kWavesPerSimd) void Node2VecWaveKernel(const WalkArgs a) {
  __shared__ N2vLds lds_all[4];
  N2vLds& S = lds_all[threadIdx.x >> 6];
  const int lane = threadIdx.x & 63;
  const int64_t waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const int64_t L = a.walk_len + 1;
  const int32_t s_end = a.step_end > 0 ? a.step_end : a.walk_len;
  for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; i < a.n;
       i += waves) {
    int64_t cur, parent;
    bool have_parent_nb;
    if (a.step_begin == 0) {
      cur = a.nodes[i];
      parent = cur;                // parent_ids_ starts as the start nodes
      have_parent_nb = false;      // parent_neighbors_ starts empty
      if (lane == 0) a.out[i * L] = cur;
    } else {
      cur = a.out[i * L + a.step_begin];
      parent = a.out[i * L + a.step_begin - 1];
      have_parent_nb = true;
    }
    for (int32_t s = a.step_begin; s < s_end; ++s) {
      const int32_t* et = a.edge_types + s * a.k;
      const int32_t* pet = s > 0 ? a.edge_types + (s - 1) * a.k : et;
      WaveSync();
      if (lane == 0) {
        N2vBuildList(&S.child, a.g, FindRow(a.g, (uint64_t)cur), et, a.k);
        N2vBuildList(&S.parent, a.g,
                     have_parent_nb ? FindRow(a.g, (uint64_t)parent) : -1, pet, a.k);
      }
      WaveSync();
      const int32_t nc = S.child.total;
      if (a.big_threshold > 0 && nc >= a.big_threshold) break;   // N2vBigStepKernel's
      int64_t sample_id = a.default_node;
      bool done = false;
      if (PAR && nc > 0) done = N2vStepParallel(a, S, lane, parent, i, s, &sample_id);
      if (lane == 0 && nc > 0) { N2vCount(done ? 0 : 2, 1); N2vCount(done ? 1 : 3, (unsigned long long)nc); }
      if (nc > 0 && !done) sample_id = N2vStepSequential(a, S, lane, parent, i, s);
      if (lane == 0) a.out[i * L + s + 1] = sample_id;
      parent = cur;
      have_parent_nb = true;
      cur = sample_id;
    }
  }
}

// ------------------------------------------------------------------------
// node2vec step by step (tuning key 7 = 3).  A walker's step is one wave's serial
// work and the metric graph has rows of 5e5 neighbours: one walker in 10^5 spends
// 12 ms on its ten steps while the rest of the chip has long finished
// (tools/prof_n2v.py: 1 000 walkers take 12.6 ms, 100 000 take 37 ms).  So the walk
// is launched per step, and a step whose child list is long goes to a WORKGROUP of
// 16 waves: 4 096 entries per round, the parent cursor and the running sum carried
// across the waves through LDS.
//   * Parent cursor: all lanes compare with the same pn[k]; the first entry that is
//     not below it is found with one ballot per wave and one LDS exchange, the
//     cursor scan runs 1 024 parent entries per round.
//   * Running sums: every wave sums its 256 entries in the binade of the round's
//     carry (WaveSumsVec's integer path), the 16 totals are exchanged and each wave
//     adds what lies before it.  The waves before the first one that cannot do that
//     (a tie, a negative entry, the sum leaving the binade inside it) are final;
//     that wave runs its entries from its real carry and the rest start over from
//     its last sum.
// ------------------------------------------------------------------------
constexpr int kN2vBigWaves = 16;
constexpr int kN2vBigCk = 1024;
constexpr int kN2vBigRound = kN2vBigWaves * kN2vChunkR;

struct alignas(16) N2vBigLds {
  N2vLds seq;                                   // lists + staging of the sequential automaton
  unsigned long long x_mask[2][kN2vBigWaves];   // exchange slots, alternating
  int64_t x_val[2][kN2vBigWaves];
  float hand;                                   // the restarting wave's last sum
  int64_t next;                                 // queue entry of this workgroup
  float ck_acc[kN2vBigCk];
  int32_t ck_k[kN2vBigCk];
};

// Every wave contributes a lane mask and the value of its first set lane; returns the
// workgroup-wide index (wave * 64 + lane) of the first set lane, -1 if none, and that
// lane's value.  One barrier; consecutive calls use alternate slots, so a wave that
// runs ahead writes the slots nobody reads any more.
__device__ __forceinline__ int32_t N2vBigFirst(N2vBigLds& S, int* phase, int wv, int lane,
                                               unsigned long long mask, int64_t v,
                                               int64_t* v_out) {
  const int b = *phase & 1;
  ++*phase;
  if (lane == 0) S.x_mask[b][wv] = mask;
  if (mask != 0 && lane == __ffsll((long long)mask) - 1) S.x_val[b][wv] = v;
  __syncthreads();
  const unsigned long long mine = lane < kN2vBigWaves ? S.x_mask[b][lane] : 0ull;
  const unsigned long long nz = __ballot(mine != 0);
  if (nz == 0) return -1;
  const int w = __ffsll((long long)nz) - 1;
  const unsigned long long m = S.x_mask[b][w];
  *v_out = S.x_val[b][w];
  return w * 64 + __ffsll((long long)m) - 1;
}

struct N2vBigState {
  N2vCursor cur;
  float acc;        // running sum before this round
};

// One round: d[] = this lane's weights after BuildWeights' comparisons, *cin = the
// running sum before this lane's first entry.  Workgroup-uniform control flow.
__device__ __forceinline__ void N2vBigRound(const WalkArgs& a, N2vBigLds& S, int* phase, int wv,
                                            int lane, int64_t parent, int32_t np,
                                            bool same_lists, const N2vVec& e, N2vBigState* st,
                                            float (&d)[kN2vR], float* cin_out,
                                            int32_t* events_out) {
  const int tid = wv * 64 + lane;
  const uint64_t* p_nbr = a.g.nbr + S.seq.parent.row_ptr;
  uint32_t keep = same_lists ? e.live : 0u;
  int res_tid = -1, res_r = -1;
  int32_t events = 0;
  int32_t k = st->cur.k;
  while (!same_lists && k < np) {
    if (st->cur.m_k != k) {
      st->cur.M = (int64_t)p_nbr[N2vPhys(S.seq.parent, k)];
      st->cur.m_k = k;
    }
    uint32_t evm = 0;
#pragma unroll
    for (int r = 0; r < kN2vR; ++r)
      if (((e.live >> r) & 1u) && (tid > res_tid || (tid == res_tid && r > res_r)) &&
          e.cid[r] >= st->cur.M)
        evm |= 1u << r;
    const int rsel = evm != 0 ? __ffs((int)evm) - 1 : 0;
    int64_t cf = 0;
    const int32_t f = N2vBigFirst(S, phase, wv, lane, __ballot(evm != 0), N2vPick(e, rsel), &cf);
    if (f < 0) break;                          // every remaining child is below pn[k]
    // first k' >= k with pn[k'] >= cf (the cursor skips the smaller entries)
    bool hit = false;
    for (;;) {
      const int32_t kk = k + tid;
      int64_t pv = 0;
      if (kk < np) pv = (int64_t)p_nbr[N2vPhys(S.seq.parent, kk)];
      int64_t mv = 0;
      const int32_t g = N2vBigFirst(S, phase, wv, lane, __ballot(kk < np && pv >= cf), pv, &mv);
      if (g >= 0) { k += g; st->cur.M = mv; st->cur.m_k = k; hit = true; break; }
      k += 64 * kN2vBigWaves;
      if (k >= np) { k = np; break; }
    }
    if (hit && st->cur.M == cf) { if (tid == f) keep |= 1u << rsel; ++k; }
    res_tid = f;
    if (tid == f) res_r = rsel;
    ++events;
  }
  st->cur.k = k;
  *events_out = events;
  if (tid == 0 && events > 0) N2vCount(4, (unsigned long long)events);
  N2vWeights(a, e, keep, parent, d);
  // ---- running sums
  float carry = st->acc;
  int w0 = 0;
  float cin = 0.f;
  for (;;) {
    const uint32_t cb = __float_as_uint(carry);
    const uint32_t ex = cb >> 23;
    const uint32_t bb = cb & 0xFF800000u;
    const float B = __uint_as_float(bb);
    const float twoB = __fadd_rn(B, B);
    const float half_ulp = __uint_as_float(bb - (24u << 23));
    const bool range_ok = ex >= 30u && ex < 254u;
    const bool active = wv >= w0;
    uint32_t N = 0;
    bool okl = true;
#pragma unroll
    for (int r = 0; r < kN2vR; ++r) {
      const float t = __fadd_rn(B, d[r]);
      const float err = __fsub_rn(d[r], __fsub_rn(t, B));
      const bool ok = d[r] >= 0.f && fabsf(err) != half_ulp && t < twoB;
      okl = okl && ok;
      N += ok ? __float_as_uint(t) - bb : 0u;
    }
    if (!active) N = 0;
    const uint32_t incl = WaveInclusiveAdd(N, lane);                // < 2^31
    const bool bad = active && (!range_ok || __ballot(!okl) != 0);
    const int b = *phase & 1;
    ++*phase;
    if (lane == 63) S.x_val[b][wv] = ((int64_t)(bad ? 1 : 0) << 32) | incl;
    __syncthreads();
    const int64_t mine = lane < kN2vBigWaves ? S.x_val[b][lane] : 0;
    int64_t pre = (int64_t)(uint32_t)mine;                          // inclusive over the waves
#pragma unroll
    for (int dd = 1; dd < kN2vBigWaves; dd <<= 1) {
      const int64_t up = __shfl_up(pre, dd);
      if (lane >= dd) pre += up;
    }
    const int64_t off0 = (int64_t)(cb - bb);
    const unsigned long long probw =
        __ballot(lane < kN2vBigWaves && lane >= w0 &&
                 ((mine >> 32) != 0 || off0 + pre >= (1 << 23)));
    const int pw = probw != 0 ? __ffsll((long long)probw) - 1 : kN2vBigWaves;
    const int64_t my_before = (wv == 0 ? 0 : __shfl(pre, wv - 1)) + off0;
    if (active && wv < pw) cin = __uint_as_float(bb + (uint32_t)(my_before + incl - N));
    if (pw == kN2vBigWaves) {
      st->acc = __uint_as_float(bb + (uint32_t)(__shfl(pre, kN2vBigWaves - 1) + off0));
      break;
    }
    const int64_t pw_before = (pw == 0 ? 0 : __shfl(pre, pw - 1)) + off0;
    if (wv == pw) {
      const float before = pw == w0 ? carry : __uint_as_float(bb + (uint32_t)pw_before);
      const float last = WaveSumsVec(before, d, lane, &cin);
      if (lane == 0) S.hand = last;
    }
    __syncthreads();
    carry = S.hand;
    w0 = pw + 1;
    if (w0 == kN2vBigWaves) { st->acc = carry; break; }
  }
  *cin_out = cin;
}

// Lane per walker: queue the walkers whose step `s` has a long child list.
__global__ __launch_bounds__(256) void N2vClassifyKernel(const WalkArgs a) {
  __shared__ int32_t base;
  __shared__ int32_t wave_cnt[4];
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int32_t s = a.step_begin;
  const int64_t L = a.walk_len + 1;
  bool big = false;
  if (i < a.n) {
    const int64_t cur = s == 0 ? a.nodes[i] : a.out[i * L + s];
    N2vList l;
    N2vBuildList(&l, a.g, FindRow(a.g, (uint64_t)cur), a.edge_types + s * a.k, a.k);
    big = l.total >= a.big_threshold;
  }
  const unsigned long long m = __ballot(big);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) wave_cnt[wv] = __popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) {
    const int32_t tot = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
    base = tot > 0 ? atomicAdd(a.big_count, tot) : 0;
  }
  __syncthreads();
  if (big) {
    int32_t off = base;
    for (int w = 0; w < wv; ++w) off += wave_cnt[w];
    off += __popcll(m & ((1ull << lane) - 1));
    a.big_queue[off] = (int32_t)i;
  }
}

__global__ __launch_bounds__(64 * kN2vBigWaves) void N2vBigStepKernel(const WalkArgs a) {
  __shared__ N2vBigLds S;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int32_t s = a.step_begin;
  const int64_t L = a.walk_len + 1;
  const int32_t queued = a.big_count[0];
  int phase = 0;
  for (;;) {
    __syncthreads();
    if (threadIdx.x == 0) S.next = atomicAdd(a.big_count + 1, 1);
    __syncthreads();
    const int64_t qe = S.next;
    if (qe >= queued) break;
    const int64_t i = a.big_queue[qe];
    const int64_t cur = s == 0 ? a.nodes[i] : a.out[i * L + s];
    const int64_t parent = s == 0 ? cur : a.out[i * L + s - 1];
    if (threadIdx.x == 0) {
      const int32_t* et = a.edge_types + s * a.k;
      const int32_t* pet = s > 0 ? a.edge_types + (s - 1) * a.k : et;
      N2vBuildList(&S.seq.child, a.g, FindRow(a.g, (uint64_t)cur), et, a.k);
      N2vBuildList(&S.seq.parent, a.g, s > 0 ? FindRow(a.g, (uint64_t)parent) : -1, pet, a.k);
    }
    __syncthreads();
    const int32_t nc = S.seq.child.total, np = S.seq.parent.total;
    const int32_t rounds = (nc + kN2vBigRound - 1) / kN2vBigRound;
    int32_t sh = 0;
    while ((rounds >> sh) > kN2vBigCk) ++sh;
    const int32_t n_slots = rounds >> sh;
    N2vBigState st{{0, -1, 0}, 0.f};
    const bool same = N2vSameLists(S.seq.child, S.seq.parent);
    const int32_t jl = threadIdx.x * kN2vR;
    float d[kN2vR], cin;
    int32_t events;
    bool ascending = false;
    // the next round's entries are requested before this round is worked on
    N2vVec e = N2vLoadVec(a, S.seq.child, nc, jl), nx = e;
    for (int32_t ri = 0; ri < rounds; ++ri) {
      if (ri + 1 < rounds) nx = N2vLoadVec(a, S.seq.child, nc, (ri + 1) * kN2vBigRound + jl);
      N2vBigRound(a, S, &phase, wv, lane, parent, np, same, e, &st, d, &cin, &events);
      e = nx;
      if (ri == 0 && events > 64) { ascending = true; break; }
      if (((ri + 1) & ((1 << sh) - 1)) == 0 && threadIdx.x == 0) {
        S.ck_acc[((ri + 1) >> sh) - 1] = st.acc;
        S.ck_k[((ri + 1) >> sh) - 1] = st.cur.k;
      }
    }
    int64_t result = a.default_node;
    if (threadIdx.x == 0) {
      N2vCount(ascending ? 2 : 6, 1);
      N2vCount(ascending ? 3 : 7, (unsigned long long)nc);
    }
    if (ascending) {
      // every child moves the parent cursor: the lane-0 automaton of one wave does it
      if (wv == 0) result = N2vStepSequential(a, S.seq, lane, parent, i, s);
    } else {
      const float total = st.acc;
      const double u = RngDraw(a.seed, a.call_id + (uint32_t)s, kDomainWalk, (uint64_t)i, 0);
      const double r = ScaleDraw(u, 0.f, total);
      __syncthreads();
      int32_t first = 0;
      if (a.g.monotone && a.p > 0.f && a.q > 0.f) {
        first = n_slots;
        for (int32_t base = 0; base < n_slots; base += 64) {
          const int32_t idx = base + lane;
          const unsigned long long gt = __ballot(idx < n_slots && (double)S.ck_acc[idx] > r);
          if (gt != 0) { first = base + __ffsll((long long)gt) - 1; break; }
        }
      }
      st.acc = first == 0 ? 0.f : S.ck_acc[first - 1];
      st.cur.k = first == 0 ? 0 : S.ck_k[first - 1];
      st.cur.m_k = -1;
      bool found = false;
      for (int32_t ri = first << sh; ri < rounds && !found; ++ri) {
        e = N2vLoadVec(a, S.seq.child, nc, ri * kN2vBigRound + jl);
        N2vBigRound(a, S, &phase, wv, lane, parent, np, same, e, &st, d, &cin, &events);
        const uint32_t hm = N2vHits(e, d, cin, r);
        int64_t hv = 0;
        if (N2vBigFirst(S, &phase, wv, lane, __ballot(hm != 0),
                        N2vPick(e, hm != 0 ? __ffs((int)hm) - 1 : 0), &hv) >= 0) {
          found = true;
          result = hv;
        }
      }
      // no interval holds r (total == 0): RandomSelect's fall-through ends on the last element
      if (!found) result = (int64_t)(a.g.nbr + S.seq.child.row_ptr)[N2vPhys(S.seq.child, nc - 1)];
    }
    if (threadIdx.x == 0) a.out[i * L + s + 1] = result;
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
  a.g = SamplingView(g); a.seed = seed; a.call_id = call_id; a.nodes = nodes_dev;
  a.edge_types = et_dev; a.out = out_dev; a.n = n; a.default_node = default_node;
  a.k = k; a.walk_len = walk_len; a.p = p; a.q = q;
  a.ablate = g_k1_ablate;
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
    if (k == 1 && g->view.monotone && g->view.blk != nullptr && g_k1_variant == 3) {
      // A/B: the 32-ary skip levels (one line per level) instead of the fanout-5 block pivots
      hipLaunchKernelGGL((RandomWalkKernel<true, true>), dim3(GridFor(n, block)), dim3(block), 0,
                         st, a);
    } else if (k == 1 && g->view.monotone && g->view.blk != nullptr && g_k1_variant >= 5) {
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
      hipLaunchKernelGGL(Node2VecWaveKernel<true>, dim3(GridFor(n * 64, block)), dim3(block), 0,
                         st, a);
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
