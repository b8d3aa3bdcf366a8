// Sampling kernels: SampleNeighbor / SampleFanout / SampleNode /
// GetFullNeighbor / RandomWalk for gfx950, plus their C-ABI entry points.
#include <hip/hip_runtime.h>

#include <cmath>
#include <vector>

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
  int32_t pad;
  int32_t et[kMaxListedTypes];
};

__global__ __launch_bounds__(256) void SampleNeighborKernel(const SampleNbArgs a) {
  const int64_t total = a.n * (int64_t)a.count;
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

namespace {
int g_k1_variant = 1;   // 1 = fast path when applicable, 0 = always generic
}

static int LaunchSampleNeighbor(const euler_gpu_graph* g, hipStream_t stream,
                                uint64_t seed, uint32_t call_id,
                                const uint64_t* roots, int64_t n,
                                const uint8_t* root_mask, int32_t root_group,
                                const int32_t* edge_types, int32_t k,
                                int32_t count, int32_t layout,
                                int64_t default_node, uint64_t* out_id,
                                float* out_w, int32_t* out_t,
                                uint8_t* out_row_mask) {
  if (g == nullptr) return Fail(EULER_GPU_ENOGRAPH, "sample_neighbor: null graph");
  if (n < 0 || count < 0 || k < 0 || k > kMaxListedTypes)
    return Fail(EULER_GPU_EINVAL, "sample_neighbor: bad n/count/k (k <= 32)");
  if (layout != EULER_GPU_LAYOUT_CORE && layout != EULER_GPU_LAYOUT_TF)
    return Fail(EULER_GPU_EINVAL, "sample_neighbor: bad layout");
  if (n == 0 || count == 0) return EULER_GPU_OK;
  if (!roots || !out_id || !out_w || !out_t)
    return Fail(EULER_GPU_EINVAL, "sample_neighbor: null buffer");
  if (k > 0 && !edge_types)
    return Fail(EULER_GPU_EINVAL, "sample_neighbor: null edge_types");
  SampleNbArgs a{};
  a.g = g->view;
  a.seed = seed; a.call_id = call_id;
  a.roots = roots; a.root_mask = root_mask;
  a.root_group = root_group > 0 ? root_group : 1;
  a.out_id = out_id; a.out_w = out_w; a.out_t = out_t;
  a.out_row_mask = out_row_mask;
  a.n = n; a.default_node = default_node;
  a.k = k; a.count = count; a.layout = layout;
  for (int i = 0; i < k; ++i) a.et[i] = edge_types[i];
  const int block = 256;
  const int grid = GridFor(n * (int64_t)count, block);
  if (g_k1_variant == 1 && k == 1 && g->view.monotone) {
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

__global__ __launch_bounds__(256) void RandomWalkKernel(const WalkArgs a) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t L = a.walk_len + 1;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n;
       i += stride) {
    uint64_t cur = (uint64_t)a.nodes[i];
    a.out[i * L] = (int64_t)cur;
    for (int32_t s = 0; s < a.walk_len; ++s) {
      RowSampler rs;
      InitRowSampler(rs, a.g, FindRow(a.g, cur), a.edge_types + s * a.k, a.k);
      uint64_t id = 0; float w; int32_t t;
      if (rs.valid) SampleAt(rs, a.seed, a.call_id + (uint32_t)s, cur, 0, &id, &w, &t);
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

extern "C" {

int euler_gpu_set_tuning(int32_t key, int32_t value) {
  if (key == 0) { g_k1_variant = value; return EULER_GPU_OK; }
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
  const uint64_t* roots = roots_dev;
  const uint8_t* mask = nullptr;
  int32_t group = 1;
  int64_t m = n;
  uint8_t* ws = (uint8_t*)workspace_dev;
  for (int32_t h = 0; h < layers; ++h) {
    // hop h: roots are the previous hop's TF-layout ids; rows the previous hop
    // marked as missing sample as the sentinel id 0, which is what the
    // reference's chained GQL feeds on (sample_fanout_op.cc:37-42).
    uint8_t* row_mask = ws;
    ws += ((size_t)m + 15) & ~(size_t)15;
    int rc = LaunchSampleNeighbor(g, (hipStream_t)stream, seed,
                                  call_id + (uint32_t)h, roots, m, mask, group,
                                  edge_types_host + (size_t)h * k, k,
                                  counts_host[h], EULER_GPU_LAYOUT_TF,
                                  default_node, out_id_dev[h], out_w_dev[h],
                                  out_t_dev[h], row_mask);
    if (rc != EULER_GPU_OK) return rc;
    roots = out_id_dev[h];
    mask = row_mask;
    group = counts_host[h];
    m *= counts_host[h];
  }
  return EULER_GPU_OK;
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
    hipLaunchKernelGGL(RandomWalkKernel, dim3(GridFor(n, block)), dim3(block), 0,
                       st, a);
  } else {
    hipLaunchKernelGGL(Node2VecKernel, dim3(GridFor(n, block)), dim3(block), 0,
                       st, a);
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
