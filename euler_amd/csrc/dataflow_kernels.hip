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

#include "common.h"
#include "device_fns.h"

namespace euler_gpu {

namespace {

constexpr uint64_t kFlowEmptyKey = ~0ULL;

struct FlowTable {
  unsigned long long* keys;   // [cap + 1]; slot `cap` is the side slot of the all-ones id
  uint32_t* minpos;           // [cap + 1]
  int32_t* rank;              // [cap + 1]
  uint64_t mask;              // cap - 1
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
  FlowTable t;
  uint32_t* slot_of;          // [cap_m]
  uint32_t* is_first;         // [cap_m + 1] (entry cap_m stays 0: the scan's total lands there)
  uint32_t* rank;             // [cap_m + 1]
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

__global__ __launch_bounds__(256) void FlowInsertKernel(const FlowHop h) {
  const int64_t cnt = (int64_t)(*h.cnt);
  const int64_t m_nb = FlowNbLen(h, cnt), m = m_nb + cnt;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += stride) {
    const uint64_t id = FlowElem(h, i, m_nb);
    uint64_t s;
    if (id == kFlowEmptyKey) {
      s = h.t.mask + 1;
    } else {
      s = Mix64(id) & h.t.mask;
      for (;;) {
        unsigned long long old =
            __hip_atomic_load(&h.t.keys[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == kFlowEmptyKey)
          old = atomicCAS(&h.t.keys[s], (unsigned long long)kFlowEmptyKey, (unsigned long long)id);
        if (old == kFlowEmptyKey || old == id) break;
        s = (s + 1) & h.t.mask;
      }
    }
    if (__hip_atomic_load(&h.t.minpos[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > (uint32_t)i)
      atomicMin(&h.t.minpos[s], (uint32_t)i);
    h.slot_of[i] = (uint32_t)s;
  }
}

// is_first over the WORST-CASE length (zeros beyond the valid prefix), so that the
// scan needs no device-side length
__global__ __launch_bounds__(256) void FlowFlagKernel(const FlowHop h) {
  const int64_t cnt = (int64_t)(*h.cnt);
  const int64_t m = FlowNbLen(h, cnt) + cnt;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= h.cap_m; i += stride)
    h.is_first[i] = (i < m && h.t.minpos[h.slot_of[i]] == (uint32_t)i) ? 1u : 0u;
}

__global__ __launch_bounds__(256) void FlowEmitKernel(const FlowHop h) {
  const int64_t cnt = (int64_t)(*h.cnt);
  const int64_t m_nb = FlowNbLen(h, cnt), m = m_nb + cnt;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t first = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (first == 0) *h.cnt_out = h.rank[h.cap_m];          // exclusive scan: the total
  for (int64_t i = first; i < m; i += stride) {
    if (h.is_first[i]) {
      h.new_n_id[h.rank[i]] = FlowElem(h, i, m_nb);
      h.t.rank[h.slot_of[i]] = (int32_t)h.rank[i];
    }
  }
}

__global__ __launch_bounds__(256) void FlowIndexKernel(const FlowHop h) {
  const int64_t cnt = (int64_t)(*h.cnt);
  const int64_t m_nb = FlowNbLen(h, cnt), m = m_nb + cnt;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += stride) {
    const int64_t dst = (int64_t)h.t.rank[h.slot_of[i]];
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

size_t euler_gpu_sage_blocks_workspace(int64_t n, const int32_t* fanouts_host, int32_t layers) {
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
                     + Al((tcap + 1) * 8) + Al((tcap + 1) * 4) * 2  // table
                     + Al((size_t)cap_m * 4)                        // slot_of
                     + Al(((size_t)cap_m + 1) * 4) * 2              // is_first, rank
                     + Al(scan_bytes) + 256;
    if (b > best) best = b;
  }
  return best + 256;
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
  if (FlowCap(n, fanouts_host, layers) >= ((int64_t)1 << 31))
    return Fail(EULER_GPU_EINVAL, "sage_blocks: worst-case layer size >= 2^31");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(FlowInitKernel, dim3(1), dim3(1), 0, st, counts_dev, (uint32_t)n);
  if (n == 0) {
    EG_HIP(hipMemsetAsync(counts_dev, 0, sizeof(uint32_t) * (layers + 1), st));
    return EULER_GPU_OK;
  }
  const uint64_t* n_id = roots_dev;
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
    f.t.keys = (unsigned long long*)p;  p += Al((tcap + 1) * 8);
    f.t.minpos = (uint32_t*)p;          p += Al((tcap + 1) * 4);
    f.t.rank = (int32_t*)p;             p += Al((tcap + 1) * 4);
    f.t.mask = tcap - 1;
    f.slot_of = (uint32_t*)p;           p += Al((size_t)cap_m * 4);
    f.is_first = (uint32_t*)p;          p += Al(((size_t)cap_m + 1) * 4);
    f.rank = (uint32_t*)p;              p += Al(((size_t)cap_m + 1) * 4);
    void* scan_tmp = p;
    size_t scan_bytes = 0;
    EG_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, f.is_first, f.rank,
                                            (int)(cap_m + 1), st));
    // 1. the hop's sampler over the first counts[h] nodes of the layer
    int rc = LaunchSampleNeighborCounted(g, st, seed, call_id + (uint32_t)h, n_id, cap_n,
                                         counts_dev + h, edge_types_host + (size_t)h * k, k, count,
                                         default_node, nb, nb_w, nb_t);
    if (rc != EULER_GPU_OK) return rc;
    // 2. first-occurrence unique of [nb | n_id]
    EG_HIP(hipMemsetAsync(f.t.keys, 0xFF, (tcap + 1) * 8, st));
    EG_HIP(hipMemsetAsync(f.t.minpos, 0xFF, (tcap + 1) * 4, st));
    f.nb = nb; f.n_id = n_id; f.cnt = counts_dev + h; f.cnt_out = counts_dev + h + 1;
    f.count = count; f.self_loops = add_self_loops ? 1 : 0; f.cap_m = cap_m;
    f.new_n_id = n_id_dev[h]; f.inv = edge_dst_dev[h]; f.edge_src = edge_src_dev[h];
    f.res_n_id = res_n_id_dev[h];
    const int block = 256;
    const int grid = GridFor(cap_m + 1, block);
    hipLaunchKernelGGL(FlowInsertKernel, dim3(grid), dim3(block), 0, st, f);
    hipLaunchKernelGGL(FlowFlagKernel, dim3(grid), dim3(block), 0, st, f);
    EG_HIP(hipcub::DeviceScan::ExclusiveSum(scan_tmp, scan_bytes, f.is_first, f.rank,
                                            (int)(cap_m + 1), st));
    hipLaunchKernelGGL(FlowEmitKernel, dim3(grid), dim3(block), 0, st, f);
    // 3. res_n_id, edge_index
    hipLaunchKernelGGL(FlowIndexKernel, dim3(grid), dim3(block), 0, st, f);
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
    if (cap_n + cap_e >= ((int64_t)1 << 31))
      return Fail(EULER_GPU_EINVAL, "full_blocks: capacities >= 2^31");
  }
  hipLaunchKernelGGL(FlowInitKernel, dim3(1), dim3(1), 0, st, counts_dev, (uint32_t)n);
  const uint64_t* n_id = roots_dev;
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
    f.t.keys = (unsigned long long*)p;  p += Al((tcap + 1) * 8);
    f.t.minpos = (uint32_t*)p;          p += Al((tcap + 1) * 4);
    f.t.rank = (int32_t*)p;             p += Al((tcap + 1) * 4);
    f.t.mask = tcap - 1;
    f.slot_of = (uint32_t*)p;           p += Al((size_t)cap_m * 4);
    f.is_first = (uint32_t*)p;          p += Al(((size_t)cap_m + 1) * 4);
    f.rank = (uint32_t*)p;              p += Al(((size_t)cap_m + 1) * 4);
    void* scan_tmp = p;
    size_t scan_bytes = 0, scan2 = 0;
    EG_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, f.is_first, f.rank, (int)(cap_m + 1), st));
    EG_HIP(hipcub::DeviceScan::ExclusiveScan(nullptr, scan2, ff.lens, ff.offs, SatAdd(), 0u, (int)(cap_n + 1), st));
    // 1. the rows of the layer's nodes: lengths -> offsets -> the hop's neighbour list
    hipLaunchKernelGGL(FlowFullCountKernel, dim3(GridFor(cap_n + 1, block)), dim3(block), 0, st, ff);
    EG_HIP(hipcub::DeviceScan::ExclusiveScan(scan_tmp, scan2, ff.lens, ff.offs, SatAdd(), 0u, (int)(cap_n + 1), st));
    hipLaunchKernelGGL(FlowFullTotalKernel, dim3(1), dim3(1), 0, st, ff);
    if (cap_e > 0)
      hipLaunchKernelGGL(FlowFullFillKernel, dim3(GridFor(cap_e, block)), dim3(block), 0, st, ff);
    // 2. first-occurrence unique of [nb | n_id], 3. res_n_id / edge_index - the kernels of
    // the Sage flow with the list length and the edge sources read from the device
    EG_HIP(hipMemsetAsync(f.t.keys, 0xFF, (tcap + 1) * 8, st));
    EG_HIP(hipMemsetAsync(f.t.minpos, 0xFF, (tcap + 1) * 4, st));
    f.nb = ff.nb; f.n_id = n_id; f.cnt = counts_dev + h; f.cnt_out = counts_dev + h + 1;
    f.m_nb_dev = ff.m_nb; f.nb_src = ff.nb_src;
    f.count = 0; f.self_loops = add_self_loops ? 1 : 0; f.cap_m = cap_m;
    f.new_n_id = n_id_dev[h]; f.inv = edge_dst_dev[h]; f.edge_src = edge_src_dev[h];
    f.res_n_id = res_n_id_dev[h];
    const int grid = GridFor(cap_m + 1, block);
    hipLaunchKernelGGL(FlowInsertKernel, dim3(grid), dim3(block), 0, st, f);
    hipLaunchKernelGGL(FlowFlagKernel, dim3(grid), dim3(block), 0, st, f);
    EG_HIP(hipcub::DeviceScan::ExclusiveSum(scan_tmp, scan_bytes, f.is_first, f.rank, (int)(cap_m + 1), st));
    hipLaunchKernelGGL(FlowEmitKernel, dim3(grid), dim3(block), 0, st, f);
    hipLaunchKernelGGL(FlowIndexKernel, dim3(grid), dim3(block), 0, st, f);
    EG_HIP(hipGetLastError());
    n_id = n_id_dev[h];
  }
  return EULER_GPU_OK;
}

}  // extern "C"
