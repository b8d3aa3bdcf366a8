// Layerwise sampling (GQL sampleLNB without a weight function) and
// SparseGetAdj for gfx950, with their C-ABI entry points.  The per-item logic
// lives in layer_fns.h; the kernels here map items to lanes / waves.
//
//   API_GET_EDGE_SUM_WEIGHT  core/kernels/get_edge_sum_weight_op.cc:33-66
//   API_SAMPLE_ROOT          core/kernels/sample_root_op.cc:33-88
//   API_SAMPLE_L             core/kernels/sample_layer_op.cc:32-72
//   API_SPARSE_GET_ADJ       core/kernels/sparse_get_adj_op.cc:35-92
//   TF SparseGetAdj          tf_euler/kernels/sparse_get_adj_op.cc:43-134
//   TF SampleNeighborLayerwiseWithAdj
//                            tf_euler/kernels/sample_neighbor_layerwise_with_adj_op.cc:56-150
#include <hip/hip_runtime.h>

#include <algorithm>
#include <thread>
#include <vector>

#include "layer_fns.h"

namespace euler_gpu {

// euler_gpu_set_tuning key 15.  API_SAMPLE_ROOT's table build (Vose's alias
// method with LIFO stacks, common/alias_method.cc:23-63) is one dependency
// chain per batch row: with many rows they run one per lane, but a call with
// FEW rows - the layerwise dataflow passes the whole frontier as ONE row
// (tf_euler/python/dataflow/layerwise_dataflow.py:44-47) - would leave a
// single lane walking n elements through HBM latency.  Calls with fewer rows
// than this build their tables with the host's cores (the same AliasBuildRow
// source, compiled for the host) between two copies; the draws stay on the
// device.  0 = always on the device.
int g_root_host_batch = 64;

int ExclusiveScanI64(hipStream_t stream, const int64_t* in, int64_t* out,
                     int64_t n);   // mp_kernels.hip

namespace {

struct TypeList {
  int32_t k;
  int32_t et[kMaxListedTypes];
};

// ---------------------------------------------------------------- sum weight
__global__ __launch_bounds__(256) void EdgeSumWeightKernel(
    const GraphView g, const TypeList tl, const uint64_t* __restrict__ ids,
    int64_t n, float* __restrict__ out) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    out[i] = EdgeSumWeight(g, ids[i], tl.et, tl.k);
}

// ---------------------------------------------------------------- root draw
struct RootScratch {
  float* wn;        // [n][batch] normalised weights (updated by the build)
  float* prob;      // [n][batch]
  int32_t* alias;   // [n][batch]
  int32_t* stack;   // [n][batch]
  float* sum;       // [batch]
};

__global__ __launch_bounds__(256) void SampleRootBuildKernel(
    const float* __restrict__ weights, int64_t batch, int32_t n, RootScratch s) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; b < batch; b += stride)
    s.sum[b] = AliasBuildRow(weights + b * n, n, batch, s.wn + b, s.prob + b,
                             s.alias + b, s.stack + b);
}

__global__ __launch_bounds__(256) void SampleRootDrawKernel(
    const uint64_t* __restrict__ roots, int64_t batch, int32_t n, int32_t m,
    uint64_t seed, uint32_t call_id, int64_t default_node, RootScratch s,
    uint64_t* __restrict__ out) {
  const int64_t total = batch * m;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int64_t b = i / m;
    const int32_t j = (int32_t)(i - b * m);
    if (s.sum[b] == 0.f) {                     // sample_root_op.cc:74-78
      out[i] = (uint64_t)default_node;
    } else {
      const int32_t slot =
          SampleRootSlot(seed, call_id, b, j, n, batch, s.prob + b, s.alias + b);
      out[i] = roots[b * n + slot];
    }
  }
}

// ---------------------------------------------------------------- layer draw
__global__ __launch_bounds__(256) void SampleLayerKernel(
    const GraphView g, const TypeList tl, const uint64_t* __restrict__ roots,
    int64_t n, uint64_t seed, uint32_t call_id, int64_t default_node,
    uint64_t* __restrict__ out_id, float* __restrict__ out_w,
    int32_t* __restrict__ out_t) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    uint64_t id; float w; int32_t t;
    SampleLayerAt(g, seed, call_id, i, roots[i], tl.et, tl.k, default_node, &id, &w, &t);
    out_id[i] = id;
    if (out_w) out_w[i] = w;
    if (out_t) out_t[i] = t;
  }
}

// ---------------------------------------------------------------- adjacency
// One wave per source node r = (batch row b, slot r % n); lanes take the m
// candidate neighbours of batch row b, 64 at a time, and ballot the hits, so
// the hits of a source keep candidate order (the push_back order of
// sparse_get_adj_op.cc:60-72).  tf != 0 adds the TF kernel's explicit zero at
// (b, n-1, m-1) when that pair is no edge (tf_euler/kernels/
// sparse_get_adj_op.cc:112-118).
struct AdjArgs {
  GraphView g;
  TypeList tl;
  const uint64_t* roots;    // [batch * n]
  const uint64_t* l_nb;     // [batch * m]
  int64_t batch;
  int32_t n, m;
  int32_t tf;
};

__device__ __forceinline__ bool AdjEmit(const AdjArgs& a, int64_t row, int64_t b,
                                        int32_t slot, int32_t j, bool* exists) {
  *exists = false;
  if (j >= a.m) return false;
  *exists = EdgeExistAny(a.g, row, a.l_nb[b * a.m + j], a.tl.et, a.tl.k);
  return *exists || (a.tf && slot == a.n - 1 && j == a.m - 1);
}

__global__ __launch_bounds__(256) void AdjCountKernel(const AdjArgs a,
                                                      int64_t* __restrict__ counts) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const int64_t total = a.batch * a.n;
  for (int64_t r = wave; r < total; r += n_waves) {
    const int64_t b = r / a.n;
    const int32_t slot = (int32_t)(r - b * a.n);
    const int64_t row = FindRow(a.g, a.roots[r]);
    int64_t c = 0;
    for (int32_t base = 0; base < a.m; base += 64) {
      bool exists;
      const bool emit = AdjEmit(a, row, b, slot, base + lane, &exists);
      c += __popcll(__ballot(emit));
    }
    if (lane == 0) counts[r] = c;
  }
}

__global__ __launch_bounds__(256) void AdjFillKernel(
    const AdjArgs a, const int64_t* __restrict__ off64,
    const int32_t* __restrict__ idx32, uint64_t* __restrict__ out_id,
    int64_t* __restrict__ indices, int64_t* __restrict__ values) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const int64_t total = a.batch * a.n;
  for (int64_t r = wave; r < total; r += n_waves) {
    const int64_t b = r / a.n;
    const int32_t slot = (int32_t)(r - b * a.n);
    const int64_t row = FindRow(a.g, a.roots[r]);
    int64_t o = off64 ? off64[r] : (int64_t)idx32[2 * r];
    for (int32_t base = 0; base < a.m; base += 64) {
      const int32_t j = base + lane;
      bool exists;
      const bool emit = AdjEmit(a, row, b, slot, j, &exists);
      const uint64_t ballot = __ballot(emit);
      if (emit) {
        const int64_t p = o + __popcll(ballot & ((1ull << lane) - 1ull));
        if (a.tf) {
          indices[3 * p] = b;
          indices[3 * p + 1] = slot;
          indices[3 * p + 2] = j;
          values[p] = exists ? 1 : 0;
        } else {
          out_id[p] = a.l_nb[b * a.m + j];
        }
      }
      o += __popcll(ballot);
    }
  }
}

__global__ void AdjOffsetsToIdxKernel(const int64_t* __restrict__ off, int64_t n,
                                      int32_t* __restrict__ idx) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    idx[2 * i] = (int32_t)off[i];
    idx[2 * i + 1] = (int32_t)off[i + 1];
  }
}

// ---------------------------------------------------------------- node types
__global__ __launch_bounds__(256) void NodeTypeKernel(
    const GraphView g, const int32_t* __restrict__ node_type,
    const uint64_t* __restrict__ ids, int64_t n, int32_t* __restrict__ out) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    out[i] = NodeTypeOf(g, node_type, ids[i]);
}

// one lane per (row i, sample j) of API_SAMPLE_N_WITH_TYPES
__global__ __launch_bounds__(256) void SampleNWithTypesKernel(
    const NodeSamplerView s, const int32_t* __restrict__ types, int64_t n,
    int32_t count, uint64_t seed, uint32_t call_id, uint64_t* __restrict__ out,
    int32_t* __restrict__ bad) {
  const int64_t total = n * count;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t x = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; x < total; x += stride) {
    const int64_t i = x / count;
    const int32_t j = (int32_t)(x - i * count);
    bool ok;
    out[x] = SampleNodeOfType(s, seed, call_id, (uint64_t)i, types[i], j, &ok);
    if (!ok && j == 0) *bad = 1;
  }
}

int FillTypes(const int32_t* edge_types_host, int32_t k, TypeList* tl, const char* who) {
  if (k < 0 || k > kMaxListedTypes || (k > 0 && !edge_types_host))
    return Fail(EULER_GPU_EINVAL, std::string(who) + ": bad edge type list (<= 32)");
  tl->k = k;
  for (int32_t i = 0; i < k; ++i) tl->et[i] = edge_types_host[i];
  for (int32_t i = k; i < kMaxListedTypes; ++i) tl->et[i] = 0;
  return EULER_GPU_OK;
}

// counts -> offsets [R + 1] (off[R] = total) on the stream
int CountAndScan(const AdjArgs& a, hipStream_t st, int64_t* counts /* [R+1] */,
                 int64_t* off /* [R+1] */) {
  const int64_t R = a.batch * a.n;
  const int block = 256;
  EG_HIP(hipMemsetAsync(counts + R, 0, sizeof(int64_t), st));
  hipLaunchKernelGGL(AdjCountKernel, dim3(GridFor(R * 64, block)), dim3(block), 0, st,
                     a, counts);
  EG_HIP(hipGetLastError());
  return ExclusiveScanI64(st, counts, off, R + 1);
}

}  // namespace
}  // namespace euler_gpu

using namespace euler_gpu;

extern "C" {

int euler_gpu_get_edge_sum_weight(const euler_gpu_graph* g, void* stream,
                                  const uint64_t* ids_dev, int64_t n,
                                  const int32_t* edge_types_host, int32_t k,
                                  float* out_w_dev) {
  if (!g) return Fail(EULER_GPU_ENOGRAPH, "get_edge_sum_weight: null graph");
  if (n < 0) return Fail(EULER_GPU_EINVAL, "get_edge_sum_weight: n < 0");
  TypeList tl;
  int rc = FillTypes(edge_types_host, k, &tl, "get_edge_sum_weight");
  if (rc != EULER_GPU_OK) return rc;
  if (n == 0) return EULER_GPU_OK;
  if (!ids_dev || !out_w_dev)
    return Fail(EULER_GPU_EINVAL, "get_edge_sum_weight: null buffer");
  const int block = 256;
  hipLaunchKernelGGL(EdgeSumWeightKernel, dim3(GridFor(n, block)), dim3(block), 0,
                     (hipStream_t)stream, g->view, tl, ids_dev, n, out_w_dev);
  EG_HIP(hipGetLastError());
  return EULER_GPU_OK;
}

int euler_gpu_sample_root(void* stream, uint64_t seed, uint32_t call_id,
                          const uint64_t* roots_dev, const float* weights_dev,
                          int64_t batch, int32_t n, int32_t m, int64_t default_node,
                          uint64_t* out_dev) {
  if (batch < 0 || n <= 0 || m < 0)
    return Fail(EULER_GPU_EINVAL, "sample_root: need batch >= 0, n > 0, m >= 0");
  if (batch == 0 || m == 0) return EULER_GPU_OK;
  if (!roots_dev || !weights_dev || !out_dev)
    return Fail(EULER_GPU_EINVAL, "sample_root: null buffer");
  hipStream_t st = (hipStream_t)stream;
  const int64_t cells = batch * n;
  uint8_t* buf = nullptr;
  EG_HIP(hipMallocAsync((void**)&buf, (size_t)(cells * 16 + batch * 4), st));
  RootScratch s;
  s.wn = reinterpret_cast<float*>(buf);
  s.prob = s.wn + cells;
  s.alias = reinterpret_cast<int32_t*>(s.prob + cells);
  s.stack = s.alias + cells;
  s.sum = reinterpret_cast<float*>(s.stack + cells);
  const int block = 256;
  const bool on_host = batch < g_root_host_batch;
  std::vector<float> h_w, h_wn, h_prob, h_sum;
  std::vector<int32_t> h_alias, h_stack;
  if (on_host) {
    h_w.resize(cells); h_wn.resize(cells); h_prob.assign(cells, 0.f); h_sum.resize(batch);
    h_alias.assign(cells, 0); h_stack.resize(cells);
    hipError_t c = hipMemcpyAsync(h_w.data(), weights_dev, (size_t)cells * 4,
                                  hipMemcpyDeviceToHost, st);
    if (c == hipSuccess) c = hipStreamSynchronize(st);
    if (c != hipSuccess) { (void)hipFreeAsync(buf, st); EG_HIP(c); }
    auto build = [&](int64_t b0, int64_t b1) {
      for (int64_t b = b0; b < b1; ++b)
        h_sum[b] = AliasBuildRow(h_w.data() + b * n, n, batch, h_wn.data() + b,
                                 h_prob.data() + b, h_alias.data() + b,
                                 h_stack.data() + b);
    };
    const int64_t hw = (int64_t)std::thread::hardware_concurrency();
    const int64_t n_thr = std::min<int64_t>(std::min<int64_t>(batch, 8),
                                            std::max<int64_t>(1, std::min<int64_t>(hw, cells >> 14)));
    if (n_thr <= 1) {
      build(0, batch);
    } else {
      std::vector<std::thread> pool;
      for (int64_t t = 0; t < n_thr; ++t)
        pool.emplace_back(build, batch * t / n_thr, batch * (t + 1) / n_thr);
      for (auto& th : pool) th.join();
    }
    c = hipMemcpyAsync(s.prob, h_prob.data(), (size_t)cells * 4, hipMemcpyHostToDevice, st);
    if (c == hipSuccess)
      c = hipMemcpyAsync(s.alias, h_alias.data(), (size_t)cells * 4, hipMemcpyHostToDevice, st);
    if (c == hipSuccess)
      c = hipMemcpyAsync(s.sum, h_sum.data(), (size_t)batch * 4, hipMemcpyHostToDevice, st);
    if (c != hipSuccess) { (void)hipFreeAsync(buf, st); EG_HIP(c); }
  } else {
    hipLaunchKernelGGL(SampleRootBuildKernel, dim3(GridFor(batch, block)), dim3(block), 0,
                       st, weights_dev, batch, n, s);
  }
  hipLaunchKernelGGL(SampleRootDrawKernel, dim3(GridFor(batch * m, block)), dim3(block),
                     0, st, roots_dev, batch, n, m, seed, call_id, default_node, s,
                     out_dev);
  hipError_t e = hipGetLastError();
  // the host vectors feed asynchronous copies: they must outlive them
  hipError_t y = on_host ? hipStreamSynchronize(st) : hipSuccess;
  hipError_t f = hipFreeAsync(buf, st);
  EG_HIP(e);
  EG_HIP(y);
  EG_HIP(f);
  return EULER_GPU_OK;
}

int euler_gpu_sample_layer(const euler_gpu_graph* g, void* stream, uint64_t seed,
                           uint32_t call_id, const uint64_t* roots_dev, int64_t n,
                           const int32_t* edge_types_host, int32_t k,
                           int64_t default_node, uint64_t* out_id_dev,
                           float* out_w_dev, int32_t* out_t_dev) {
  if (!g) return Fail(EULER_GPU_ENOGRAPH, "sample_layer: null graph");
  if (n < 0) return Fail(EULER_GPU_EINVAL, "sample_layer: n < 0");
  TypeList tl;
  int rc = FillTypes(edge_types_host, k, &tl, "sample_layer");
  if (rc != EULER_GPU_OK) return rc;
  if (n == 0) return EULER_GPU_OK;
  if (!roots_dev || !out_id_dev)
    return Fail(EULER_GPU_EINVAL, "sample_layer: null buffer");
  const int block = 256;
  hipLaunchKernelGGL(SampleLayerKernel, dim3(GridFor(n, block)), dim3(block), 0,
                     (hipStream_t)stream, g->view, tl, roots_dev, n, seed, call_id,
                     default_node, out_id_dev, out_w_dev, out_t_dev);
  EG_HIP(hipGetLastError());
  return EULER_GPU_OK;
}

int euler_gpu_sample_neighbor_layerwise(const euler_gpu_graph* g, void* stream,
                                        uint64_t seed, uint32_t call_id,
                                        const uint64_t* nodes_dev, int64_t batch,
                                        int32_t n, const int32_t* edge_types_host,
                                        int32_t k, int32_t count,
                                        int64_t default_node, uint64_t* out_dev) {
  if (!g) return Fail(EULER_GPU_ENOGRAPH, "sample_neighbor_layerwise: null graph");
  if (batch < 0 || n <= 0 || count < 0)
    return Fail(EULER_GPU_EINVAL,
                "sample_neighbor_layerwise: need batch >= 0, n > 0, count >= 0");
  if (batch == 0 || count == 0) return EULER_GPU_OK;
  if (!nodes_dev || !out_dev)
    return Fail(EULER_GPU_EINVAL, "sample_neighbor_layerwise: null buffer");
  hipStream_t st = (hipStream_t)stream;
  uint8_t* buf = nullptr;
  const int64_t cells = batch * n, draws = batch * (int64_t)count;
  EG_HIP(hipMallocAsync((void**)&buf, (size_t)(draws * 8 + cells * 4), st));
  uint64_t* l_root = reinterpret_cast<uint64_t*>(buf);
  float* weights = reinterpret_cast<float*>(l_root + draws);
  int rc = euler_gpu_get_edge_sum_weight(g, stream, nodes_dev, cells, edge_types_host,
                                         k, weights);
  if (rc == EULER_GPU_OK)
    rc = euler_gpu_sample_root(stream, seed, call_id, nodes_dev, weights, batch, n,
                               count, default_node, l_root);
  if (rc == EULER_GPU_OK)
    rc = euler_gpu_sample_layer(g, stream, seed, call_id, l_root, draws,
                                edge_types_host, k, default_node, out_dev, nullptr,
                                nullptr);
  hipError_t f = hipFreeAsync(buf, st);
  if (rc != EULER_GPU_OK) return rc;
  EG_HIP(f);
  return EULER_GPU_OK;
}

int euler_gpu_get_node_type(const euler_gpu_graph* g, void* stream,
                            const uint64_t* ids_dev, int64_t n, int32_t* out_dev) {
  if (!g) return Fail(EULER_GPU_ENOGRAPH, "get_node_type: null graph");
  if (n < 0) return Fail(EULER_GPU_EINVAL, "get_node_type: n < 0");
  if (n == 0) return EULER_GPU_OK;
  if (!ids_dev || !out_dev) return Fail(EULER_GPU_EINVAL, "get_node_type: null buffer");
  const int block = 256;
  hipLaunchKernelGGL(NodeTypeKernel, dim3(GridFor(n, block)), dim3(block), 0,
                     (hipStream_t)stream, g->view, g->node_type_dev, ids_dev, n, out_dev);
  EG_HIP(hipGetLastError());
  return EULER_GPU_OK;
}

int euler_gpu_sample_n_with_types(const euler_gpu_graph* g, void* stream, uint64_t seed,
                                  uint32_t call_id, const int32_t* types_dev, int64_t n,
                                  int32_t count, uint64_t* out_dev) {
  if (!g) return Fail(EULER_GPU_ENOGRAPH, "sample_n_with_types: null graph");
  if (!g->has_sampler)
    return Fail(EULER_GPU_ENOGRAPH, "sample_n_with_types: graph has no global sampler");
  if (n < 0 || count < 0) return Fail(EULER_GPU_EINVAL, "sample_n_with_types: bad sizes");
  if (n == 0 || count == 0) return EULER_GPU_OK;
  if (!types_dev || !out_dev)
    return Fail(EULER_GPU_EINVAL, "sample_n_with_types: null buffer");
  hipStream_t st = (hipStream_t)stream;
  int32_t* bad = nullptr;
  EG_HIP(hipMallocAsync((void**)&bad, sizeof(int32_t), st));
  EG_HIP(hipMemsetAsync(bad, 0, sizeof(int32_t), st));
  const int block = 256;
  hipLaunchKernelGGL(SampleNWithTypesKernel, dim3(GridFor(n * count, block)), dim3(block),
                     0, st, g->sampler, types_dev, n, count, seed, call_id, out_dev, bad);
  hipError_t e = hipGetLastError();
  int32_t bad_host = 0;
  hipError_t c = hipMemcpyAsync(&bad_host, bad, sizeof(int32_t), hipMemcpyDeviceToHost, st);
  hipError_t y = hipStreamSynchronize(st);
  hipError_t f = hipFreeAsync(bad, st);
  EG_HIP(e); EG_HIP(c); EG_HIP(y); EG_HIP(f);
  if (bad_host)
    return Fail(EULER_GPU_EEMPTY,
                "sample_n_with_types: a listed type is unknown or has zero weight");
  return EULER_GPU_OK;
}

int euler_gpu_sparse_get_adj(const euler_gpu_graph* g, void* stream,
                             const uint64_t* roots_dev, const uint64_t* l_nb_dev,
                             int64_t batch, int32_t n, int32_t m,
                             const int32_t* edge_types_host, int32_t k,
                             int32_t* idx_dev, int64_t* total_host,
                             uint64_t* out_id_dev) {
  if (!g) return Fail(EULER_GPU_ENOGRAPH, "sparse_get_adj: null graph");
  if (batch < 0 || n < 0 || m < 0)
    return Fail(EULER_GPU_EINVAL, "sparse_get_adj: negative size");
  AdjArgs a{};
  int rc = FillTypes(edge_types_host, k, &a.tl, "sparse_get_adj");
  if (rc != EULER_GPU_OK) return rc;
  const int64_t R = batch * n;
  if (R == 0) { if (total_host) *total_host = 0; return EULER_GPU_OK; }
  if (!roots_dev || !idx_dev || (m > 0 && !l_nb_dev))
    return Fail(EULER_GPU_EINVAL, "sparse_get_adj: null buffer");
  if ((int64_t)m * n * batch > 0x7fffffffLL)
    return Fail(EULER_GPU_EINVAL, "sparse_get_adj: result offsets exceed int32");
  hipStream_t st = (hipStream_t)stream;
  a.g = g->view; a.roots = roots_dev; a.l_nb = l_nb_dev;
  a.batch = batch; a.n = n; a.m = m; a.tf = 0;
  const int block = 256;
  if (out_id_dev == nullptr) {
    int64_t* counts = nullptr;
    EG_HIP(hipMallocAsync((void**)&counts, (size_t)(2 * (R + 1)) * sizeof(int64_t), st));
    int64_t* off = counts + R + 1;
    rc = CountAndScan(a, st, counts, off);
    if (rc != EULER_GPU_OK) { (void)hipFreeAsync(counts, st); return rc; }
    hipLaunchKernelGGL(AdjOffsetsToIdxKernel, dim3((unsigned)((R + block - 1) / block)),
                       dim3(block), 0, st, off, R, idx_dev);
    int64_t total = 0;
    EG_HIP(hipMemcpyAsync(&total, off + R, 8, hipMemcpyDeviceToHost, st));
    EG_HIP(hipStreamSynchronize(st));
    EG_HIP(hipFreeAsync(counts, st));
    if (total_host) *total_host = total;
    return EULER_GPU_OK;
  }
  hipLaunchKernelGGL(AdjFillKernel, dim3(GridFor(R * 64, block)), dim3(block), 0, st, a,
                     (const int64_t*)nullptr, (const int32_t*)idx_dev, out_id_dev,
                     (int64_t*)nullptr, (int64_t*)nullptr);
  EG_HIP(hipGetLastError());
  return EULER_GPU_OK;
}

int euler_gpu_sparse_get_adj_tf(const euler_gpu_graph* g, void* stream,
                                const uint64_t* nodes_dev, const uint64_t* nb_nodes_dev,
                                int64_t batch, int32_t n, int32_t m,
                                const int32_t* edge_types_host, int32_t k,
                                int64_t* row_off_dev, int64_t* nnz_host,
                                int64_t* indices_dev, int64_t* values_dev) {
  if (!g) return Fail(EULER_GPU_ENOGRAPH, "sparse_get_adj_tf: null graph");
  if (batch < 0 || n < 0 || m < 0)
    return Fail(EULER_GPU_EINVAL, "sparse_get_adj_tf: negative size");
  AdjArgs a{};
  int rc = FillTypes(edge_types_host, k, &a.tl, "sparse_get_adj_tf");
  if (rc != EULER_GPU_OK) return rc;
  const int64_t R = batch * n;
  if (R == 0 || m == 0) { if (nnz_host) *nnz_host = 0; return EULER_GPU_OK; }
  if (!nodes_dev || !nb_nodes_dev || !row_off_dev)
    return Fail(EULER_GPU_EINVAL, "sparse_get_adj_tf: null buffer");
  hipStream_t st = (hipStream_t)stream;
  a.g = g->view; a.roots = nodes_dev; a.l_nb = nb_nodes_dev;
  a.batch = batch; a.n = n; a.m = m; a.tf = 1;
  const int block = 256;
  if (indices_dev == nullptr) {
    int64_t* counts = nullptr;
    EG_HIP(hipMallocAsync((void**)&counts, (size_t)(R + 1) * sizeof(int64_t), st));
    rc = CountAndScan(a, st, counts, row_off_dev);
    if (rc != EULER_GPU_OK) { (void)hipFreeAsync(counts, st); return rc; }
    int64_t total = 0;
    EG_HIP(hipMemcpyAsync(&total, row_off_dev + R, 8, hipMemcpyDeviceToHost, st));
    EG_HIP(hipStreamSynchronize(st));
    EG_HIP(hipFreeAsync(counts, st));
    if (nnz_host) *nnz_host = total;
    return EULER_GPU_OK;
  }
  if (!values_dev) return Fail(EULER_GPU_EINVAL, "sparse_get_adj_tf: null values");
  hipLaunchKernelGGL(AdjFillKernel, dim3(GridFor(R * 64, block)), dim3(block), 0, st, a,
                     (const int64_t*)row_off_dev, (const int32_t*)nullptr,
                     (uint64_t*)nullptr, indices_dev, values_dev);
  EG_HIP(hipGetLastError());
  return EULER_GPU_OK;
}

}  // extern "C"
