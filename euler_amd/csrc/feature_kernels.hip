// Sparse (uint64) node features for gfx950 with their C-ABI entry point:
// TF GetSparseFeature (tf_euler/kernels/get_sparse_feature_op.cc:52-131) over
// Node::GetUint64Feature (core/graph/node.cc:330-372).  The dense (float)
// feature kernel lives in mp_kernels.hip.
#include <hip/hip_runtime.h>

#include "device_fns.h"

namespace euler_gpu {

int ExclusiveScanI64(hipStream_t stream, const int64_t* in, int64_t* out,
                     int64_t n);   // mp_kernels.hip

namespace {

struct SparseFeatArgs {
  GraphView g;
  const int64_t* ufeat_ptr;
  const int32_t* ufeat_idx;
  const uint64_t* ufeat_val;
  const uint64_t* nodes;
  int64_t n;
  int32_t n_u64;
  int32_t fid;
};

// Values of slot `fid` of the node (GET_NODE_FEATURE, node.cc:330-351): an
// unknown node or slot has none.
__device__ __forceinline__ int32_t SlotRange(const SparseFeatArgs& a, uint64_t id,
                                             const uint64_t** first) {
  *first = nullptr;
  if (a.fid < 0 || a.fid >= a.n_u64) return 0;
  const int64_t row = FindRow(a.g, id);
  if (row < 0) return 0;
  const int32_t* idx = a.ufeat_idx + row * (int64_t)a.n_u64;
  const int32_t pre = a.fid == 0 ? 0 : idx[a.fid - 1];
  const int32_t now = idx[a.fid];
  *first = a.ufeat_val + a.ufeat_ptr[row] + pre;
  return now - pre;
}

// counts[i] = entries node i contributes: its values, or the one default entry
__global__ __launch_bounds__(256) void SparseFeatCountKernel(
    const SparseFeatArgs a, int64_t* __restrict__ counts,
    unsigned long long* __restrict__ max_len) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int32_t local_max = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += stride) {
    const uint64_t* first;
    int32_t len = SlotRange(a, a.nodes[i], &first);
    if (len < 1) len = 1;
    counts[i] = len;
    local_max = max(local_max, len);
  }
  // one atomic per wave
  for (int off = 32; off > 0; off >>= 1) local_max = max(local_max, __shfl_xor(local_max, off));
  if ((threadIdx.x & 63) == 0 && local_max > 0) atomicMax(max_len, (unsigned long long)local_max);
}

// The GQL `values(...)` form (core/kernels/get_feature_op.cc:34-70, "fea:2i" /
// "fea:2i+1"): counts without the default entry, idx pairs, packed values.
__global__ __launch_bounds__(256) void SparseFeatCoreCountKernel(
    const SparseFeatArgs a, int64_t* __restrict__ counts) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += stride) {
    const uint64_t* first;
    counts[i] = SlotRange(a, a.nodes[i], &first);
  }
}

__global__ __launch_bounds__(256) void SparseFeatCoreFillKernel(
    const SparseFeatArgs a, const int32_t* __restrict__ idx, uint64_t* __restrict__ values) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t i = wave; i < a.n; i += n_waves) {
    const uint64_t* first;
    const int32_t len = SlotRange(a, a.nodes[i], &first);
    const int64_t o = idx[2 * i];
    for (int32_t k = lane; k < len; k += 64) values[o + k] = first[k];
  }
}

__global__ void SparseFeatOffsetsToIdxKernel(const int64_t* __restrict__ off, int64_t n,
                                             int32_t* __restrict__ idx) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    idx[2 * i] = (int32_t)off[i];
    idx[2 * i + 1] = (int32_t)off[i + 1];
  }
}

// One wave per node: lanes over its values (lists are short; the row offsets
// make the writes of consecutive nodes contiguous).
__global__ __launch_bounds__(256) void SparseFeatFillKernel(
    const SparseFeatArgs a, const int64_t* __restrict__ off, int64_t default_value,
    int64_t* __restrict__ indices, int64_t* __restrict__ values) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t i = wave; i < a.n; i += n_waves) {
    const uint64_t* first;
    const int32_t len = SlotRange(a, a.nodes[i], &first);
    const int64_t o = off[i];
    if (len < 1) {
      if (lane == 0) {
        indices[2 * o] = i;
        indices[2 * o + 1] = 0;
        values[o] = default_value;
      }
      continue;
    }
    for (int32_t k = lane; k < len; k += 64) {
      indices[2 * (o + k)] = i;
      indices[2 * (o + k) + 1] = k;
      values[o + k] = (int64_t)first[k];
    }
  }
}

}  // namespace
}  // namespace euler_gpu

using namespace euler_gpu;

extern "C" {

int32_t euler_gpu_graph_num_u64_features(const euler_gpu_graph* g) {
  return g ? g->n_u64 : -1;
}

int euler_gpu_get_sparse_feature(const euler_gpu_graph* g, void* stream,
                                 const uint64_t* nodes_dev, int64_t n, int32_t fid,
                                 int64_t default_value, int64_t* row_off_dev,
                                 int64_t* nnz_host, int64_t* max_len_host,
                                 int64_t* indices_dev, int64_t* values_dev) {
  if (!g) return Fail(EULER_GPU_ENOGRAPH, "get_sparse_feature: null graph");
  if (n < 0) return Fail(EULER_GPU_EINVAL, "get_sparse_feature: n < 0");
  if (n == 0) {
    if (nnz_host) *nnz_host = 0;
    if (max_len_host) *max_len_host = 0;
    return EULER_GPU_OK;
  }
  if (!nodes_dev || !row_off_dev)
    return Fail(EULER_GPU_EINVAL, "get_sparse_feature: null buffer");
  hipStream_t st = (hipStream_t)stream;
  SparseFeatArgs a{};
  a.g = g->view;
  a.ufeat_ptr = g->ufeat_ptr; a.ufeat_idx = g->ufeat_idx; a.ufeat_val = g->ufeat_val;
  a.nodes = nodes_dev; a.n = n; a.n_u64 = g->n_u64; a.fid = fid;
  const int block = 256;
  if (indices_dev == nullptr) {
    int64_t* counts = nullptr;
    EG_HIP(hipMallocAsync((void**)&counts, (size_t)(n + 2) * sizeof(int64_t), st));
    unsigned long long* max_len = reinterpret_cast<unsigned long long*>(counts + n + 1);
    EG_HIP(hipMemsetAsync(counts + n, 0, 2 * sizeof(int64_t), st));
    hipLaunchKernelGGL(SparseFeatCountKernel, dim3(GridFor(n, block)), dim3(block), 0, st, a,
                       counts, max_len);
    int rc = ExclusiveScanI64(st, counts, row_off_dev, n + 1);
    if (rc != EULER_GPU_OK) { (void)hipFreeAsync(counts, st); return rc; }
    int64_t total = 0;
    unsigned long long ml = 0;
    EG_HIP(hipMemcpyAsync(&total, row_off_dev + n, 8, hipMemcpyDeviceToHost, st));
    EG_HIP(hipMemcpyAsync(&ml, max_len, 8, hipMemcpyDeviceToHost, st));
    EG_HIP(hipStreamSynchronize(st));
    EG_HIP(hipFreeAsync(counts, st));
    if (nnz_host) *nnz_host = total;
    if (max_len_host) *max_len_host = (int64_t)ml;
    return EULER_GPU_OK;
  }
  if (!values_dev) return Fail(EULER_GPU_EINVAL, "get_sparse_feature: null values");
  hipLaunchKernelGGL(SparseFeatFillKernel, dim3(GridFor(n * 64, block)), dim3(block), 0, st,
                     a, row_off_dev, default_value, indices_dev, values_dev);
  EG_HIP(hipGetLastError());
  return EULER_GPU_OK;
}

int euler_gpu_get_sparse_feature_core(const euler_gpu_graph* g, void* stream,
                                      const uint64_t* nodes_dev, int64_t n, int32_t fid,
                                      int32_t* idx_dev, int64_t* total_host,
                                      uint64_t* values_dev) {
  if (!g) return Fail(EULER_GPU_ENOGRAPH, "get_sparse_feature_core: null graph");
  if (n < 0) return Fail(EULER_GPU_EINVAL, "get_sparse_feature_core: n < 0");
  if (n == 0) { if (total_host) *total_host = 0; return EULER_GPU_OK; }
  if (!nodes_dev || !idx_dev)
    return Fail(EULER_GPU_EINVAL, "get_sparse_feature_core: null buffer");
  hipStream_t st = (hipStream_t)stream;
  SparseFeatArgs a{};
  a.g = g->view;
  a.ufeat_ptr = g->ufeat_ptr; a.ufeat_idx = g->ufeat_idx; a.ufeat_val = g->ufeat_val;
  a.nodes = nodes_dev; a.n = n; a.n_u64 = g->n_u64; a.fid = fid;
  const int block = 256;
  if (values_dev == nullptr) {
    int64_t* counts = nullptr;
    EG_HIP(hipMallocAsync((void**)&counts, (size_t)(2 * n + 2) * sizeof(int64_t), st));
    int64_t* off = counts + n + 1;
    EG_HIP(hipMemsetAsync(counts + n, 0, sizeof(int64_t), st));
    hipLaunchKernelGGL(SparseFeatCoreCountKernel, dim3(GridFor(n, block)), dim3(block), 0, st,
                       a, counts);
    int rc = ExclusiveScanI64(st, counts, off, n + 1);
    if (rc != EULER_GPU_OK) { (void)hipFreeAsync(counts, st); return rc; }
    hipLaunchKernelGGL(SparseFeatOffsetsToIdxKernel, dim3((unsigned)((n + block - 1) / block)),
                       dim3(block), 0, st, off, n, idx_dev);
    int64_t total = 0;
    EG_HIP(hipMemcpyAsync(&total, off + n, 8, hipMemcpyDeviceToHost, st));
    EG_HIP(hipStreamSynchronize(st));
    EG_HIP(hipFreeAsync(counts, st));
    if (total_host) *total_host = total;
    return EULER_GPU_OK;
  }
  hipLaunchKernelGGL(SparseFeatCoreFillKernel, dim3(GridFor(n * 64, block)), dim3(block), 0, st,
                     a, idx_dev, values_dev);
  EG_HIP(hipGetLastError());
  return EULER_GPU_OK;
}

}  // extern "C"
