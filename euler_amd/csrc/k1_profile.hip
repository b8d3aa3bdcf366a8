// Measurement only: the single-type flat-array sample_neighbor kernel with
// s_memtime stamps between its phases (every stamp waits for all outstanding
// memory operations first), to see where a wave's time goes.  Not on any
// product path; exported as euler_gpu_debug_k1_phases for tools/k1_phases.py.
#include <hip/hip_runtime.h>

#include "device_fns.h"

namespace euler_gpu {

__device__ __forceinline__ uint64_t Stamp() {
  __builtin_amdgcn_s_waitcnt(0);
  return __builtin_amdgcn_s_memtime();
}

// acc[0..7] phase cycle sums (one stamp set per wave per iteration),
// acc[8] wave-iterations, acc[9] probe steps (wave level)
__global__ __launch_bounds__(256) void K1PhaseKernel(
    const GraphView g, uint64_t seed, uint32_t call_id, const uint64_t* roots,
    int64_t n, int32_t count, uint64_t* out_id, float* out_w,
    unsigned long long* acc) {
  const int64_t total = n * (int64_t)count;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long iters = 0, probes = 0;
  for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < total;
       s += stride) {
    const uint64_t t0 = Stamp();
    const int64_t r = s / count;
    const int32_t j = (int32_t)(s - r * count);
    const uint64_t node = roots[r];
    const uint64_t t1 = Stamp();
    const int64_t row = FindRow(g, node);
    uint64_t id = 0;
    float w = 0.f;
    uint64_t t2 = t1, t3 = t1, t4 = t1, t5 = t1, t6 = t1;
    if (row >= 0) {
      const uint4 q = *reinterpret_cast<const uint4*>(g.row_meta + row * 16);
      const int64_t row_ptr = (int64_t)(((uint64_t)q.y << 32) | q.x);
      const int32_t e = (int32_t)q.z - 1;
      t2 = Stamp();
      const float* nw = g.prefix_w + row_ptr;
      const float limit_end = e >= 0 ? nw[e] : 0.f;
      t3 = Stamp();
      const Philox4 blk = RngBlock(seed, call_id, kDomainNeighbor, node,
                                   ((uint32_t)j) >> 1);
      const double u = (j & 1) ? UnitFromWords(blk.w[2], blk.w[3])
                               : UnitFromWords(blk.w[0], blk.w[1]);
      const double rr = ScaleDraw(u, 0.f, limit_end);
      t4 = Stamp();
      int32_t lo = 0, hi = e;
      while (lo < hi) {
        const int32_t mid = (lo + hi) >> 1;
        const float v = nw[mid];
        __builtin_amdgcn_s_waitcnt(0);
        if ((double)v > rr) hi = mid; else lo = mid + 1;
        ++probes;
      }
      t5 = Stamp();
      if (e >= 0) {
        id = g.nbr[row_ptr + lo];
        w = __fsub_rn(nw[lo], lo == 0 ? 0.f : nw[lo - 1]);
      }
      t6 = Stamp();
    }
    out_id[s] = id;
    out_w[s] = w;
    const uint64_t t7 = Stamp();
    ph[0] += t1 - t0; ph[1] += t2 - t1; ph[2] += t3 - t2; ph[3] += t4 - t3;
    ph[4] += t5 - t4; ph[5] += t6 - t5; ph[6] += t7 - t6;
    ++iters;
  }
  // the wave's time is what its last lane saw; take lane 0 as the witness of
  // the stamps (they are wave-uniform) and the max probe count of the wave
  unsigned long long pmax = probes;
  for (int off = 32; off > 0; off >>= 1) {
    const unsigned long long o = __shfl_xor(pmax, off);
    pmax = o > pmax ? o : pmax;
  }
  if ((threadIdx.x & 63) == 0) {
    for (int i = 0; i < 7; ++i) atomicAdd(&acc[i], ph[i]);
    atomicAdd(&acc[8], iters);
    atomicAdd(&acc[9], pmax);
  }
}

}  // namespace euler_gpu

extern "C" int euler_gpu_debug_k1_phases(const euler_gpu_graph* g, void* stream,
                                         uint64_t seed, const uint64_t* roots_dev,
                                         int64_t n, int32_t count,
                                         uint64_t* out_id_dev, float* out_w_dev,
                                         int32_t grid, uint64_t* acc_host) {
  using namespace euler_gpu;
  if (!g || g->view.T != 1 || g->view.map_mode != 0)
    return Fail(EULER_GPU_EINVAL, "debug_k1_phases: needs a single-type identity graph");
  hipStream_t st = (hipStream_t)stream;
  unsigned long long* acc = nullptr;
  EG_HIP(hipMalloc((void**)&acc, 16 * 8));
  EG_HIP(hipMemsetAsync(acc, 0, 16 * 8, st));
  hipLaunchKernelGGL(K1PhaseKernel, dim3(grid), dim3(256), 0, st, g->view, seed, 0u,
                     roots_dev, n, count, out_id_dev, out_w_dev, acc);
  EG_HIP(hipGetLastError());
  EG_HIP(hipMemcpyAsync(acc_host, acc, 16 * 8, hipMemcpyDeviceToHost, st));
  EG_HIP(hipStreamSynchronize(st));
  EG_HIP(hipFree(acc));
  return EULER_GPU_OK;
}
