// Micro-benchmark 2: vector-L1 (TCP) throughput of gather patterns whose
// addresses do NOT depend on loaded data (throughput, not latency).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_tcp.hip -o tools/ubench_tcp
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// MODE 0 divergent dword | 1 divergent dwordx4 | 2 16-lane group dword (same line)
// 3 4-lane group dwordx4 (same line) | 4 wave-coalesced dword | 5 8-lane group dwordx2
// 6 divergent dwordx2 | 7 all lanes same address dword
template <int MODE>
__global__ __launch_bounds__(256) void Gather(const uint32_t* __restrict__ base,
                                              uint32_t line_mask, int iters,
                                              uint32_t* __restrict__ sink) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t lane = threadIdx.x & 63;
  uint32_t g = tid;
  if (MODE == 2) g = tid >> 4;
  if (MODE == 3) g = tid >> 2;
  if (MODE == 4 || MODE == 7) g = tid >> 6;
  if (MODE == 5) g = tid >> 3;
  uint32_t x = g * 2654435761u + 12345u;
  uint32_t acc = 0;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      x = x * 1664525u + 1013904223u;
      uint32_t line = (x >> 7) & line_mask;
      if (MODE == 4) line &= ~3u;
      const uint32_t* p = base + (size_t)line * 16;
      if (MODE == 0) acc += p[lane & 15];
      else if (MODE == 1) { uint4 q = *reinterpret_cast<const uint4*>(p + 4 * (lane & 3)); acc += q.x ^ q.y ^ q.z ^ q.w; }
      else if (MODE == 2) acc += p[lane & 15];
      else if (MODE == 3) { uint4 q = *reinterpret_cast<const uint4*>(p + 4 * (lane & 3)); acc += q.x ^ q.y ^ q.z ^ q.w; }
      else if (MODE == 4) acc += p[lane];
      else if (MODE == 5) { uint2 q = *reinterpret_cast<const uint2*>(p + 2 * (lane & 7)); acc += q.x ^ q.y; }
      else if (MODE == 6) { uint2 q = *reinterpret_cast<const uint2*>(p + 2 * (lane & 7)); acc += q.x ^ q.y; }
      else acc += p[3];
    }
  }
  if (acc == 0x12345678u) sink[tid] = acc;
}

template <int MODE>
static void Run(const uint32_t* buf, uint64_t bytes, const char* name, uint32_t* sink) {
  const int block = 256, grid = 256 * 8, iters = 32;
  const uint32_t mask = (uint32_t)(bytes / 64 - 1);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(Gather<MODE>, dim3(grid), dim3(block), 0, 0, buf, mask, iters, sink);
  CK(hipEventRecord(e0, 0));
  const int reps = 5;
  for (int r = 0; r < reps; ++r)
    hipLaunchKernelGGL(Gather<MODE>, dim3(grid), dim3(block), 0, 0, buf, mask, iters, sink);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= reps;
  const double wave_instr = (double)grid * block / 64 * iters * 8;
  printf("%-26s ws=%9.3f MB %8.3f ms  %8.2f clk/CU per wave-load  %8.1f G lane-loads/s\n",
         name, bytes / 1e6, ms, ms * 1e-3 * 2.4e9 * 256 / wave_instr,
         wave_instr * 64 / ms / 1e6);
}

__global__ void Fill(uint32_t* p, uint64_t n) {
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n;
       i += (uint64_t)gridDim.x * blockDim.x) p[i] = (uint32_t)(i * 2654435761u);
}

int main() {
  const uint64_t max_bytes = 4ULL << 30;
  uint32_t* buf; uint32_t* sink;
  CK(hipMalloc(&buf, max_bytes));
  CK(hipMalloc(&sink, 256 * 8 * 256 * 4));
  hipLaunchKernelGGL(Fill, dim3(4096), dim3(256), 0, 0, buf, max_bytes / 4);
  CK(hipDeviceSynchronize());
  const uint64_t sizes[] = {8ULL << 10, 1ULL << 20, 64ULL << 20, 4ULL << 30};
  for (uint64_t s : sizes) {
    Run<0>(buf, s, "divergent dword", sink);
    Run<6>(buf, s, "divergent dwordx2", sink);
    Run<1>(buf, s, "divergent dwordx4", sink);
    Run<3>(buf, s, "4-lane line, dwordx4", sink);
    Run<5>(buf, s, "8-lane line, dwordx2", sink);
    Run<2>(buf, s, "16-lane line, dword", sink);
    Run<4>(buf, s, "wave 256B, dword", sink);
    Run<7>(buf, s, "wave same addr, dword", sink);
    printf("\n");
  }
  return 0;
}
