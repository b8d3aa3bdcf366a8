// Micro-benchmark: what does one random line lookup cost on gfx950 depending
// on how the 64 lanes of a wave share lines?  Drives the K1 search layout.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_gather.hip -o gpurun_out/ubench_gather
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ uint64_t Mix(uint64_t z) {
  z ^= z >> 30; z *= 0xbf58476d1ce4e5b9ULL;
  z ^= z >> 27; z *= 0x94d049bb133111ebULL;
  z ^= z >> 31;
  return z;
}

// MODE 0: every lane its own line, one dword
// MODE 1: every lane its own line, one dwordx4
// MODE 2: 16-lane groups share a line, one dword per lane (4 lines / instr)
// MODE 3: whole wave reads 256 contiguous bytes (4 lines / instr)
// MODE 4: every lane its own line, 4 x dwordx4 (whole line per lane)
// MODE 5: 4-lane groups share a line (one dwordx4 per lane = whole line per quad)
template <int MODE>
__global__ __launch_bounds__(256) void Chase(const uint32_t* __restrict__ base,
                                             uint64_t n_lines, int iters,
                                             uint32_t* __restrict__ sink) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t lane = threadIdx.x & 63;
  uint64_t x;
  if (MODE == 2) x = Mix(tid >> 4);
  else if (MODE == 3) x = Mix(tid >> 6);
  else if (MODE == 5) x = Mix(tid >> 2);
  else x = Mix(tid);
  uint32_t acc = 0;
  for (int i = 0; i < iters; ++i) {
    uint64_t line = (x >> 8) % n_lines;
    if (MODE == 3) line &= ~3ULL;
    const uint32_t* p = base + line * 16;
    uint32_t v;
    if (MODE == 0) {
      v = p[lane & 15];
    } else if (MODE == 1) {
      const uint4 q = *reinterpret_cast<const uint4*>(p + 4 * (lane & 3));
      v = q.x ^ q.y ^ q.z ^ q.w;
    } else if (MODE == 2) {
      v = p[lane & 15];
      v = __shfl(v, lane & 48);   // group leader's value: keeps the group in step
    } else if (MODE == 3) {
      v = p[lane];
      v = __shfl(v, 0);
    } else if (MODE == 4) {
      const uint4* q4 = reinterpret_cast<const uint4*>(p);
      const uint4 a = q4[0], b = q4[1], c = q4[2], d = q4[3];
      v = a.x ^ b.y ^ c.z ^ d.w;
    } else {
      const uint4 q = *reinterpret_cast<const uint4*>(p + 4 * (lane & 3));
      v = q.x ^ q.y ^ q.z ^ q.w;
      v = __shfl(v, lane & 60);
    }
    acc += v;
    x = Mix(x + v + 0x9E3779B97F4A7C15ULL);
  }
  if (acc == 0x12345678u) sink[tid] = acc;
}

__global__ void Fill(uint32_t* p, uint64_t n) {
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n;
       i += (uint64_t)gridDim.x * blockDim.x)
    p[i] = (uint32_t)Mix(i);
}

template <int MODE>
static void Run(const uint32_t* buf, uint64_t n_lines, const char* name,
                uint32_t* sink) {
  const int block = 256, grid = 256 * 8, iters = 64;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(Chase<MODE>, dim3(grid), dim3(block), 0, 0, buf, n_lines, iters, sink);
  CK(hipEventRecord(e0, 0));
  const int reps = 5;
  for (int r = 0; r < reps; ++r)
    hipLaunchKernelGGL(Chase<MODE>, dim3(grid), dim3(block), 0, 0, buf, n_lines, iters, sink);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= reps;
  const double lane_loads = (double)grid * block * iters;
  double lines_per_lane = 1.0;
  if (MODE == 2) lines_per_lane = 1.0 / 16;
  if (MODE == 3) lines_per_lane = 4.0 / 64;
  if (MODE == 5) lines_per_lane = 1.0 / 4;
  const double lines = lane_loads * lines_per_lane;
  printf("%-28s ws=%8.1f MB  %7.3f ms  %7.1f G lane-steps/s  %7.1f G lines/s  "
         "%6.2f clk/CU per wave-step (2.4GHz)\n",
         name, n_lines * 64.0 / 1e6, ms, lane_loads / ms / 1e6, lines / ms / 1e6,
         ms * 1e-3 * 2.4e9 * 256 / (lane_loads / 64));
}

int main() {
  const uint64_t max_bytes = 8ULL << 30;
  uint32_t* buf; uint32_t* sink;
  CK(hipMalloc(&buf, max_bytes));
  CK(hipMalloc(&sink, 256 * 8 * 256 * 4));
  hipLaunchKernelGGL(Fill, dim3(4096), dim3(256), 0, 0, buf, max_bytes / 4);
  CK(hipDeviceSynchronize());
  const uint64_t sizes[] = {2ULL << 20, 16ULL << 20, 128ULL << 20, 1ULL << 30, 8ULL << 30};
  for (uint64_t s : sizes) {
    const uint64_t nl = s / 64;
    Run<0>(buf, nl, "lane-divergent dword", sink);
    Run<1>(buf, nl, "lane-divergent dwordx4", sink);
    Run<4>(buf, nl, "lane-divergent 4xdwordx4", sink);
    Run<5>(buf, nl, "quad-shared line dwordx4", sink);
    Run<2>(buf, nl, "16-lane-shared line dword", sink);
    Run<3>(buf, nl, "wave-coalesced 256B dword", sink);
    printf("\n");
  }
  return 0;
}
