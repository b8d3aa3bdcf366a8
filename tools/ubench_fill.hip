// Write-only bandwidth of the chip: what bounds DedupExpandKernel (0.58 GB of
// row writes per hop-2 launch).  hipcc --offload-arch=gfx950 -O3 tools/ubench_fill.hip -o /tmp/ubench_fill
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s\n", hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void Fill(uint64_t* out, int64_t n16, uint64_t v) {
  typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) {
    u64x2 x = {v + (uint64_t)i, v};
    if (MODE == 0) *reinterpret_cast<u64x2*>(out + 2 * i) = x;
    else __builtin_nontemporal_store(x, reinterpret_cast<u64x2*>(out + 2 * i));
  }
}

// three streams like the expand: 8-byte ids (16-B stores), 4-byte w, 4-byte t (8-B stores)
template <int MODE>
__global__ __launch_bounds__(256) void Fill3(uint64_t* o_id, float* o_w, int32_t* o_t,
                                             int64_t n2, uint64_t v) {
  typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  typedef int i32x2 __attribute__((ext_vector_type(2)));
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += stride) {
    u64x2 x = {v + (uint64_t)i, v};
    f32x2 w = {(float)i, 1.f};
    i32x2 t = {(int)i, 0};
    if (MODE == 0) {
      *reinterpret_cast<u64x2*>(o_id + 2 * i) = x;
      *reinterpret_cast<f32x2*>(o_w + 2 * i) = w;
      *reinterpret_cast<i32x2*>(o_t + 2 * i) = t;
    } else {
      __builtin_nontemporal_store(x, reinterpret_cast<u64x2*>(o_id + 2 * i));
      __builtin_nontemporal_store(w, reinterpret_cast<f32x2*>(o_w + 2 * i));
      __builtin_nontemporal_store(t, reinterpret_cast<i32x2*>(o_t + 2 * i));
    }
  }
}

int main() {
  const int64_t n = 32768000;             // output edges of the metric's hop 2
  uint64_t* id; float* w; int32_t* t;
  CK(hipMalloc(&id, n * 8)); CK(hipMalloc(&w, n * 4)); CK(hipMalloc(&t, n * 4));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const int iters = 20;
  for (int grid : {2048, 8192, 32768, 65536}) {
    for (int mode = 0; mode < 2; ++mode) {
      float ms1, ms3;
      for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(a));
        for (int i = 0; i < iters; ++i) {
          if (mode == 0) hipLaunchKernelGGL(Fill<0>, dim3(grid), dim3(256), 0, 0, id, n / 2, 7ull);
          else hipLaunchKernelGGL(Fill<1>, dim3(grid), dim3(256), 0, 0, id, n / 2, 7ull);
        }
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        CK(hipEventElapsedTime(&ms1, a, b));
        CK(hipEventRecord(a));
        for (int i = 0; i < iters; ++i) {
          if (mode == 0) hipLaunchKernelGGL(Fill3<0>, dim3(grid), dim3(256), 0, 0, id, w, t, n / 2, 7ull);
          else hipLaunchKernelGGL(Fill3<1>, dim3(grid), dim3(256), 0, 0, id, w, t, n / 2, 7ull);
        }
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        CK(hipEventElapsedTime(&ms3, a, b));
      }
      printf("grid %6d %s: ids only %.4f ms = %.2f TB/s | ids+w+t %.4f ms = %.2f TB/s\n", grid,
             mode ? "nontemporal" : "plain      ", ms1 / iters, n * 8.0 / (ms1 / iters * 1e-3) / 1e12,
             ms3 / iters, n * 16.0 / (ms3 / iters * 1e-3) / 1e12);
    }
  }
  return 0;
}
