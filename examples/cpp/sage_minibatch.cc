// A C++ host - what tf_euler's op kernels are - building one GraphSAGE
// minibatch through include/euler_gpu.h and the HIP runtime ONLY (no Python, no
// torch): the drop-in boundary of this backend used the way a maintainer of the
// reference would use it from tf_euler/kernels/*.cc.
//
//   InitQueryProxy("mode=local;data_path=...")   tf_euler/utils/init_query_proxy.cc:19-37
//   SampleNode     -> euler_gpu_sample_node      tf_euler/kernels/sample_node_op.cc
//   SampleFanout   -> euler_gpu_sample_fanout    tf_euler/kernels/sample_fanout_op.cc
//   GetDenseFeature-> euler_gpu_get_dense_feature  tf_euler/kernels/get_dense_feature_op.cc
//   MPScatterAdd   -> euler_gpu_gather_segment_reduce (euler_gpu_scatter_add for arbitrary indices)
//                                               tf_euler/kernels/scatter_op.cc
//
// usage: sage_minibatch <data_path> <seed> <batch> <fanout1> <fanout2> <feature id> <dim>
// Prints one line per result array (ids in decimal, floats as hex bit patterns)
// so that a test can compare it bit for bit with the Python surface / the oracle.
// Build: see examples/cpp/Makefile (hipcc; links ../../euler_amd/lib/libeuler_gpu.so).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "euler_gpu.h"

#define HIP_OK(expr)                                                            \
  do {                                                                          \
    hipError_t e_ = (expr);                                                     \
    if (e_ != hipSuccess) {                                                     \
      fprintf(stderr, "%s: %s\n", #expr, hipGetErrorString(e_));                \
      return 2;                                                                 \
    }                                                                           \
  } while (0)
#define EULER_OK(expr)                                                          \
  do {                                                                          \
    if ((expr) != EULER_GPU_OK) {                                               \
      fprintf(stderr, "%s: %s\n", #expr, euler_gpu_last_error());               \
      return 3;                                                                 \
    }                                                                           \
  } while (0)

template <typename T>
static int DeviceAlloc(T** p, size_t count) {
  return hipMalloc(reinterpret_cast<void**>(p), (count ? count : 1) * sizeof(T)) == hipSuccess ? 0 : 1;
}

template <typename T>
static std::vector<T> ToHost(const T* dev, size_t count) {
  std::vector<T> h(count);
  if (count) (void)hipMemcpy(h.data(), dev, count * sizeof(T), hipMemcpyDeviceToHost);
  return h;
}

static void PrintIds(const char* name, const std::vector<uint64_t>& v) {
  printf("%s", name);
  for (uint64_t x : v) printf(" %llu", (unsigned long long)x);
  printf("\n");
}

static void PrintBits(const char* name, const std::vector<float>& v) {
  printf("%s", name);
  for (float x : v) {
    uint32_t b;
    memcpy(&b, &x, 4);
    printf(" %08x", b);
  }
  printf("\n");
}

int main(int argc, char** argv) {
  if (argc < 8) {
    fprintf(stderr, "usage: %s data_path seed batch fanout1 fanout2 feature_id dim\n", argv[0]);
    return 1;
  }
  const std::string conf = std::string("mode=local;data_path=") + argv[1] + ";device=0";
  const uint64_t seed = strtoull(argv[2], nullptr, 10);
  const int32_t batch = atoi(argv[3]);
  const int32_t counts[2] = {atoi(argv[4]), atoi(argv[5])};
  const int32_t fid = atoi(argv[6]), dim = atoi(argv[7]);
  if (!InitQueryProxy(conf.c_str())) {
    fprintf(stderr, "InitQueryProxy failed: %s\n", euler_gpu_last_error());
    return 3;
  }
  euler_gpu_graph* g = euler_gpu_default_graph();
  hipStream_t stream;
  HIP_OK(hipStreamCreate(&stream));
  const int64_t n1 = (int64_t)batch * counts[0], n2 = n1 * counts[1];

  // roots: SampleNode over all node types (call id 0)
  uint64_t* roots;
  if (DeviceAlloc(&roots, batch)) return 2;
  const int32_t all_types[1] = {-1};
  EULER_OK(euler_gpu_sample_node(g, stream, seed, 0, all_types, 1, batch, roots));

  // two hops over edge types {0, 1} (call ids 1, 2), default node -1
  const int32_t edge_types[4] = {0, 1, 0, 1};
  uint64_t* ids[2]; float* w[2]; int32_t* t[2];
  const int64_t sizes[2] = {n1, n2};
  for (int h = 0; h < 2; ++h)
    if (DeviceAlloc(&ids[h], sizes[h]) || DeviceAlloc(&w[h], sizes[h]) || DeviceAlloc(&t[h], sizes[h]))
      return 2;
  void* workspace;
  if (DeviceAlloc(reinterpret_cast<uint8_t**>(&workspace),
                  euler_gpu_sample_fanout_workspace(batch, counts, 2)))
    return 2;
  EULER_OK(euler_gpu_sample_fanout(g, stream, seed, 1, roots, batch, edge_types, 2, counts, 2,
                                   -1, ids, w, t, workspace));

  // features of the second hop's nodes, summed into their first-hop parents
  float *feat, *agg;
  if (DeviceAlloc(&feat, n2 * dim) || DeviceAlloc(&agg, n1 * dim)) return 2;
  EULER_OK(euler_gpu_get_dense_feature(g, stream, ids[1], n2, fid, dim, feat));
  // the counts[1] second-hop rows of a first-hop node are consecutive: MPScatterAdd with the
  // index i / counts[1] is a segment reduce with a fixed segment length (same adds, same order)
  EULER_OK(euler_gpu_gather_segment_reduce(stream, /*mode=add*/ 0, feat, /*gather_indices=*/nullptr,
                                           /*seg_ptr=*/nullptr, counts[1], dim, (int32_t)n1, agg));
  HIP_OK(hipStreamSynchronize(stream));

  PrintIds("roots", ToHost(roots, batch));
  PrintIds("hop1", ToHost(ids[0], n1));
  PrintIds("hop2", ToHost(ids[1], n2));
  PrintBits("w2", ToHost(w[1], n2));
  PrintBits("agg", ToHost(agg, n1 * dim));
  return 0;
}
