// A C++ client of the Query boundary (include/euler_query.h) the way the reference's TF kernels use it:
// T caller threads keep queries in flight (client/query_proxy.cc:205-210 runs 8), each query the 2-hop
// chain tf_euler/kernels/sample_fanout_op.cc:37-42 builds, host tensors in and out.  Prints the time per
// query for 1 and for T callers - the aggregate rate of the boundary without a Python harness.
//   query_clients [nodes = 10000000] [batch = 1024] [threads = 8] [queries per thread = 200]
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <random>
#include <thread>
#include <vector>

#include "euler_gpu.h"
#include "euler_query.h"

static void RunOne(euler::QueryProxy* proxy, const std::vector<uint64_t>& roots, int c1, int c2) {
  euler::Query q("v(nodes).sampleNB(et_0,nb_count_0,-1).as(nb_0).sampleNB(et_1,nb_count_1,-1).as(nb_1)");
  euler::Tensor* t = q.AllocInput("nodes", {roots.size()}, euler::kUInt64);
  std::copy(roots.begin(), roots.end(), t->Raw<uint64_t>());
  *q.AllocInput("et_0", {1}, euler::kInt32)->Raw<int32_t>() = 0;
  *q.AllocInput("et_1", {1}, euler::kInt32)->Raw<int32_t>() = 0;
  *q.AllocInput("nb_count_0", {}, euler::kInt32)->Raw<int32_t>() = c1;
  *q.AllocInput("nb_count_1", {}, euler::kInt32)->Raw<int32_t>() = c2;
  std::mutex mu;
  std::condition_variable cv;
  bool done = false;
  proxy->RunAsyncGremlin(&q, [&] {
    std::lock_guard<std::mutex> lk(mu);
    done = true;
    cv.notify_one();
  });
  std::unique_lock<std::mutex> lk(mu);
  cv.wait(lk, [&] { return done; });
  euler::Tensor* r = q.GetResult("nb_1:1");
  if (r == nullptr || (size_t)r->NumElements() != roots.size() * c1 * c2) {
    fprintf(stderr, "query produced no nb_1:1\n");
    exit(2);
  }
}

int main(int argc, char** argv) {
  const int64_t n = argc > 1 ? atoll(argv[1]) : 10000000;
  const int batch = argc > 2 ? atoi(argv[2]) : 1024;
  const int threads = argc > 3 ? atoi(argv[3]) : 8;
  const int per = argc > 4 ? atoi(argv[4]) : 200;
  euler_gpu_synth_params p{};
  p.seed = 7; p.n_nodes = n; p.n_edges_target = 10 * n; p.n_types = 1; p.weighted = 1;
  p.scale = 1;
  while ((1LL << p.scale) < n) ++p.scale;
  for (int z = 0; z < 64; ++z) p.deg_table[z] = 9.0;        // min degree 1 + 9: ~10 edges a node
  euler_gpu_graph* g = nullptr;
  if (euler_gpu_graph_create_synthetic(&p, 0, 1, 0, 1, &g) != 0) {
    fprintf(stderr, "graph: %s\n", euler_gpu_last_error());
    return 1;
  }
  euler::QueryProxy::Init(g);
  euler::QueryProxy* proxy = euler::QueryProxy::GetInstance();
  std::vector<std::vector<uint64_t>> roots(threads, std::vector<uint64_t>(batch));
  std::mt19937_64 rng(5);
  for (auto& r : roots)
    for (auto& x : r) x = 1 + rng() % (uint64_t)n;
  for (int i = 0; i < 32; ++i) RunOne(proxy, roots[0], 25, 10);     // every proxy thread warm
  const double edges = (double)batch * (25 + 250);
  for (int T : {1, threads}) {
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> th;
    for (int k = 0; k < T; ++k)
      th.emplace_back([&, k] { for (int i = 0; i < per; ++i) RunOne(proxy, roots[k], 25, 10); });
    for (auto& t : th) t.join();
    const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("callers %d batch %d: %.4f ms per query, %.3f G sampled edges/s\n", T, batch,
           s / (T * per) * 1e3, edges * T * per / s / 1e9);
  }
  return 0;
}
