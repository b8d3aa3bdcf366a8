// A C++ multi-GPU host over the C ABI: one thread per GPU of the node, the graph
// hash-partitioned over them (owner(id) = (id % N) % N, core/kernels/
// id_split_op.cc:46-49), one euler_gpu_sharded_sample_fanout call per rank and
// minibatch - the place of the reference's ID_SPLIT -> REMOTE (gRPC) ->
// IDX_MERGE / DATA_MERGE sub-DAG (core/kernels/remote_op.cc:60-142), with RCCL
// (ncclSend / ncclRecv groups over xGMI) as the transport.  No Python, no torch.
//
//   usage: sharded_fanout [n_gpus (default: all visible)] [nodes] [batch]
// Every rank samples its own batch with fanout [25, 10] and walks 40 steps from the same
// roots (euler_gpu_sharded_random_walk, BASELINE configs[3]); rank 0's GPU also holds the
// UNSHARDED graph and the program checks that each rank's results equal the unsharded
// euler_gpu_sample_fanout / euler_gpu_random_walk of the same roots, bit for bit.
// Build: examples/cpp/Makefile (hipcc, -lrccl, ../../euler_amd/lib/libeuler_gpu.so).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <mutex>
#include <thread>
#include <vector>

#include "euler_gpu.h"

#define CHECK(expr)                                                                   \
  do {                                                                                \
    if (!(expr)) { fprintf(stderr, "FAILED %s (%s:%d) %s\n", #expr, __FILE__, __LINE__, \
                           euler_gpu_last_error()); exit(2); }                        \
  } while (0)

// deg_table of the synthetic power-law graph (euler_amd/graph.py: synth_params)
static void FillParams(euler_gpu_synth_params* p, uint64_t seed, int64_t n, int64_t e) {
  memset(p, 0, sizeof(*p));
  int scale = 1;
  while ((1LL << scale) < n) ++scale;
  p->seed = seed; p->n_nodes = n; p->n_edges_target = e; p->scale = scale;
  p->n_types = 1; p->weighted = 1;
  std::vector<double> cnt(65, 0.0);
  const uint64_t mask = (1ULL << scale) - 1;
  for (int64_t x = 0; x < n; ++x) cnt[__builtin_popcountll((uint64_t)x & mask)] += 1.0;
  double norm = 0, wz[65];
  for (int z = 0; z <= scale; ++z) { wz[z] = pow(0.76, scale - z) * pow(0.24, z); norm += cnt[z] * wz[z]; }
  const double extra = e > n ? (double)(e - n) : 0.0;
  for (int z = 0; z < 64; ++z) p->deg_table[z] = (z <= scale && norm > 0) ? extra * wz[z] / norm : 0.0;
}

int main(int argc, char** argv) {
  int visible = 0;
  CHECK(hipGetDeviceCount(&visible) == hipSuccess && visible > 0);
  const int N = argc > 1 ? atoi(argv[1]) : visible;
  const int64_t nodes = argc > 2 ? atoll(argv[2]) : 1000000;
  const int64_t B = argc > 3 ? atoll(argv[3]) : 4096;
  CHECK(N >= 1 && N <= visible);
  euler_gpu_synth_params sp;
  FillParams(&sp, 20240521, nodes, 10 * nodes);
  const int32_t counts[2] = {25, 10}, et[2] = {0, 0};
  const uint64_t seed = 7;
  const int64_t default_node = nodes + 1;

  std::vector<ncclComm_t> comms(N);
  std::vector<int> devs(N);
  for (int r = 0; r < N; ++r) devs[r] = r;
  CHECK(ncclCommInitAll(comms.data(), N, devs.data()) == ncclSuccess);

  CHECK(hipSetDevice(0) == hipSuccess);
  euler_gpu_graph* whole = nullptr;
  CHECK(euler_gpu_graph_create_synthetic(&sp, 0, 1, 0, 1, &whole) == EULER_GPU_OK);

  std::atomic<int> bad(0);
  std::atomic<long long> checked(0), walked(0);
  auto rank_main = [&](int r) {
    CHECK(hipSetDevice(r) == hipSuccess);
    hipStream_t st;
    CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess);
    euler_gpu_graph* shard = nullptr;
    CHECK(euler_gpu_graph_create_synthetic(&sp, r, N, r, N, &shard) == EULER_GPU_OK);
    euler_gpu_transport tr;
    CHECK(euler_gpu_transport_rccl(comms[r], r, N, nullptr, &tr) == EULER_GPU_OK);
    // this rank's batch
    std::vector<uint64_t> roots((size_t)B);
    uint64_t x = 0x9E3779B97F4A7C15ULL * (uint64_t)(r + 1);
    for (auto& v : roots) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; v = x % (uint64_t)nodes + 1; }
    const int64_t n1 = B * counts[0], n2 = n1 * counts[1];
    uint64_t *d_roots, *id1, *id2; float *w1, *w2; int32_t *t1, *t2; uint8_t* ws;
    CHECK(hipMalloc((void**)&d_roots, B * 8) == hipSuccess);
    CHECK(hipMalloc((void**)&id1, n1 * 8) == hipSuccess && hipMalloc((void**)&id2, n2 * 8) == hipSuccess);
    CHECK(hipMalloc((void**)&w1, n1 * 4) == hipSuccess && hipMalloc((void**)&w2, n2 * 4) == hipSuccess);
    CHECK(hipMalloc((void**)&t1, n1 * 4) == hipSuccess && hipMalloc((void**)&t2, n2 * 4) == hipSuccess);
    CHECK(hipMalloc((void**)&ws, euler_gpu_sample_fanout_workspace(B, counts, 2)) == hipSuccess);
    CHECK(hipMemcpy(d_roots, roots.data(), B * 8, hipMemcpyHostToDevice) == hipSuccess);
    uint64_t* ids[2] = {id1, id2}; float* ws_[2] = {w1, w2}; int32_t* ts[2] = {t1, t2};
    CHECK(euler_gpu_sharded_sample_fanout(shard, &tr, st, seed, 100, d_roots, B, et, 1, counts, 2,
                                          default_node, N, ids, ws_, ts, ws) == EULER_GPU_OK);
    CHECK(hipStreamSynchronize(st) == hipSuccess);
    std::vector<uint64_t> got((size_t)n2);
    std::vector<float> gotw((size_t)n2);
    CHECK(hipMemcpy(got.data(), id2, n2 * 8, hipMemcpyDeviceToHost) == hipSuccess);
    CHECK(hipMemcpy(gotw.data(), w2, n2 * 4, hipMemcpyDeviceToHost) == hipSuccess);
    // configs[3] from the same host: DeepWalk, random_walk of length 40 (p = q = 1) from the
    // batch's roots over the sharded graph (tf_euler/kernels/random_walk_op.cc:207-247) - one C
    // call per rank (two cohorts of walkers whose steps alternate on the stream)
    const int32_t WL = 40;
    std::vector<int32_t> wet((size_t)WL, 0);
    int64_t* d_walk = nullptr;
    int64_t wstats[4] = {0, 0, 0, 0};
    CHECK(hipMalloc((void**)&d_walk, (size_t)B * (WL + 1) * 8) == hipSuccess);
    CHECK(euler_gpu_sharded_random_walk(shard, &tr, st, seed, 200, (const int64_t*)d_roots, B, wet.data(), 1,
                                        WL, default_node, N, 2, nullptr, 0, d_walk, wstats) == EULER_GPU_OK);
    CHECK(hipStreamSynchronize(st) == hipSuccess);
    std::vector<int64_t> gotwalk((size_t)B * (WL + 1));
    CHECK(hipMemcpy(gotwalk.data(), d_walk, gotwalk.size() * 8, hipMemcpyDeviceToHost) == hipSuccess);
    // the walk is ENQUEUED: one host exchange per call (the ranks' walker counts; none when this rank
    // is alone and skips its exchanges) - euler_gpu_set_tuning(63, 0) waits once per step and cohort
    CHECK(wstats[0] <= 1 && wstats[1] > 0);
    // the same roots on the unsharded graph (device 0)
    static std::mutex mu;
    std::lock_guard<std::mutex> lk(mu);
    CHECK(hipSetDevice(0) == hipSuccess);
    uint64_t *u_roots, *u1, *u2; float *uw1, *uw2; int32_t *ut1, *ut2; uint8_t* uws;
    CHECK(hipMalloc((void**)&u_roots, B * 8) == hipSuccess);
    CHECK(hipMalloc((void**)&u1, n1 * 8) == hipSuccess && hipMalloc((void**)&u2, n2 * 8) == hipSuccess);
    CHECK(hipMalloc((void**)&uw1, n1 * 4) == hipSuccess && hipMalloc((void**)&uw2, n2 * 4) == hipSuccess);
    CHECK(hipMalloc((void**)&ut1, n1 * 4) == hipSuccess && hipMalloc((void**)&ut2, n2 * 4) == hipSuccess);
    CHECK(hipMalloc((void**)&uws, euler_gpu_sample_fanout_workspace(B, counts, 2)) == hipSuccess);
    CHECK(hipMemcpy(u_roots, roots.data(), B * 8, hipMemcpyHostToDevice) == hipSuccess);
    uint64_t* uids[2] = {u1, u2}; float* uw[2] = {uw1, uw2}; int32_t* ut[2] = {ut1, ut2};
    CHECK(euler_gpu_sample_fanout(whole, nullptr, seed, 100, u_roots, B, et, 1, counts, 2,
                                  default_node, uids, uw, ut, uws) == EULER_GPU_OK);
    CHECK(hipDeviceSynchronize() == hipSuccess);
    std::vector<uint64_t> want((size_t)n2);
    std::vector<float> wantw((size_t)n2);
    CHECK(hipMemcpy(want.data(), u2, n2 * 8, hipMemcpyDeviceToHost) == hipSuccess);
    CHECK(hipMemcpy(wantw.data(), uw2, n2 * 4, hipMemcpyDeviceToHost) == hipSuccess);
    if (memcmp(got.data(), want.data(), (size_t)n2 * 8) != 0 ||
        memcmp(gotw.data(), wantw.data(), (size_t)n2 * 4) != 0) bad = 1;
    checked += n1 + n2;
    int64_t* u_walk = nullptr;
    CHECK(hipMalloc((void**)&u_walk, (size_t)B * (WL + 1) * 8) == hipSuccess);
    CHECK(euler_gpu_random_walk(whole, nullptr, seed, 200, (const int64_t*)u_roots, B, wet.data(), 1, WL,
                                1.0f, 1.0f, default_node, u_walk) == EULER_GPU_OK);
    CHECK(hipDeviceSynchronize() == hipSuccess);
    std::vector<int64_t> wantwalk((size_t)B * (WL + 1));
    CHECK(hipMemcpy(wantwalk.data(), u_walk, wantwalk.size() * 8, hipMemcpyDeviceToHost) == hipSuccess);
    if (memcmp(gotwalk.data(), wantwalk.data(), wantwalk.size() * 8) != 0) bad = 1;
    walked += B * WL;
    (void)hipFree(u_walk);
    (void)hipFree(u_roots); (void)hipFree(u1); (void)hipFree(u2); (void)hipFree(uw1);
    (void)hipFree(uw2); (void)hipFree(ut1); (void)hipFree(ut2); (void)hipFree(uws);
    euler_gpu_transport_rccl_release(&tr);
    CHECK(hipSetDevice(r) == hipSuccess);
    euler_gpu_graph_destroy(shard);
  };
  std::vector<std::thread> pool;
  for (int r = 0; r < N; ++r) pool.emplace_back(rank_main, r);
  for (auto& t : pool) t.join();
  for (auto& c : comms) ncclCommDestroy(c);
  if (bad) { printf("sharded_fanout MISMATCH\n"); return 1; }
  printf("sharded_fanout OK: %d rank(s), %lld sampled edges and %lld walker steps (random_walk of length 40) "
         "identical to the unsharded graph\n", N, (long long)checked.load(), (long long)walked.load());
  return 0;
}
