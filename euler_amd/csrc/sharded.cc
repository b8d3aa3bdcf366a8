// Multi-GPU sampling for C / C++ hosts: one hop = front end (distinct ids bucketed
// by owner) -> id exchange -> owner-side sampling into wire rows -> row exchange ->
// expansion, all behind one C entry point (include/euler_gpu.h,
// euler_gpu_sharded_sample_fanout).  This is what the reference's
// ID_SPLIT -> REMOTE -> IDX_MERGE / DATA_MERGE sub-DAG does over gRPC
// (core/kernels/id_split_op.cc:46-99, remote_op.cc:60-142, idx_merge_op.cc:32-78,
// data_merge_op.cc:44-67); here the exchange is an all-to-all(v) between the GPUs
// of one node through a caller-supplied transport: RCCL (ncclSend / ncclRecv
// groups over xGMI, euler_gpu_transport_rccl) in production, anything else - the
// tests use a host-staged one so that several ranks can share one GPU - through
// the two callbacks of euler_gpu_transport.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "common.h"

using euler_gpu::Fail;

namespace {

// ---- RCCL, resolved at run time: the library never links librccl itself (the
// host process - a C++ trainer linked with -lrccl, or torch, which brings its own
// copy - already has one loaded, and two RCCL copies in one process do not mix).
typedef int (*nccl_send_t)(const void*, size_t, int, int, void*, hipStream_t);
typedef int (*nccl_recv_t)(void*, size_t, int, int, void*, hipStream_t);
typedef int (*nccl_group_t)(void);
struct Rccl {
  nccl_send_t send = nullptr;
  nccl_recv_t recv = nullptr;
  nccl_group_t group_start = nullptr, group_end = nullptr;
  bool ok = false;
};

const Rccl& GetRccl() {
  static Rccl r = [] {
    Rccl x;
    // the global scope first, then copies that were loaded RTLD_LOCAL (torch's)
    void* handles[3] = {RTLD_DEFAULT, dlopen("librccl.so.1", RTLD_NOLOAD | RTLD_LAZY),
                        dlopen("librccl.so", RTLD_NOLOAD | RTLD_LAZY)};
    for (void* h : handles) {
      x.send = (nccl_send_t)dlsym(h, "ncclSend");
      x.recv = (nccl_recv_t)dlsym(h, "ncclRecv");
      x.group_start = (nccl_group_t)dlsym(h, "ncclGroupStart");
      x.group_end = (nccl_group_t)dlsym(h, "ncclGroupEnd");
      if (x.send && x.recv && x.group_start && x.group_end) { x.ok = true; break; }
    }
    if (!x.ok) {
      // nothing loaded yet: load the system copy
      void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
      if (h == nullptr) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
      if (h != nullptr) {
        x.send = (nccl_send_t)dlsym(h, "ncclSend");
        x.recv = (nccl_recv_t)dlsym(h, "ncclRecv");
        x.group_start = (nccl_group_t)dlsym(h, "ncclGroupStart");
        x.group_end = (nccl_group_t)dlsym(h, "ncclGroupEnd");
        x.ok = x.send && x.recv && x.group_start && x.group_end;
      }
    }
    return x;
  }();
  return r;
}

}  // namespace (reopened below)
namespace euler_gpu {
// One rank's ids never leave it, so its hops make no exchange at all - which would leave the
// transport (ncclSend / ncclRecv groups) unexecuted on a one-GPU box.  EULER_GPU_SELF_EXCHANGE=1 (or
// tuning key 52) makes a lone rank send to itself what N ranks send to one another: the
// coverage switch of the tests, never a production setting.
// Process-wide: the per-rank threads of a C++ host, a Python dataloader thread and
// ShardedSampler.force_exchange all read and write this one value (key 52 both ways).
std::atomic<int> g_sharded_self_exchange{-1};
std::atomic<long long> g_sharded_exchanges{0};     // exchanges made (euler_gpu_sharded_exchange_count)
// Tuning key 63: the sharded DeepWalk is ENQUEUED - levels and buckets in slab layout, sizes on the
// device, fixed-size messages - instead of waiting for every step's bucket sizes on the host (0).
std::atomic<int> g_sharded_walk_enqueued{1};
// Tuning key 66: first step of the enqueued walk whose level is sent as it is - no search for
// nodes that several entries share (the single-GPU walk's key 43 rule: late levels hold few NEW
// mergers, and a step's mark pass + table look cost more than the duplicate draws they save).
// 0 = every step deduplicates.  1M x 40 on one rank: 1.86 ms with 0, 1.90 / 1.76 / 1.715 / 1.72 /
// 1.76 / 1.80 with 9 / 12 / 16 / 20 / 28 / 32 (profiles/r6_walk_tail_ab.txt).
std::atomic<int> g_sharded_walk_tail{16};
// Tuning key 67: level T at which the enqueued walk's path writer splits into two passes
// (WalkPathsFromLevelsTail; walks of at least T + 8 steps).  0 = one pass.  1M x 40 on one rank:
// 1.70 ms in one pass, 1.615 / 1.54 / 1.52 / 1.53 / 1.63 / 1.64 with T = 4 / 6 / 10 / 12 / 16 / 24
// (profiles/r6_walk_split_ab.txt).
std::atomic<int> g_sharded_walk_split{10};
}
namespace {
bool SelfExchange() {
  int v = euler_gpu::g_sharded_self_exchange.load();
  if (v < 0) {
    const char* e = getenv("EULER_GPU_SELF_EXCHANGE");
    int want = (e != nullptr && e[0] == '1') ? 1 : 0;
    int expect = -1;
    euler_gpu::g_sharded_self_exchange.compare_exchange_strong(expect, want);
    v = euler_gpu::g_sharded_self_exchange.load();
  }
  return v != 0;
}

struct RcclUser {
  void* comm;
  int32_t rank, world;
  euler_shm* shm;           // host-side counts mailbox, or null
  int64_t* pinned;          // [2 * world] when the counts travel through RCCL
  int64_t* dev_counts;      // [2 * world]
};

int RcclCounts(void* user, const int64_t* send, int64_t* recv) {
  RcclUser* u = (RcclUser*)user;
  if (u->world == 1) { recv[0] = send[0]; return EULER_GPU_OK; }
  if (u->shm != nullptr) return euler_shm_alltoall_i64(u->shm, send, recv, 1, 60000);
  // no mailbox: the counts make a round trip through the GPUs (one sync per hop)
  const Rccl& r = GetRccl();
  for (int p = 0; p < u->world; ++p) u->pinned[p] = send[p];
  EG_HIP(hipMemcpy(u->dev_counts, u->pinned, sizeof(int64_t) * u->world, hipMemcpyHostToDevice));
  if (r.group_start() != 0) return Fail(EULER_GPU_EHIP, "ncclGroupStart failed");
  // a group left open defers or hangs every later collective of this rank while the peers
  // block: remember the first failure, always close the group
  bool failed = false;
  for (int p = 0; p < u->world && !failed; ++p) {
    failed = r.send(u->dev_counts + p, 8, 0, p, u->comm, nullptr) != 0 ||
             r.recv(u->dev_counts + u->world + p, 8, 0, p, u->comm, nullptr) != 0;
  }
  const bool end_failed = r.group_end() != 0;
  if (failed) return Fail(EULER_GPU_EHIP, "ncclSend / ncclRecv (counts) failed");
  if (end_failed) return Fail(EULER_GPU_EHIP, "ncclGroupEnd failed");
  EG_HIP(hipMemcpy(u->pinned + u->world, u->dev_counts + u->world, sizeof(int64_t) * u->world,
                   hipMemcpyDeviceToHost));
  for (int p = 0; p < u->world; ++p) recv[p] = u->pinned[u->world + p];
  return EULER_GPU_OK;
}

int RcclAllToAllV(void* user, const void* send_dev, const int64_t* send_rows, void* recv_dev,
                  const int64_t* recv_rows, int64_t row_bytes, void* stream) {
  RcclUser* u = (RcclUser*)user;
  const Rccl& r = GetRccl();
  hipStream_t st = (hipStream_t)stream;
  if (r.group_start() != 0) return Fail(EULER_GPU_EHIP, "ncclGroupStart failed");
  int64_t so = 0, ro = 0;
  const char* failed = nullptr;          // first failure; the group is closed regardless
  for (int p = 0; p < u->world && failed == nullptr; ++p) {
    const size_t sb = (size_t)(send_rows[p] * row_bytes), rb = (size_t)(recv_rows[p] * row_bytes);
    if (sb && r.send((const uint8_t*)send_dev + so, sb, 0 /* ncclInt8 */, p, u->comm, st) != 0)
      failed = "ncclSend failed";
    else if (rb && r.recv((uint8_t*)recv_dev + ro, rb, 0, p, u->comm, st) != 0)
      failed = "ncclRecv failed";
    so += (int64_t)sb; ro += (int64_t)rb;
  }
  const bool end_failed = r.group_end() != 0;
  if (failed != nullptr) return Fail(EULER_GPU_EHIP, failed);
  if (end_failed) return Fail(EULER_GPU_EHIP, "ncclGroupEnd failed");
  return EULER_GPU_OK;
}

// Device scratch of one call.  Blocks come from a cache of hipMalloc'd blocks kept per (device,
// stream) and go back to it when the call releases them: a block is only ever reused by later work
// of the SAME stream, so stream order makes the reuse safe without a stream-ordered free.
// Why not hipMallocAsync / hipFreeAsync (rounds 4-5): with the per-step buffers of a node2vec walk
// (the rows fetched in a step: gigabytes, a different size every step) one hipMallocAsync in a few
// hundred took 1.5-5 s although the pool kept all its memory (hip-trace of
// tools/sharded_n2v_ab.py: every third to ninth 0.12 s walk took 1.7-5.0 s; identical calls,
// constant free memory) - the driver's bench takes medians and never showed it.  Sizes are rounded
// up to eight classes per octave so that a step finds the blocks of the steps before it.
struct BlockCache {
  std::mutex mu;
  std::map<std::pair<int, void*>, std::multimap<size_t, void*>> free_blocks;
  size_t cached_bytes = 0;
};
BlockCache& Blocks() { static BlockCache* c = new BlockCache(); return *c; }
constexpr size_t kBlockCacheCap = (size_t)32 << 30;       // beyond this, released blocks go back to the driver
                                                          // (a node2vec walk of 100 000 hub walkers holds ~10 GB)

struct Scratch {
  hipStream_t st;
  int dev = 0;
  std::vector<void*> ptrs;                            // stream-ordered allocations of others, freed with the call
  std::vector<std::pair<void*, size_t>> blocks;       // cached blocks in use
  explicit Scratch(hipStream_t s) : st(s) { (void)hipGetDevice(&dev); }
  ~Scratch() {
    for (void* p : ptrs) (void)hipFreeAsync(p, st);
    while (!blocks.empty()) Release(blocks.back().first);
  }
  static size_t SizeClass(size_t bytes) {
    if (bytes < 256) return 256;
    int e = 0;
    while (((size_t)1 << (e + 1)) <= bytes) ++e;            // 2^e <= bytes < 2^(e + 1)
    const size_t step = (size_t)1 << (e > 11 ? e - 3 : 8);
    return (bytes + step - 1) & ~(step - 1);
  }
  void* Get(size_t bytes) {
    const size_t cls = SizeClass(bytes);
    void* p = nullptr;
    {
      BlockCache& c = Blocks();
      std::lock_guard<std::mutex> lk(c.mu);
      auto& fm = c.free_blocks[std::make_pair(dev, (void*)st)];
      auto it = fm.find(cls);
      if (it != fm.end()) { p = it->second; fm.erase(it); c.cached_bytes -= cls; }
    }
    if (p == nullptr && hipMalloc(&p, cls) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    blocks.emplace_back(p, cls);
    return p;
  }
  // a block the call is done with: back to the cache (later work of this stream may take it)
  void Release(void* p) {
    if (p == nullptr) return;
    for (size_t i = 0; i < blocks.size(); ++i) {
      if (blocks[i].first != p) continue;
      const size_t cls = blocks[i].second;
      blocks[i] = blocks.back(); blocks.pop_back();
      BlockCache& c = Blocks();
      bool keep;
      {
        std::lock_guard<std::mutex> lk(c.mu);
        keep = c.cached_bytes + cls <= kBlockCacheCap;
        if (keep) { c.free_blocks[std::make_pair(dev, (void*)st)].emplace(cls, p); c.cached_bytes += cls; }
      }
      if (!keep) { (void)hipStreamSynchronize(st); (void)hipFree(p); }
      return;
    }
    for (size_t i = 0; i < ptrs.size(); ++i)
      if (ptrs[i] == p) { ptrs[i] = ptrs.back(); ptrs.pop_back(); (void)hipFreeAsync(p, st); return; }
  }
};

int32_t PackedWordsHost(int32_t count, int32_t tcol) { return ((3 + tcol) * count + 2 + 1) & ~1; }

// The sharded DeepWalk WITHOUT a host wait per step.  The polled form sizes every step's exchange
// from the bucket sizes, which the host has to see first: 40 steps x (front end -> wait ->
// exchange) made the call host-bound (65 us per step for 33 of kernels on one rank, round 5).
// Here a level never leaves the device:
//   slab      stride = the largest walker count of a rank's cohort + 1, rounded up to the front
//             end's chunk: word 0 = how many entries follow, then the entries, then padding.  A
//             level holds at most as many distinct nodes as the cohort has walkers, so no bucket
//             outgrows its slab - whatever the batch looks like (all roots on one owner included);
//   step s    front end over level s (FrontSlabs, ONE kernel with an id-indexed table: W slabs
//             out, one per owner, + the sizes as the next level's lens) -> W equal messages -> the
//             owners draw over the slabs they got (WalkOwnedSlabs reads the headers) -> W equal
//             messages back INTO level s + 1, whose slabs are the answers in the order asked;
//             next[s][e] = the word of e's answer (or ~ the entry that holds the same node).
// One host exchange per CALL (the ranks' walker counts, which fix the stride) instead of one per
// step; the price is padding on the wire - W x stride words per message round where the polled
// form sends the level's distinct nodes (1M walkers on 8 ranks: 8 x 1 MB per rank, step and
// direction ~ 10 us of xGMI, against a host round trip + the counts mailbox).
int WalkEnqueued(const euler_gpu_graph* shard, const euler_gpu_transport* tr, hipStream_t st,
                 uint64_t seed, uint32_t call_id, const int64_t* starts_dev, int64_t n,
                 const int32_t* et_dev, int32_t k, int32_t L, int64_t default_node, int32_t partitions,
                 int32_t K, uint32_t* dense_owner_dev, int64_t dense_limit, int64_t* out_dev,
                 int64_t* stats_host, bool lone, Scratch& sc) {
  const int32_t W = tr->world;
  int64_t waits = 0;
  std::vector<int64_t> n_of((size_t)W, n);
  if (!lone) {
    std::vector<int64_t> mine((size_t)W, n);
    const int rc = tr->alltoall_counts(tr->user, mine.data(), n_of.data());
    if (rc != EULER_GPU_OK) return rc;
    ++waits;
  }
  struct Cohort {
    int64_t lo = 0, n = 0;
    uint32_t stride = 0;                         // 0: no rank has a walker in this cohort
    std::vector<const uint64_t*> ids;            // [L + 1] levels (0 = the walkers' start nodes, plain)
    std::vector<const int32_t*> next;            // [L]
    uint64_t* bucketed = nullptr;                // [W, stride] the step's ids by owner
    uint64_t* owned = nullptr;                   // [W, stride] what the peers ask of this rank
    uint64_t* drawn = nullptr;                   // [W, stride] its answers
    uint32_t* lens = nullptr;                    // [L + 1, W] lens[s] = entries of level s's slabs (s >= 1)
  };
  std::vector<Cohort> co((size_t)K);
  for (int32_t c = 0; c < K; ++c) {
    Cohort& q = co[(size_t)c];
    q.lo = n * c / K; q.n = n * (c + 1) / K - q.lo;
    int64_t cap = 0;
    for (int32_t p = 0; p < W; ++p) {
      const int64_t np = n_of[(size_t)p] * (c + 1) / K - n_of[(size_t)p] * c / K;
      if (np > cap) cap = np;
    }
    q.ids.assign((size_t)L + 1, nullptr); q.next.assign((size_t)(L > 0 ? L : 1), nullptr);
    q.ids[0] = (const uint64_t*)starts_dev + q.lo;
    if (cap == 0 || L == 0) continue;
    const int64_t stride = (cap + 1 + 2047) & ~(int64_t)2047;
    if (stride * W >= ((int64_t)1 << 30))
      return Fail(EULER_GPU_EINVAL, "sharded_random_walk: walkers x ranks >= 2^30 (key 63 = 0 takes the polled form)");
    q.stride = (uint32_t)stride;
    const size_t slab = (size_t)stride * W;
    // levels 1 .. L and the step's buckets (8 bytes a word), next 0 .. L - 1 (4), the lens
    const size_t bytes = slab * 8 * ((size_t)L + 1 + (lone ? 0 : 2)) + ((size_t)(q.n > 0 ? q.n : 1) + slab * (L - 1)) * 4 +
                         (size_t)(L + 1) * W * 4 + 1024;
    uint8_t* p8 = (uint8_t*)sc.Get(bytes);
    if (!p8) { (void)hipGetLastError(); return Fail(EULER_GPU_ENOMEM, "sharded_random_walk: scratch"); }
    for (int32_t s = 1; s <= L; ++s) { q.ids[(size_t)s] = (const uint64_t*)p8; p8 += slab * 8; }
    q.bucketed = (uint64_t*)p8; p8 += slab * 8;
    if (!lone) { q.owned = (uint64_t*)p8; p8 += slab * 8; q.drawn = (uint64_t*)p8; p8 += slab * 8; }
    q.next[0] = (const int32_t*)p8; p8 += (size_t)(q.n > 0 ? q.n : 1) * 4;
    for (int32_t s = 1; s < L; ++s) { q.next[(size_t)s] = (const int32_t*)p8; p8 += slab * 4; }
    p8 = (uint8_t*)(((uintptr_t)p8 + 255) & ~(uintptr_t)255);
    q.lens = (uint32_t*)p8;
    // (the front end adds its buckets' sizes to lens[s + 1]: one clear per call)
    if (hipMemsetAsync(q.lens, 0, (size_t)(L + 1) * W * 4, st) != hipSuccess)
      return Fail(EULER_GPU_EHIP, "sharded_random_walk: clearing the level sizes failed");
  }
  std::vector<int64_t> rows((size_t)W);
  const int32_t tail = euler_gpu::g_sharded_walk_tail.load();
  for (int32_t s = 0; s < L; ++s) {
    for (int32_t c = 0; c < K; ++c) {
      Cohort& q = co[(size_t)c];
      if (q.stride == 0u) continue;              // (the same on every rank: nobody exchanges)
      const int64_t slab = (int64_t)q.stride * W;
      uint64_t* level = const_cast<uint64_t*>(q.ids[(size_t)s + 1]);
      int rc = euler_gpu::FrontSlabs(st, q.ids[(size_t)s], s == 0 ? q.n : slab, s == 0 ? nullptr : q.lens + (size_t)s * W,
                                     q.stride, partitions, W, dense_owner_dev, dense_limit, q.bucketed, q.stride,
                                     q.lens + (size_t)(s + 1) * W, !lone, tail == 0 || s < tail,
                                     const_cast<int32_t*>(q.next[(size_t)s]));
      if (rc != EULER_GPU_OK) return rc;
      if (lone) {
        rc = euler_gpu::WalkOwnedSlabs(shard, st, seed, call_id, et_dev, k, L, s, q.bucketed,
                                       q.lens + (size_t)(s + 1) * W, W, q.stride, level);
        if (rc != EULER_GPU_OK) return rc;
        continue;
      }
      for (int32_t p = 0; p < W; ++p) rows[(size_t)p] = (int64_t)q.stride;
      rc = tr->alltoallv(tr->user, q.bucketed, rows.data(), q.owned, rows.data(), 8, st);
      if (rc != EULER_GPU_OK) return rc;
      rc = euler_gpu::WalkOwnedSlabs(shard, st, seed, call_id, et_dev, k, L, s, q.owned, nullptr, W, q.stride, q.drawn);
      if (rc != EULER_GPU_OK) return rc;
      rc = tr->alltoallv(tr->user, q.drawn, rows.data(), level, rows.data(), 8, st);
      if (rc != EULER_GPU_OK) return rc;
    }
  }
  // the walkers' paths; long walks in two passes: the levels from T on once per ENTRY of level T,
  // the walkers through the levels before it + the row of the entry they reach (walk_kernels.hip)
  const int32_t T = euler_gpu::g_sharded_walk_split.load();
  for (int32_t c = 0; c < K; ++c) {
    Cohort& q = co[(size_t)c];
    if (q.n == 0) continue;
    int rc;
    void* tail_scratch = nullptr;
    if (T > 0 && L >= T + 8 && q.stride != 0u && q.n * (int64_t)(L - T + 1) < ((int64_t)1 << 40))
      tail_scratch = sc.Get((size_t)q.stride * W * (size_t)(L - T + 1) * 8 + (size_t)q.n * 4 + 256);
    if (tail_scratch != nullptr) {
      rc = euler_gpu::WalkPathsFromLevelsTail(st, starts_dev + q.lo, q.n, L, q.ids.data(), q.next.data(), default_node,
                                              out_dev + q.lo * ((int64_t)L + 1), T, q.lens + (size_t)T * W, q.stride,
                                              W, tail_scratch);
    } else {
      (void)hipGetLastError();
      rc = euler_gpu::WalkPathsFromLevels(st, starts_dev + q.lo, q.n, L, q.ids.data(), q.next.data(),
                                          default_node, out_dev + q.lo * ((int64_t)L + 1));
    }
    if (rc != EULER_GPU_OK) return rc;
  }
  if (stats_host) {
    // (asked for: one wait at the END of the call for the sizes the device kept to itself)
    int64_t entries = 0, wire_ids = 0;
    std::vector<uint32_t> lens((size_t)(L + 1) * W);
    for (int32_t c = 0; c < K; ++c) {
      const Cohort& q = co[(size_t)c];
      if (q.stride == 0u) continue;
      EG_HIP(hipMemcpyAsync(lens.data(), q.lens, lens.size() * 4, hipMemcpyDeviceToHost, st));
      EG_HIP(hipStreamSynchronize(st));
      entries += q.n;
      for (int32_t s = 1; s <= L; ++s)
        for (int32_t p = 0; p < W; ++p) {
          const int64_t v = (int64_t)lens[(size_t)s * W + p];
          if (s < L) entries += v;
          if (p != tr->rank) wire_ids += v;
        }
    }
    stats_host[0] = waits; stats_host[1] = entries; stats_host[2] = wire_ids; stats_host[3] = K;
  }
  return EULER_GPU_OK;
}

}  // namespace

extern "C" {

int euler_gpu_transport_rccl(void* nccl_comm, int32_t rank, int32_t world, euler_shm* counts,
                             euler_gpu_transport* out) {
  if (!out || !nccl_comm || world < 1 || rank < 0 || rank >= world)
    return Fail(EULER_GPU_EINVAL, "transport_rccl: bad arguments");
  if (!GetRccl().ok)
    return Fail(EULER_GPU_EHIP, "transport_rccl: no RCCL (ncclSend / ncclRecv) in this process");
  RcclUser* u = new RcclUser();
  u->comm = nccl_comm; u->rank = rank; u->world = world; u->shm = counts;
  u->pinned = nullptr; u->dev_counts = nullptr;
  if (counts == nullptr && world > 1) {
    hipError_t e = hipHostMalloc((void**)&u->pinned, sizeof(int64_t) * 2 * world);
    if (e == hipSuccess) e = hipMalloc((void**)&u->dev_counts, sizeof(int64_t) * 2 * world);
    if (e != hipSuccess) {
      if (u->pinned) (void)hipHostFree(u->pinned);
      delete u;
      return Fail(EULER_GPU_ENOMEM, std::string("transport_rccl: ") + hipGetErrorString(e));
    }
  }
  out->rank = rank; out->world = world; out->user = u;
  out->alltoall_counts = RcclCounts;
  out->alltoallv = RcclAllToAllV;
  return EULER_GPU_OK;
}

void euler_gpu_transport_rccl_release(euler_gpu_transport* t) {
  if (t == nullptr || t->user == nullptr) return;
  RcclUser* u = (RcclUser*)t->user;
  if (u->pinned) (void)hipHostFree(u->pinned);
  if (u->dev_counts) (void)hipFree(u->dev_counts);
  delete u;
  t->user = nullptr;
}

int euler_gpu_sharded_sample_neighbor(const euler_gpu_graph* shard, const euler_gpu_transport* tr,
                                      void* stream, uint64_t seed, uint32_t call_id,
                                      const uint64_t* roots_dev, int64_t n,
                                      const uint8_t* root_mask_dev, int32_t root_group,
                                      const int32_t* edge_types_host, int32_t k, int32_t count,
                                      int64_t default_node, int32_t partitions,
                                      uint64_t* out_id_dev, float* out_w_dev, int32_t* out_t_dev,
                                      uint8_t* out_mask_dev) {
  if (!shard) return Fail(EULER_GPU_ENOGRAPH, "sharded_sample_neighbor: null graph");
  if (!tr || !tr->alltoall_counts || !tr->alltoallv || tr->world < 1 || n < 0 || count <= 0 ||
      partitions < tr->world)
    return Fail(EULER_GPU_EINVAL, "sharded_sample_neighbor: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const int32_t W = tr->world;
  Scratch sc(st);
  // 1. the DISTINCT ids of the batch bucketed by owner + every position's place
  //    among the answers (ID_UNIQUE precedes ID_SPLIT: parser/compiler.cc:76-90)
  uint64_t* shard_ids = (uint64_t*)sc.Get((size_t)(n > 0 ? n : 1) * 8);
  int32_t* pos = (int32_t*)sc.Get((size_t)(n > 0 ? n : 1) * 4);
  if (!shard_ids || !pos) return Fail(EULER_GPU_ENOMEM, "sharded_sample_neighbor: scratch");
  std::vector<int64_t> off((size_t)W + 1, 0);
  if (n > 0) {
    const int rc = euler_gpu_dedup_split(stream, roots_dev, n, root_mask_dev, root_group, partitions,
                                         W, nullptr, 0, off.data(), shard_ids, pos);
    if (rc != EULER_GPU_OK) return rc;
  }
  std::vector<int64_t> send_rows((size_t)W), recv_rows((size_t)W);
  for (int32_t s = 0; s < W; ++s) send_rows[s] = off[s + 1] - off[s];
  // 2. who gets how many ids from whom.  (One rank: every id is its own - no exchange, the
  //    buckets ARE the owned ids and the sampled rows ARE the answers.)
  int rc = EULER_GPU_OK;
  const bool lone = W == 1 && !SelfExchange();
  if (lone) recv_rows[0] = send_rows[0];
  else rc = tr->alltoall_counts(tr->user, send_rows.data(), recv_rows.data());
  if (rc != EULER_GPU_OK) return rc;
  int64_t m = 0;
  for (int32_t s = 0; s < W; ++s) m += recv_rows[s];
  // 3. ids to their owners
  uint64_t* owned = shard_ids;
  if (!lone) {
    owned = (uint64_t*)sc.Get((size_t)(m > 0 ? m : 1) * 8);
    if (!owned) return Fail(EULER_GPU_ENOMEM, "sharded_sample_neighbor: scratch");
    rc = tr->alltoallv(tr->user, shard_ids, send_rows.data(), owned, recv_rows.data(), 8, stream);
    if (rc != EULER_GPU_OK) return rc;
  }
  // 4. the owner samples its rows straight into wire rows
  const int32_t single_type = k == 1 ? edge_types_host[0] : -1;
  const int32_t words = PackedWordsHost(count, single_type >= 0 ? 0 : 1);
  int32_t* rows = (int32_t*)sc.Get((size_t)(m > 0 ? m : 1) * words * 4);
  int64_t asked = 0;
  for (int32_t s = 0; s < W; ++s) asked += send_rows[s];
  int32_t* back = lone ? rows : (int32_t*)sc.Get((size_t)(asked > 0 ? asked : 1) * words * 4);
  if (!rows || !back) return Fail(EULER_GPU_ENOMEM, "sharded_sample_neighbor: scratch");
  if (m > 0) {
    rc = euler_gpu_sample_neighbor_packed(shard, stream, seed, call_id, owned, m, edge_types_host, k,
                                          count, default_node, rows);
    if (rc != EULER_GPU_OK) return rc;
  }
  // 5. rows back along the reversed split; the shards answered in the order they
  //    were asked, so row pos[i] of `back` is position i's row
  if (!lone) {
    rc = tr->alltoallv(tr->user, rows, recv_rows.data(), back, send_rows.data(), (int64_t)words * 4,
                       stream);
    if (rc != EULER_GPU_OK) return rc;
  }
  // 6. IDX_MERGE / DATA_MERGE / DATA_GATHER / unpack in one pass
  if (n > 0) {
    rc = euler_gpu_expand_packed(stream, pos, n, count, single_type, back, out_id_dev, out_w_dev,
                                 out_t_dev, out_mask_dev);
    if (rc != EULER_GPU_OK) return rc;
  }
  return EULER_GPU_OK;
}

int euler_gpu_sharded_sample_fanout(const euler_gpu_graph* shard, const euler_gpu_transport* tr,
                                    void* stream, uint64_t seed, uint32_t call_id,
                                    const uint64_t* roots_dev, int64_t n,
                                    const int32_t* edge_types_host, int32_t k,
                                    const int32_t* counts_host, int32_t layers,
                                    int64_t default_node, int32_t partitions,
                                    uint64_t* const* out_id_dev, float* const* out_w_dev,
                                    int32_t* const* out_t_dev, void* workspace_dev) {
  if (layers < 0 || (layers > 0 && (!counts_host || !out_id_dev || !out_w_dev || !out_t_dev)))
    return Fail(EULER_GPU_EINVAL, "sharded_sample_fanout: bad arguments");
  if (layers > 0 && n > 0 && !workspace_dev)
    return Fail(EULER_GPU_EINVAL, "sharded_sample_fanout: workspace required");
  const uint64_t* roots = roots_dev;
  const uint8_t* mask = nullptr;
  int32_t group = 1;
  int64_t m = n;
  uint8_t* ws = (uint8_t*)workspace_dev;
  for (int32_t h = 0; h < layers; ++h) {
    uint8_t* row_mask = ws;
    ws += ((size_t)m + 15) & ~(size_t)15;
    // every rank makes every hop's exchanges, also with an empty batch
    const int rc = euler_gpu_sharded_sample_neighbor(
        shard, tr, stream, seed, call_id + (uint32_t)h, roots, m, mask, group,
        edge_types_host + (size_t)h * k, k, counts_host[h], default_node, partitions,
        out_id_dev[h], out_w_dev[h], out_t_dev[h], row_mask);
    if (rc != EULER_GPU_OK) return rc;
    roots = out_id_dev[h];
    mask = row_mask;
    group = counts_host[h];
    m *= counts_host[h];
  }
  return EULER_GPU_OK;
}

// DeepWalk (p = q = 1) over the sharded graph, tf_euler/kernels/random_walk_op.cc:207-247: one
// `sampleNB(edge_types, 1)` query per step, each ID_UNIQUE -> ID_SPLIT -> REMOTE -> MERGE ->
// GATHER (parser/compiler.cc:76-90).  Here the GATHER is deferred to the end of the walk:
//   level s   = the DISTINCT nodes the rank's walkers stand on at step s (level 0 = the
//               walkers), an entry = a group of walkers that have merged - two walkers on one
//               node draw the same next node (the draw is keyed by (seed, call_id + s, node)),
//               so they stay together for good; 1M walkers are 63 % distinct nodes after one
//               step, 11 % after ten (walk_kernels.hip);
//   step s    = front end over level s (distinct ids bucketed by owner; next[s][e] = the place
//               of entry e's answer) -> ids to the owners -> the OWNERS draw (WalkOwnedStep: the
//               draw of the single-GPU walk) -> the answers, in the order asked, ARE level s + 1;
//   paths     = every walker follows next[0], next[1], ... once (WalkPathsFromLevels).
// A step costs what its level holds, not what the batch holds, and the wire carries every
// distinct node once per rank and step.  The host waits once per step (the bucket sizes, which
// size the exchange); `cohorts` > 1 splits the walkers into that many independent walks whose
// steps alternate on the stream, so that while the host waits for one cohort's sizes the GPU
// runs the other cohorts' kernels.  Bit-identical to euler_gpu_random_walk on the unsharded graph.
int euler_gpu_sharded_random_walk(const euler_gpu_graph* shard, const euler_gpu_transport* tr,
                                  void* stream, uint64_t seed, uint32_t call_id,
                                  const int64_t* starts_dev, int64_t n,
                                  const int32_t* edge_types_host, int32_t k, int32_t walk_len,
                                  int64_t default_node, int32_t partitions, int32_t cohorts,
                                  uint32_t* dense_owner_dev, int64_t dense_limit,
                                  int64_t* out_dev, int64_t* stats_host) {
  if (!shard) return Fail(EULER_GPU_ENOGRAPH, "sharded_random_walk: null graph");
  if (!tr || tr->world < 1 || (tr->world > 1 && (!tr->alltoall_counts || !tr->alltoallv)) || n < 0 ||
      walk_len < 0 || k < 0 || k > 32 || partitions < tr->world || cohorts < 1 || cohorts > 16 ||
      n >= ((int64_t)1 << 30) || (n > 0 && (!starts_dev || !out_dev)) ||
      (k > 0 && walk_len > 0 && !edge_types_host))
    return Fail(EULER_GPU_EINVAL, "sharded_random_walk: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const int32_t W = tr->world, L = walk_len, K = cohorts;
  const bool lone = W == 1 && (!SelfExchange() || !tr->alltoall_counts || !tr->alltoallv);
  Scratch sc(st);
  int32_t* et_dev = nullptr;
  {
    const int rc = euler_gpu::WalkEdgeTypes(st, edge_types_host, k, L, &et_dev);
    if (rc != EULER_GPU_OK) return rc;
    sc.ptrs.push_back(et_dev);
  }
  if (euler_gpu::g_sharded_walk_enqueued.load() != 0)
    return WalkEnqueued(shard, tr, st, seed, call_id, starts_dev, n, et_dev, k, L, default_node, partitions,
                        K, dense_owner_dev, dense_limit, out_dev, stats_host, lone, sc);
  struct Cohort {
    int64_t lo = 0, n = 0, m = 0;                // walkers; entries of the current level
    euler_gpu_front* front = nullptr;
    std::vector<const uint64_t*> ids;            // [L + 1] levels (0 = the walkers' start nodes)
    std::vector<const int32_t*> next;            // [L]
    uint64_t* bucketed = nullptr;                // [n] the current step's distinct ids by owner
    uint8_t* arena = nullptr;
    ~Cohort() { if (front) euler_gpu_front_destroy(front); }
  };
  std::vector<Cohort> co((size_t)K);
  for (int32_t c = 0; c < K; ++c) {
    Cohort& q = co[(size_t)c];
    q.lo = n * c / K; q.n = n * (c + 1) / K - q.lo; q.m = q.n;
    q.ids.assign((size_t)L + 1, nullptr); q.next.assign((size_t)(L > 0 ? L : 1), nullptr);
    q.ids[0] = (const uint64_t*)starts_dev + q.lo;
    int rc = euler_gpu_front_create(&q.front);
    if (rc != EULER_GPU_OK) return rc;
    if (q.n > 0 && L > 0) {
      // levels 1 .. L (8 bytes an entry), next 0 .. L - 1 (4), the step's bucketed ids (8): a
      // level never holds more entries than the cohort has walkers
      q.arena = (uint8_t*)sc.Get((size_t)q.n * ((size_t)L * 12 + 8) + 256);
      if (!q.arena) { (void)hipGetLastError(); return Fail(EULER_GPU_ENOMEM, "sharded_random_walk: scratch"); }
      uint8_t* p8 = q.arena;
      for (int32_t s = 1; s <= L; ++s) { q.ids[(size_t)s] = (const uint64_t*)p8; p8 += (size_t)q.n * 8; }
      q.bucketed = (uint64_t*)p8; p8 += (size_t)q.n * 8;
      for (int32_t s = 0; s < L; ++s) { q.next[(size_t)s] = (const int32_t*)p8; p8 += (size_t)q.n * 4; }
    }
  }
  int64_t waits = 0, entries = 0, wire_ids = 0;
  auto begin = [&](Cohort& q, int32_t s) -> int {
    // (an empty level still takes part: its rank must make the step's exchanges)
    return euler_gpu_dedup_split_begin(q.front, stream, q.ids[(size_t)s], q.m, nullptr, 1, partitions, W,
                                       dense_owner_dev, dense_limit, q.bucketed,
                                       const_cast<int32_t*>(q.next[(size_t)s]));
  };
  std::vector<int64_t> off((size_t)W + 1), send_rows((size_t)W), recv_rows((size_t)W);
  auto finish = [&](Cohort& q, int32_t s) -> int {
    int rc = euler_gpu_dedup_split_end(q.front, off.data());        // the step's one host wait
    if (rc != EULER_GPU_OK) return rc;
    ++waits;
    const int64_t asked = off[(size_t)W];
    entries += q.m;
    uint64_t* level = const_cast<uint64_t*>(q.ids[(size_t)s + 1]);
    if (lone) {
      // one rank: the buckets are the owned ids, the draws are the next level
      rc = euler_gpu::WalkOwnedStep(shard, st, seed, call_id, et_dev, k, L, s, q.bucketed, asked, level);
      if (rc != EULER_GPU_OK) return rc;
    } else {
      for (int32_t p = 0; p < W; ++p) send_rows[(size_t)p] = off[(size_t)p + 1] - off[(size_t)p];
      rc = tr->alltoall_counts(tr->user, send_rows.data(), recv_rows.data());
      if (rc != EULER_GPU_OK) return rc;
      int64_t m_in = 0;
      for (int32_t p = 0; p < W; ++p) m_in += recv_rows[(size_t)p];
      wire_ids += asked - send_rows[(size_t)tr->rank];
      uint64_t* owned = (uint64_t*)sc.Get((size_t)(m_in > 0 ? m_in : 1) * 16);
      if (!owned) { (void)hipGetLastError(); return Fail(EULER_GPU_ENOMEM, "sharded_random_walk: scratch"); }
      uint64_t* drawn = owned + (m_in > 0 ? m_in : 1);
      rc = tr->alltoallv(tr->user, q.bucketed, send_rows.data(), owned, recv_rows.data(), 8, stream);
      if (rc != EULER_GPU_OK) return rc;
      rc = euler_gpu::WalkOwnedStep(shard, st, seed, call_id, et_dev, k, L, s, owned, m_in, drawn);
      if (rc != EULER_GPU_OK) return rc;
      // (an empty cohort has no arena and asks for nothing: any valid pointer receives its 0 rows)
      rc = tr->alltoallv(tr->user, drawn, recv_rows.data(), level ? (void*)level : (void*)owned,
                         send_rows.data(), 8, stream);
      if (rc != EULER_GPU_OK) return rc;
      // (stream-ordered: the block is back in the pool before the next step asks for its own -
      // the call's footprint is the largest step's, not the sum over the walk's steps)
      sc.Release(owned);
    }
    q.m = asked;
    return EULER_GPU_OK;
  };
  if (L > 0) {
    for (int32_t c = 0; c < K; ++c) {
      const int rc = begin(co[(size_t)c], 0);
      if (rc != EULER_GPU_OK) return rc;
    }
    for (int32_t s = 0; s < L; ++s) {
      for (int32_t c = 0; c < K; ++c) {
        int rc = finish(co[(size_t)c], s);
        if (rc == EULER_GPU_OK && s + 1 < L) rc = begin(co[(size_t)c], s + 1);
        if (rc != EULER_GPU_OK) return rc;
      }
    }
  }
  // the walkers' paths: one chain walk per walker through the levels
  for (int32_t c = 0; c < K; ++c) {
    Cohort& q = co[(size_t)c];
    if (q.n == 0) continue;
    const int rc = euler_gpu::WalkPathsFromLevels(st, starts_dev + q.lo, q.n, L, q.ids.data(), q.next.data(),
                                                  default_node, out_dev + q.lo * ((int64_t)L + 1));
    if (rc != EULER_GPU_OK) return rc;
  }
  if (stats_host) { stats_host[0] = waits; stats_host[1] = entries; stats_host[2] = wire_ids; stats_host[3] = K; }
  return EULER_GPU_OK;
}

// node2vec (p or q != 1) over the sharded graph, tf_euler/kernels/random_walk_op.cc:83-168: the
// reference's CLIENT runs the walk - per step one `v(nodes).outV(edge_types)` query for the
// walkers' current nodes (ID_UNIQUE -> ID_SPLIT -> REMOTE -> MERGE: one row per distinct node),
// the previous step's rows kept as the parents', BuildWeights + the draw on the client.  Here,
// per step and rank: front end over the walkers' nodes (distinct ids bucketed by owner + the
// row of every walker) -> ids to the owners -> the owners' full rows (euler_gpu_get_full_neighbor)
// -> row lengths, ids and weights back (three all-to-all(v)s, sizes from the lengths) -> the
// requester's draw on the fetched rows (euler_gpu_node2vec_step, keyed by the walker's index)
// -> column s + 1 of the result.  Bit-identical to euler_gpu_random_walk on the unsharded graph.
int euler_gpu_sharded_node2vec_walk(const euler_gpu_graph* shard, const euler_gpu_transport* tr,
                                    void* stream, uint64_t seed, uint32_t call_id,
                                    const int64_t* starts_dev, int64_t n,
                                    const int32_t* edge_types_host, int32_t k, int32_t walk_len,
                                    float p, float q, int64_t default_node, int32_t partitions,
                                    uint32_t* dense_owner_dev, int64_t dense_limit,
                                    int64_t* out_dev, int64_t* stats_host) {
  if (!shard) return Fail(EULER_GPU_ENOGRAPH, "sharded_node2vec_walk: null graph");
  if (!tr || tr->world < 1 || (tr->world > 1 && (!tr->alltoall_counts || !tr->alltoallv)) || n < 0 ||
      walk_len < 0 || k < 0 || k > 32 || partitions < tr->world || n >= ((int64_t)1 << 30) ||
      (n > 0 && (!starts_dev || !out_dev)) || (k > 0 && walk_len > 0 && !edge_types_host))
    return Fail(EULER_GPU_EINVAL, "sharded_node2vec_walk: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const int32_t W = tr->world, L = walk_len;
  const bool lone = W == 1 && (!SelfExchange() || !tr->alltoall_counts || !tr->alltoallv);
  Scratch sc(st);
  const int64_t n1 = n > 0 ? n : 1;
  int rc = euler_gpu::N2vStoreColumn(st, starts_dev, n, (int64_t)L + 1, 0, out_dev);
  if (rc != EULER_GPU_OK) return rc;
  // per walker: current and previous node, next node; the step's front end
  int64_t* cur = (int64_t*)sc.Get((size_t)n1 * 8);
  int64_t* parent = (int64_t*)sc.Get((size_t)n1 * 8);
  int64_t* nxt = (int64_t*)sc.Get((size_t)n1 * 8);
  uint64_t* bucketed = (uint64_t*)sc.Get((size_t)n1 * 8);
  int32_t* c_row = (int32_t*)sc.Get((size_t)n1 * 4);
  int32_t* p_row = (int32_t*)sc.Get((size_t)n1 * 4);
  int64_t* bounds_dev = (int64_t*)sc.Get((size_t)(W + 1) * 16);
  if (!cur || !parent || !nxt || !bucketed || !c_row || !p_row || !bounds_dev) {
    (void)hipGetLastError();
    return Fail(EULER_GPU_ENOMEM, "sharded_node2vec_walk: scratch");
  }
  int64_t* bound_off_dev = bounds_dev + (W + 1);
  if (n > 0) {
    EG_HIP(hipMemcpyAsync(cur, starts_dev, (size_t)n * 8, hipMemcpyDeviceToDevice, st));
    EG_HIP(hipMemcpyAsync(parent, starts_dev, (size_t)n * 8, hipMemcpyDeviceToDevice, st));   // parent_ids_ starts as the start nodes
  }
  // the previous step's rows (the parents' neighbour lists)
  int32_t* p_idx = nullptr; uint64_t* p_ids = nullptr; float* p_w = nullptr;
  bool have_parent = false;
  std::vector<int64_t> off((size_t)W + 1), send_rows((size_t)W), recv_rows((size_t)W), val_send((size_t)W),
      val_recv((size_t)W), bounds((size_t)W + 1), bound_off((size_t)W + 1);
  int64_t waits = 0, rows_total = 0, entries_total = 0, wire_ids = 0;
  for (int32_t s = 0; s < L; ++s) {
    const int32_t* et = edge_types_host + (size_t)s * k;
    // 1. distinct nodes bucketed by owner; c_row[i] = the row of walker i among the answers
    for (auto& x : off) x = 0;
    if (n > 0) {              // (an empty batch still makes the step's exchanges)
      rc = euler_gpu_dedup_split(stream, (const uint64_t*)cur, n, nullptr, 1, partitions, W, dense_owner_dev,
                                 dense_limit, off.data(), bucketed, c_row);
      if (rc != EULER_GPU_OK) return rc;
      ++waits;
    }
    const int64_t asked = off[(size_t)W];
    for (int32_t r = 0; r < W; ++r) send_rows[(size_t)r] = off[(size_t)r + 1] - off[(size_t)r];
    rows_total += asked;
    // 2. ids to their owners
    int64_t m_in = asked;
    uint64_t* owned = bucketed;
    if (lone) recv_rows[0] = send_rows[0];
    else {
      rc = tr->alltoall_counts(tr->user, send_rows.data(), recv_rows.data());
      if (rc != EULER_GPU_OK) return rc;
      m_in = 0;
      for (int32_t r = 0; r < W; ++r) m_in += recv_rows[(size_t)r];
      wire_ids += asked - send_rows[(size_t)tr->rank];
      owned = (uint64_t*)sc.Get((size_t)(m_in > 0 ? m_in : 1) * 8);
      if (!owned) { (void)hipGetLastError(); return Fail(EULER_GPU_ENOMEM, "sharded_node2vec_walk: scratch"); }
      rc = tr->alltoallv(tr->user, bucketed, send_rows.data(), owned, recv_rows.data(), 8, stream);
      if (rc != EULER_GPU_OK) return rc;
    }
    // 3. the owners' rows (offsets first - the one host wait of the owners' side - then the values)
    const int64_t m1 = m_in > 0 ? m_in : 1;
    int32_t* o_idx = (int32_t*)sc.Get((size_t)m1 * 8);
    if (!o_idx) { (void)hipGetLastError(); return Fail(EULER_GPU_ENOMEM, "sharded_node2vec_walk: scratch"); }
    int64_t o_total = 0;
    if (m_in > 0) {
      rc = euler_gpu_get_full_neighbor(shard, stream, owned, m_in, et, k, o_idx, &o_total, nullptr, nullptr, nullptr);
      if (rc != EULER_GPU_OK) return rc;
      ++waits;
    }
    if (o_total >= ((int64_t)1 << 31))
      return Fail(EULER_GPU_EINVAL, "sharded_node2vec_walk: a step's rows hold more than 2^31 neighbours");
    const int64_t t1 = o_total > 0 ? o_total : 1;
    uint64_t* o_ids = (uint64_t*)sc.Get((size_t)t1 * 8);
    float* o_w = (float*)sc.Get((size_t)t1 * 4);
    if (!o_ids || !o_w) { (void)hipGetLastError(); return Fail(EULER_GPU_ENOMEM, "sharded_node2vec_walk: scratch"); }
    if (m_in > 0 && o_total > 0) {
      // (BuildWeights does not read the edge types: not written - 0.56 GB a step on the metric graph)
      rc = euler_gpu_get_full_neighbor(shard, stream, owned, m_in, et, k, o_idx, &o_total, o_ids, o_w, nullptr);
      if (rc != EULER_GPU_OK) return rc;
    }
    // 4. the rows back to whoever asked: lengths, then ids and weights sized from them
    int32_t* c_idx = o_idx; uint64_t* c_ids = o_ids; float* c_w = o_w;
    int64_t c_entries = o_total;
    if (!lone) {
      int32_t* o_len = (int32_t*)sc.Get((size_t)m1 * 4);
      int32_t* b_len = (int32_t*)sc.Get((size_t)(asked > 0 ? asked : 1) * 8);     // lengths, then their running ends
      if (!o_len || !b_len) { (void)hipGetLastError(); return Fail(EULER_GPU_ENOMEM, "sharded_node2vec_walk: scratch"); }
      rc = euler_gpu::N2vRowLens(st, o_idx, m_in, o_len);
      if (rc != EULER_GPU_OK) return rc;
      rc = tr->alltoallv(tr->user, o_len, recv_rows.data(), b_len, send_rows.data(), 4, stream);
      if (rc != EULER_GPU_OK) return rc;
      // values per requester = the offsets at the bounds of its rows
      bounds[0] = 0;
      for (int32_t r = 0; r < W; ++r) bounds[(size_t)r + 1] = bounds[(size_t)r] + recv_rows[(size_t)r];
      EG_HIP(hipMemcpyAsync(bounds_dev, bounds.data(), (size_t)(W + 1) * 8, hipMemcpyHostToDevice, st));
      rc = euler_gpu::N2vBoundOffsets(st, o_idx, bounds_dev, W, bound_off_dev);
      if (rc != EULER_GPU_OK) return rc;
      EG_HIP(hipMemcpyAsync(bound_off.data(), bound_off_dev, (size_t)(W + 1) * 8, hipMemcpyDeviceToHost, st));
      EG_HIP(hipStreamSynchronize(st));
      ++waits;
      for (int32_t r = 0; r < W; ++r) val_send[(size_t)r] = bound_off[(size_t)r + 1] - bound_off[(size_t)r];
      rc = tr->alltoall_counts(tr->user, val_send.data(), val_recv.data());
      if (rc != EULER_GPU_OK) return rc;
      c_entries = 0;
      for (int32_t r = 0; r < W; ++r) c_entries += val_recv[(size_t)r];
      if (c_entries >= ((int64_t)1 << 31))
        return Fail(EULER_GPU_EINVAL, "sharded_node2vec_walk: a step's rows hold more than 2^31 neighbours");
      const int64_t e1 = c_entries > 0 ? c_entries : 1;
      c_ids = (uint64_t*)sc.Get((size_t)e1 * 8);
      c_w = (float*)sc.Get((size_t)e1 * 4);
      c_idx = (int32_t*)sc.Get((size_t)(asked > 0 ? asked : 1) * 8);
      if (!c_ids || !c_w || !c_idx) { (void)hipGetLastError(); return Fail(EULER_GPU_ENOMEM, "sharded_node2vec_walk: scratch"); }
      rc = tr->alltoallv(tr->user, o_ids, val_send.data(), c_ids, val_recv.data(), 8, stream);
      if (rc != EULER_GPU_OK) return rc;
      rc = tr->alltoallv(tr->user, o_w, val_send.data(), c_w, val_recv.data(), 4, stream);
      if (rc != EULER_GPU_OK) return rc;
      rc = euler_gpu::N2vIdxFromLens(st, b_len, asked, b_len + (asked > 0 ? asked : 1), c_idx);
      if (rc != EULER_GPU_OK) return rc;
      sc.Release(o_len); sc.Release(b_len); sc.Release(o_idx); sc.Release(o_ids); sc.Release(o_w);
      sc.Release(owned);
    }
    entries_total += c_entries;
    // 5. the draw, on the requester (RWCallback::operator(), random_walk_op.cc:83-168)
    rc = euler_gpu_node2vec_step(stream, seed, call_id + (uint32_t)s, n, c_row, c_idx, c_ids, c_w, c_entries,
                                 have_parent ? p_row : nullptr, p_idx, p_ids, parent, p, q, default_node, nxt);
    if (rc != EULER_GPU_OK) return rc;
    rc = euler_gpu::N2vStoreColumn(st, nxt, n, (int64_t)L + 1, (int64_t)s + 1, out_dev);
    if (rc != EULER_GPU_OK) return rc;
    // 6. this step's rows are the next step's parents' rows
    sc.Release(p_idx); sc.Release(p_ids); sc.Release(p_w);
    p_idx = c_idx; p_ids = c_ids; p_w = c_w;
    have_parent = true;
    { int32_t* t = p_row; p_row = c_row; c_row = t; }
    { int64_t* t = parent; parent = cur; cur = nxt; nxt = t; }
  }
  if (stats_host) { stats_host[0] = waits; stats_host[1] = rows_total; stats_host[2] = entries_total; stats_host[3] = wire_ids; }
  return EULER_GPU_OK;
}

}  // extern "C"
