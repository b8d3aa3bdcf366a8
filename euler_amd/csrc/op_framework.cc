// Implementation of the plugin-API mirror and of the hot-path op kernels
// registered under the reference's op names.  See include/euler_op_framework.h.
#include "euler_op_framework.h"

#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <map>
#include <memory>
#include <random>

namespace euler {
inline namespace gpu_abi {

namespace {

void LogError(const std::string& msg) {
  // EULER_LOG(ERROR) semantics: log, leave outputs unallocated, return.
  fprintf(stderr, "[euler_gpu] ERROR %s\n", msg.c_str());
}

struct Registry {
  std::mutex mu;
  std::unordered_map<std::string, OpKernelRegistrar::Factory> factories;
  std::unordered_map<std::string, std::unique_ptr<OpKernel>> kernels;
};

Registry* GlobalRegistry() {
  static Registry* r = new Registry();
  return r;
}

// ---- process-wide sampling state (OpKernelContext::seed / NextCallId)
std::atomic<uint32_t> g_call_seq{0};
std::atomic<uint64_t> g_process_seed{0};
std::once_flag g_seed_once;
uint64_t ProcessSeed() {
  std::call_once(g_seed_once, [] {
    if (g_process_seed.load() == 0) {
      std::random_device rd;
      g_process_seed.store(((uint64_t)rd() << 32) ^ (uint64_t)rd() ^ 0x9E3779B97F4A7C15ULL);
    }
  });
  return g_process_seed.load();
}

// ---- device staging of one op invocation.  Every host thread (the reference's
// client pool runs 8 of them, client/query_proxy.cc:205-210) owns, per device, a
// non-blocking HIP stream and a grow-only arena: an op bump-allocates its
// staging buffers from the arena, enqueues copies and kernels on the stream and
// synchronises once (twice when a size has to come back first).  No hipMalloc /
// hipFree in steady state, nothing on the null stream, every HIP call checked:
// a failure marks the scope and the op logs and returns WITHOUT allocating its
// outputs.
struct Arena {
  hipStream_t stream = nullptr;
  struct Chunk { void* p; size_t cap; size_t used; };
  std::vector<Chunk> chunks;
};

Arena* ThreadArena(int device) {
  // leaked on purpose: freeing device memory from thread_local destructors races
  // with the HIP runtime's own teardown at process exit
  thread_local std::map<int, Arena*>* arenas = new std::map<int, Arena*>();
  Arena*& a = (*arenas)[device];
  if (a == nullptr) a = new Arena();
  return a;
}

class OpScope {
 public:
  explicit OpScope(euler_gpu_graph* g) {
    int dev = 0;
    if (g != nullptr) dev = euler_gpu_graph_device(g);
    else (void)hipGetDevice(&dev);
    Check(hipSetDevice(dev), "hipSetDevice");
    arena_ = ThreadArena(dev);
    if (ok_ && arena_->stream == nullptr)
      Check(hipStreamCreateWithFlags(&arena_->stream, hipStreamNonBlocking), "hipStreamCreate");
  }
  ~OpScope() {
    if (arena_ == nullptr) return;
    // an op that leaves early (a failed call, a bad argument) may have copies and kernels
    // queued that read or write its locals, its just-freed outputs or this arena: wait
    // for them before any of that memory is reused
    if (arena_->stream != nullptr && dirty_) (void)hipStreamSynchronize(arena_->stream);
    if (arena_->chunks.size() > 1) {
      // the op outgrew the arena: one chunk of the total size for the next call
      size_t total = 0;
      (void)hipStreamSynchronize(arena_->stream);
      for (auto& c : arena_->chunks) { total += c.cap; (void)hipFree(c.p); }
      arena_->chunks.clear();
      void* p = nullptr;
      if (hipMalloc(&p, total) == hipSuccess) arena_->chunks.push_back({p, total, 0});
    }
    for (auto& c : arena_->chunks) c.used = 0;
  }
  bool ok() const { return ok_; }
  const std::string& error() const { return err_; }
  void* stream() const { return (void*)arena_->stream; }
  void* AllocBytes(size_t bytes) {
    if (!ok_) return nullptr;
    bytes = (bytes + 255) & ~(size_t)255;
    if (bytes == 0) bytes = 256;
    if (!arena_->chunks.empty()) {
      auto& c = arena_->chunks.back();
      if (c.cap - c.used >= bytes) { void* p = (uint8_t*)c.p + c.used; c.used += bytes; return p; }
    }
    size_t have = 0;
    for (auto& c : arena_->chunks) have += c.cap;
    size_t want = bytes > 2 * have ? bytes : 2 * have;
    if (want < ((size_t)1 << 20)) want = (size_t)1 << 20;
    void* p = nullptr;
    if (!Check(hipMalloc(&p, want), "hipMalloc")) return nullptr;
    arena_->chunks.push_back({p, want, bytes});
    return p;
  }
  template <typename T> T* Alloc(size_t n) { return reinterpret_cast<T*>(AllocBytes(n * sizeof(T))); }
  bool Upload(void* dst, const void* src, size_t bytes) {
    if (!ok_ || dst == nullptr) return ok_ = false;
    if (bytes == 0) return true;
    dirty_ = true;
    return Check(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, arena_->stream), "copy to device");
  }
  bool Download(void* dst, const void* src, size_t bytes) {
    if (!ok_ || src == nullptr) return ok_ = false;
    if (bytes == 0) return true;
    dirty_ = true;
    return Check(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, arena_->stream), "copy to host");
  }
  // whatever the scope's state: nothing of this op is in flight afterwards
  void Drain() {
    if (arena_ != nullptr && arena_->stream != nullptr) (void)hipStreamSynchronize(arena_->stream);
  }
  bool Sync() {
    if (!ok_) return false;
    const bool r = Check(hipStreamSynchronize(arena_->stream), "stream sync");
    if (r) dirty_ = false;
    return r;
  }
  void MarkDirty() { dirty_ = true; }     // the caller enqueues on stream() itself
  // the C ABI's return code
  bool Call(int rc) {
    dirty_ = true;              // the call may have enqueued kernels on the stream
    if (rc != 0) { ok_ = false; err_ = euler_gpu_last_error(); }
    return ok_;
  }

 private:
  bool Check(hipError_t e, const char* what) {
    if (e != hipSuccess) { ok_ = false; err_ = std::string(what) + ": " + hipGetErrorString(e); }
    return ok_;
  }
  Arena* arena_ = nullptr;
  bool ok_ = true;
  bool dirty_ = false;        // work was enqueued on the stream since the last Sync()
  std::string err_;
};

bool GetIntArg(const NodeDef& nd, int i, OpKernelContext* ctx,
               std::vector<int32_t>* out) {
  Tensor* t = nullptr;
  if ((int)nd.inputs.size() <= i || ctx->tensor(nd.inputs[i], &t) != 0) return false;
  out->assign(t->Raw<int32_t>(), t->Raw<int32_t>() + t->NumElements());
  return true;
}

// DAGNodeProto.post_process entry -> (order_by, desc, limit); false = skip it
// (get_neighbor_op.cc:117-168, sample_neighbor_op.cc:86-132: same grammar)
bool ParsePostProcess(const std::string& post, int32_t* order_by, int32_t* desc, int64_t* limit) {
  std::vector<std::string> vec;
  std::string cur;
  for (char ch : post) {
    if (ch == ' ') { if (!cur.empty()) vec.push_back(cur); cur.clear(); }
    else cur.push_back(ch);
  }
  if (!cur.empty()) vec.push_back(cur);
  *order_by = 0; *desc = 0; *limit = -1;
  if (vec.empty()) return false;
  if (vec[0] == "order_by") {
    if (vec.size() < 2 || vec.size() > 3) { LogError("Invalid post process: " + post); return false; }
    *desc = vec.size() == 3 && vec[2] == "desc" ? 1 : 0;
    if (vec[1] == "id") *order_by = 1;
    else if (vec[1] == "weight") *order_by = 2;
    else { LogError("Invalid order_by field: " + vec[1]); return false; }
    return true;
  }
  if (vec[0] == "limit") {
    if (vec.size() != 2) { LogError("Invalid post process: " + post); return false; }
    *limit = atoll(vec[1].c_str());
    return true;
  }
  return false;
}

}  // namespace

uint64_t OpKernelContext::seed() const { return seed_set_ ? seed_ : ProcessSeed(); }
uint32_t OpKernelContext::NextCallId() {
  return call_id_set_ ? call_id_++ : g_call_seq.fetch_add(1, std::memory_order_relaxed);
}
void OpKernelContext::SetProcessSeed(uint64_t seed) {
  (void)ProcessSeed();                 // consume the once-flag first
  g_process_seed.store(seed);
  g_call_seq.store(0);
}

size_t SizeOfType(DataType t) {
  switch (t) {
    case kInt8: case kUInt8: case kBool: case kString: return 1;
    case kInt16: case kUInt16: return 2;
    case kInt32: case kUInt32: case kFloat: return 4;
    default: return 8;
  }
}

// ---- host memory of the tensors.  The reference's Tensor is malloc'ed (op_kernel.cc:92-105) and
// its results travel by gRPC; here they travel by PCIe, and a copy from the device into PAGEABLE
// memory is staged by the runtime at a few GB/s (the metric step's 577 MB of results: 130-140 ms
// per query, tools/host_boundary_rate.py) where pinned memory takes the link's 55 GB/s.  Tensors of
// kPinnedMin bytes or more therefore come from a process-wide cache of pinned blocks (four size
// classes per octave; hipHostMalloc itself costs ~0.1 ms per MB, so freed blocks are kept - up to
// EULER_GPU_PINNED_POOL_MB, default 4096, per process); smaller ones, and every tensor of a process
// without a GPU or with the pool set to 0, stay malloc'ed.  Ownership is unchanged: the Tensor owns
// its memory and gives it back in its destructor.
namespace {
constexpr size_t kPinnedMin = (size_t)16 << 10;
struct PinnedPool {
  std::mutex mu;
  std::unordered_map<void*, size_t> live;               // pinned blocks handed out -> class bytes
  std::map<size_t, std::vector<void*>> free_blocks;      // class bytes -> cached blocks
  size_t cached = 0, cap = 0;
  bool off = false;
  PinnedPool() {
    const char* e = getenv("EULER_GPU_PINNED_POOL_MB");
    const long mb = e ? atol(e) : 4096;
    off = mb <= 0;
    cap = off ? 0 : (size_t)mb << 20;
  }
};
PinnedPool* Pool() {
  static PinnedPool* p = new PinnedPool();     // leaked: see ThreadArena
  return p;
}
void* HostAlloc(size_t bytes) {
  PinnedPool* P = Pool();
  bool off;
  { std::lock_guard<std::mutex> lk(P->mu); off = P->off; }
  if (bytes >= kPinnedMin && !off) {
    // size classes: four per octave (at most 25 % of a block unused)
    size_t q = kPinnedMin >> 2;
    while ((q << 3) < bytes) q <<= 1;
    const size_t cls = (bytes + q - 1) / q * q;
    {
      std::lock_guard<std::mutex> lk(P->mu);
      auto it = P->free_blocks.find(cls);
      if (it != P->free_blocks.end() && !it->second.empty()) {
        void* p = it->second.back();
        it->second.pop_back();
        P->cached -= cls;
        P->live[p] = cls;
        return p;
      }
    }
    void* p = nullptr;
    if (hipHostMalloc(&p, cls, hipHostMallocPortable) == hipSuccess && p != nullptr) {
      std::lock_guard<std::mutex> lk(P->mu);
      P->live[p] = cls;
      return p;
    }
    (void)hipGetLastError();                   // no device / no pinned memory left: pageable
    {
      std::lock_guard<std::mutex> lk(P->mu);
      if (P->live.empty() && P->cached == 0) P->off = true;   // never worked here: do not ask again
    }
  }
  return malloc(bytes);
}
void HostFree(void* p) {
  if (p == nullptr) return;
  PinnedPool* P = Pool();
  size_t cls = 0;
  bool keep = false;
  {
    std::lock_guard<std::mutex> lk(P->mu);
    auto it = P->live.find(p);
    if (it != P->live.end()) {
      cls = it->second;
      P->live.erase(it);
      keep = P->cached + cls <= P->cap;
      if (keep) { P->free_blocks[cls].push_back(p); P->cached += cls; }
    }
  }
  if (cls == 0) free(p);
  else if (!keep) (void)hipHostFree(p);
}
}  // namespace

Tensor::Tensor(const TensorShape& shape, DataType type)
    : shape_(shape), type_(type),
      data_(HostAlloc(shape.NumElements() * SizeOfType(type) + 16)) {}
Tensor::~Tensor() { HostFree(data_); }

std::string OutputName(const NodeDef& node_def, int i) {
  return node_def.name + ":" + std::to_string(i);
}
std::string OutputName(const std::string& name, int i) { return name + ":" + std::to_string(i); }

OpStatus OpKernelContext::RemoveAlias(const std::string& name) {
  std::lock_guard<std::mutex> lk(mu_);
  return tensor_map_.erase(name) ? 0 : -1;         // (the tensor stays: op_kernel.cc RemoveAlias)
}

OpKernelContext::~OpKernelContext() {
  // aliases share a tensor: free every distinct pointer once
  std::vector<Tensor*> seen;
  for (auto& kv : tensor_map_) {
    bool dup = false;
    for (Tensor* t : seen) dup = dup || t == kv.second;
    if (!dup) { seen.push_back(kv.second); delete kv.second; }
  }
}

OpStatus OpKernelContext::AddAlias(const std::string& name, Tensor* tensor) {
  std::lock_guard<std::mutex> lk(mu_);
  if (tensor_map_.count(name)) return -1;
  tensor_map_[name] = tensor;
  return 0;
}

OpStatus OpKernelContext::Allocate(const std::string& name, const TensorShape& shape,
                              DataType type, Tensor** tensor) {
  std::lock_guard<std::mutex> lk(mu_);
  if (tensor_map_.count(name)) return -1;        // already exists
  *tensor = new Tensor(shape, type);
  tensor_map_[name] = *tensor;
  return 0;
}

OpStatus OpKernelContext::tensor(const std::string& name, Tensor** tensor) {
  std::lock_guard<std::mutex> lk(mu_);
  auto it = tensor_map_.find(name);
  if (it == tensor_map_.end()) return -1;
  *tensor = it->second;
  return 0;
}

OpStatus OpKernelContext::Deallocate(const std::string& name) {
  std::lock_guard<std::mutex> lk(mu_);
  auto it = tensor_map_.find(name);
  if (it == tensor_map_.end()) return -1;
  Tensor* t = it->second;
  tensor_map_.erase(it);
  for (auto& kv : tensor_map_)
    if (kv.second == t) return 0;      // still reachable through an alias
  delete t;
  return 0;
}

euler_gpu_graph* OpKernelContext::graph() const {
  return graph_ ? graph_ : euler_gpu_default_graph();
}

// op_kernel.cc (AsyncOpKernel::Compute): the synchronous entry blocks until the kernel's
// done callback has run - on this thread for the GPU kernels of this library (they complete
// inside AsyncCompute), on whichever thread a user-registered asynchronous kernel calls it.
void AsyncOpKernel::Compute(const NodeDef& node_def, OpKernelContext* ctx) {
  std::mutex mu;
  std::condition_variable cv;
  bool done = false;
  AsyncCompute(node_def, ctx, [&] {
    std::lock_guard<std::mutex> lk(mu);
    done = true;
    cv.notify_all();
  });
  std::unique_lock<std::mutex> lk(mu);
  cv.wait(lk, [&] { return done; });
}

OpKernelRegistrar::OpKernelRegistrar(const std::string& name, Factory factory) {
  Registry* r = GlobalRegistry();
  std::lock_guard<std::mutex> lk(r->mu);
  if (r->factories.count(name)) {
    // op_kernel.cc:203-207: duplicate registration is fatal
    fprintf(stderr, "[euler_gpu] FATAL op kernel '%s' registered twice\n",
            name.c_str());
    abort();
  }
  r->factories[name] = factory;
}

OpStatus LookupOpKernel(const std::string& name) {
  Registry* r = GlobalRegistry();
  std::lock_guard<std::mutex> lk(r->mu);
  return r->factories.count(name) ? 0 : -1;
}

OpStatus CreateOpKernel(const std::string& name, OpKernel** kernel) {
  Registry* r = GlobalRegistry();
  std::lock_guard<std::mutex> lk(r->mu);
  auto k = r->kernels.find(name);
  if (k == r->kernels.end()) {
    auto f = r->factories.find(name);
    if (f == r->factories.end()) return -1;
    k = r->kernels.emplace(name, std::unique_ptr<OpKernel>(f->second(name))).first;
  }
  *kernel = k->second.get();
  return 0;
}

// ---------------------------------------------------------------- kernels
#define OP_FAIL(scope, what)                                                   \
  do { LogError(std::string(what) + ": " + (scope).error()); return; } while (0)

// API_SAMPLE_NB (core/kernels/sample_neighbor_op.cc:37-147, no-condition
// path): inputs node_ids (uint64), edge_types (int32), count (int32[1]),
// default_node (ignored, :134); outputs "<name>:0" idx [n,2] int32,
// ":1" ids uint64, ":2" weights float, ":3" types int32 (FillNeighbor).
// Post process (:86-132): `order_by id|weight [desc]` and `limit k` act on the
// rows that HAVE samples; rows without are filled with count x (0, 0.0, 0)
// afterwards (:134-143), so they keep `count` entries whatever the limit.
class GpuSampleNeighborOp : public OpKernel {
 public:
  explicit GpuSampleNeighborOp(const std::string& name) : OpKernel(name) {}
  void Compute(const NodeDef& nd, OpKernelContext* ctx) override {
    if (nd.inputs.size() < 4) {
      LogError("Argment 'node_ids', 'edge_types', 'count', 'default_node' must be specified!");
      return;
    }
    Tensor* ids_t = nullptr;
    if (ctx->tensor(nd.inputs[0], &ids_t) != 0) { LogError("Invalid argment 'node_ids'"); return; }
    std::vector<int32_t> edge_types, arg;
    if (!GetIntArg(nd, 1, ctx, &edge_types)) { LogError("Invalid argment 'edge_types'"); return; }
    if (!GetIntArg(nd, 2, ctx, &arg) || arg.empty()) { LogError("Invalid argment 'count'"); return; }
    euler_gpu_graph* g = ctx->graph();
    if (!g) { LogError("API_SAMPLE_NB: no graph initialised"); return; }
    const int64_t n = ids_t->NumElements();
    const int32_t count = arg[0];
    const int64_t total = n * count;
    bool post = false;
    for (const std::string& pp : nd.post_process) {
      int32_t ob, de; int64_t li;
      post = post || ParsePostProcess(pp, &ob, &de, &li);
    }
    OpScope sc(g);
    uint64_t* d_ids = sc.Alloc<uint64_t>(n);
    uint64_t* d_oid = sc.Alloc<uint64_t>(total);
    float* d_ow = sc.Alloc<float>(total);
    int32_t* d_ot = sc.Alloc<int32_t>(total);
    uint8_t* d_mask = sc.Alloc<uint8_t>(n);
    if (!sc.Upload(d_ids, ids_t->Raw<uint64_t>(), n * 8)) OP_FAIL(sc, "API_SAMPLE_NB");
    if (!sc.Call(euler_gpu_sample_neighbor(
            g, sc.stream(), ctx->seed(), ctx->NextCallId(), d_ids, n, nullptr, 1,
            edge_types.data(), (int32_t)edge_types.size(), count, EULER_GPU_LAYOUT_CORE, 0,
            d_oid, d_ow, d_ot, d_mask)))
      OP_FAIL(sc, "API_SAMPLE_NB");
    std::vector<int32_t> idx;
    std::vector<uint8_t> mask;
    int64_t out_total = total;
    if (post && n > 0) {
      idx.resize((size_t)n * 2);
      mask.assign((size_t)n, 0);
      // rows with samples: [i * count, (i + 1) * count); rows without: empty, so that
      // order_by / limit leave them alone; the device post-process repacks in place
      if (!sc.Download(mask.data(), d_mask, (size_t)n) || !sc.Sync()) OP_FAIL(sc, "API_SAMPLE_NB");
      // compact the valid rows to the front (the post-process expects packed rows)
      int64_t w = 0;
      for (int64_t i = 0; i < n; ++i) {
        idx[2 * i] = (int32_t)w;
        if (!mask[i]) w += count;
        idx[2 * i + 1] = (int32_t)w;
      }
      uint64_t* p_id = sc.Alloc<uint64_t>(w);
      float* p_w = sc.Alloc<float>(w);
      int32_t* p_t = sc.Alloc<int32_t>(w);
      int32_t* d_idx = sc.Alloc<int32_t>(n * 2);
      if (!sc.ok()) OP_FAIL(sc, "API_SAMPLE_NB");
      // runs of consecutive valid rows move with one copy each
      sc.MarkDirty();
      for (int64_t i = 0; i < n;) {
        if (mask[i]) { ++i; continue; }
        int64_t j = i;
        while (j < n && !mask[j]) ++j;
        const size_t src = (size_t)i * count, dst = (size_t)idx[2 * i], len = (size_t)(j - i) * count;
        if (hipMemcpyAsync(p_id + dst, d_oid + src, len * 8, hipMemcpyDeviceToDevice, (hipStream_t)sc.stream()) != hipSuccess ||
            hipMemcpyAsync(p_w + dst, d_ow + src, len * 4, hipMemcpyDeviceToDevice, (hipStream_t)sc.stream()) != hipSuccess ||
            hipMemcpyAsync(p_t + dst, d_ot + src, len * 4, hipMemcpyDeviceToDevice, (hipStream_t)sc.stream()) != hipSuccess) {
          LogError("API_SAMPLE_NB: device copy failed");
          return;
        }
        i = j;
      }
      if (!sc.Upload(d_idx, idx.data(), (size_t)n * 8)) OP_FAIL(sc, "API_SAMPLE_NB");
      int64_t cur = w;
      for (const std::string& pp : nd.post_process) {
        int32_t order_by, desc; int64_t limit;
        if (!ParsePostProcess(pp, &order_by, &desc, &limit)) continue;
        if (!sc.Call(euler_gpu_neighbor_post_process(sc.stream(), n, d_idx, cur, p_id, p_w, p_t,
                                                     order_by, desc, limit, &cur)))
          OP_FAIL(sc, "API_SAMPLE_NB post process");
      }
      std::vector<int32_t> pidx((size_t)n * 2);
      std::vector<uint64_t> hid((size_t)cur);
      std::vector<float> hw((size_t)cur);
      std::vector<int32_t> ht((size_t)cur);
      if (!sc.Download(pidx.data(), d_idx, (size_t)n * 8) || !sc.Download(hid.data(), p_id, (size_t)cur * 8) ||
          !sc.Download(hw.data(), p_w, (size_t)cur * 4) || !sc.Download(ht.data(), p_t, (size_t)cur * 4) ||
          !sc.Sync())
        OP_FAIL(sc, "API_SAMPLE_NB");
      // a row the post-process emptied (`limit 0`) is an empty row like one without
      // samples: the reference refills EVERY empty row with count x (0, 0.0, 0)
      // (core/kernels/sample_neighbor_op.cc:134-143)
      for (int64_t i = 0; i < n; ++i)
        if (pidx[2 * i + 1] == pidx[2 * i]) mask[i] = 1;
      out_total = 0;
      for (int64_t i = 0; i < n; ++i) out_total += mask[i] ? count : pidx[2 * i + 1] - pidx[2 * i];
      Tensor *t_idx = nullptr, *oid = nullptr, *ow = nullptr, *ot = nullptr;
      if (ctx->Allocate(OutputName(nd, 0), {(size_t)n, 2}, kInt32, &t_idx) != 0 ||
          ctx->Allocate(OutputName(nd, 1), {(size_t)out_total}, kUInt64, &oid) != 0 ||
          ctx->Allocate(OutputName(nd, 2), {(size_t)out_total}, kFloat, &ow) != 0 ||
          ctx->Allocate(OutputName(nd, 3), {(size_t)out_total}, kInt32, &ot) != 0) {
        LogError("Allocate output tensor failed!");
        return;
      }
      int64_t o = 0;
      for (int64_t i = 0; i < n; ++i) {
        t_idx->Raw<int32_t>()[2 * i] = (int32_t)o;
        if (mask[i]) {
          for (int32_t j = 0; j < count; ++j, ++o) {
            oid->Raw<uint64_t>()[o] = 0; ow->Raw<float>()[o] = 0.f; ot->Raw<int32_t>()[o] = 0;
          }
        } else {
          for (int32_t s2 = pidx[2 * i]; s2 < pidx[2 * i + 1]; ++s2, ++o) {
            oid->Raw<uint64_t>()[o] = hid[s2]; ow->Raw<float>()[o] = hw[s2]; ot->Raw<int32_t>()[o] = ht[s2];
          }
        }
        t_idx->Raw<int32_t>()[2 * i + 1] = (int32_t)o;
      }
      return;
    }
    Tensor *t_idx = nullptr, *oid = nullptr, *ow = nullptr, *ot = nullptr;
    if (ctx->Allocate(OutputName(nd, 0), {(size_t)n, 2}, kInt32, &t_idx) != 0 ||
        ctx->Allocate(OutputName(nd, 1), {(size_t)total}, kUInt64, &oid) != 0 ||
        ctx->Allocate(OutputName(nd, 2), {(size_t)total}, kFloat, &ow) != 0 ||
        ctx->Allocate(OutputName(nd, 3), {(size_t)total}, kInt32, &ot) != 0) {
      LogError("Allocate output tensor failed!");
      return;
    }
    // the copies first, the row offsets while they are in flight
    const bool queued = sc.Download(oid->Raw<uint64_t>(), d_oid, (size_t)total * 8) &&
                        sc.Download(ow->Raw<float>(), d_ow, (size_t)total * 4) &&
                        sc.Download(ot->Raw<int32_t>(), d_ot, (size_t)total * 4);
    for (int64_t i = 0; i < n; ++i) {
      t_idx->Raw<int32_t>()[2 * i] = (int32_t)(i * count);
      t_idx->Raw<int32_t>()[2 * i + 1] = (int32_t)((i + 1) * count);
    }
    if (!queued || !sc.Sync()) {
      sc.Drain();            // a copy may still be writing into the tensors
      for (int i = 0; i < 4; ++i) ctx->Deallocate(OutputName(nd, i));
      OP_FAIL(sc, "API_SAMPLE_NB");
    }
  }
};
REGISTER_OP_KERNEL("API_SAMPLE_NB", GpuSampleNeighborOp);

// API_SAMPLE_NODE (core/kernels/sample_node_op.cc:37-137): inputs node_type
// (int32[k]), count (int32); output "<name>:0" [count] int64.  A size
// mismatch logs and produces no output (:118-122).
class GpuSampleNodeOp : public OpKernel {
 public:
  explicit GpuSampleNodeOp(const std::string& name) : OpKernel(name) {}
  void Compute(const NodeDef& nd, OpKernelContext* ctx) override {
    if (nd.inputs.size() != 2) { LogError("Invalid input arguments for SampleNode"); return; }
    std::vector<int32_t> types, cnt;
    if (!GetIntArg(nd, 0, ctx, &types)) { LogError("Retrieve node_type input for SampleNode failed!"); return; }
    if (!GetIntArg(nd, 1, ctx, &cnt) || cnt.empty()) { LogError("Retrieve count input for SampleNode failed!"); return; }
    euler_gpu_graph* g = ctx->graph();
    if (!g) { LogError("API_SAMPLE_NODE: no graph initialised"); return; }
    const int32_t count = cnt[0];
    OpScope sc(g);
    uint64_t* d_out = sc.Alloc<uint64_t>((size_t)(count > 0 ? count : 0));
    if (!sc.ok()) OP_FAIL(sc, "API_SAMPLE_NODE");
    if (!sc.Call(euler_gpu_sample_node(g, sc.stream(), ctx->seed(), ctx->NextCallId(), types.data(),
                                       (int32_t)types.size(), count, d_out))) {
      LogError("Expect sample count: " + std::to_string(count) + ", real got:0 (" + sc.error() + ")");
      return;
    }
    Tensor* out = nullptr;
    if (ctx->Allocate(OutputName(nd, 0), {(size_t)count}, kInt64, &out) != 0) {
      LogError("Allocate output tensor failed!");
      return;
    }
    if (!sc.Download(out->Raw<int64_t>(), d_out, (size_t)count * 8) || !sc.Sync()) {
      ctx->Deallocate(OutputName(nd, 0));
      OP_FAIL(sc, "API_SAMPLE_NODE");
    }
  }
};
REGISTER_OP_KERNEL("API_SAMPLE_NODE", GpuSampleNodeOp);

// ID_UNIQUE (core/kernels/id_unique_op.cc:35-64), node-id branch.
class GpuIdUniqueOp : public OpKernel {
 public:
  explicit GpuIdUniqueOp(const std::string& name) : OpKernel(name) {}
  void Compute(const NodeDef& nd, OpKernelContext* ctx) override {
    Tensor* ids_t = nullptr;
    if (nd.inputs.empty() || ctx->tensor(nd.inputs[0], &ids_t) != 0) { LogError("ID_UNIQUE: missing input"); return; }
    const int64_t n = ids_t->NumElements();
    OpScope sc(ctx->graph());
    uint64_t* d_ids = sc.Alloc<uint64_t>(n);
    uint64_t* d_uq = sc.Alloc<uint64_t>(n);
    int32_t* d_gi = sc.Alloc<int32_t>(n);
    if (!sc.Upload(d_ids, ids_t->Raw<uint64_t>(), (size_t)n * 8)) OP_FAIL(sc, "ID_UNIQUE");
    int64_t nu = 0;
    if (!sc.Call(euler_gpu_id_unique(sc.stream(), d_ids, n, d_uq, d_gi, &nu))) OP_FAIL(sc, "ID_UNIQUE");
    Tensor *uq = nullptr, *gi = nullptr;
    if (ctx->Allocate(OutputName(nd, 0), {(size_t)nu}, kUInt64, &uq) != 0 ||
        ctx->Allocate(OutputName(nd, 1), {(size_t)n}, kInt32, &gi) != 0) {
      LogError("ID_UNIQUE: allocate failed");
      return;
    }
    if (!sc.Download(uq->Raw<uint64_t>(), d_uq, (size_t)nu * 8) ||
        !sc.Download(gi->Raw<int32_t>(), d_gi, (size_t)n * 4) || !sc.Sync()) {
      ctx->Deallocate(OutputName(nd, 0)); ctx->Deallocate(OutputName(nd, 1));
      OP_FAIL(sc, "ID_UNIQUE");
    }
  }
};
REGISTER_OP_KERNEL("ID_UNIQUE", GpuIdUniqueOp);

// IDX_GATHER (core/kernels/idx_gather_op.cc:33-55).
class GpuIdxGatherOp : public OpKernel {
 public:
  explicit GpuIdxGatherOp(const std::string& name) : OpKernel(name) {}
  void Compute(const NodeDef& nd, OpKernelContext* ctx) override {
    Tensor *idx_t = nullptr, *gi_t = nullptr;
    if (nd.inputs.size() < 2 || ctx->tensor(nd.inputs[0], &idx_t) != 0 ||
        ctx->tensor(nd.inputs[1], &gi_t) != 0) { LogError("IDX_GATHER: missing input"); return; }
    const int64_t n = gi_t->NumElements();
    OpScope sc(ctx->graph());
    int32_t* d_idx = (int32_t*)sc.AllocBytes(idx_t->TotalBytes());
    int32_t* d_gi = sc.Alloc<int32_t>(n);
    int32_t* d_out = sc.Alloc<int32_t>(n * 2);
    if (!sc.Upload(d_idx, idx_t->Raw<int32_t>(), idx_t->TotalBytes()) ||
        !sc.Upload(d_gi, gi_t->Raw<int32_t>(), (size_t)n * 4)) OP_FAIL(sc, "IDX_GATHER");
    int64_t total = 0;
    if (!sc.Call(euler_gpu_idx_gather(sc.stream(), d_idx, d_gi, n, d_out, &total))) OP_FAIL(sc, "IDX_GATHER");
    Tensor* out = nullptr;
    if (ctx->Allocate(OutputName(nd, 0), {(size_t)n, 2}, kInt32, &out) != 0) return;
    if (!sc.Download(out->Raw<int32_t>(), d_out, (size_t)n * 8) || !sc.Sync()) {
      ctx->Deallocate(OutputName(nd, 0));
      OP_FAIL(sc, "IDX_GATHER");
    }
  }
};
REGISTER_OP_KERNEL("IDX_GATHER", GpuIdxGatherOp);

// DATA_GATHER (core/kernels/data_gather_op.cc:33-80).
class GpuDataGatherOp : public OpKernel {
 public:
  explicit GpuDataGatherOp(const std::string& name) : OpKernel(name) {}
  void Compute(const NodeDef& nd, OpKernelContext* ctx) override {
    Tensor *data_t = nullptr, *idx_t = nullptr, *gi_t = nullptr;
    if (nd.inputs.size() < 3 || ctx->tensor(nd.inputs[0], &data_t) != 0 ||
        ctx->tensor(nd.inputs[1], &idx_t) != 0 ||
        ctx->tensor(nd.inputs[2], &gi_t) != 0) { LogError("DATA_GATHER: missing input"); return; }
    const DataType type = data_t->Type();
    if (type != kUInt64 && type != kFloat && type != kInt8 && type != kInt32) {
      LogError("error data type");
      return;
    }
    const int64_t n = gi_t->NumElements();
    const int32_t es = (int32_t)SizeOfType(type);
    OpScope sc(ctx->graph());
    void* d_data = sc.AllocBytes(data_t->TotalBytes());
    int32_t* d_idx = (int32_t*)sc.AllocBytes(idx_t->TotalBytes());
    int32_t* d_gi = sc.Alloc<int32_t>(n);
    int32_t* d_oidx = sc.Alloc<int32_t>(n * 2);
    if (!sc.Upload(d_data, data_t->Raw<char>(), data_t->TotalBytes()) ||
        !sc.Upload(d_idx, idx_t->Raw<int32_t>(), idx_t->TotalBytes()) ||
        !sc.Upload(d_gi, gi_t->Raw<int32_t>(), (size_t)n * 4)) OP_FAIL(sc, "DATA_GATHER");
    int64_t total = 0;
    if (!sc.Call(euler_gpu_idx_gather(sc.stream(), d_idx, d_gi, n, d_oidx, &total))) OP_FAIL(sc, "DATA_GATHER");
    void* d_out = sc.AllocBytes((size_t)total * es);
    if (!sc.ok()) OP_FAIL(sc, "DATA_GATHER");
    if (!sc.Call(euler_gpu_data_gather(sc.stream(), d_data, es, d_idx, d_gi, d_oidx, n, d_out)))
      OP_FAIL(sc, "DATA_GATHER");
    Tensor* out = nullptr;
    if (ctx->Allocate(OutputName(nd, 0), {(size_t)total}, type, &out) != 0) return;
    if (!sc.Download(out->Raw<char>(), d_out, (size_t)total * es) || !sc.Sync()) {
      ctx->Deallocate(OutputName(nd, 0));
      OP_FAIL(sc, "DATA_GATHER");
    }
  }
};
REGISTER_OP_KERNEL("DATA_GATHER", GpuDataGatherOp);

// API_GET_NB_NODE (core/kernels/get_neighbor_op.cc, no-condition path).
class GpuGetNeighborOp : public OpKernel {
 public:
  explicit GpuGetNeighborOp(const std::string& name) : OpKernel(name) {}
  void Compute(const NodeDef& nd, OpKernelContext* ctx) override {
    if (nd.inputs.size() < 2) { LogError("Argments 'node_ids' and 'edge_type' must be specified!"); return; }
    Tensor* ids_t = nullptr;
    std::vector<int32_t> et;
    if (ctx->tensor(nd.inputs[0], &ids_t) != 0) { LogError("Invalid argment 'node_ids'"); return; }
    if (!GetIntArg(nd, 1, ctx, &et)) { LogError("Invalid argment 'edge_types'"); return; }
    euler_gpu_graph* g = ctx->graph();
    if (!g) { LogError("API_GET_NB_NODE: no graph initialised"); return; }
    const int64_t n = ids_t->NumElements();
    OpScope sc(g);
    uint64_t* d_ids = sc.Alloc<uint64_t>(n);
    int32_t* d_idx = sc.Alloc<int32_t>(n * 2);
    if (!sc.Upload(d_ids, ids_t->Raw<uint64_t>(), (size_t)n * 8)) OP_FAIL(sc, "API_GET_NB_NODE");
    int64_t total = 0;
    if (!sc.Call(euler_gpu_get_full_neighbor(g, sc.stream(), d_ids, n, et.data(), (int32_t)et.size(),
                                             d_idx, &total, nullptr, nullptr, nullptr)))
      OP_FAIL(sc, "API_GET_NB_NODE");
    uint64_t* d_oid = sc.Alloc<uint64_t>(total);
    float* d_ow = sc.Alloc<float>(total);
    int32_t* d_ot = sc.Alloc<int32_t>(total);
    if (!sc.ok()) OP_FAIL(sc, "API_GET_NB_NODE");
    if (!sc.Call(euler_gpu_get_full_neighbor(g, sc.stream(), d_ids, n, et.data(), (int32_t)et.size(),
                                             d_idx, &total, d_oid, d_ow, d_ot)))
      OP_FAIL(sc, "API_GET_NB_NODE");
    // Post process (get_neighbor_op.cc:117-168): applied in order
    for (const std::string& post : nd.post_process) {
      int32_t order_by, desc; int64_t limit;
      if (!ParsePostProcess(post, &order_by, &desc, &limit)) continue;
      if (!sc.Call(euler_gpu_neighbor_post_process(sc.stream(), n, d_idx, total, d_oid, d_ow, d_ot,
                                                   order_by, desc, limit, &total)))
        OP_FAIL(sc, "API_GET_NB_NODE post process");
    }
    Tensor *idx = nullptr, *oid = nullptr, *ow = nullptr, *ot = nullptr;
    if (ctx->Allocate(OutputName(nd, 0), {(size_t)n, 2}, kInt32, &idx) != 0 ||
        ctx->Allocate(OutputName(nd, 1), {(size_t)total}, kUInt64, &oid) != 0 ||
        ctx->Allocate(OutputName(nd, 2), {(size_t)total}, kFloat, &ow) != 0 ||
        ctx->Allocate(OutputName(nd, 3), {(size_t)total}, kInt32, &ot) != 0) {
      LogError("Allocate output tensor failed!");
      return;
    }
    if (!sc.Download(idx->Raw<int32_t>(), d_idx, (size_t)n * 8) ||
        !sc.Download(oid->Raw<uint64_t>(), d_oid, (size_t)total * 8) ||
        !sc.Download(ow->Raw<float>(), d_ow, (size_t)total * 4) ||
        !sc.Download(ot->Raw<int32_t>(), d_ot, (size_t)total * 4) || !sc.Sync()) {
      for (int i = 0; i < 4; ++i) ctx->Deallocate(OutputName(nd, i));
      OP_FAIL(sc, "API_GET_NB_NODE");
    }
  }
};
REGISTER_OP_KERNEL("API_GET_NB_NODE", GpuGetNeighborOp);

// ------------------------------------------------------- layerwise chain
// The DAG of `v(nodes).sampleLNB(edge_types, n, m, default_node)`
// (euler/parser/translator.cc:338-386,489-527).  Literal attributes
// (default_node) arrive as the input STRING itself, as in the reference
// (`atol(node_def.inputs(i).c_str())`).

// API_GET_EDGE_SUM_WEIGHT (core/kernels/get_edge_sum_weight_op.cc:33-66):
// inputs roots (uint64), edge_types; outputs ":0" roots [n,1], ":1" sums [n,1].
class GpuGetEdgeSumWeightOp : public OpKernel {
 public:
  explicit GpuGetEdgeSumWeightOp(const std::string& name) : OpKernel(name) {}
  void Compute(const NodeDef& nd, OpKernelContext* ctx) override {
    Tensor* root_t = nullptr;
    std::vector<int32_t> et;
    if (nd.inputs.size() < 2 || ctx->tensor(nd.inputs[0], &root_t) != 0 ||
        !GetIntArg(nd, 1, ctx, &et)) { LogError("API_GET_EDGE_SUM_WEIGHT: bad inputs"); return; }
    euler_gpu_graph* g = ctx->graph();
    if (!g) { LogError("API_GET_EDGE_SUM_WEIGHT: no graph initialised"); return; }
    const int64_t n = root_t->NumElements();
    OpScope sc(g);
    uint64_t* d_ids = sc.Alloc<uint64_t>(n);
    float* d_w = sc.Alloc<float>(n);
    if (!sc.Upload(d_ids, root_t->Raw<uint64_t>(), (size_t)n * 8)) OP_FAIL(sc, "API_GET_EDGE_SUM_WEIGHT");
    if (!sc.Call(euler_gpu_get_edge_sum_weight(g, sc.stream(), d_ids, n, et.data(), (int32_t)et.size(), d_w)))
      OP_FAIL(sc, "API_GET_EDGE_SUM_WEIGHT");
    Tensor *o_root = nullptr, *o_w = nullptr;
    if (ctx->Allocate(OutputName(nd, 0), {(size_t)n, 1}, kUInt64, &o_root) != 0 ||
        ctx->Allocate(OutputName(nd, 1), {(size_t)n, 1}, kFloat, &o_w) != 0) {
      LogError("Allocate output tensor failed!");
      return;
    }
    memcpy(o_root->Raw<uint64_t>(), root_t->Raw<uint64_t>(), (size_t)n * 8);
    if (!sc.Download(o_w->Raw<float>(), d_w, (size_t)n * 4) || !sc.Sync()) {
      ctx->Deallocate(OutputName(nd, 0)); ctx->Deallocate(OutputName(nd, 1));
      OP_FAIL(sc, "API_GET_EDGE_SUM_WEIGHT");
    }
  }
};
REGISTER_OP_KERNEL("API_GET_EDGE_SUM_WEIGHT", GpuGetEdgeSumWeightOp);

// API_SAMPLE_ROOT (core/kernels/sample_root_op.cc:33-88): inputs roots,
// weights, n, m, default_node (literal); output ":0" [batch * m] uint64.
class GpuSampleRootOp : public OpKernel {
 public:
  explicit GpuSampleRootOp(const std::string& name) : OpKernel(name) {}
  void Compute(const NodeDef& nd, OpKernelContext* ctx) override {
    Tensor *roots_t = nullptr, *w_t = nullptr;
    std::vector<int32_t> n_v, m_v;
    if (nd.inputs.size() < 5 || ctx->tensor(nd.inputs[0], &roots_t) != 0 ||
        ctx->tensor(nd.inputs[1], &w_t) != 0 || !GetIntArg(nd, 2, ctx, &n_v) ||
        !GetIntArg(nd, 3, ctx, &m_v) || n_v.empty() || m_v.empty() || n_v[0] <= 0) {
      LogError("API_SAMPLE_ROOT: bad inputs");
      return;
    }
    const int64_t default_node = atol(nd.inputs[4].c_str());
    const int32_t n = n_v[0], m = m_v[0];
    const int64_t batch = roots_t->NumElements() / n;
    if (batch == 0) { LogError("batch size is zero!"); abort(); }   // EULER_LOG(FATAL)
    const int64_t cells = batch * n, draws = batch * m;
    OpScope sc(ctx->graph());
    uint64_t* d_r = sc.Alloc<uint64_t>(cells);
    float* d_w = sc.Alloc<float>(cells);
    uint64_t* d_o = sc.Alloc<uint64_t>(draws);
    if (!sc.Upload(d_r, roots_t->Raw<uint64_t>(), (size_t)cells * 8) ||
        !sc.Upload(d_w, w_t->Raw<float>(), (size_t)cells * 4)) OP_FAIL(sc, "API_SAMPLE_ROOT");
    if (!sc.Call(euler_gpu_sample_root(sc.stream(), ctx->seed(), ctx->NextCallId(), d_r, d_w, batch, n, m,
                                       default_node, d_o)))
      OP_FAIL(sc, "API_SAMPLE_ROOT");
    Tensor* out = nullptr;
    if (ctx->Allocate(OutputName(nd, 0), {(size_t)draws}, kUInt64, &out) != 0) {
      LogError("Allocate output tensor failed!");
      return;
    }
    if (!sc.Download(out->Raw<uint64_t>(), d_o, (size_t)draws * 8) || !sc.Sync()) {
      ctx->Deallocate(OutputName(nd, 0));
      OP_FAIL(sc, "API_SAMPLE_ROOT");
    }
  }
};
REGISTER_OP_KERNEL("API_SAMPLE_ROOT", GpuSampleRootOp);

// API_SAMPLE_L (core/kernels/sample_layer_op.cc:32-72): inputs l_root,
// edge_types, default_node (literal); outputs ":0" ids, ":1" weights,
// ":2" types, each [n, 1].
class GpuSampleLayerOp : public OpKernel {
 public:
  explicit GpuSampleLayerOp(const std::string& name) : OpKernel(name) {}
  void Compute(const NodeDef& nd, OpKernelContext* ctx) override {
    Tensor* root_t = nullptr;
    std::vector<int32_t> et;
    if (nd.inputs.size() < 3 || ctx->tensor(nd.inputs[0], &root_t) != 0 ||
        !GetIntArg(nd, 1, ctx, &et)) { LogError("API_SAMPLE_L: bad inputs"); return; }
    const int64_t default_node = atol(nd.inputs[2].c_str());
    euler_gpu_graph* g = ctx->graph();
    if (!g) { LogError("API_SAMPLE_L: no graph initialised"); return; }
    const int64_t n = root_t->NumElements();
    OpScope sc(g);
    uint64_t* d_r = sc.Alloc<uint64_t>(n);
    uint64_t* d_id = sc.Alloc<uint64_t>(n);
    float* d_w = sc.Alloc<float>(n);
    int32_t* d_t = sc.Alloc<int32_t>(n);
    if (!sc.Upload(d_r, root_t->Raw<uint64_t>(), (size_t)n * 8)) OP_FAIL(sc, "API_SAMPLE_L");
    if (!sc.Call(euler_gpu_sample_layer(g, sc.stream(), ctx->seed(), ctx->NextCallId(), d_r, n, et.data(),
                                        (int32_t)et.size(), default_node, d_id, d_w, d_t)))
      OP_FAIL(sc, "API_SAMPLE_L");
    Tensor *o_nb = nullptr, *o_w = nullptr, *o_t = nullptr;
    if (ctx->Allocate(OutputName(nd, 0), {(size_t)n, 1}, kUInt64, &o_nb) != 0 ||
        ctx->Allocate(OutputName(nd, 1), {(size_t)n, 1}, kFloat, &o_w) != 0 ||
        ctx->Allocate(OutputName(nd, 2), {(size_t)n, 1}, kInt32, &o_t) != 0) {
      LogError("Allocate output tensor failed!");
      return;
    }
    if (!sc.Download(o_nb->Raw<uint64_t>(), d_id, (size_t)n * 8) ||
        !sc.Download(o_w->Raw<float>(), d_w, (size_t)n * 4) ||
        !sc.Download(o_t->Raw<int32_t>(), d_t, (size_t)n * 4) || !sc.Sync()) {
      for (int i = 0; i < 3; ++i) ctx->Deallocate(OutputName(nd, i));
      OP_FAIL(sc, "API_SAMPLE_L");
    }
  }
};
REGISTER_OP_KERNEL("API_SAMPLE_L", GpuSampleLayerOp);

// API_LOCAL_SAMPLE_L (core/kernels/local_sample_layer_op.cc:43-146): inputs the
// four API_GET_NB_NODE outputs (idx, ids, weights, types), n, m, weight_func and
// default_node (literals); outputs ":0" ids, ":1" weights, ":2" types [batch*m, 1].
class GpuLocalSampleLayerOp : public OpKernel {
 public:
  explicit GpuLocalSampleLayerOp(const std::string& name) : OpKernel(name) {}
  void Compute(const NodeDef& nd, OpKernelContext* ctx) override {
    Tensor *idx_t = nullptr, *id_t = nullptr, *w_t = nullptr, *t_t = nullptr;
    std::vector<int32_t> n_v, m_v;
    if (nd.inputs.size() < 8 || ctx->tensor(nd.inputs[0], &idx_t) != 0 ||
        ctx->tensor(nd.inputs[1], &id_t) != 0 || ctx->tensor(nd.inputs[2], &w_t) != 0 ||
        ctx->tensor(nd.inputs[3], &t_t) != 0 || !GetIntArg(nd, 4, ctx, &n_v) ||
        !GetIntArg(nd, 5, ctx, &m_v) || n_v.empty() || m_v.empty() || n_v[0] <= 0) {
      LogError("API_LOCAL_SAMPLE_L: bad inputs");
      return;
    }
    const std::string weight_func = nd.inputs[6];
    const int64_t default_node = atol(nd.inputs[7].c_str());
    if (weight_func != "sqrt") LogError("weight function: " + weight_func + " not support");
    const int32_t n = n_v[0], m = m_v[0];
    const int64_t batch = idx_t->NumElements() / (n * 2);
    const int64_t total = id_t->NumElements(), draws = batch * m;
    OpScope sc(ctx->graph());
    int32_t* d_idx = sc.Alloc<int32_t>(idx_t->NumElements());
    uint64_t* d_id = sc.Alloc<uint64_t>(total);
    float* d_w = sc.Alloc<float>(total);
    int32_t* d_t = sc.Alloc<int32_t>(total);
    uint64_t* o_id = sc.Alloc<uint64_t>(draws);
    float* o_w = sc.Alloc<float>(draws);
    int32_t* o_t = sc.Alloc<int32_t>(draws);
    if (!sc.Upload(d_idx, idx_t->Raw<int32_t>(), (size_t)idx_t->NumElements() * 4) ||
        !sc.Upload(d_id, id_t->Raw<uint64_t>(), (size_t)total * 8) ||
        !sc.Upload(d_w, w_t->Raw<float>(), (size_t)total * 4) ||
        !sc.Upload(d_t, t_t->Raw<int32_t>(), (size_t)total * 4)) OP_FAIL(sc, "API_LOCAL_SAMPLE_L");
    if (!sc.Call(euler_gpu_local_sample_layer(sc.stream(), ctx->seed(), ctx->NextCallId(), d_idx, d_id, d_w,
                                              d_t, total, batch, n, m, weight_func.c_str(),
                                              default_node, o_id, o_w, o_t)))
      OP_FAIL(sc, "API_LOCAL_SAMPLE_L");
    Tensor *r_id = nullptr, *r_w = nullptr, *r_t = nullptr;
    if (ctx->Allocate(OutputName(nd, 0), {(size_t)draws, 1}, kUInt64, &r_id) != 0 ||
        ctx->Allocate(OutputName(nd, 1), {(size_t)draws, 1}, kFloat, &r_w) != 0 ||
        ctx->Allocate(OutputName(nd, 2), {(size_t)draws, 1}, kInt32, &r_t) != 0) {
      LogError("Allocate output tensor failed!");
      return;
    }
    if (!sc.Download(r_id->Raw<uint64_t>(), o_id, (size_t)draws * 8) ||
        !sc.Download(r_w->Raw<float>(), o_w, (size_t)draws * 4) ||
        !sc.Download(r_t->Raw<int32_t>(), o_t, (size_t)draws * 4) || !sc.Sync()) {
      for (int i = 0; i < 3; ++i) ctx->Deallocate(OutputName(nd, i));
      OP_FAIL(sc, "API_LOCAL_SAMPLE_L");
    }
  }
};
REGISTER_OP_KERNEL("API_LOCAL_SAMPLE_L", GpuLocalSampleLayerOp);

// API_SPARSE_GEN_ADJ (core/kernels/sparse_gen_adj_op.cc:35-64): host-only
// reshaping - (root id, batch row) pairs + an alias of l_nb.
class SparseGenAdjOp : public OpKernel {
 public:
  explicit SparseGenAdjOp(const std::string& name) : OpKernel(name) {}
  void Compute(const NodeDef& nd, OpKernelContext* ctx) override {
    Tensor *roots_t = nullptr, *l_nb_t = nullptr;
    std::vector<int32_t> n_v;
    if (nd.inputs.size() < 3 || ctx->tensor(nd.inputs[0], &roots_t) != 0 ||
        ctx->tensor(nd.inputs[1], &l_nb_t) != 0 || !GetIntArg(nd, 2, ctx, &n_v) ||
        n_v.empty() || n_v[0] <= 0) { LogError("API_SPARSE_GEN_ADJ: bad inputs"); return; }
    const int32_t n = n_v[0];
    const int32_t batch = roots_t->NumElements() / n;
    if (batch == 0) { LogError("batch size is zero!"); abort(); }
    Tensor* rb = nullptr;
    if (ctx->Allocate(OutputName(nd, 0), {(size_t)roots_t->NumElements(), 2}, kUInt64, &rb) != 0) {
      LogError("Allocate output tensor failed!");
      return;
    }
    for (int32_t i = 0; i < batch; ++i)
      for (int32_t j = 0; j < n; ++j) {
        const int32_t cnt = i * n + j;
        rb->Raw<uint64_t>()[cnt * 2] = roots_t->Raw<uint64_t>()[cnt];
        rb->Raw<uint64_t>()[cnt * 2 + 1] = (uint64_t)i;
      }
    ctx->AddAlias(OutputName(nd, 1), l_nb_t);
  }
};
REGISTER_OP_KERNEL("API_SPARSE_GEN_ADJ", SparseGenAdjOp);

// API_SPARSE_GET_ADJ (core/kernels/sparse_get_adj_op.cc:35-92): inputs
// root_batch [R,2] (id, batch row), l_nb, edge_types, m; outputs ":0" idx
// [R,2] int32, ":1" ids.  The device entry point takes the batch row as
// r / n, which is what API_SPARSE_GEN_ADJ writes; any other root_batch is
// rejected.
class GpuSparseGetAdjOp : public OpKernel {
 public:
  explicit GpuSparseGetAdjOp(const std::string& name) : OpKernel(name) {}
  void Compute(const NodeDef& nd, OpKernelContext* ctx) override {
    Tensor *rb_t = nullptr, *l_nb_t = nullptr;
    std::vector<int32_t> et, m_v;
    if (nd.inputs.size() < 4 || ctx->tensor(nd.inputs[0], &rb_t) != 0 ||
        ctx->tensor(nd.inputs[1], &l_nb_t) != 0 || !GetIntArg(nd, 2, ctx, &et) ||
        !GetIntArg(nd, 3, ctx, &m_v) || m_v.empty() || m_v[0] < 0) {
      LogError("API_SPARSE_GET_ADJ: bad inputs");
      return;
    }
    euler_gpu_graph* g = ctx->graph();
    if (!g) { LogError("API_SPARSE_GET_ADJ: no graph initialised"); return; }
    const int32_t m = m_v[0];
    const int64_t R = rb_t->NumElements() / 2;
    const uint64_t* rb = rb_t->Raw<uint64_t>();
    const int64_t batch = R ? (int64_t)rb[2 * (R - 1) + 1] + 1 : 0;
    const int64_t n = batch ? R / batch : 0;
    bool regular = batch > 0 && n * batch == R &&
                   (int64_t)l_nb_t->NumElements() >= batch * m;
    std::vector<uint64_t> roots((size_t)R);
    for (int64_t r = 0; r < R && regular; ++r) {
      roots[r] = rb[2 * r];
      regular = (int64_t)rb[2 * r + 1] == r / n;
    }
    if (R > 0 && !regular) { LogError("API_SPARSE_GET_ADJ: root_batch is not API_SPARSE_GEN_ADJ's"); return; }
    int64_t total = 0;
    std::vector<int32_t> idx((size_t)R * 2);
    std::vector<uint64_t> vals;
    if (R > 0) {
      OpScope sc(g);
      uint64_t* d_r = sc.Alloc<uint64_t>(R);
      uint64_t* d_nb = sc.Alloc<uint64_t>((size_t)batch * m);
      int32_t* d_idx = sc.Alloc<int32_t>(R * 2);
      void* d_ws = sc.AllocBytes(euler_gpu_sparse_get_adj_workspace(batch, (int32_t)n, m));
      if (!sc.Upload(d_r, roots.data(), (size_t)R * 8) ||
          !sc.Upload(d_nb, l_nb_t->Raw<uint64_t>(), (size_t)batch * m * 8)) OP_FAIL(sc, "API_SPARSE_GET_ADJ");
      if (!sc.Call(euler_gpu_sparse_get_adj(g, sc.stream(), d_r, d_nb, batch, (int32_t)n, m, et.data(),
                                            (int32_t)et.size(), d_ws, d_idx, &total, nullptr)))
        OP_FAIL(sc, "API_SPARSE_GET_ADJ");
      uint64_t* d_v = sc.Alloc<uint64_t>((size_t)total);
      if (!sc.ok()) OP_FAIL(sc, "API_SPARSE_GET_ADJ");
      if (total > 0 &&
          !sc.Call(euler_gpu_sparse_get_adj(g, sc.stream(), d_r, d_nb, batch, (int32_t)n, m, et.data(),
                                            (int32_t)et.size(), d_ws, d_idx, &total, d_v)))
        OP_FAIL(sc, "API_SPARSE_GET_ADJ");
      vals.resize((size_t)total);
      if (!sc.Download(idx.data(), d_idx, (size_t)R * 8) ||
          (total && !sc.Download(vals.data(), d_v, (size_t)total * 8)) || !sc.Sync())
        OP_FAIL(sc, "API_SPARSE_GET_ADJ");
    }
    Tensor *o_idx = nullptr, *o_data = nullptr;
    if (ctx->Allocate(OutputName(nd, 0), {(size_t)R, 2}, kInt32, &o_idx) != 0 ||
        ctx->Allocate(OutputName(nd, 1), {(size_t)total}, kUInt64, &o_data) != 0) {
      LogError("Allocate output tensor failed!");
      return;
    }
    if (R) memcpy(o_idx->Raw<int32_t>(), idx.data(), (size_t)R * 8);
    if (total) memcpy(o_data->Raw<uint64_t>(), vals.data(), (size_t)total * 8);
  }
};
REGISTER_OP_KERNEL("API_SPARSE_GET_ADJ", GpuSparseGetAdjOp);

// API_GATHER_RESULT (core/kernels/gather_result_op.cc:25-47): three aliases.
class GatherResultOp : public OpKernel {
 public:
  explicit GatherResultOp(const std::string& name) : OpKernel(name) {}
  void Compute(const NodeDef& nd, OpKernelContext* ctx) override {
    for (int i = 0; i < 3; ++i) {
      Tensor* t = nullptr;
      if ((int)nd.inputs.size() <= i || ctx->tensor(nd.inputs[i], &t) != 0) {
        LogError("API_GATHER_RESULT: missing input");
        return;
      }
      ctx->AddAlias(OutputName(nd, i), t);
    }
  }
};
REGISTER_OP_KERNEL("API_GATHER_RESULT", GatherResultOp);

}  // namespace gpu_abi
}  // namespace euler

extern "C" {

int euler_op_registered(const char* op_name) {
  return euler::LookupOpKernel(op_name) == 0 ? 1 : 0;
}

int64_t euler_op_run_sample_nb(euler_gpu_graph* g, uint64_t seed,
                               const uint64_t* node_ids, int64_t n,
                               const int32_t* edge_types, int32_t k,
                               int32_t count, int32_t* idx_out, uint64_t* id_out,
                               float* w_out, int32_t* t_out) {
  using namespace euler;
  return euler_op_run_sample_nb_post(g, seed, 0, node_ids, n, edge_types, k, count, nullptr,
                                     idx_out, id_out, w_out, t_out);
}

int64_t euler_op_run_sample_nb_post(euler_gpu_graph* g, uint64_t seed, uint32_t call_id,
                                    const uint64_t* node_ids, int64_t n,
                                    const int32_t* edge_types, int32_t k, int32_t count,
                                    const char* post_process, int32_t* idx_out,
                                    uint64_t* id_out, float* w_out, int32_t* t_out) {
  using namespace euler;
  OpKernelContext ctx;
  ctx.SetGraph(g);
  ctx.SetSeed(seed);
  ctx.SetCallId(call_id);
  Tensor *t_ids = nullptr, *t_et = nullptr, *t_cnt = nullptr, *t_def = nullptr;
  ctx.Allocate("nodes", {(size_t)n}, kUInt64, &t_ids);
  ctx.Allocate("edge_types", {(size_t)k}, kInt32, &t_et);
  ctx.Allocate("nb_count", {1}, kInt32, &t_cnt);
  ctx.Allocate("default_node", {1}, kInt32, &t_def);
  memcpy(t_ids->Raw<uint64_t>(), node_ids, (size_t)n * 8);
  if (k) memcpy(t_et->Raw<int32_t>(), edge_types, (size_t)k * 4);
  *t_cnt->Raw<int32_t>() = count;
  *t_def->Raw<int32_t>() = -1;
  NodeDef nd{"API_SAMPLE_NB,0", "API_SAMPLE_NB",
             {"nodes", "edge_types", "nb_count", "default_node"}, {}};
  {
    std::string cur;
    for (const char* p = post_process ? post_process : ""; ; ++p) {
      if (*p == ';' || *p == 0) {
        if (!cur.empty()) nd.post_process.push_back(cur);
        cur.clear();
        if (*p == 0) break;
      } else {
        cur.push_back(*p);
      }
    }
  }
  OpKernel* kernel = nullptr;
  if (CreateOpKernel("API_SAMPLE_NB", &kernel) != 0) return -1;
  kernel->Compute(nd, &ctx);
  Tensor *idx = nullptr, *oid = nullptr, *ow = nullptr, *ot = nullptr;
  if (ctx.tensor(OutputName(nd, 0), &idx) != 0 || ctx.tensor(OutputName(nd, 1), &oid) != 0 ||
      ctx.tensor(OutputName(nd, 2), &ow) != 0 || ctx.tensor(OutputName(nd, 3), &ot) != 0)
    return -2;   // op logged an error and produced no output
  memcpy(idx_out, idx->Raw<int32_t>(), idx->TotalBytes());
  memcpy(id_out, oid->Raw<uint64_t>(), oid->TotalBytes());
  memcpy(w_out, ow->Raw<float>(), ow->TotalBytes());
  memcpy(t_out, ot->Raw<int32_t>(), ot->TotalBytes());
  return oid->NumElements();
}

// Runs the registered API_GET_NB_NODE kernel (with its post-process strings,
// ';'-separated) through the plugin API on host tensors; returns the number of
// neighbours or < 0.
int64_t euler_op_run_get_nb(euler_gpu_graph* g, const uint64_t* node_ids, int64_t n,
                            const int32_t* edge_types, int32_t k,
                            const char* post_process, int64_t capacity,
                            int32_t* idx_out, uint64_t* id_out, float* w_out,
                            int32_t* t_out) {
  using namespace euler;
  OpKernelContext ctx;
  ctx.SetGraph(g);
  Tensor *t_ids = nullptr, *t_et = nullptr;
  ctx.Allocate("nodes", {(size_t)n}, kUInt64, &t_ids);
  ctx.Allocate("edge_types", {(size_t)k}, kInt32, &t_et);
  memcpy(t_ids->Raw<uint64_t>(), node_ids, (size_t)n * 8);
  if (k) memcpy(t_et->Raw<int32_t>(), edge_types, (size_t)k * 4);
  NodeDef nd{"API_GET_NB_NODE,0", "API_GET_NB_NODE", {"nodes", "edge_types"}, {}};
  std::string cur;
  for (const char* p = post_process ? post_process : ""; ; ++p) {
    if (*p == ';' || *p == 0) {
      if (!cur.empty()) nd.post_process.push_back(cur);
      cur.clear();
      if (*p == 0) break;
    } else {
      cur.push_back(*p);
    }
  }
  OpKernel* kernel = nullptr;
  if (CreateOpKernel("API_GET_NB_NODE", &kernel) != 0) return -1;
  kernel->Compute(nd, &ctx);
  Tensor *idx = nullptr, *oid = nullptr, *ow = nullptr, *ot = nullptr;
  if (ctx.tensor(OutputName(nd, 0), &idx) != 0 || ctx.tensor(OutputName(nd, 1), &oid) != 0 ||
      ctx.tensor(OutputName(nd, 2), &ow) != 0 || ctx.tensor(OutputName(nd, 3), &ot) != 0)
    return -2;
  if (oid->NumElements() > capacity) return -3;
  memcpy(idx_out, idx->Raw<int32_t>(), idx->TotalBytes());
  memcpy(id_out, oid->Raw<uint64_t>(), oid->TotalBytes());
  memcpy(w_out, ow->Raw<float>(), ow->TotalBytes());
  memcpy(t_out, ot->Raw<int32_t>(), ot->TotalBytes());
  return oid->NumElements();
}

// Runs the DAG the reference's translator builds for
// `v(nodes).sampleLNB(edge_types, n, m, default_node).as(nb)` (euler/parser/
// translator.cc:338-386,489-527) through the plugin API on host tensors:
// nb:0 = adjacency idx [batch*n, 2], nb:1 = adjacency ids (<= capacity),
// nb:2 = the sampled layer [batch*m].  Returns the number of adjacency ids or
// < 0.  call ids: API_SAMPLE_ROOT takes the context's next one, API_SAMPLE_L
// the one after.
int64_t euler_op_run_sample_lnb(euler_gpu_graph* g, uint64_t seed, uint32_t first_call_id,
                                const uint64_t* node_ids, int64_t batch, int32_t n,
                                const int32_t* edge_types, int32_t k, int32_t m,
                                const char* weight_func, int64_t default_node,
                                int64_t capacity,
                                int32_t* adj_idx_out, uint64_t* adj_id_out,
                                uint64_t* l_nb_out) {
  using namespace euler;
  OpKernelContext ctx;
  ctx.SetGraph(g);
  ctx.SetSeed(seed);
  ctx.SetCallId(first_call_id);
  const int64_t R = batch * n;
  Tensor *t_ids = nullptr, *t_et = nullptr, *t_n = nullptr, *t_m = nullptr;
  ctx.Allocate("nodes", {(size_t)R}, kUInt64, &t_ids);
  ctx.Allocate("edge_types", {(size_t)k}, kInt32, &t_et);
  ctx.Allocate("n", {1}, kInt32, &t_n);
  ctx.Allocate("m", {1}, kInt32, &t_m);
  memcpy(t_ids->Raw<uint64_t>(), node_ids, (size_t)R * 8);
  if (k) memcpy(t_et->Raw<int32_t>(), edge_types, (size_t)k * 4);
  *t_n->Raw<int32_t>() = n;
  *t_m->Raw<int32_t>() = m;
  const std::string dn = std::to_string(default_node);
  const std::string wf = weight_func ? weight_func : "";
  std::vector<NodeDef> dag;
  std::string roots_name, l_nb_name;
  if (wf.empty()) {           // Translator::TrivialSampleLayer
    dag.push_back({"API_GET_EDGE_SUM_WEIGHT,0", "API_GET_EDGE_SUM_WEIGHT", {"nodes", "edge_types"}, {}});
    dag.push_back({"API_SAMPLE_ROOT,1", "API_SAMPLE_ROOT",
                   {"API_GET_EDGE_SUM_WEIGHT,0:0", "API_GET_EDGE_SUM_WEIGHT,0:1", "n", "m", dn}, {}});
    dag.push_back({"API_SAMPLE_L,2", "API_SAMPLE_L", {"API_SAMPLE_ROOT,1:0", "edge_types", dn}, {}});
    roots_name = "API_GET_EDGE_SUM_WEIGHT,0:0";
    l_nb_name = "API_SAMPLE_L,2:0";
  } else {                    // Translator::GeneralSampleLayer (API_RESHAPE of the roots
                              // is host plumbing: the roots are fed as they are)
    dag.push_back({"API_GET_NB_NODE,1", "API_GET_NB_NODE", {"nodes", "edge_types"}, {}});
    dag.push_back({"API_LOCAL_SAMPLE_L,2", "API_LOCAL_SAMPLE_L",
                   {"API_GET_NB_NODE,1:0", "API_GET_NB_NODE,1:1", "API_GET_NB_NODE,1:2",
                    "API_GET_NB_NODE,1:3", "n", "m", wf, dn}, {}});
    roots_name = "nodes";
    l_nb_name = "API_LOCAL_SAMPLE_L,2:0";
  }
  dag.push_back({"API_SPARSE_GEN_ADJ,3", "API_SPARSE_GEN_ADJ", {roots_name, l_nb_name, "n"}, {}});
  dag.push_back({"API_SPARSE_GET_ADJ,4", "API_SPARSE_GET_ADJ",
                 {"API_SPARSE_GEN_ADJ,3:0", "API_SPARSE_GEN_ADJ,3:1", "edge_types", "m"}, {}});
  dag.push_back({"API_GATHER_RESULT,5", "API_GATHER_RESULT",
                 {"API_SPARSE_GET_ADJ,4:0", "API_SPARSE_GET_ADJ,4:1", l_nb_name}, {}});
  for (const NodeDef& nd : dag) {
    OpKernel* kernel = nullptr;
    if (CreateOpKernel(nd.op, &kernel) != 0) return -1;
    kernel->Compute(nd, &ctx);
  }
  Tensor *idx = nullptr, *ids = nullptr, *l_nb = nullptr;
  if (ctx.tensor("API_GATHER_RESULT,5:0", &idx) != 0 ||
      ctx.tensor("API_GATHER_RESULT,5:1", &ids) != 0 ||
      ctx.tensor("API_GATHER_RESULT,5:2", &l_nb) != 0)
    return -2;   // an op logged an error and produced no output
  if (ids->NumElements() > capacity) return -3;
  memcpy(adj_idx_out, idx->Raw<int32_t>(), idx->TotalBytes());
  memcpy(adj_id_out, ids->Raw<uint64_t>(), ids->TotalBytes());
  memcpy(l_nb_out, l_nb->Raw<uint64_t>(), l_nb->TotalBytes());
  return ids->NumElements();
}

}  // extern "C"
