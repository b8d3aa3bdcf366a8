// Host-side all-to-all of a few int64 per peer between the ranks of ONE node
// through POSIX shared memory - the split sizes of a sharded hop.
//
// The reference ships its split sizes inside the gRPC requests themselves
// (euler/core/kernels/remote_op.cc:62-142: one ExecuteRequest per shard holding its
// inputs, the reply carries its own length).  RCCL's all-to-all needs both sides'
// counts on the host BEFORE the data moves, and exchanging 8 numbers through a
// GPU collective costs a kernel launch plus a device synchronisation per hop
// (~70 us on one rank, more on eight).  The ranks of this framework are
// processes of one node (one per GPU), so the counts travel through a mailbox in
// /dev/shm instead: no GPU work, no stream synchronisation, a few microseconds.
//
// Layout: world x world slots; slot (src, dst) = {seq, ack, payload[kWidth]}.
// Round t (1, 2, ...): src waits until dst acknowledged round t - 1, writes the
// payload, publishes seq = t (release); dst waits for seq == t (acquire), reads,
// publishes ack = t.  Every rank runs the same sequence of rounds (SPMD), sends
// to all peers first and then receives from all, so no cycle of waits exists.
#include <atomic>
#include <chrono>
#include <cerrno>
#include <cstring>
#include <new>
#include <string>
#include <thread>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include "common.h"

namespace {

constexpr int kWidth = 8;   // int64 per (src, dst) message

struct alignas(128) Slot {
  std::atomic<uint64_t> seq;
  std::atomic<uint64_t> ack;
  int64_t payload[kWidth];
};

struct Header {
  std::atomic<uint32_t> magic;
  int32_t world;
  std::atomic<int32_t> attached;
};

constexpr uint32_t kMagic = 0x45554c52u;   // "EULR"

size_t RegionBytes(int32_t world) {
  return sizeof(Slot) + sizeof(Slot) * (size_t)world * (size_t)world;   // header in slot 0's size
}

}  // namespace

struct euler_shm {
  std::string name;
  void* base = nullptr;
  size_t bytes = 0;
  int32_t rank = 0, world = 0;
  uint64_t round = 0;
  Slot* slots = nullptr;
  bool owner = false;
};

using euler_gpu::Fail;

extern "C" {

int euler_shm_open(const char* name, int32_t rank, int32_t world, int32_t create,
                   euler_shm** out) {
  if (!name || !out || world <= 0 || world > 64 || rank < 0 || rank >= world)
    return Fail(EULER_GPU_EINVAL, "shm_open: bad arguments (world <= 64)");
  static_assert(sizeof(Header) <= sizeof(Slot), "header must fit the first slot");
  const size_t bytes = RegionBytes(world);
  int fd = -1;
  if (create) {
    fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0) return Fail(EULER_GPU_EIO, std::string("shm_open(create) ") + name + ": " + strerror(errno));
    if (ftruncate(fd, (off_t)bytes) != 0) {
      close(fd); shm_unlink(name);
      return Fail(EULER_GPU_EIO, std::string("ftruncate ") + name + ": " + strerror(errno));
    }
  } else {
    fd = shm_open(name, O_RDWR, 0600);
    if (fd < 0) return Fail(EULER_GPU_EIO, std::string("shm_open ") + name + ": " + strerror(errno));
    struct stat sb;
    if (fstat(fd, &sb) != 0 || (size_t)sb.st_size < bytes) {
      close(fd);
      return Fail(EULER_GPU_EIO, std::string("shm region too small: ") + name);
    }
  }
  void* base = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (base == MAP_FAILED) {
    if (create) shm_unlink(name);
    return Fail(EULER_GPU_EIO, std::string("mmap ") + name + ": " + strerror(errno));
  }
  auto* h = reinterpret_cast<Header*>(base);
  if (create) {
    // a fresh region is zero-filled: every seq / ack starts at 0 = "round 0 done"
    h->world = world;
    h->attached.store(0, std::memory_order_relaxed);
    h->magic.store(kMagic, std::memory_order_release);
  } else if (h->magic.load(std::memory_order_acquire) != kMagic || h->world != world) {
    munmap(base, bytes);
    return Fail(EULER_GPU_EIO, std::string("shm region of another job: ") + name);
  }
  h->attached.fetch_add(1, std::memory_order_acq_rel);
  auto* s = new (std::nothrow) euler_shm();
  if (!s) { munmap(base, bytes); return Fail(EULER_GPU_ENOMEM, "shm_open: out of memory"); }
  s->name = name; s->base = base; s->bytes = bytes; s->rank = rank; s->world = world;
  s->slots = reinterpret_cast<Slot*>(reinterpret_cast<uint8_t*>(base) + sizeof(Slot));
  s->owner = create != 0;
  *out = s;
  return EULER_GPU_OK;
}

// send / recv: [world * width] int64, message of peer p at [p * width, (p + 1) * width)
int euler_shm_alltoall_i64(euler_shm* s, const int64_t* send, int64_t* recv, int32_t width,
                           int64_t timeout_ms) {
  if (!s || !send || !recv || width <= 0 || width > kWidth)
    return Fail(EULER_GPU_EINVAL, "shm_alltoall: bad arguments (width <= 8)");
  const uint64_t t = ++s->round;
  const auto deadline = std::chrono::steady_clock::now() +
                        std::chrono::milliseconds(timeout_ms > 0 ? timeout_ms : 60000);
  auto wait_for = [&](const std::atomic<uint64_t>& a, uint64_t want) -> bool {
    for (uint32_t spins = 0;; ++spins) {
      if (a.load(std::memory_order_acquire) >= want) return true;
      if (spins > 2000) {
        if (std::chrono::steady_clock::now() > deadline) return false;
        std::this_thread::yield();
      }
    }
  };
  const int32_t W = s->world, me = s->rank;
  for (int32_t k = 0; k < W; ++k) {              // start with the next rank: spreads the load
    const int32_t dst = (me + k) % W;
    Slot& sl = s->slots[(size_t)me * W + dst];
    if (!wait_for(sl.ack, t - 1)) return Fail(EULER_GPU_EIO, "shm_alltoall: peer did not consume the previous round (timeout)");
    std::memcpy(sl.payload, send + (size_t)dst * width, sizeof(int64_t) * (size_t)width);
    sl.seq.store(t, std::memory_order_release);
  }
  for (int32_t k = 0; k < W; ++k) {
    const int32_t src = (me + W - k) % W;
    Slot& sl = s->slots[(size_t)src * W + me];
    if (!wait_for(sl.seq, t)) return Fail(EULER_GPU_EIO, "shm_alltoall: peer did not send (timeout)");
    std::memcpy(recv + (size_t)src * width, sl.payload, sizeof(int64_t) * (size_t)width);
    sl.ack.store(t, std::memory_order_release);
  }
  return EULER_GPU_OK;
}

// number of ranks that have attached so far (the creator unlinks the name once
// everyone has: the mapping stays valid, the name cannot leak)
int32_t euler_shm_attached(const euler_shm* s) {
  if (!s) return -1;
  return reinterpret_cast<const Header*>(s->base)->attached.load(std::memory_order_acquire);
}

int euler_shm_unlink(euler_shm* s) {
  if (!s) return Fail(EULER_GPU_EINVAL, "shm_unlink: null");
  if (shm_unlink(s->name.c_str()) != 0 && errno != ENOENT)
    return Fail(EULER_GPU_EIO, std::string("shm_unlink ") + s->name + ": " + strerror(errno));
  return EULER_GPU_OK;
}

void euler_shm_close(euler_shm* s) {
  if (!s) return;
  if (s->base) munmap(s->base, s->bytes);
  delete s;
}

}  // extern "C"
