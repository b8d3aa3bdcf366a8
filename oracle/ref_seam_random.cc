// ORACLE (test infrastructure, NOT product code).
//
// Link-time seam: this translation unit is linked INSTEAD OF the reference's
// euler/common/random.cc (random.cc:22-27) when building oracle/_ref.  The
// reference sampler sources are compiled unmodified from /root/reference and
// call euler::common::ThreadLocalRandom() (random.h:24), the only RNG entry
// point on the hot path (callers: compact_weighted_collection.h:35,
// alias_method.cc:68,77, core/kernels/sample_node_split_op.cc:83).  Here it is
// backed by the counter RNG contract of oracle/eo_rng.h, with the stream
// context held in thread-local storage and set by the harness before every
// reference call.
#include "euler/common/random.h"

#include "eo_rng.h"

namespace {
thread_local eo_rng_ctx g_ctx = {0, 0, 0, 0, 0};
}  // namespace

extern "C" void euler_ref_set_rng(uint64_t seed, uint32_t call_id,
                                  uint32_t domain, uint64_t stream) {
  g_ctx.seed = seed;
  g_ctx.call_id = call_id;
  g_ctx.domain = domain;
  g_ctx.stream = stream;
  g_ctx.draw_idx = 0;
}

extern "C" uint64_t euler_ref_rng_draws() { return g_ctx.draw_idx; }

namespace euler {
namespace common {

double ThreadLocalRandom() { return eo_next_uniform(&g_ctx); }

}  // namespace common
}  // namespace euler
