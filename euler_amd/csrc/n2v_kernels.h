// node2vec (tf_euler/kernels/random_walk_op.cc:83-168, RWCallback / BuildWeights /
// RandomSelect) on gfx950: the lane-per-walker reference loop, the whole-wave two-cursor
// walk with integer running sums (default), and the per-step launch with a workgroup per
// hub row.  Included by walk_kernels.hip after WalkArgs (inside namespace euler_gpu).
#ifndef EULER_AMD_CSRC_N2V_KERNELS_H_
#define EULER_AMD_CSRC_N2V_KERNELS_H_

// Iterator over GetFullNeighbor(node, listed types) in the reference order
// (listed-type order, storage order inside a type) without materialising it.
struct NbIter {
  const uint64_t* nbr;
  const float* nw;
  const int32_t* type_end;
  const int32_t* et;
  int32_t k, T;
  int32_t x;       // current listed-type slot
  int32_t p, e;    // current position / end inside the row
  __device__ __forceinline__ void Seek() {
    while (x < k) {
      const int32_t t = et[x];
      if (t >= 0 && t < T) {
        p = t == 0 ? 0 : type_end[t - 1];
        e = type_end[t];
        if (p < e) return;
      }
      ++x;
    }
  }
  __device__ __forceinline__ void Init(const GraphView& g, int64_t row,
                                       const int32_t* et_, int32_t k_) {
    et = et_; k = k_; T = g.T; x = 0; p = 0; e = 0;
    if (row < 0) { x = k; return; }
    const RowMeta m = LoadRowMeta(g, row);
    nbr = g.nbr + m.row_ptr; nw = g.prefix_w + m.row_ptr; type_end = m.type_end;
    Seek();
  }
  __device__ __forceinline__ bool Done() const { return x >= k; }
  __device__ __forceinline__ int64_t Id() const { return (int64_t)nbr[p]; }
  __device__ __forceinline__ float Weight() const {
    return __fsub_rn(nw[p], p == 0 ? 0.f : nw[p - 1]);
  }
  __device__ __forceinline__ void Next() {
    if (++p >= e) { ++x; Seek(); }
  }
};

// node2vec step weights (BuildWeights, random_walk_op.cc:140-168) streamed:
// the child list is merged against the parent's list with two cursors and the
// biased weight of each child is produced in order.
struct BiasedStream {
  NbIter c, pn;
  int64_t parent_id;
  float p, q;
  __device__ __forceinline__ bool Done() const { return c.Done(); }
  // weight of the current child (advances the parent cursor as the reference)
  __device__ __forceinline__ float Take(int64_t* id) {
    const int64_t cid = c.Id();
    float w = c.Weight();
    for (;;) {
      if (pn.Done()) {
        w = cid != parent_id ? __fdiv_rn(w, q) : __fdiv_rn(w, p);
        break;
      }
      const int64_t pid = pn.Id();
      if (cid < pid) {
        w = cid != parent_id ? __fdiv_rn(w, q) : __fdiv_rn(w, p);
        break;
      } else if (cid == pid) {
        pn.Next();
        break;
      } else {
        pn.Next();
      }
    }
    *id = cid;
    c.Next();
    return w;
  }
};

// node2vec (RWCallback, random_walk_op.cc:83-138).  One lane per walker.  The
// reference materialises w[], builds f32 running sums and binary-searches
// them; with non-negative weights the hit interval is unique, so the same
// index is found by one sequential pass for the total and a second pass that
// stops at the first running sum > r.  The running sums are the same
// sequential f32 adds, hence bit-identical.  (All-zero totals follow the
// reference's fall-through: every probe moves `low` up, ending on the last
// element.)
__global__ __launch_bounds__(256, kWavesPerSimd) void Node2VecKernel(const WalkArgs a) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t L = a.walk_len + 1;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n;
       i += stride) {
    int64_t cur = a.nodes[i];
    int64_t parent = cur;         // parent_ids_ starts as the start nodes
    bool have_parent_nb = false;  // parent_neighbors_ starts empty
    a.out[i * L] = cur;
    for (int32_t s = 0; s < a.walk_len; ++s) {
      const int32_t* et = a.edge_types + s * a.k;
      const int64_t row = FindRow(a.g, (uint64_t)cur);
      const int64_t prow = have_parent_nb ? FindRow(a.g, (uint64_t)parent) : -1;
      const int32_t* pet = s > 0 ? a.edge_types + (s - 1) * a.k : et;
      BiasedStream bs;
      bs.parent_id = parent; bs.p = a.p; bs.q = a.q;
      bs.c.Init(a.g, row, et, a.k);
      bs.pn.Init(a.g, prow, pet, a.k);
      int64_t sample_id = a.default_node;
      if (!bs.Done()) {
        float total = 0.f;
        int64_t nc = 0, id;
        while (!bs.Done()) { total = __fadd_rn(total, bs.Take(&id)); ++nc; }
        const double u = RngDraw(a.seed, a.call_id + (uint32_t)s, kDomainWalk,
                                 (uint64_t)i, 0);
        const double r = ScaleDraw(u, 0.f, total);
        bs.c.Init(a.g, row, et, a.k);
        bs.pn.Init(a.g, prow, pet, a.k);
        float acc = 0.f;
        bool found = false;
        while (!bs.Done()) {
          const float w = bs.Take(&id);
          const float prev = acc;
          acc = __fadd_rn(acc, w);
          if ((double)prev <= r && r < (double)acc) { found = true; break; }
        }
        if (!found) {
          // fall-through of RandomSelect: no interval holds r (total == 0).
          // Every probe then takes `interval_end <= r`: low = mid + 1, so the
          // search ends on mid = nc - 1; `id` already is that last element.
        }
        sample_id = id;
      }
      a.out[i * L + s + 1] = sample_id;
      parent = cur;
      have_parent_nb = true;
      cur = sample_id;
    }
  }
}

// ------------------------------------------------------------------------
// node2vec, one WAVE per walker (default).  The step's weights come out of a
// two-cursor walk over the child's and the parent's neighbour lists in storage
// order (BuildWeights, random_walk_op.cc:140-168) - a sequential recurrence
// that cannot be split across lanes without changing which parent entry each
// child is compared with.  What can be shared is the memory traffic: the 64
// lanes copy both lists into LDS in coalesced chunks (ids, and the weights as
// differences of the running sums), and lane 0 runs the recurrence out of LDS
// (tens of cycles per step instead of a dependent HBM round trip per lane and
// step, and no lane waits for a neighbour's hub row).  Pass 1 accumulates the
// total with the reference's sequential f32 adds, pass 2 stops at the first
// running sum > r - the same index the reference's bisection of those sums
// returns (and its last element when the total is 0).  Measured on the metric
// graph (100 K walkers x 10 steps, walkers sit on hubs of 1e5+ neighbours):
// 1.69 s -> 1.15 s; prefetching the next entries by hand made it slower (1.34 s).
// ------------------------------------------------------------------------
constexpr int kN2vChunk = 128;
constexpr int kN2vMaxSeg = kMaxListedTypes;

struct N2vList {           // one neighbour list = listed type segments of a row
  int64_t row_ptr;         // row start in nbr / prefix_w
  // where the list lives: the graph's arrays (ids = nbr + row_ptr, nw = prefix_w + row_ptr:
  // the weights are differences of the running sums), or a row FETCHED from its owner shard
  // (euler_gpu_node2vec_step: ids and the weights themselves, w != nullptr)
  const uint64_t* ids;
  const float* nw;
  const float* w;
  int32_t n_seg;
  int32_t total;           // entries
  int32_t seg_b[kN2vMaxSeg];
  int32_t seg_len[kN2vMaxSeg];
};

constexpr int kN2vCk = 2 * kN2vChunk;   // checkpoints of the whole-wave path (they reuse the staging arrays)

struct alignas(16) N2vLds {
  union { uint64_t c_id[kN2vChunk]; float ck_acc[kN2vCk]; };
  union { uint64_t p_id[kN2vChunk]; int32_t ck_k[kN2vCk]; };
  float c_w[kN2vChunk];
  N2vList child, parent;
};

// Built by lane 0, read by all lanes after a wave sync.
__device__ __forceinline__ void N2vBuildList(N2vList* L, const GraphView& g, int64_t row,
                                             const int32_t* et, int32_t k) {
  L->n_seg = 0; L->total = 0; L->row_ptr = 0;
  L->ids = g.nbr; L->nw = g.prefix_w; L->w = nullptr;
  if (row < 0) return;
  const RowMeta m = LoadRowMeta(g, row);
  L->row_ptr = m.row_ptr;
  L->ids = g.nbr + m.row_ptr; L->nw = g.prefix_w + m.row_ptr;
  for (int32_t x = 0; x < k; ++x) {
    const int32_t t = et[x];
    if (t < 0 || t >= g.T) continue;
    const int32_t b = t == 0 ? 0 : m.type_end[t - 1];
    const int32_t len = m.type_end[t] - b;
    if (len <= 0) continue;
    L->seg_b[L->n_seg] = b;
    L->seg_len[L->n_seg] = len;
    ++L->n_seg;
    L->total += len;
  }
}

// the running sums of a step never decrease: the graph says so, or - lists fetched from their
// owners - a pass over the fetched weights found none negative (euler_gpu_node2vec_step)
__device__ __forceinline__ bool N2vMonotone(const WalkArgs& a) {
  return a.nonneg_flag != nullptr ? *a.nonneg_flag != 0 : a.g.monotone != 0;
}

// A fetched row (packed set: row r = entries [idx[2 r], idx[2 r + 1]) of ids / w - the GQL
// result's (begin, end) pairs); row < 0: empty.
__device__ __forceinline__ void N2vBuildListFetched(N2vList* L, const int32_t* idx,
                                                    const uint64_t* ids, const float* w, int32_t row) {
  L->n_seg = 0; L->total = 0; L->row_ptr = 0;
  L->ids = ids; L->nw = nullptr; L->w = w;
  if (row < 0 || idx == nullptr) return;
  const int32_t b = idx[2 * row], len = idx[2 * row + 1] - b;
  if (len <= 0) return;
  L->row_ptr = b;
  L->ids = ids + b;
  L->w = w != nullptr ? w + b : nullptr;
  L->n_seg = 1; L->seg_b[0] = 0; L->seg_len[0] = len; L->total = len;
}

// weight of the entry at row-relative position ph
__device__ __forceinline__ float N2vWeightAt(const N2vList& L, int32_t ph) {
  if (L.w != nullptr) return L.w[ph];
  return __fsub_rn(L.nw[ph], ph == 0 ? 0.f : L.nw[ph - 1]);
}

// The child list IS the parent list (the walker took a self loop and the steps list the
// same types): the two cursors then move in lockstep and every child is common.
__device__ __forceinline__ bool N2vSameLists(const N2vList& c, const N2vList& p) {
  if (c.total == 0 || c.total != p.total || c.ids != p.ids || c.n_seg != p.n_seg)
    return false;
  for (int32_t x = 0; x < c.n_seg; ++x)
    if (c.seg_b[x] != p.seg_b[x] || c.seg_len[x] != p.seg_len[x]) return false;
  return true;
}

// The same for lists FETCHED from their owners (one segment each, different buffers even when they
// are one node's row): the cursors move in lockstep exactly when the two id sequences are equal,
// so the wave compares them - lengths first (almost every pair ends there), then 64 ids a step.
// Without it a walker that took a self loop on a long row has an event per entry, looks
// "ascending" and goes to the sequential automaton: 57 650 entries by one lane = 27 ms of a
// 100 000-walker step that otherwise takes 3-7 (profiles/r6_sharded_n2v_steps.txt).
__device__ __forceinline__ bool N2vSameFetched(const N2vList& c, const N2vList& p, int lane) {
  if (c.total == 0 || c.total != p.total || c.n_seg != 1 || p.n_seg != 1) return false;
  if (c.ids == p.ids) return true;
  for (int32_t b = 0; b < c.total; b += 64) {
    const int32_t j = b + lane;
    const bool ne = j < c.total && c.ids[j] != p.ids[j];
    if (__ballot(ne) != 0ull) return false;
  }
  return true;
}

// ... by a whole workgroup (every thread calls; the lists' headers in LDS)
__device__ __forceinline__ bool N2vSameFetchedBlock(const N2vList& c, const N2vList& p) {
  if (c.total == 0 || c.total != p.total || c.n_seg != 1 || p.n_seg != 1) return false;
  if (c.ids == p.ids) return true;
  int ne = 0;
  for (int32_t j = threadIdx.x; j < c.total && !ne; j += blockDim.x) ne = c.ids[j] != p.ids[j] ? 1 : 0;
  return __syncthreads_or(ne) == 0;
}

// row-relative position of logical entry j
__device__ __forceinline__ int32_t N2vPhys(const N2vList& L, int32_t j) {
  for (int32_t x = 0; x < L.n_seg; ++x) {
    if (j < L.seg_len[x]) return L.seg_b[x] + j;
    j -= L.seg_len[x];
  }
  return 0;
}

// ------------------------------------------------------------------------
// node2vec, the two-cursor walk done by the WHOLE wave (tuning key 7 >= 2).
// BuildWeights (random_walk_op.cc:140-168) moves a parent cursor k forward only:
// child j is resolved against pn[k] -
//   cn[j] <  pn[k] : "not a common neighbour", weight / q (or / p), j++
//   cn[j] == pn[k] : common neighbour, weight kept,            j++, k++
//   cn[j] >  pn[k] : k++ and look again
// - so between two moves of k every child compares with the SAME pn[k], and a run
// of children below it is resolved by all lanes at once (one ballot).  Only a
// child that is >= pn[k] is an event: the wave then scans pn from k in 64-entry
// steps for the first entry >= that child (another ballot) and goes on.  On lists
// in storage order - what the reference's `outV` returns and what this backend's
// synthetic graphs hold - pn[k] soon sits on a large id and events are rare (3 per
// step on the metric graph, after which the parent list is used up).  Lists that
// ARE ascending make every child an event; a step whose first chunk has many is
// handed to the lane-0 automaton (same results, different speed).
//
// A lane holds kN2vR CONSECUTIVE child entries (a chunk is 64 * kN2vR entries): the
// per-chunk work - ballots, the scan across lanes, the loop - is paid once per
// four entries, and the kernel is bound by instruction issue (a 64-entry chunk
// cost ~200 wave instructions).
//
// The running sums stay the reference's sequential f32 adds, mostly without doing
// them one by one.  For a carry m * ulp in [2^e, 2^(e+1)) and carry + d below
// 2^(e+1), fl(carry + d) = (m + n) * ulp with n = d / ulp rounded to nearest - the
// same n for every m unless d / ulp ends in exactly .5 (then the tie goes to the
// even m + n and depends on m).  So with no such tie and no sum reaching 2^(e+1),
// the f32 chain IS an integer running sum of the n's over the mantissa.  n comes
// out of the adder itself: fl(2^e + d) has mantissa offset n (2^e is an even m, and
// without a tie the parity does not matter); d - (fl(2^e + d) - 2^e) is exact and
// equals +-ulp/2 exactly on a tie.  A lane the integer sum cannot pass - the sum
// leaves the binade there, a tie, a negative entry, a carry of 0 - does its entries
// by real f32 adds from its predecessor's sum and the lanes after it start over
// from there (binade crossings: ~20 per list); a chunk with more than four such
// lanes runs the add chain lane after lane (a row_shr DPP chain).  On the metric
// graph's hub rows 92 % of the 64-entry chunks need no real add at all
// (tools/n2v_binade_model.py restates the scheme in numpy against the sequential
// sums).
//
// Pass 1 runs the chunks once for the total and leaves (running sum, parent cursor)
// checkpoints in LDS - one per 2^sh chunks; the draw r then names the first
// checkpoint whose sum exceeds it, and only the chunks after the previous
// checkpoint are run again to find the entry (the sums never decrease when the
// weights, p and q are non-negative; otherwise the second pass starts from the
// first entry as the reference's scan would).  That entry is the index
// RandomSelect's bisection of the sums returns (its last element when the total
// is 0).
// ------------------------------------------------------------------------
constexpr int kN2vR = 4;
constexpr int kN2vChunkR = 64 * kN2vR;

// Diagnostic counters of the node2vec kernels (euler_gpu_random_walk_stats): 0 steps by
// the whole-wave path, 1 their child entries, 2 steps handed to the sequential
// automaton, 3 their child entries, 4 moves of the parent cursor, 5 chunks / wave-chunks
// whose running sums needed the add chain, 6 steps by the workgroup kernel, 7 their entries.
__device__ unsigned long long g_n2v_stats[8];
__device__ int g_n2v_stats_on;          // set by euler_gpu_random_walk_stats(.., reset = 2)
__device__ __forceinline__ void N2vCount(int slot, unsigned long long v) {
  if (g_n2v_stats_on) atomicAdd(&g_n2v_stats[slot], v);
}

__device__ __forceinline__ int64_t ReadLane64(int64_t v, int src) {
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, src);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)v >> 32), src);
  return (int64_t)(((uint64_t)hi << 32) | lo);
}

// The add chain: lane after lane, a lane's kN2vR entries one after the other.
// *cin = the sum before this lane's first entry; returns the sum after the last lane.
__device__ __forceinline__ float ChunkChainVec(float carry, const float (&d)[kN2vR], int lane,
                                               float* cin_out) {
  float s = 0.f, cin = 0.f, my_rin = 0.f;
#pragma unroll
  for (int row = 0; row < 4; ++row) {
    const float rin = row == 0 ? carry : ReadLaneF(s, 16 * row - 1);
    if ((lane >> 4) == row) my_rin = rin;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      // after iteration t the first t + 1 lanes of this row hold their sums; the
      // rows before it recompute theirs unchanged
      float in = __int_as_float(__builtin_amdgcn_update_dpp(
          0, __float_as_int(s), 0x111 /* row_shr:1 */, 0xf, 0xf, true));
      if ((lane & 15) == 0) in = my_rin;
      cin = in;
      float x = in;
#pragma unroll
      for (int r = 0; r < kN2vR; ++r) x = __fadd_rn(x, d[r]);
      s = x;
    }
  }
  *cin_out = cin;
  return ReadLaneF(s, 63);
}

// Running sums of one chunk: *cin = the sum before this lane's first entry (its
// entries' sums follow by kN2vR adds); returns the sum after the chunk.
__device__ __forceinline__ float WaveSumsVec(float carry, const float (&d)[kN2vR], int lane,
                                             float* cin_out) {
  const float carry0 = carry;
  float cin = 0.f, lout = 0.f;
  int start = 0;
  for (int iter = 0; iter < 4; ++iter) {
    const uint32_t cb = __float_as_uint(carry);
    const uint32_t e = cb >> 23;                     // sign bit set => e >= 256
    const bool range_ok = e >= 30u && e < 254u;
    const uint32_t bb = cb & 0xFF800000u;
    const float B = __uint_as_float(bb);
    const float twoB = __fadd_rn(B, B);
    const float half_ulp = __uint_as_float(bb - (24u << 23));
    uint32_t N = 0;
    bool okl = true;
#pragma unroll
    for (int r = 0; r < kN2vR; ++r) {
      const float t = __fadd_rn(B, d[r]);
      const float err = __fsub_rn(d[r], __fsub_rn(t, B));
      const bool ok = d[r] >= 0.f && fabsf(err) != half_ulp && t < twoB;
      okl = okl && ok;
      N += ok ? __float_as_uint(t) - bb : 0u;        // each < 2^23
    }
    const bool active = lane >= start;
    if (!active) N = 0;
    // <= (2^23 - 1) * (64 * 4 + 1): fits 32 bits unsigned
    const uint32_t off_out = (cb - bb) + WaveInclusiveAdd(N, lane);
    const unsigned long long prob =
        __ballot(active && (!range_ok || !okl || off_out >= (1u << 23)));
    const int c = prob != 0 ? __ffsll((long long)prob) - 1 : 64;
    if (active && lane < c) {
      cin = __uint_as_float(bb + off_out - N);
      lout = __uint_as_float(bb + off_out);
    }
    if (c == 64) { *cin_out = cin; return ReadLaneF(lout, 63); }
    const float cin_c = c == start ? carry : ReadLaneF(lout, c - 1);
    float x = cin_c;
#pragma unroll
    for (int r = 0; r < kN2vR; ++r) x = __fadd_rn(x, ReadLaneF(d[r], c));
    if (lane == c) { cin = cin_c; lout = x; }
    carry = x;
    start = c + 1;
    if (start == 64) { *cin_out = cin; return carry; }
  }
  if (lane == 0) N2vCount(5, 1);
  return ChunkChainVec(carry0, d, lane, cin_out);
}

// BuildWeights' weight of a child that is not a common neighbour (random_walk_op.cc:
// 152-160): w / p for the parent itself, w / q otherwise.
__device__ __forceinline__ float N2vScaled(const WalkArgs& a, float w, bool is_parent) {
  const float inv = is_parent ? a.inv_p : a.inv_q;
  if (a.inv_p != 0.f && a.inv_q != 0.f) return __fmul_rn(w, inv);
  return is_parent ? __fdiv_rn(w, a.p) : __fdiv_rn(w, a.q);
}

// A lane's kN2vR consecutive entries of the child list from logical entry jl (ids,
// and the weights as differences of the row's running sums - what `outV` hands the
// reference).
struct N2vVec {
  int64_t cid[kN2vR];
  float w[kN2vR];
  uint32_t live;                  // bit r: the entry exists
};

__device__ __forceinline__ N2vVec N2vLoadVec(const WalkArgs& a, const N2vList& L, int32_t nc,
                                             int32_t jl) {
  N2vVec v;
  v.live = 0;
  const float* c_nw = L.nw;
  const uint64_t* c_nbr = L.ids;
  if (L.w != nullptr) {
    // a fetched row: the weights themselves
#pragma unroll
    for (int r = 0; r < kN2vR; ++r) {
      v.cid[r] = 0;
      v.w[r] = 0.f;
      if (jl + r < nc) {
        const int32_t ph = N2vPhys(L, jl + r);
        v.cid[r] = (int64_t)c_nbr[ph];
        v.w[r] = L.w[ph];
        v.live |= 1u << r;
      }
    }
  } else if (L.n_seg == 1 && jl + kN2vR <= nc) {
    // one listed type (or one non-empty): the entries are adjacent in the row
    const int32_t ph = L.seg_b[0] + jl;
    float prev = ph == 0 ? 0.f : c_nw[ph - 1];
#pragma unroll
    for (int r = 0; r < kN2vR; ++r) {
      v.cid[r] = (int64_t)c_nbr[ph + r];
      const float x = c_nw[ph + r];
      v.w[r] = __fsub_rn(x, prev);
      prev = x;
    }
    v.live = (1u << kN2vR) - 1;
  } else {
#pragma unroll
    for (int r = 0; r < kN2vR; ++r) {
      v.cid[r] = 0;
      v.w[r] = 0.f;
      if (jl + r < nc) {
        const int32_t ph = N2vPhys(L, jl + r);
        v.cid[r] = (int64_t)c_nbr[ph];
        v.w[r] = __fsub_rn(c_nw[ph], ph == 0 ? 0.f : c_nw[ph - 1]);
        v.live |= 1u << r;
      }
    }
  }
  return v;
}

// lane-local pick of entry r.  Written as a masked OR so that it stays a select
// network: an if-chain over e.cid[q] is folded into a dynamically indexed load,
// which puts the whole struct into scratch / LDS.
__device__ __forceinline__ int64_t N2vPick(const N2vVec& e, int r) {
  uint64_t x = 0;
#pragma unroll
  for (int q = 0; q < kN2vR; ++q) x |= (uint64_t)e.cid[q] & (r == q ? ~0ull : 0ull);
  return (int64_t)x;
}

// The parent cursor: k, and pn[k] once it has been read (a run of chunks whose
// children all sit below pn[k] reads it once).
struct N2vCursor {
  int32_t k;
  int32_t m_k;      // the k that M belongs to, -1 = none
  int64_t M;
};

// BuildWeights' comparisons for one chunk: returns the mask of this lane's entries
// that are common neighbours; *events = moves of the parent cursor.
__device__ __forceinline__ uint32_t N2vEventsWave(const WalkArgs& a, const N2vList& P, int lane,
                                                  int32_t np, const N2vVec& e, N2vCursor* c,
                                                  int32_t* events_out) {
  const uint64_t* p_nbr = P.ids;
  int32_t k = c->k;
  uint32_t keep = 0;
  int res_lane = -1, res_r = -1;    // entries up to (res_lane, res_r) are resolved
  int32_t events = 0;
  while (k < np) {
    if (c->m_k != k) { c->M = (int64_t)p_nbr[N2vPhys(P, k)]; c->m_k = k; }
    uint32_t evm = 0;
#pragma unroll
    for (int r = 0; r < kN2vR; ++r)
      if (((e.live >> r) & 1u) && (lane > res_lane || (lane == res_lane && r > res_r)) &&
          e.cid[r] >= c->M)
        evm |= 1u << r;
    const unsigned long long ev = __ballot(evm != 0);
    if (ev == 0) break;                       // every remaining child is below pn[k]
    const int f = __ffsll((long long)ev) - 1;
    const int rsel = evm != 0 ? __ffs((int)evm) - 1 : 0;
    const int64_t cf = ReadLane64(N2vPick(e, rsel), f);
    // first k' >= k with pn[k'] >= cf (the cursor skips the smaller entries)
    // (128 entries a dependent trip, the two loads in flight together: the sharded walk 45.5 -> 42.7 ms,
    // the single-GPU walk 19.4 -> 19.0; four loads - 256 entries - cost registers: 34.3 ms)
    bool eq = false;
    for (;;) {
      const int32_t kk = k + lane, kk1 = kk + 64;
      int64_t pv = 0, pv1 = 0;
      if (kk < np) pv = (int64_t)p_nbr[N2vPhys(P, kk)];
      if (kk1 < np) pv1 = (int64_t)p_nbr[N2vPhys(P, kk1)];
      const unsigned long long ge = __ballot(kk < np && pv >= cf);
      if (ge != 0) {
        const int g = __ffsll((long long)ge) - 1;
        k += g;
        c->M = ReadLane64(pv, g);
        c->m_k = k;
        eq = c->M == cf;
        break;
      }
      const unsigned long long ge1 = __ballot(kk1 < np && pv1 >= cf);
      if (ge1 != 0) {
        const int g = __ffsll((long long)ge1) - 1;
        k += 64 + g;
        c->M = ReadLane64(pv1, g);
        c->m_k = k;
        eq = c->M == cf;
        break;
      }
      k += 128;
      if (k >= np) { k = np; break; }
    }
    if (eq) { if (lane == f) keep |= 1u << rsel; ++k; }
    res_lane = f;
    if (lane == f) res_r = rsel;
    ++events;
  }
  c->k = k;
  *events_out = events;
  if (lane == 0 && events > 0) N2vCount(4, (unsigned long long)events);
  return keep;
}

__device__ __forceinline__ void N2vWeights(const WalkArgs& a, const N2vVec& e, uint32_t keep,
                                           int64_t parent, float (&d)[kN2vR]) {
#pragma unroll
  for (int r = 0; r < kN2vR; ++r)
    d[r] = !((e.live >> r) & 1u) ? 0.f
           : ((keep >> r) & 1u)  ? e.w[r]
                                 : N2vScaled(a, e.w[r], e.cid[r] == parent);
}

// The entry whose interval [sum before, sum after) holds r, if this lane has one:
// bit r of the result.
__device__ __forceinline__ uint32_t N2vHits(const N2vVec& e, const float (&d)[kN2vR], float cin,
                                            double r) {
  uint32_t hit = 0;
  float x = cin;
#pragma unroll
  for (int q = 0; q < kN2vR; ++q) {
    const float prev = x;
    x = __fadd_rn(x, d[q]);
    if (((e.live >> q) & 1u) && (double)prev <= r && r < (double)x) hit |= 1u << q;
  }
  return hit;
}

// One step of one walker by the whole wave; returns false when the lists look
// ascending (the caller then runs the sequential automaton).  *out = sampled id.
__device__ __forceinline__ bool N2vStepParallel(const WalkArgs& a, N2vLds& S, int lane,
                                                int64_t parent, int64_t walker, int32_t step,
                                                int64_t* out, const int same_in = -1) {
  const int32_t nc = S.child.total, np = S.parent.total;
  const int32_t nchunks = (nc + kN2vChunkR - 1) / kN2vChunkR;
  int32_t sh = 0;
  while ((nchunks >> sh) > kN2vCk) ++sh;
  const int32_t n_slots = nchunks >> sh;
  const bool same = same_in >= 0 ? same_in != 0 : N2vSameLists(S.child, S.parent);
  int32_t events = 0;
  float acc = 0.f, cin;
  float d[kN2vR];
  N2vCursor cur{0, -1, 0};
  // the next chunk's entries are requested before this chunk is worked on: a wave
  // has one dependent round trip per chunk, not two.  (Two chunks ahead costs 20
  // more registers: 23.7 ms against 18.8 on the hub workload, and the slowest
  // walker is no faster.)
  N2vVec e = N2vLoadVec(a, S.child, nc, lane * kN2vR), nx = e;
  for (int32_t ci = 0; ci < nchunks; ++ci) {
    if (ci + 1 < nchunks) nx = N2vLoadVec(a, S.child, nc, (ci + 1) * kN2vChunkR + lane * kN2vR);
    const uint32_t keep = same ? e.live : N2vEventsWave(a, S.parent, lane, np, e, &cur, &events);
    if (ci == 0 && nc >= 64 && events > 16) return false;   // ascending lists
    N2vWeights(a, e, keep, parent, d);
    acc = WaveSumsVec(acc, d, lane, &cin);
    if (((ci + 1) & ((1 << sh) - 1)) == 0 && lane == 0) {
      S.ck_acc[((ci + 1) >> sh) - 1] = acc;
      S.ck_k[((ci + 1) >> sh) - 1] = cur.k;
    }
    e = nx;
  }
  const float total = acc;
  const double u = RngDraw(a.seed, a.call_id + (uint32_t)step, kDomainWalk, (uint64_t)walker, 0);
  const double r = ScaleDraw(u, 0.f, total);
  WaveSync();
  int32_t first = 0;
  if (N2vMonotone(a) && a.p > 0.f && a.q > 0.f) {
    first = n_slots;
    for (int32_t base = 0; base < n_slots; base += 64) {
      const int32_t idx = base + lane;
      const unsigned long long gt = __ballot(idx < n_slots && (double)S.ck_acc[idx] > r);
      if (gt != 0) { first = base + __ffsll((long long)gt) - 1; break; }
    }
  }
  acc = first == 0 ? 0.f : S.ck_acc[first - 1];
  cur.k = first == 0 ? 0 : S.ck_k[first - 1];
  cur.m_k = -1;
  bool found = false;
  int64_t result = a.default_node;
  for (int32_t ci = first << sh; ci < nchunks && !found; ++ci) {
    e = N2vLoadVec(a, S.child, nc, ci * kN2vChunkR + lane * kN2vR);
    const uint32_t keep = same ? e.live : N2vEventsWave(a, S.parent, lane, np, e, &cur, &events);
    N2vWeights(a, e, keep, parent, d);
    acc = WaveSumsVec(acc, d, lane, &cin);
    const uint32_t hm = N2vHits(e, d, cin, r);
    const unsigned long long hit = __ballot(hm != 0);
    if (hit != 0) {
      found = true;
      result = ReadLane64(N2vPick(e, hm != 0 ? __ffs((int)hm) - 1 : 0), __ffsll((long long)hit) - 1);
    }
  }
  // no interval holds r (total == 0): RandomSelect's fall-through ends on the last element
  if (!found) result = (int64_t)S.child.ids[N2vPhys(S.child, nc - 1)];
  WaveSync();          // the checkpoints share LDS with the next step's lists
  *out = result;
  return true;
}

// One step of one walker, lists in LDS chunk by chunk, lane 0 runs the two-cursor
// recurrence (see the header comment above); every lane returns the sampled id.
__device__ __forceinline__ int64_t N2vStepSequential(const WalkArgs& a, N2vLds& S, int lane,
                                                     int64_t parent, int64_t walker,
                                                     int32_t step) {
  const int32_t nc = S.child.total, np = S.parent.total;
  const uint64_t* c_nbr = S.child.ids;
  const uint64_t* p_nbr = S.parent.ids;
  float total = 0.f;
  double r = 0.0;
  uint64_t last_id = 0;
  for (int pass = 0; pass < 2; ++pass) {
    int32_t j = 0, k = 0;           // cursors (logical entries)
    int32_t cj0 = 0, pk0 = 0;       // chunk bases
    int32_t c_have = 0, p_have = 0; // entries loaded in each chunk
    float acc = 0.f;
    bool found = false;
    bool need_c = true, need_p = np > 0;
    while (j < nc && !found) {
      if (need_c) {
        WaveSync();
        cj0 = j;
        c_have = min(kN2vChunk, nc - cj0);
        for (int32_t t = lane; t < c_have; t += 64) {
          const int32_t ph = N2vPhys(S.child, cj0 + t);
          S.c_id[t] = c_nbr[ph];
          S.c_w[t] = N2vWeightAt(S.child, ph);
        }
        need_c = false;
      }
      if (need_p) {
        WaveSync();
        pk0 = k;
        p_have = min(kN2vChunk, np - pk0);
        for (int32_t t = lane; t < p_have; t += 64)
          S.p_id[t] = p_nbr[N2vPhys(S.parent, pk0 + t)];
        need_p = false;
      }
      WaveSync();
      if (lane == 0) {
        const int32_t c_end = cj0 + c_have;
        const int32_t p_end = pk0 + p_have;
        while (j < c_end) {
          const int64_t cid = (int64_t)S.c_id[j - cj0];
          float w = S.c_w[j - cj0];
          if (k < np) {
            if (k >= p_end) break;               // next parent chunk
            const int64_t pid = (int64_t)S.p_id[k - pk0];
            if (cid > pid) { ++k; continue; }    // parent cursor only
            if (cid == pid) ++k;                 // common neighbour: weight kept
            else w = cid != parent ? __fdiv_rn(w, a.q) : __fdiv_rn(w, a.p);
          } else {
            w = cid != parent ? __fdiv_rn(w, a.q) : __fdiv_rn(w, a.p);
          }
          const float prev = acc;
          acc = __fadd_rn(acc, w);
          last_id = (uint64_t)cid;
          ++j;
          if (pass == 1 && (double)prev <= r && r < (double)acc) { found = true; break; }
        }
      }
      j = __shfl(j, 0);
      k = __shfl(k, 0);
      found = __shfl((int)found, 0) != 0;
      need_c = j >= cj0 + c_have;
      need_p = k < np && k >= pk0 + p_have;
    }
    if (pass == 0) {
      total = __shfl(acc, 0);
      const double u = RngDraw(a.seed, a.call_id + (uint32_t)step, kDomainWalk,
                               (uint64_t)walker, 0);
      r = ScaleDraw(u, 0.f, total);
    }
  }
  // found: last_id is the hit; not found (total == 0): RandomSelect's
  // fall-through ends on the last element, which is last_id as well
  const uint32_t lo32 = __shfl((uint32_t)last_id, 0);
  const uint32_t hi32 = __shfl((uint32_t)(last_id >> 32), 0);
  return (int64_t)(((uint64_t)hi32 << 32) | lo32);
}

// 64 registers (8 waves per SIMD) and 12 KB of LDS per workgroup: 18.8 ms against
// 20.4 with 82 registers on the hub workload.
template <bool PAR>
__global__ __launch_bounds__(256, kWavesPerSimd) void Node2VecWaveKernel(const WalkArgs a) {
  __shared__ N2vLds lds_all[4];
  N2vLds& S = lds_all[threadIdx.x >> 6];
  const int lane = threadIdx.x & 63;
  const int64_t waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const int64_t L = a.walk_len + 1;
  const int32_t s_end = a.step_end > 0 ? a.step_end : a.walk_len;
  // a.walk_ticket != NULL: the walkers are handed out by ticket (a walk's cost varies with the rows it
  // crosses far more than a step's) instead of every `waves`-th walker to a wave
  int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  for (;; i += waves) {
    if (a.walk_ticket != nullptr) {
      unsigned long long t = 0;
      if (lane == 0) t = atomicAdd(a.walk_ticket, 1ull);
      i = (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(t >> 32)) << 32) |
                    (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)t));
    }
    if (i >= a.n) break;
    int64_t cur, parent;
    bool have_parent_nb;
    if (a.step_begin == 0) {
      cur = a.nodes[i];
      parent = cur;                // parent_ids_ starts as the start nodes
      have_parent_nb = false;      // parent_neighbors_ starts empty
      if (lane == 0) a.out[i * L] = cur;
    } else {
      cur = a.out[i * L + a.step_begin];
      parent = a.out[i * L + a.step_begin - 1];
      have_parent_nb = true;
    }
    for (int32_t s = a.step_begin; s < s_end; ++s) {
      const int32_t* et = a.edge_types + s * a.k;
      const int32_t* pet = s > 0 ? a.edge_types + (s - 1) * a.k : et;
      WaveSync();
      if (lane == 0) {
        N2vBuildList(&S.child, a.g, FindRow(a.g, (uint64_t)cur), et, a.k);
        N2vBuildList(&S.parent, a.g,
                     have_parent_nb ? FindRow(a.g, (uint64_t)parent) : -1, pet, a.k);
      }
      WaveSync();
      const int32_t nc = S.child.total;
      if (a.big_threshold > 0 && nc >= a.big_threshold) break;   // N2vBigStepKernel's
      int64_t sample_id = a.default_node;
      bool done = false;
      if (PAR && nc > 0) done = N2vStepParallel(a, S, lane, parent, i, s, &sample_id);
      if (lane == 0 && nc > 0) { N2vCount(done ? 0 : 2, 1); N2vCount(done ? 1 : 3, (unsigned long long)nc); }
      if (nc > 0 && !done) sample_id = N2vStepSequential(a, S, lane, parent, i, s);
      if (lane == 0) a.out[i * L + s + 1] = sample_id;
      parent = cur;
      have_parent_nb = true;
      cur = sample_id;
    }
  }
}

// ------------------------------------------------------------------------
// node2vec step by step (tuning key 7 = 3).  A walker's step is one wave's serial
// work and the metric graph has rows of 5e5 neighbours: one walker in 10^5 spends
// 12 ms on its ten steps while the rest of the chip has long finished
// (tools/prof_n2v.py: 1 000 walkers take 12.6 ms, 100 000 take 37 ms).  So the walk
// is launched per step, and a step whose child list is long goes to a WORKGROUP of
// 16 waves: 4 096 entries per round, the parent cursor and the running sum carried
// across the waves through LDS.
//   * Parent cursor: all lanes compare with the same pn[k]; the first entry that is
//     not below it is found with one ballot per wave and one LDS exchange, the
//     cursor scan runs 1 024 parent entries per round.
//   * Running sums: every wave sums its 256 entries in the binade of the round's
//     carry (WaveSumsVec's integer path), the 16 totals are exchanged and each wave
//     adds what lies before it.  The waves before the first one that cannot do that
//     (a tie, a negative entry, the sum leaving the binade inside it) are final;
//     that wave runs its entries from its real carry and the rest start over from
//     its last sum.
// ------------------------------------------------------------------------
constexpr int kN2vBigWaves = 16;
constexpr int kN2vBigCk = 1024;
constexpr int kN2vBigRound = kN2vBigWaves * kN2vChunkR;

struct alignas(16) N2vBigLds {
  N2vLds seq;                                   // lists + staging of the sequential automaton
  unsigned long long x_mask[2][kN2vBigWaves];   // exchange slots, alternating
  int64_t x_val[2][kN2vBigWaves];
  float hand;                                   // the restarting wave's last sum
  int64_t next;                                 // queue entry of this workgroup
  float ck_acc[kN2vBigCk];
  int32_t ck_k[kN2vBigCk];
};

// Every wave contributes a lane mask and the value of its first set lane; returns the
// workgroup-wide index (wave * 64 + lane) of the first set lane, -1 if none, and that
// lane's value.  One barrier; consecutive calls use alternate slots, so a wave that
// runs ahead writes the slots nobody reads any more.
__device__ __forceinline__ int32_t N2vBigFirst(N2vBigLds& S, int* phase, int wv, int lane,
                                               unsigned long long mask, int64_t v,
                                               int64_t* v_out) {
  const int b = *phase & 1;
  ++*phase;
  if (lane == 0) S.x_mask[b][wv] = mask;
  if (mask != 0 && lane == __ffsll((long long)mask) - 1) S.x_val[b][wv] = v;
  __syncthreads();
  const unsigned long long mine = lane < kN2vBigWaves ? S.x_mask[b][lane] : 0ull;
  const unsigned long long nz = __ballot(mine != 0);
  if (nz == 0) return -1;
  const int w = __ffsll((long long)nz) - 1;
  const unsigned long long m = S.x_mask[b][w];
  *v_out = S.x_val[b][w];
  return w * 64 + __ffsll((long long)m) - 1;
}

struct N2vBigState {
  N2vCursor cur;
  float acc;        // running sum before this round
};

// One round: d[] = this lane's weights after BuildWeights' comparisons, *cin = the
// running sum before this lane's first entry.  Workgroup-uniform control flow.
__device__ __forceinline__ void N2vBigRound(const WalkArgs& a, N2vBigLds& S, int* phase, int wv,
                                            int lane, int64_t parent, int32_t np,
                                            bool same_lists, const N2vVec& e, N2vBigState* st,
                                            float (&d)[kN2vR], float* cin_out,
                                            int32_t* events_out) {
  const int tid = wv * 64 + lane;
  const uint64_t* p_nbr = S.seq.parent.ids;
  uint32_t keep = same_lists ? e.live : 0u;
  int res_tid = -1, res_r = -1;
  int32_t events = 0;
  int32_t k = st->cur.k;
  while (!same_lists && k < np) {
    if (st->cur.m_k != k) {
      st->cur.M = (int64_t)p_nbr[N2vPhys(S.seq.parent, k)];
      st->cur.m_k = k;
    }
    uint32_t evm = 0;
#pragma unroll
    for (int r = 0; r < kN2vR; ++r)
      if (((e.live >> r) & 1u) && (tid > res_tid || (tid == res_tid && r > res_r)) &&
          e.cid[r] >= st->cur.M)
        evm |= 1u << r;
    const int rsel = evm != 0 ? __ffs((int)evm) - 1 : 0;
    int64_t cf = 0;
    const int32_t f = N2vBigFirst(S, phase, wv, lane, __ballot(evm != 0), N2vPick(e, rsel), &cf);
    if (f < 0) break;                          // every remaining child is below pn[k]
    // first k' >= k with pn[k'] >= cf (the cursor skips the smaller entries)
    // (two loads of 1 024 entries in flight per trip measured the same: 42.6 against 42.7 ms)
    bool hit = false;
    for (;;) {
      const int32_t kk = k + tid;
      int64_t pv = 0;
      if (kk < np) pv = (int64_t)p_nbr[N2vPhys(S.seq.parent, kk)];
      int64_t mv = 0;
      const int32_t g = N2vBigFirst(S, phase, wv, lane, __ballot(kk < np && pv >= cf), pv, &mv);
      if (g >= 0) { k += g; st->cur.M = mv; st->cur.m_k = k; hit = true; break; }
      k += 64 * kN2vBigWaves;
      if (k >= np) { k = np; break; }
    }
    if (hit && st->cur.M == cf) { if (tid == f) keep |= 1u << rsel; ++k; }
    res_tid = f;
    if (tid == f) res_r = rsel;
    ++events;
  }
  st->cur.k = k;
  *events_out = events;
  if (tid == 0 && events > 0) N2vCount(4, (unsigned long long)events);
  N2vWeights(a, e, keep, parent, d);
  // ---- running sums
  float carry = st->acc;
  int w0 = 0;
  float cin = 0.f;
  for (;;) {
    const uint32_t cb = __float_as_uint(carry);
    const uint32_t ex = cb >> 23;
    const uint32_t bb = cb & 0xFF800000u;
    const float B = __uint_as_float(bb);
    const float twoB = __fadd_rn(B, B);
    const float half_ulp = __uint_as_float(bb - (24u << 23));
    const bool range_ok = ex >= 30u && ex < 254u;
    const bool active = wv >= w0;
    uint32_t N = 0;
    bool okl = true;
#pragma unroll
    for (int r = 0; r < kN2vR; ++r) {
      const float t = __fadd_rn(B, d[r]);
      const float err = __fsub_rn(d[r], __fsub_rn(t, B));
      const bool ok = d[r] >= 0.f && fabsf(err) != half_ulp && t < twoB;
      okl = okl && ok;
      N += ok ? __float_as_uint(t) - bb : 0u;
    }
    if (!active) N = 0;
    const uint32_t incl = WaveInclusiveAdd(N, lane);                // < 2^31
    const bool bad = active && (!range_ok || __ballot(!okl) != 0);
    const int b = *phase & 1;
    ++*phase;
    if (lane == 63) S.x_val[b][wv] = ((int64_t)(bad ? 1 : 0) << 32) | incl;
    __syncthreads();
    const int64_t mine = lane < kN2vBigWaves ? S.x_val[b][lane] : 0;
    int64_t pre = (int64_t)(uint32_t)mine;                          // inclusive over the waves
#pragma unroll
    for (int dd = 1; dd < kN2vBigWaves; dd <<= 1) {
      const int64_t up = __shfl_up(pre, dd);
      if (lane >= dd) pre += up;
    }
    const int64_t off0 = (int64_t)(cb - bb);
    const unsigned long long probw =
        __ballot(lane < kN2vBigWaves && lane >= w0 &&
                 ((mine >> 32) != 0 || off0 + pre >= (1 << 23)));
    const int pw = probw != 0 ? __ffsll((long long)probw) - 1 : kN2vBigWaves;
    const int64_t my_before = (wv == 0 ? 0 : __shfl(pre, wv - 1)) + off0;
    if (active && wv < pw) cin = __uint_as_float(bb + (uint32_t)(my_before + incl - N));
    if (pw == kN2vBigWaves) {
      st->acc = __uint_as_float(bb + (uint32_t)(__shfl(pre, kN2vBigWaves - 1) + off0));
      break;
    }
    const int64_t pw_before = (pw == 0 ? 0 : __shfl(pre, pw - 1)) + off0;
    if (wv == pw) {
      const float before = pw == w0 ? carry : __uint_as_float(bb + (uint32_t)pw_before);
      const float last = WaveSumsVec(before, d, lane, &cin);
      if (lane == 0) S.hand = last;
    }
    __syncthreads();
    carry = S.hand;
    w0 = pw + 1;
    if (w0 == kN2vBigWaves) { st->acc = carry; break; }
  }
  *cin_out = cin;
}

// One walker's step by the whole workgroup, the lists already in S.seq (N2vBuildList from the
// graph's rows, or N2vBuildListFetched from rows fetched from their owners): the sample, on
// every thread.  `s`: the step (the draw's call id is a.call_id + s; explicit lists pass 0).
__device__ __forceinline__ int64_t N2vBigStepBody(const WalkArgs& a, N2vBigLds& S, int* phase_p, const int wv,
                                                   const int lane, const int64_t parent, const int64_t i,
                                                   const int32_t s, const int same_in = -1) {
  int& phase_ref = *phase_p;
  const int32_t nc = S.seq.child.total, np = S.seq.parent.total;
  const int32_t rounds = (nc + kN2vBigRound - 1) / kN2vBigRound;
  int32_t sh = 0;
  while ((rounds >> sh) > kN2vBigCk) ++sh;
  const int32_t n_slots = rounds >> sh;
  N2vBigState st{{0, -1, 0}, 0.f};
  const bool same = same_in >= 0 ? same_in != 0 : N2vSameLists(S.seq.child, S.seq.parent);
  const int32_t jl = threadIdx.x * kN2vR;
  float d[kN2vR], cin;
  int32_t events;
  bool ascending = false;
  // the next round's entries are requested before this round is worked on
  N2vVec e = N2vLoadVec(a, S.seq.child, nc, jl), nx = e;
  for (int32_t ri = 0; ri < rounds; ++ri) {
    if (ri + 1 < rounds) nx = N2vLoadVec(a, S.seq.child, nc, (ri + 1) * kN2vBigRound + jl);
    N2vBigRound(a, S, &phase_ref, wv, lane, parent, np, same, e, &st, d, &cin, &events);
    e = nx;
    if (ri == 0 && events > 64) { ascending = true; break; }
    if (((ri + 1) & ((1 << sh) - 1)) == 0 && threadIdx.x == 0) {
      S.ck_acc[((ri + 1) >> sh) - 1] = st.acc;
      S.ck_k[((ri + 1) >> sh) - 1] = st.cur.k;
    }
  }
  int64_t result = a.default_node;
  if (threadIdx.x == 0) {
    N2vCount(ascending ? 2 : 6, 1);
    N2vCount(ascending ? 3 : 7, (unsigned long long)nc);
  }
  if (ascending) {
    // every child moves the parent cursor: the lane-0 automaton of one wave does it
    if (wv == 0) result = N2vStepSequential(a, S.seq, lane, parent, i, s);
  } else {
    const float total = st.acc;
    const double u = RngDraw(a.seed, a.call_id + (uint32_t)s, kDomainWalk, (uint64_t)i, 0);
    const double r = ScaleDraw(u, 0.f, total);
    __syncthreads();
    int32_t first = 0;
    if (N2vMonotone(a) && a.p > 0.f && a.q > 0.f) {
      first = n_slots;
      for (int32_t base = 0; base < n_slots; base += 64) {
        const int32_t idx = base + lane;
        const unsigned long long gt = __ballot(idx < n_slots && (double)S.ck_acc[idx] > r);
        if (gt != 0) { first = base + __ffsll((long long)gt) - 1; break; }
      }
    }
    st.acc = first == 0 ? 0.f : S.ck_acc[first - 1];
    st.cur.k = first == 0 ? 0 : S.ck_k[first - 1];
    st.cur.m_k = -1;
    bool found = false;
    for (int32_t ri = first << sh; ri < rounds && !found; ++ri) {
      e = N2vLoadVec(a, S.seq.child, nc, ri * kN2vBigRound + jl);
      N2vBigRound(a, S, &phase_ref, wv, lane, parent, np, same, e, &st, d, &cin, &events);
      const uint32_t hm = N2vHits(e, d, cin, r);
      int64_t hv = 0;
      if (N2vBigFirst(S, &phase_ref, wv, lane, __ballot(hm != 0),
                      N2vPick(e, hm != 0 ? __ffs((int)hm) - 1 : 0), &hv) >= 0) {
        found = true;
        result = hv;
      }
    }
    // no interval holds r (total == 0): RandomSelect's fall-through ends on the last element
    if (!found) result = (int64_t)S.seq.child.ids[N2vPhys(S.seq.child, nc - 1)];
  }
  return result;
}

// Lane per walker: queue the walkers whose step `s` has a long child list.
__global__ __launch_bounds__(256) void N2vClassifyKernel(const WalkArgs a) {
  __shared__ int32_t base;
  __shared__ int32_t wave_cnt[4];
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int32_t s = a.step_begin;
  const int64_t L = a.walk_len + 1;
  bool big = false;
  if (i < a.n) {
    const int64_t cur = s == 0 ? a.nodes[i] : a.out[i * L + s];
    N2vList l;
    N2vBuildList(&l, a.g, FindRow(a.g, (uint64_t)cur), a.edge_types + s * a.k, a.k);
    big = l.total >= a.big_threshold;
  }
  const unsigned long long m = __ballot(big);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) wave_cnt[wv] = __popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) {
    const int32_t tot = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
    base = tot > 0 ? atomicAdd(a.big_count, tot) : 0;
  }
  __syncthreads();
  if (big) {
    int32_t off = base;
    for (int w = 0; w < wv; ++w) off += wave_cnt[w];
    off += __popcll(m & ((1ull << lane) - 1));
    a.big_queue[off] = (int32_t)i;
  }
}

__global__ __launch_bounds__(64 * kN2vBigWaves) void N2vBigStepKernel(const WalkArgs a) {
  __shared__ N2vBigLds S;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int32_t s = a.step_begin;
  const int64_t L = a.walk_len + 1;
  const int32_t queued = a.big_count[0];
  int phase = 0;
  for (;;) {
    __syncthreads();
    if (threadIdx.x == 0) S.next = atomicAdd(a.big_count + 1, 1);
    __syncthreads();
    const int64_t qe = S.next;
    if (qe >= queued) break;
    const int64_t i = a.big_queue[qe];
    const int64_t cur = s == 0 ? a.nodes[i] : a.out[i * L + s];
    const int64_t parent = s == 0 ? cur : a.out[i * L + s - 1];
    if (threadIdx.x == 0) {
      const int32_t* et = a.edge_types + s * a.k;
      const int32_t* pet = s > 0 ? a.edge_types + (s - 1) * a.k : et;
      N2vBuildList(&S.seq.child, a.g, FindRow(a.g, (uint64_t)cur), et, a.k);
      N2vBuildList(&S.seq.parent, a.g, s > 0 ? FindRow(a.g, (uint64_t)parent) : -1, pet, a.k);
    }
    __syncthreads();
    const int64_t result = N2vBigStepBody(a, S, &phase, wv, lane, parent, i, s);
    if (threadIdx.x == 0) a.out[i * L + s + 1] = result;
  }
}

#endif  // EULER_AMD_CSRC_N2V_KERNELS_H_
