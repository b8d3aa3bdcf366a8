/*
 * euler_gpu_measure.h - measurement and A/B surface of libeuler_gpu.so: tuning keys,
 * HIP-event timers around the product's own launches, algorithmic byte counters.
 *
 * Nothing here belongs to the drop-in boundary (include/euler_gpu.h): a host that replaces
 * the reference's sampler never needs this header.  bench.py, tools/ and the tests use it.
 * Every tuning setting produces identical results.
 */
#ifndef EULER_GPU_MEASURE_H_
#define EULER_GPU_MEASURE_H_

#include "euler_gpu.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- tuning ---------------------------------------------------------------
 * (keys 1, 6, 19, 21, 22, 26 belonged to kernels retired in round 3: EINVAL)
 * key 0: sample_neighbor kernel for single-type calls on graphs with non-decreasing
 *        running sums: 6 = block pivots over EdgeBlocks [default], 5 = pivot levels over
 *        the flat arrays, 0 = always the generic reference-loop kernel.
 * key 2: measurement only (ablation mask).
 * key 3: workgroup cap of the sample_neighbor launches: -1 = by concurrency [default]: a
 *        caller that alternates streams between calls (several minibatches in flight)
 *        gets 4096 = 16 waves per CU, so that the kernels of two streams fit on the
 *        chip together, every other call 32 768; 0 = always 32 768; > 0 = that many.
 * key 4: the pivot kernels draw two adjacent samples per lane when count is even (1).
 * key 5: duplicate roots: 0 = never look, 1 = look when a call has >= 200 000
 *        roots [default], 2 = always look.
 * key 7: node2vec kernel: 2 = one wave per walker, runs of children below the
 *        parent cursor resolved by all lanes at once, running sums as integer sums
 *        inside a binade [default]; 3 = the same launched per step, child lists of
 *        key 25 entries or more (default 8192, 0 = none) by a 16-wave workgroup;
 *        1 = one wave per walker, lane 0 walks LDS-staged lists; 0 = one lane per
 *        walker.
 * key 8: dense-feature kernel: 16-byte loads when the slots allow it (1).
 * key 9: fanout, hop by hop: a hop's kernels enter their ids into the next hop's owner
 *        table, 1 [default]; 0 = every hop runs its own mark pass.
 * key 10: expansion of the distinct roots' rows: grid-stride steps in flight per
 *        lane (1, 2 [default], 4).  key 11: rebuild the type column of
 *        single-type calls from the row mask instead of gathering it (1).
 *        key 12: measurement only (workgroup cap of the expansion).
 * key 13: both gated passes of a duplicate-root call in one launch (1).
 * key 14: duplicate roots, numbering of the distinct ones: 2 = ONE pass, every
 *        workgroup takes its numbers from the call's counter with one atomic and the
 *        representative leaves its number in the owner table [default]; 1 =
 *        per-workgroup counts + one small scan + assign + resolve; 0 = device-wide
 *        scan over the positions with the flags evaluated in its loads.
 * key 15: euler_gpu_sample_root: who builds the alias tables (one sequential
 *        chain per batch row): 0 = one lane per row on the device, 2 = the host's
 *        cores between two copies, 1 = by a measured cost model [default] (few
 *        long rows go to the host); the draws always run on the device.
 * key 16: SparseGetAdj: 0 = candidates in an LDS hash table, sources stream
 *        their rows once [default]; 1 = every candidate compared with the row.
 * key 17: SparseGetAdj: sources with more listed edges than this (default
 *        16384) are cut into segments handled by separate workgroups.
 * key 18: edge weight sums of long rows: 0 = lane-shifting DPP chain [default],
 *        1 = scalar loads and a wave-uniform chain (measured 2x slower).
 * key 20: last hop of a fanout with key 14 = 2: 1 = the expansion reads every
 *        position's row number from the owner table itself; 0 = a separate resolve
 *        kernel fills an index array first [default: measured 9 us faster].
 * key 23: a 2-hop fanout of single listed types below key 33's batch size
 *        (the reference examples' batch of 1 024) runs as ONE launch: a workgroup
 *        draws a root's first-hop samples and, from LDS, their second-hop samples
 *        (1 [default]); 0 = one launch per hop.
 * key 24: get_full_neighbor fill pass: a lane owns 4 consecutive output entries,
 *        whatever rows they belong to (1 [default]); 0 = one wave per queried node.
 * key 25: node2vec with key 7 = 3: child lists of this many entries or more go to the
 *        workgroup kernel (default 8192; 0 = none).
 * key 27: the one-kernel fanout (csrc/fanout_local.h: 2 hops, one listed type each - or,
 *        on graphs with the weight-bucket index and at most 127 edge types, or of uniform
 *        weights and at most 4, the same number of several listed types each: a type draw
 *        per sample -, >= key 33 roots): 1 = on weighted graphs [default], 2 = on every graph,
 *        0 = off (hop by hop).  Its geometry: key 28 roots per wave (1..16; 0 [default] =
 *        the launcher chooses: 4, or 8 for a caller that alternates streams on a graph with
 *        the weight-bucket index), key 29 distinct children sampled per pass (0 [default] =
 *        64 with the weight-bucket index - 48 under alternating streams -, else 8 per root),
 *        key 30 threads per workgroup (64, 128, 256; 0 [default] = 128 with the index and one
 *        caller stream, else 64), key 31 weights / types as 16-byte
 *        stores (1), key 32 cap on launched waves (-1 = 16 384 when the caller
 *        alternates streams, else one tile per wave [default]; 0 = never; > 0 = that
 *        many), key 33 smallest batch it takes (32768), key 34 plain graphs: 2 = the
 *        lean build [default], 1 = the general build constant-folded, 0 = general,
 *        key 35 register budget in waves per SIMD (5 [default: nothing spilled], 6, 8),
 *        key 36 ablation bits - exists only in a library built with `make MEASURE=1`
 *        (-DEULER_GPU_MEASURE); the shipped library answers EINVAL, as for key 2.
 * key 37: calls that draw the edge type (k != 1) on monotone graphs search the
 *        neighbour with the block pivots (1 [default]); 0 = the reference loop.
 * key 38: DeepWalk (p = q = 1) of at least this many walkers runs over groups of merged
 *        walkers (default 262144 - below, one lane per walker is faster: 131 072 walkers x 40 steps
 *        0.45 vs 0.64 ms, 262 144: 0.84 vs 0.77 -; 0 = never).  key 39: workgroups of its per-step
 *        launches (1024; 0 = one per 256 walkers).  key 43: first step from which the
 *        groups stop looking for mergers and finish the walk in one launch (9; 0 =
 *        never).  key 44: plain graphs draw with the lean search of the one-kernel
 *        fanout (1 [default]).
 * key 45: 1 [default] = searches go through the weight-bucket index (csrc/wb_index.h: the
 *        bucket of a draw in its row's running-sum range names ONE 128-byte line; built on
 *        first use for graphs with non-decreasing, non-uniform running sums and < 2^32
 *        edges, ~40 bytes per edge of HBM); 0 = the pivot-level search of rounds 2-3.  (A graph
 *        whose weights are so uneven that more than 2 buckets in a thousand overflow their
 *        block keeps the pivot levels for the one-kernel fanout and the merged walk by itself.)
 * key 47: euler_gpu_sample_neighbor_sets stages the roots' records in LDS once per workgroup
 *        (1 [default]; 2 = the same, built for 6 waves per SIMD instead of 8: no register
 *        spills, measured equal); 0 = every sample lane reads them.
 * key 48: typed hops of the one-kernel fanout on graphs with at most 4 edge-type groups keep
 *        the row record in registers (1 [default]); 0 = walk it in memory, as graphs with more
 *        groups do.
 * key 49: graphs with a hash id map and at most two edge-type groups: the general builds of the
 *        one-kernel fanout find a root's record in its 64-byte hash slot (1 [default]: one cold
 *        line per root / child); 0 = the 16-byte slot, then the row's record (two).
 * key 51: 1 = a graph that has a weight-bucket index is served by it alone even when more than
 *        2 of its buckets in a thousand overflow: no EdgeBlocks are built for it and every draw
 *        its block does not bracket bisects the flat running sums (the fallback of graphs the
 *        index serves, made frequent for tests); 0 [default] = such graphs get the EdgeBlocks
 *        and their pivot levels.  Takes effect for graphs whose EdgeBlocks are not built yet.
 * key 52: 1 = a lone rank of the sharded entry points (euler_gpu_sharded_*) makes the id and row
 *        exchanges anyway - a send to itself through the transport - instead of skipping them
 *        (0 [default]; the environment variable EULER_GPU_SELF_EXCHANGE=1 sets the default): the
 *        tests' way to execute the RCCL transport on a box with one GPU.  PROCESS-WIDE (the
 *        sharded calls of every host thread see it), unlike the other keys.
 * key 53: plain graphs served by the weight-bucket index take the 2-hop fanout kernel of
 *        csrc/fanout_plain.h when the caller keeps ONE stream (1 [default]; 2 = also when it
 *        alternates streams); 0 = round 5's lean build (fanout_local.h).
 * key 54: ... with a block's keys fetched by three lanes through LDS-DMA and staged in LDS (1);
 *        0 [default] = by the lane that owns the draw.  key 55: its register budget in waves per
 *        SIMD (4 .. 8, default 5).  key 57: its hop 2 asks for two of a block's three key chunks
 *        and for the third only when the pick lands on the block's first edge or past its
 *        eighth (1); 0 [default] = all three at once.
 * key 56: test hook, PROCESS-WIDE: the next `value` builds of the EdgeBlocks fail the way an
 *        allocation failure would (the graph is then served without them: same results).
 * key 60: PROCESS-WIDE: a Sage flow whose hops list one edge type runs as three launches per hop -
 *        sampler + insert in one kernel, flag, emit + index, the hash tables cleared by the kernels
 *        before them (1 [default]); 0 = sampler, clear, insert as separate launches.
 * key 62: PROCESS-WIDE: fused Sage flows send the ids WITH a graph row through the hop's own
 *        {row, smallest position} hash table instead of the stream's row-indexed table: 2 [default] =
 *        flows of at most 32 768 roots (16 384 roots x [25, 10]: 0.117 -> 0.111 ms; at 131 072 roots
 *        the hop's table is itself 128 MB and the flow 3 % slower), 1 = always, 0 = never.
 * key 63: PROCESS-WIDE: euler_gpu_sharded_random_walk ENQUEUED - levels and buckets in slab layout,
 *        sizes on the device, fixed-size messages, one host exchange per call (1 [default]); 0 = the
 *        polled form, one host wait per step and cohort.  key 66: first step of the enqueued walk
 *        whose level is sent as it is, without looking for entries that share a node (16; 0 =
 *        every step deduplicates).  key 64: columns its path kernel parks in LDS at a time (16).
 *        key 67: level T at which its path writer splits into two passes - the levels from T on
 *        are walked once per entry of level T, the walkers walk the levels before it and append
 *        the row of the entry they reach (10; walks of at least T + 8 steps; 0 = one pass).
 * key 69: PROCESS-WIDE: euler_gpu_node2vec_step (the sharded node2vec walk's step on fetched rows):
 *        child rows of at least this many entries are drawn by a workgroup of 16 waves each
 *        (65536; 0 = every row by one wave).  key 71: ... and so are walkers whose PARENT's row
 *        has at least this many entries (65536; 0 = by the child row alone: a wave moves the
 *        parent cursor 64 entries a dependent step).  Its waves take the walkers by ticket, in index order.  key 72: 1 (default) = the long rows' workgroups and the
 *        other walkers' waves are ONE launch (a workgroup's 16 waves go on as single waves when the
 *        long rows are done), 0 = two launches.  key 73: PROCESS-WIDE: 1 (default) = the node2vec
 *        walk of more than 16 384 walkers (one launch, a wave per walker) hands the walkers out by
 *        ticket, 0 = every 16 384th walker to a wave (100 000 x 10 on the metric graph: 17.2 / 19.1 ms).
 * All settings produce identical results; the knobs exist for A/B measurements
 * and tests.  They are THREAD-LOCAL: a call changes the launches the calling host
 * thread enqueues afterwards and nobody else's (new threads start from the
 * defaults), so concurrent query threads cannot disturb one another. */
int euler_gpu_set_tuning(int32_t key, int32_t value);
/* Measurement builds only (`make MEASURE=1`; the shipped library returns EINVAL and its
 * kernels carry no such code): a device buffer of 8 x uint64 per wave tile in which the
 * one-kernel fanout (fanout_local.h, lean build) leaves its phase time stamps; NULL = off.
 * Thread-local like the tuning keys. */
int euler_gpu_set_debug_buffer(void* dev);

/* The kernel the calling thread's last euler_gpu_sample_fanout* call of a 2-hop fanout launched
 * ("SampleFanoutPlainKernel", "SampleFanoutLeanKernel", "SampleFanoutLocalKernel", "hop by hop";
 * "" before any): what bench.py names in its roofline object. */
const char* euler_gpu_last_fanout_kernel(void);

/* Rows of the graph (their node ids, identity id maps only) in which some bucket of the
 * weight-bucket index overflows its block - a draw that lands there takes the fallback search
 * (csrc/wb_index.h) - as found while the index was built: at most 4 096 of them.  Builds the
 * index if it is not built yet.  Tests draw roots from these rows on purpose. */
int euler_gpu_graph_index_overflow_rows(const euler_gpu_graph* g, uint64_t* ids_host, int64_t cap,
                                        int64_t* n_host);

/* ---- measurement helper -------------------------------------------------------
 * Runs the sample_neighbor kernel `iters` times on `stream` between two HIP
 * events recorded on that same stream and returns the mean kernel time in
 * milliseconds (bench.py's roofline leg). */
int euler_gpu_time_sample_neighbor(const euler_gpu_graph* g, void* stream,
                                   uint64_t seed, const uint64_t* roots_dev,
                                   int64_t n, const int32_t* edge_types_host,
                                   int32_t k, int32_t count, int32_t layout,
                                   uint64_t* out_id_dev, float* out_w_dev,
                                   int32_t* out_t_dev, int32_t iters,
                                   float* mean_ms_host);
/* Same, split by phase: mean_ms3_host[0] duplicate detection, [1] sampling
 * kernel(s), [2] expansion of the unique rows (0 when the call did not take
 * the unique path).  dedup: 0 = sample the given roots directly, 1 = the
 * launcher's automatic policy.  *n_unique_host (optional) receives the number
 * of distinct roots the last launch counted (-1 if it did not count). */
int euler_gpu_time_sample_neighbor_phases(const euler_gpu_graph* g, void* stream,
                                          uint64_t seed, const uint64_t* roots_dev,
                                          int64_t n, const int32_t* edge_types_host,
                                          int32_t k, int32_t count, int32_t layout,
                                          int32_t dedup, uint64_t* out_id_dev,
                                          float* out_w_dev, int32_t* out_t_dev,
                                          int32_t iters, float* mean_ms3_host,
                                          int64_t* n_unique_host);
/* euler_gpu_sample_fanout's arguments, the call repeated `iters` times between two HIP
 * events recorded on `stream` (after one untimed call): the mean time of one call in
 * milliseconds.  What bench.py quotes for the one-kernel 2-hop fanout. */
int euler_gpu_time_sample_fanout(const euler_gpu_graph* g, void* stream, uint64_t seed,
                                 const uint64_t* roots_dev, int64_t n,
                                 const int32_t* edge_types_host, int32_t k,
                                 const int32_t* counts_host, int32_t layers,
                                 int64_t default_node, uint64_t* const* out_id_dev,
                                 float* const* out_w_dev, int32_t* const* out_t_dev,
                                 void* workspace_dev, int32_t iters, float* mean_ms_host);
/* The whole fanout (euler_gpu_sample_fanout's arguments), timed in place:
 * mean_ms_host[3*h + {0,1,2}] = hop h's duplicate detection / sampling
 * kernel(s) / expansion, with the hop chaining the product uses (hop h's
 * kernels enter their ids into hop h+1's owner table).  n_unique_host
 * (optional, [layers]): distinct roots the LAST hop counted, -1 elsewhere. */
int euler_gpu_time_sample_fanout_phases(const euler_gpu_graph* g, void* stream,
                                        uint64_t seed, const uint64_t* roots_dev,
                                        int64_t n, const int32_t* edge_types_host,
                                        int32_t k, const int32_t* counts_host,
                                        int32_t layers, int64_t default_node,
                                        uint64_t* const* out_id_dev,
                                        float* const* out_w_dev,
                                        int32_t* const* out_t_dev, void* workspace_dev,
                                        int32_t iters, float* mean_ms_host,
                                        int64_t* n_unique_host);
/* Exact algorithmic byte count of one sample_neighbor launch (SURVEY §8d
 * formula evaluated on the actual roots' degrees); synchronises. */
int euler_gpu_sample_neighbor_algo_bytes(const euler_gpu_graph* g, void* stream,
                                         const uint64_t* roots_dev, int64_t n,
                                         const int32_t* edge_types_host,
                                         int32_t k, int32_t count,
                                         double* bytes_host);

/* Diagnostic counters of the node2vec kernels on the current device since the last
 * reset (synchronises the device): out8_host[0] steps done by the whole-wave path,
 * [1] their child-list entries, [2] steps handed to the sequential automaton, [3]
 * their entries, [4] moves of the parent cursor, [5] chunks whose running sums fell
 * back to the add chain, [6] steps done by the workgroup kernel, [7] their entries.
 * reset: 2 = clear the counters and count from now on (the kernels then pay an atomic
 * per counted event), 1 = clear and stop counting (the default state), 0 = read only.
 * out8_host may be NULL. */
int euler_gpu_random_walk_stats(uint64_t* out8_host, int32_t reset);

/* Exact algorithmic byte count of a finished walk (walks_dev = the
 * [n, walk_len + 1] output of euler_gpu_random_walk): p = q = 1 - per step the
 * SampleNeighbor(count = 1) terms of euler_gpu_sample_neighbor_algo_bytes over the
 * degree of the node the walker stands on; node2vec - (deg(cur) + deg(prev)) * 12
 * + 8 per step.  Synchronises. */
int euler_gpu_random_walk_algo_bytes(const euler_gpu_graph* g, void* stream,
                                     const int64_t* walks_dev, int64_t n,
                                     const int32_t* edge_types_host, int32_t k,
                                     int32_t walk_len, float p, float q,
                                     double* bytes_host);

#ifdef __cplusplus
}
#endif
#endif  /* EULER_GPU_MEASURE_H_ */
