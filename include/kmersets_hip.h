/*
 * kmersets_hip.h -- C ABI of the MI355X (gfx950) k-mer-set hot path.
 *
 * The reference (kkty/kmer-sets-compression) has no FFI: its hot path is a set of
 * header-only C++17 templates (the lib/core headers).  This header is the boundary the
 * build introduces underneath them: flat extern "C" functions over
 * plain device pointers and sizes, parameterised at run time by
 * (k, n_bucket_bits, key_bytes) instead of the reference's template parameters
 * <K, N, KeyType>.  The C++17 class templates in
 * kmer-sets-compression_amd/cpp/core/ (KmerSet / KmerSetCompact / KmerSetSet, same
 * names and argument meaning as the reference) are thin wrappers over these calls;
 * INTEGRATION.md shows the binding a reference maintainer would add.
 *
 * Device layout of a k-mer set (replaces 2^N absl::flat_hash_set<KeyType>,
 * lib/core/kmer_set.h:247-251):
 *     offsets : int64[2^N + 1]   bucket b holds keys[offsets[b] .. offsets[b+1])
 *     keys    : u16, u32 or u64  low (2K - N) bits of the k-mer, ascending inside a bucket;
 *                                key_bytes = 2 when 2K-N <= 16 (the reference's (15, 14, uint16_t)),
 *                                4 when 2K-N <= 32, else 8
 * i.e. the set is one ascending array of 2K-bit k-mers with a bucket index.
 *
 * All pointers named d_* are device pointers on the context's GPU.  Functions
 * that return host scalars synchronise the context's stream before returning;
 * the others only enqueue work.  There is no CPU fallback anywhere behind this
 * header: without a usable HIP device every call fails with KSH_INTERNAL.
 *
 * Status codes mirror the three absl codes the reference uses
 * (lib/core/io.h:26,43,63; lib/core/kmer_set_set.h:465,525,610).
 */
#ifndef KMERSETS_HIP_H_
#define KMERSETS_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KSH_OK 0
#define KSH_INVALID_ARGUMENT 3
#define KSH_FAILED_PRECONDITION 9
#define KSH_INTERNAL 13

typedef struct ksh_ctx ksh_ctx;

/* <K, N, KeyType> of the reference, as run-time values. */
typedef struct ksh_geom {
  int32_t k;             /* k-mer length, 2 <= k <= 31                         */
  int32_t n_bucket_bits; /* N: top N bits of the 2K-bit k-mer pick the bucket  */
  int32_t key_bytes;     /* device key width: 2, 4 or 8 (>= ceil((2k - N) / 8)) */
  int32_t reserved;
} ksh_geom;

/* One k-mer set resident in HBM.  d_keys is 16-byte aligned and its allocation runs to a
 * multiple of 16 bytes (ksh_malloc, hipMalloc and torch all round up further): the merge
 * kernels read whole 16-byte vectors, so the last vector of a key array may reach up to 12
 * bytes past its last key (never into another allocation's bytes they would use). */
typedef struct ksh_set_view {
  const int64_t* d_offsets; /* int64[2^N + 1]                */
  const void* d_keys;       /* key_bytes * n_keys bytes      */
  int64_t n_keys;
} ksh_set_view;

/* ---- library / errors --------------------------------------------------------- */
int ksh_version(void);
/* Message of the last failing call on this thread ("" if none). */
const char* ksh_last_error(void);

/* ---- device plumbing (so that host C++ needs no HIP headers) ---------------------- */
int ksh_device_count(int* count);
int ksh_malloc(int device, size_t bytes, void** d_ptr);
int ksh_free(int device, void* d_ptr);
/* The three copies wait for the whole device first (hipDeviceSynchronize): they are ordered after
 * everything enqueued on any context's stream, blocking or not, and done when they return. */
int ksh_memcpy_h2d(int device, void* d_dst, const void* src, size_t bytes);
int ksh_memcpy_d2h(int device, void* dst, const void* d_src, size_t bytes);
int ksh_memcpy_d2d(int device, void* d_dst, const void* d_src, size_t bytes);
/* The same copies ordered after ONE context's stream only (the work of other contexts goes on): for
 * a host thread with a context of its own and buffers it produced itself or received complete --
 * the per-node writer threads of KmerSetSet::Dump (lib/core/kmer_set_set.h:497-519), the rank threads
 * of a rehearsal. */
int ksh_ctx_memcpy_h2d(ksh_ctx* ctx, void* d_dst, const void* src, size_t bytes);
int ksh_ctx_memcpy_d2h(ksh_ctx* ctx, void* dst, const void* d_src, size_t bytes);
int ksh_ctx_memcpy_d2d(ksh_ctx* ctx, void* d_dst, const void* d_src, size_t bytes);

/* A context = one GPU + one HIP stream + a scratch arena.  `stream` may be an
 * existing hipStream_t (e.g. torch's current stream) or NULL for a new one.
 * Contexts are not thread-safe; use one per host thread (the reference calls its
 * set operations concurrently from pool threads, lib/core/kmer_set_set.h:146-149). */
int ksh_ctx_create(int device, void* stream, ksh_ctx** out);
int ksh_ctx_destroy(ksh_ctx* ctx);
int ksh_ctx_sync(ksh_ctx* ctx);
/* Pre-sizes the scratch arena (otherwise it grows on demand). */
int ksh_ctx_reserve(ksh_ctx* ctx, size_t bytes);
/* Kernel timing for bench.py's roofline leg.  enable = n > 0: every n-th launch of a
 * timed kernel kind (the 1st, the (n+1)-th, ...) is bracketed by its own pair of HIP
 * events on the context's stream; an event pair costs the stream about 10 us of idle time,
 * so a caller that also measures throughput samples (n > 1) rather than timing every
 * launch.  ksh_ctx_timing_read synchronises the stream and returns the summed duration and
 * the number of timed launches since the last reset.
 * kinds: 0 = pair merge, write pass   1 = pair merge, count pass
 *        2 = sampled-bucket weight count pass
 *        3 = SPSS encode, neighbour probe stage (partition + LDS-staged probes: the loop's dominant stage)
 *        4 = SPSS encode, chain ranking walks (k_rank_walk / k_rank_heads, or the stamping k_ruler_walk /
 *            k_ruler_heads)   5 = SPSS encode, base emit (k_emit_rulers / k_emit_heads, or k_emit)
 * ksh_ctx_timing_units: the k-mers (kinds 3..5) the timed launches of a kind covered. */
int ksh_ctx_enable_timing(ksh_ctx* ctx, int enable);
int ksh_ctx_timing_reset(ksh_ctx* ctx);
int ksh_ctx_timing_read(ksh_ctx* ctx, int kind, float* total_ms, int64_t* launches);
int ksh_ctx_timing_units(ksh_ctx* ctx, int kind, int64_t* units);
/* Lanes.  Calls that hold independent jobs -- the SPSS encodes of the stale nodes at a convergence check and at
 * the end of ksh_kss_build (lib/core/kmer_set_set.h:287,345-360), the decodes of its inputs (:138-153, which the
 * reference runs on its thread pool too) -- run them on up to n_lanes HIP streams of the context's GPU at once,
 * each from a host thread of its own with scratch of its own (one job's LDS-bound probe kernels then overlap
 * another's walks, which wait on HBM, and nobody's host round trips leave the GPU idle); results are the same,
 * node ids being fixed before the jobs start.  The lanes are helper contexts that live inside the context
 * (created on first use, mapped scratch kept, freed by ksh_ctx_destroy); every job runs on one of them and
 * allocates its result from the context's own pool, which only hands out while jobs run.  Lanes whose scratch
 * does not fit the free memory are left out; with fewer than two the jobs run on the context's stream, one
 * after the other.  n_lanes = 1: everything on the context's stream, as before; 0: the default (KSH_LANES, else 4).  The lanes run on two disjoint shares of the CUs, odd and even lanes (KSH_LANE_CUS=1: every lane on all CUs).
 * With lanes, ksh_ctx_timing_read / _units sum over the lanes (stream time: spans of different lanes overlap);
 * ksh_ctx_timing_wall is the length of the UNION of a kind's timed spans over all lanes, i.e. the wall time
 * during which at least one launch of the kind was running. */
int ksh_ctx_set_lanes(ksh_ctx* ctx, int n_lanes);
/* Device memory behind a context, bytes: stats = { pooled buffers handed out and not yet given back (the sets,
 * containers and samples of a KmerSetSet live here), the maximum of that since the context was made or the peak was
 * last reset, pooled buffers cached for reuse, the context's own scratch (encode / decode / text slots, arena, pair
 * plan), everything its helper lanes hold, number of helper lanes }.  reset_peak != 0: the peak restarts at the
 * current value.  (DESIGN.md 7.2's memory table is checked against these in tests/test_gpu_full_size.py.) */
int ksh_ctx_mem_stats(ksh_ctx* ctx, int64_t stats[6], int reset_peak);
int ksh_ctx_timing_wall(ksh_ctx* ctx, int kind, float* wall_ms);

/* ---- KmerSet::Size / Hash  (lib/core/kmer_set.h:65-71, :224-244) --------------------- */
/* XOR of all k-mer bit patterns in the set. */
int ksh_set_hash(ksh_ctx* ctx, const ksh_geom* g, const ksh_set_view* s, uint64_t* hash);

/* ---- KmerSet::Contains / Find  (lib/core/kmer_set.h:99-105, :116-161) ------------------------
 * contains: d_found[i] = 1 if the 2K-bit pattern d_kmers[i] is in the set, else 0 (batched: one
 * launch for n queries; enqueued on the context's stream).
 * kmers: every k-mer of the set as its full 2K-bit pattern, ascending, into d_kmers[n_keys] --
 * KmerSet::Find(n_workers); a Find with a host predicate filters this one download. */
int ksh_set_contains(ksh_ctx* ctx, const ksh_geom* g, const ksh_set_view* s, const uint64_t* d_kmers, int64_t n,
                     uint8_t* d_found);
int ksh_set_kmers(ksh_ctx* ctx, const ksh_geom* g, const ksh_set_view* s, uint64_t* d_kmers);

/* ---- ParallelDisjointSet  (lib/core/parallel_disjoint_set.h:15-111) ---------------------------
 * The wait-free union-find the path cover's loop detection runs on (spss.h:1541-1625), as an entry
 * of its own: n nodes, m pairs (d_x[i], d_y[i]) united concurrently (one thread per pair, 64-bit
 * compare-and-swap on rank << 32 | parent), then d_root[i] = Find(i).  WHICH node represents a
 * component depends on the interleaving, as in the reference with n_workers > 1; the partition
 * does not.  Synchronises the stream. */
int ksh_dsu_components(ksh_ctx* ctx, int64_t n, const int32_t* d_x, const int32_t* d_y, int64_t m, int32_t* d_root);

/* ---- set algebra  (lib/core/kmer_set.h:164-187, :286-305; the loop needs
 *      A&B, A\B and B\A of one pair, lib/core/kmer_set_set.h:339-343) ------------------ */
/* Pass 1.  Counts |A & B| per bucket and derives the bucket offsets of the three
 * results: d_off_i / d_off_amb / d_off_bma are int64[2^N + 1] outputs.
 * totals = { |A & B|, |A \ B|, |B \ A| } (host).  The plan stays valid for the
 * next ksh_pair_write on the same context with the same A and B. */
int ksh_pair_plan(ksh_ctx* ctx, const ksh_geom* g, const ksh_set_view* a, const ksh_set_view* b,
                  int64_t* d_off_i, int64_t* d_off_amb, int64_t* d_off_bma, int64_t totals[3]);
/* Pass 2.  Writes the keys of the three results (buffers sized from `totals`;
 * any of them may be NULL to skip that result). */
int ksh_pair_write(ksh_ctx* ctx, const ksh_geom* g, const ksh_set_view* a, const ksh_set_view* b,
                   void* d_keys_i, void* d_keys_amb, void* d_keys_bma);

/* Both passes in ONE call, enqueued back to back (one stream synchronisation, for the
 * totals, and no allocation between the passes).  The key buffers are allocated by the
 * caller before the sizes are known: d_keys_i >= min(|A|, |B|) keys, d_keys_amb >= |A|,
 * d_keys_bma >= |B|. */
int ksh_pair_algebra(ksh_ctx* ctx, const ksh_geom* g, const ksh_set_view* a, const ksh_set_view* b,
                     int64_t* d_off_i, int64_t* d_off_amb, int64_t* d_off_bma, void* d_keys_i,
                     void* d_keys_amb, void* d_keys_bma, int64_t totals[3]);

/* Many independent pairs (a pairwise diff matrix, the reference's pooled loops over pairs,
 * lib/core/kmer_set_set.h:205-216): the buckets of all pairs are tiled together, so the batch is
 * one plan, one count launch and one write launch, and one stream synchronisation returns all
 * totals.  Buffers as for ksh_pair_algebra. */
typedef struct ksh_pair_job {
  ksh_set_view a, b;
  int64_t *d_off_i, *d_off_amb, *d_off_bma; /* int64[2^N + 1] each                         */
  void *d_keys_i, *d_keys_amb, *d_keys_bma; /* >= min(|A|,|B|), |A|, |B| keys               */
  int64_t totals[3];                        /* out: |A & B|, |A \ B|, |B \ A|               */
} ksh_pair_job;
int ksh_pair_algebra_batch(ksh_ctx* ctx, const ksh_geom* g, ksh_pair_job* jobs, int32_t n_jobs);

/* KmerSet::Add(other) / free Add (lib/core/kmer_set.h:164-174,286-290): A | B, same
 * two-call shape.  d_off_u is int64[2^N + 1]; d_keys_u holds `total` keys. */
int ksh_set_union_plan(ksh_ctx* ctx, const ksh_geom* g, const ksh_set_view* a,
                       const ksh_set_view* b, int64_t* d_off_u, int64_t* total);
int ksh_set_union_write(ksh_ctx* ctx, const ksh_geom* g, const ksh_set_view* a,
                        const ksh_set_view* b, void* d_keys_u);

/* ---- KmerSet::Diff / Equals  (lib/core/kmer_set.h:191-219) ----------------------------- */
/* |A \ B| + |B \ A|. */
int ksh_set_diff(ksh_ctx* ctx, const ksh_geom* g, const ksh_set_view* a, const ksh_set_view* b,
                 int64_t* diff);

/* ---- GetEdgeWeight over sampled buckets  (lib/core/kmer_set_set.h:158-219,385-425) ----- */
/* weights[p] = sum over the listed buckets of |bucket(sets[pairs[2p]]) &
 * bucket(sets[pairs[2p+1]])|.  A "sampled set" of the reference
 * (KmerSetCompact::GetSampledKmerSet, kmer_set_compact.h:120-203) is a slice of
 * the resident sorted buckets here, so no separate structure is built.
 * sets, bucket_ids, pairs and weights are HOST arrays. */
int ksh_pair_weights(ksh_ctx* ctx, const ksh_geom* g, const ksh_set_view* sets, int32_t n_sets,
                     const int32_t* bucket_ids, int32_t n_ids, const int32_t* pairs,
                     int32_t n_pairs, int64_t* weights);

/* ---- SPSS container and decode -------------------------------------------------------
 * KmerSetCompact (lib/core/kmer_set_compact.h:339-347) on device: the strings'
 * bases, 2 bits each (A=0 C=1 G=2 T=3), concatenated without separators in 64-bit
 * words -- base j of the stream at bits [63-2(j%32), 62-2(j%32)] of word j/32, i.e.
 * the reference's vector<bool> index 2j is the high bit (:236-251) -- plus
 * len - K per string (:222-223). */
typedef struct ksh_spss_view {
  const uint64_t* d_words; /* ceil(n_bases / 32) words            */
  const uint32_t* d_lens;  /* len - K per string                  */
  int64_t n_strings;
  int64_t n_bases;         /* = KmerSetCompact::Weight() (:115)   */
} ksh_spss_view;

/* KmerSetCompact::Size (kmer_set_compact.h:90-112): sum of (len - K + 1). */
int ksh_spss_size(ksh_ctx* ctx, const ksh_geom* g, const ksh_spss_view* s, int64_t* n_kmers);

/* KmerSetCompact::ToKmerSet = ToStrings + GetKmerSetFromSPSS
 * (kmer_set_compact.h:52-55,290-336; lib/core/spss.h:1861-1941).
 * plan : bucket histogram -> d_offsets (int64[2^N + 1]); n_keys = k-mer positions
 *        (an upper bound on the set size: repeated k-mers collapse in write).
 * write: scatter + per-bucket sort, duplicates dropped; d_keys holds n_keys(plan)
 *        keys; d_offsets is rewritten if duplicates were dropped; n_keys = set size.
 * Limit: n_bucket_bits <= 14 (KSH_INVALID_ARGUMENT beyond): the counting pass keeps one 4-byte counter per
 * bucket in a workgroup's 64 KB of LDS.  The reference's template takes any N (lib/core/kmer_set.h:20-31); its
 * CLIs instantiate N = 10 and 14 (src/kmerset-multiple-compress.cc:156-157).  The same limit holds for
 * ksh_kmer_count_write and for everything that decodes (ksh_kss_build*, ksh_kss_get, ksh_spss_from_text_*). */
int ksh_spss_decode_plan(ksh_ctx* ctx, const ksh_geom* g, const ksh_spss_view* s, int canonical,
                         int64_t* d_offsets, int64_t* n_keys);
int ksh_spss_decode_write(ksh_ctx* ctx, const ksh_geom* g, const ksh_spss_view* s, int canonical,
                          int64_t* d_offsets, void* d_keys, int64_t* n_keys);

/* ---- Text form of an SPSS: the file format of KmerSetCompact::Dump / Load ------------------
 * (lib/core/kmer_set_compact.h:62-87 over WriteLines / ReadLines, lib/core/io.h:20-126): one
 * string over ACGT per line, every line closed by '\n'; the reference spells / parses it base
 * by base on the host (ToStrings :290-336, the private constructor :206-266).
 * to_text: d_text receives exactly n_bases + n_strings bytes.
 * from_text: plan counts the lines and bases of n_bytes of text in device memory (a last line
 * without '\n' counts); the caller allocates ceil(n_bases / 32) words and n_strings lengths;
 * write fills them (ksh_spss_view layout).  A byte other than A, C, G, T, '\n' or a line
 * shorter than K is KSH_INVALID_ARGUMENT (the reference asserts / underflows).  K >= 4. */
int ksh_spss_to_text(ksh_ctx* ctx, const ksh_geom* g, const ksh_spss_view* s, char* d_text);
int ksh_spss_from_text_plan(ksh_ctx* ctx, const ksh_geom* g, const char* d_text, int64_t n_bytes,
                            int64_t* n_strings, int64_t* n_bases);
int ksh_spss_from_text_write(ksh_ctx* ctx, uint64_t* d_words, uint32_t* d_lens);

/* ---- KmerCounter: FASTA -> counted k-mers -> KmerSet (the step before the loop) ---------------
 * lib/core/kmer_counter.h:136-206 FromFASTA (lines 0, 2, ... are headers starting with '>',
 * lines 1, 3, ... reads over ACGTN), :64-133 FromReads (reads split at 'N', every K-long window
 * is one k-mer, canonical or not), :209-243 ToKmerSet (k-mers seen at least `cutoff` times; the
 * second result counts the distinct k-mers seen less often).
 * ksh_fasta_plan / ksh_fasta_write turn n_bytes of FASTA text in device memory into the reads'
 * ACGT fragments of length >= K as an SPSS-shaped 2-bit stream (ksh_spss_view layout; the
 * caller allocates ceil(n_bases / 32) words and n_fragments lengths between the two calls).
 * An odd number of lines, a header that is empty or does not start with '>', or a read byte
 * outside ACGTN is KSH_FAILED_PRECONDITION with the reference's message.
 * Counting is the decode pipeline with multiplicities: ksh_spss_decode_plan on the fragments
 * (*n_keys = k-mer occurrences, the size of the key buffer), then ksh_kmer_count_write:
 * offsets + sorted keys of the k-mers whose count >= cutoff, *n_keys of them, *n_cut distinct
 * k-mers below the cutoff (cutoff in 0..255: uint8 counts saturate in the reference). */
int ksh_fasta_plan(ksh_ctx* ctx, const ksh_geom* g, const char* d_text, int64_t n_bytes,
                   int64_t* n_fragments, int64_t* n_bases);
int ksh_fasta_write(ksh_ctx* ctx, uint64_t* d_words, uint32_t* d_lens);
int ksh_kmer_count_write(ksh_ctx* ctx, const ksh_geom* g, const ksh_spss_view* reads, int canonical,
                         int32_t cutoff, int64_t* d_offsets, void* d_keys, int64_t* n_keys,
                         int64_t* n_cut);

/* ---- StreamVByte "0124" pack of the string lengths ---------------------------------------
 * The in-memory form of KmerSetCompact::lengths_compressed_
 * (lib/core/kmer_set_compact.h:257-265 streamvbyte_encode_0124, :269-287 decode; format in
 * csrc/ksh_svb.hip).  encode: d_out holds ceil(n/4) + 4n bytes (streamvbyte_max_compressedbytes)
 * or is NULL to get the size only; *bytes = compressed size.  decode: n values out. */
int ksh_svb_encode_0124(ksh_ctx* ctx, const uint32_t* d_in, int64_t n, uint8_t* d_out, int64_t* bytes);
int ksh_svb_decode_0124(ksh_ctx* ctx, const uint8_t* d_in, int64_t n, uint32_t* d_out,
                        int64_t* bytes_read);

/* ---- SPSS encode ----------------------------------------------------------------------
 * KmerSetCompact::FromKmerSet(set, canonical, fast, n_workers)
 * (lib/core/kmer_set_compact.h:36-47) = the SPSS + the 2-bit packing constructor (:206-266).
 * canonical != 0 (the set holds canonical k-mers only):
 *   mode 0: GetSPSSCanonical(fast = true) (lib/core/spss.h:1835-1858: unitigs, greedy path
 *           cover :1358-1539, loop cut :1541-1647, stitch :1649-1829).
 *   mode 1: GetUnitigsCanonical (lib/core/spss.h:230-615), every unitig a string.
 *   mode 2: GetSPSSCanonical(fast = false) (lib/core/spss.h:1208-1356): the reference's
 *           one-thread path extension, replayed by one device thread over the edge table
 *           (sequential by definition; everything around it is the parallel pipeline).
 * canonical == 0 (k-mers as they are, edges only forward):
 *   mode 0 / 2: GetSPSS (lib/core/spss.h:697-1036; FromKmerSet ignores `fast` here).
 *   mode 1: GetUnitigs (lib/core/spss.h:73-227).
 * The strings and their order are the oracle's (the reference's n_workers == 1
 * control flow with ascending iteration).
 * KSH_INVALID_ARGUMENT for a canonical set that holds a k-mer equal to its own reverse
 * complement (even k only).
 * plan : everything up to the string layout; returns the container's sizes.
 * write: d_words = ceil(n_bases / 32) words, d_lens = n_strings values (len - K). */
int ksh_spss_encode_plan(ksh_ctx* ctx, const ksh_geom* g, const ksh_set_view* set, int canonical,
                         int mode, int64_t* n_strings, int64_t* n_bases);
int ksh_spss_encode_write(ksh_ctx* ctx, uint64_t* d_words, uint32_t* d_lens);
/* stats = { unitigs, matching rounds, strings, bases } of the current plan. */
int ksh_spss_encode_stats(ksh_ctx* ctx, int64_t stats[4]);
/* Which variants of the encode's kernels the current plan ran (they are chosen by set size and geometry;
 * the parity tests assert the route so that a case cannot silently take another one): a mask of KSH_ROUTE_*. */
enum {
  KSH_ROUTE_PROBE_STAGED = 1 << 0,       /* neighbour probe staged in LDS (k_rc_*, k_adj_rc1 / k_adj_rc), not k_adjacency */
  KSH_ROUTE_RC_1024 = 1 << 1,            /* k_adj_rc1 / k_adj_rc with 1024 / 512 / 256 / 64 threads per group */
  KSH_ROUTE_RC_512 = 1 << 2,
  KSH_ROUTE_RC_256 = 1 << 3,
  KSH_ROUTE_RC_64 = 1 << 4,
  KSH_ROUTE_RC_BATCHED = 1 << 5,         /* some group took several batches: of its records (k_adj_rc1) or of its ranges (k_adj_rc) */
  KSH_ROUTE_SCATTER_TWO_LEVEL = 1 << 6,  /* k_rc_scatter_l1 / _l2 instead of k_rc_scatter */
  KSH_ROUTE_FWD_STAGED = 1 << 7,         /* k_adj_fwd_staged (KSH_FWD=staged) instead of k_adj_fwd */
  KSH_ROUTE_RANK_ONE_LAUNCH = 1 << 8,    /* all ruler walkers in one launch (mirror images racing) */
  KSH_ROUTE_HEADS_ONE_LAUNCH = 1 << 9,   /* the same for the chain-start walkers */
  KSH_ROUTE_JUMP_TWO_LEVEL = 1 << 10,    /* pointer jumping over the level-2 rulers (k_l2_*) */
  KSH_ROUTE_RANK_STAMPED = 1 << 11,      /* the stamping walks (sets with a non-branching loop, KSH_RANK=stamp) */
  KSH_ROUTE_EMIT_LOGS = 1 << 12,         /* strings written from the ranking walks' logs */
  KSH_ROUTE_LONG_STRETCHES = 1 << 13,    /* ... and some stretch outgrew its log (the long list) */
  KSH_ROUTE_MATCH_MORE_ROUNDS = 1 << 14, /* the matching needed more than its first batch of rounds */
  KSH_ROUTE_FWD_TARGETS = 1 << 15,       /* k_adj_fwd_targets: forward probes marked at their targets, one search per k-mer */
  KSH_ROUTE_RC1_STREAMED = 1 << 16,      /* k_adj_rc1: k_adj_rc turned round (a group's records in LDS, its ranges streamed) */
  KSH_ROUTE_RC_MARKS_GROUPS = 1 << 17    /* ... and k_adj_rc ran for some group whose records did not fit k_adj_rc1 */
};
int ksh_spss_encode_routes(ksh_ctx* ctx, int64_t* routes);
/* Frees the current plan's device memory (also done by the next plan / ctx_destroy). */
int ksh_spss_encode_release(ksh_ctx* ctx);

/* ---- KmerSetSet: the loop kmerset-multiple-compress runs -------------------------------
 * ksh_kss_build = KmerSetSet(vector<KmerSetCompact>, canonical, n_workers)
 * (lib/core/kmer_set_set.h:109-427) on device-resident sets.  inputs are the
 * KmerSetCompact containers (device); bucket_ids replaces the unseeded
 * GetRandomInts((1 << N) / 50, ...) of :123-124 (HOST array, ascending);
 * max_iterations < 0 runs to the reference's stopping rule.  The input containers
 * must stay alive while they are nodes of the result, and the context must outlive the
 * result.  The three re-encodes of a merge (:345-360) are deferred to the points where the
 * reference reads them (Weight() at the convergence checks, :287, and the final nodes): same
 * values, fewer encodes.  Synchronises the stream before returning. */
typedef struct ksh_kss ksh_kss;
int ksh_kss_build(ksh_ctx* ctx, const ksh_geom* g, const ksh_spss_view* inputs, int32_t n_inputs,
                  const int32_t* bucket_ids, int32_t n_ids, int canonical, int32_t max_iterations,
                  ksh_kss** out);
/* Multi-GPU build, one process per GPU.  Every rank calls this with the same inputs and runs the
 * same (deterministic) loop on its own resident copies of the sets; the SPSS encodes, ~85 % of
 * the loop's time, are dealt out node by node (largest set first, to the least loaded rank), and at every point
 * where the loop reads the SPSS weights (kmer_set_set.h:287 and the end) the ranks exchange
 * (n_strings, n_bases) of the freshly encoded nodes through `gather`: an all-gather of `count`
 * int64 per rank into recv[world * count], rank-major, returning 0 on success (RCCL or any other
 * transport; KB-scale, a few times per build).  Trace, checkpoints, DAG and every node's set
 * are identical on all ranks and equal to the single-GPU build; a node's SPSS is held by one
 * rank (ksh_kss_node_holder; -1 = every rank, the inputs), and ksh_kss_node answers
 * KSH_FAILED_PRECONDITION for the SPSS of a node held elsewhere. */
typedef int (*ksh_allgather_i64)(void* user, const int64_t* send, int64_t count, int64_t* recv);
int ksh_kss_build_sharded(ksh_ctx* ctx, const ksh_geom* g, const ksh_spss_view* inputs, int32_t n_inputs,
                          const int32_t* bucket_ids, int32_t n_ids, int canonical,
                          int32_t max_iterations, int32_t rank, int32_t world, ksh_allgather_i64 gather,
                          void* gather_user, ksh_kss** out);
int ksh_kss_node_holder(const ksh_kss* k, int32_t i, int32_t* rank);

/* ---- owner-sharded multi-GPU build --------------------------------------------------------
 * One process per GPU; no rank holds every node (config 5: 256 sets of 5 * 10^8 48-bit keys are
 * 1 TB decoded, 128 GB per GPU of an 8-GPU node).
 *   * every input set is decoded and kept by ONE rank, its owner (owners[i]); a rank passes the
 *     containers of the inputs it owns, the others' entries are ignored;
 *   * what the loop's decisions read -- the 2 % sampled buckets of every node
 *     (kmer_set_set.h:138-153) -- is all-gathered once for the inputs; after that every rank runs
 *     the same control loop on its copy of the samples (arg-max :308-316, the merged pair's new
 *     samples :349-363 = the same merge on the sampled buckets, the re-weighting :385-425) with no
 *     exchange at all: the samples of a merge's three results ARE the merge of the samples;
 *   * the full-size Intersection + Sub + Sub of iteration (j, k) (:339-343) runs on the owner of
 *     j; if k lives elsewhere its keys travel there point to point (xGMI) and k's remainder stays
 *     with j's owner, which also owns the new node;
 *   * at the points where the loop reads SPSS weights (:287 and the end) every rank encodes the
 *     stale nodes it owns and one all-gather of (n_strings, n_bases, size) per stale node puts
 *     the sum on all ranks.  A check only decides whether the loop stops, so its exchange is taken
 *     one check later: the ranks go on with the next interval meanwhile and undo it if the answer
 *     was "stop" (same result; a loaded rank no longer holds up the others at every check), and
 *     before a check's encodes run they are dealt out again: the most loaded rank hands its largest
 *     stale node to the least loaded one (the set travels under the encodes, on a second
 *     communicator), which owns it from then on.
 * Trace, checkpoints, DAG and every node's set and SPSS equal the single-GPU build's; a node's
 * set and SPSS live on its owner only (ksh_kss_node_holder; ksh_kss_node answers
 * KSH_FAILED_PRECONDITION elsewhere, ksh_kss_get when a reachable node lives elsewhere).
 *
 * Transport: RCCL, opened at run time (librccl.so), collectives enqueued on the context's stream
 * with device buffers -- rank 0 draws the id, the caller carries its 128 bytes to the other ranks
 * (MPI, torch.distributed, a file) -- or caller-supplied functions over device buffers, which
 * return when the transfer is complete (rehearsals with several ranks on one GPU, where RCCL
 * cannot run; the library drains its stream before calling them). */
#define KSH_COMM_ID_BYTES 128
typedef struct ksh_comm ksh_comm;
typedef struct ksh_comm_fns {
  /* = sizeof(ksh_comm_fns) of the header the caller was compiled against (ksh_version() >= 2): members beyond it
   * are taken as NULL, so the struct can grow without old callers being read past their end; zero-initialise
   * the struct and set the members you have */
  size_t struct_size;
  void* user;
  /* d_recv receives world * bytes, rank-major */
  int (*allgather)(void* user, const void* d_send, void* d_recv, size_t bytes);
  int (*send)(void* user, const void* d_buf, size_t bytes, int32_t peer);
  int (*recv)(void* user, void* d_buf, size_t bytes, int32_t peer);
  /* optional (may be NULL): called once when this rank gives the build up at a point where it cannot
   * follow the exchange protocol any longer; it should make the peers' pending calls fail */
  void (*abort)(void* user);
} ksh_comm_fns;
int ksh_comm_unique_id(unsigned char id[KSH_COMM_ID_BYTES]);
int ksh_comm_create_rccl(ksh_ctx* ctx, int32_t rank, int32_t world, const unsigned char id[KSH_COMM_ID_BYTES],
                         ksh_comm** out);
int ksh_comm_create_custom(ksh_ctx* ctx, int32_t rank, int32_t world, const ksh_comm_fns* fns, ksh_comm** out);
int ksh_comm_destroy(ksh_comm* comm);
/* Collective: every rank's id through the transport's all-gather (device buffers); *n_ranks = distinct
 * ids received: the ranks that actually took part (bench.py quotes it as multi_gpu.ranks_seen). */
int ksh_comm_ranks_seen(ksh_comm* comm, int32_t* n_ranks);
/* owners[i] in [0, world): the rank that holds input i.  inputs[i] is read on that rank only. */
int ksh_kss_build_owned(ksh_ctx* ctx, ksh_comm* comm, const ksh_geom* g, const ksh_spss_view* inputs,
                        int32_t n_inputs, const int32_t* owners, const int32_t* bucket_ids, int32_t n_ids,
                        int canonical, int32_t max_iterations, ksh_kss** out);
/* stats = { bytes this rank sent point to point, bytes it received, sets it sent (merged pairs whose
 * members lived on different ranks), bytes it contributed to all-gathers, convergence checks whose
 * exchange was deferred by one check (the ranks go on with the next interval instead of waiting for the
 * slowest encoder; KSH_OWNED_LOOKAHEAD=0 turns that off), intervals undone because a deferred check
 * said "stop", sets this rank handed to a less loaded rank for encoding at a check (they live there
 * afterwards; KSH_OWNED_MIGRATE=0 turns that off), all-gathers of per-pair int64 weights (one per iteration
 * with KSH_OWNED_WEIGHTS=sharded: the weight tables dealt out by pair list, lib/core/kmer_set_set.h:205-218,
 * 385-425; 0 by default: every rank weighs its replica of the samples and nothing is exchanged) }. */
int ksh_kss_comm_stats(const ksh_kss* k, int64_t stats[8]);
/* SPSS encodes this process ran for the build, and the k-mers they covered (the sharded build's
 * balance; in a single-GPU build: how many encodes the deferral left). */
int ksh_kss_encode_counts(const ksh_kss* k, int64_t* n_encodes, int64_t* n_encoded_kmers);
/* Of those encodes, the ones that only computed Weight() (the encode's plan without the write): with
 * KSH_KSS_LOOP=ahead ksh_kss_build runs the control loop on the samples two intervals ahead of the sets
 * (every decision of the loop reads sampled buckets only) and only weighs the stale nodes it will merge
 * again.  Off by default: on loops of a few checks the strings owed when the last check stops the loop
 * cost what the skipped writes saved. */
int ksh_kss_weighed_counts(const ksh_kss* k, int64_t* n_weighed, int64_t* n_weighed_kmers);
/* Wall seconds the build spent in: [0] decode of the inputs, [1] weight computations,
 * [2] merges (pair plan + write), [3] SPSS encodes (with, in a sharded build, the exchanges). */
int ksh_kss_phase_seconds(const ksh_kss* k, double seconds[4]);
int ksh_kss_destroy(ksh_kss* k);
/* KmerSetSet::Size (:430): number of nodes. */
int ksh_kss_size(const ksh_kss* k, int32_t* n_nodes);
/* Node i: its SPSS container, its resident set and its k-mer count (any may be NULL). */
int ksh_kss_node(const ksh_kss* k, int32_t i, ksh_spss_view* compact, ksh_set_view* set,
                 int64_t* size);
/* children_[i] (:365-366); the pointer stays valid until ksh_kss_destroy. */
int ksh_kss_children(const ksh_kss* k, int32_t i, const int32_t** children, int32_t* n_children);
/* Line 0 of meta.<ext> (SerializeAdjacencyList, :45-56), keys ascending. */
const char* ksh_kss_meta(const ksh_kss* k);
/* Per-iteration rows {j, k, weight, |S_j| + |S_k|, size_diff} (the values the reference
 * logs at :324,:380) and per-checkpoint rows {iteration, previous, updated, stopped}
 * with the float improvement of :289-295. */
int ksh_kss_trace(const ksh_kss* k, int64_t* n_iterations, const int64_t** rows,
                  int64_t* n_checkpoints, const int64_t** checkpoint_rows,
                  const float** improvements);
int ksh_kss_initial_weights(const ksh_kss* k, const int64_t** weights, int64_t* n);
/* stats = { initial total_size, final total_size, initial total_spss_weight, N_proc
 * (SURVEY.md 8d), final total_spss_weight, sum over nodes of ceil(2 Weight / 8) bytes, sum
 * over nodes of the StreamVByte-0124 size of the lengths, nodes }: bytes/k-mer after SPSS =
 * (stats[5] + stats[6]) / sum of the input set sizes (metric 2). */
int ksh_kss_stats(const ksh_kss* k, int64_t stats[8]);
/* KmerSetSet::Get(i) (:433-454): union over the nodes reachable from i.  Returns new
 * device buffers (release with ksh_free). */
int ksh_kss_get(const ksh_kss* k, int32_t i, int64_t** d_offsets, void** d_keys, int64_t* n_keys);

#ifdef __cplusplus
}
#endif

#endif /* KMERSETS_HIP_H_ */
