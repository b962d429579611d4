// KmerSetSet on device: the greedy pair-merge loop that kmerset-multiple-compress runs
// (lib/core/kmer_set_set.h:109-427), Get (:433-454) and the node accessors.
//
// Host control flow is the reference's, statement for statement (same float
// arithmetic for the stopping rule, same pair lists), with the oracle's two
// ordering rules: bucket_ids are an input and the arg-max takes the first maximal
// (j, k) in lexicographic order.  What changes is where the data lives:
//
//   * every node's k-mer set stays resident in HBM as bucketed sorted keys, so the
//     reference's two ToKmerSet decodes per iteration (:333-336) disappear -- inputs
//     are decoded once -- and GetSampledKmerSet (:349-363) is a slice, not a pass;
//   * Intersection + Sub + Sub (:339-343) is one ksh_pair_plan/ksh_pair_write;
//   * the three FromKmerSet re-encodes (:345-360) are ksh_spss_encode_*;
//   * the weight table (:191-219,:385-425) is ksh_pair_weights.
#include "ksh_internal.h"
#include "ksh_comm.h"

#include <cstring>
#include <map>
#include <queue>
#include <unordered_set>
#include <algorithm>
#include <chrono>
#include <utility>
#include <vector>

struct KssSet {
  int64_t* off = nullptr;
  void* keys = nullptr;
  int64_t n = 0;
};

struct KssCompact {
  uint64_t* words = nullptr;
  uint32_t* lens = nullptr;
  int64_t n_strings = 0, n_bases = 0, size = 0;
  bool owned = false;
  bool valid = true;  // false: the node's set changed and its SPSS has not been re-encoded yet
  bool weight_only = false;  // valid, but only (n_strings, n_bases) are known: the strings were not written
  int holder = -1;    // sharded build: the rank that holds the SPSS (-1: every rank does)
};

struct ksh_kss {
  ksh_ctx* ctx = nullptr;
  ksh_geom g{};
  int canonical = 1;
  std::vector<KssSet> sets;
  std::vector<KssCompact> compacts;
  std::map<int, std::vector<int>> children;
  std::vector<int64_t> trace;        // 5 per iteration: j, k, weight, original_size, size_diff
  std::vector<int64_t> checkpoints;  // 4 per checkpoint: iteration, previous, updated, stopped
  std::vector<float> improvements;
  std::vector<int64_t> initial_weights;
  int64_t initial_total_size = 0, final_total_size = 0, initial_spss_weight = 0, n_processed = 0;
  int64_t final_spss_weight = 0;
  int64_t n_encodes = 0, n_encoded_kmers = 0;
  int64_t n_weighed = 0, n_weighed_kmers = 0;  // of them: weight only (encode plan without the write)
  double phase_seconds[4] = {0, 0, 0, 0};  // decode of the inputs, weights, merges, encodes
  std::string meta;
  // sharded build (ksh_kss_build_sharded): every rank runs the same loop on resident copies of
  // the sets; the SPSS encodes, the dominant cost, are dealt out node by node
  int rank = 0, world = 1;
  ksh_allgather_i64 gather = nullptr;
  void* gather_user = nullptr;
  // owner-sharded build (ksh_kss_build_owned): a node's set and SPSS live on owner[node] only; every
  // rank keeps the sampled buckets of every node (samples[i]: a set whose other buckets are empty)
  ksh_comm* comm = nullptr;
  bool owned = false;
  std::vector<int> owner;
  std::vector<KssSet> samples;
  std::vector<bool> sample_pooled;   // false: the keys point into sample_block
  char* sample_block = nullptr;      // the all-gathered samples of the inputs
  int64_t p2p_bytes_sent = 0, p2p_bytes_received = 0, p2p_sets = 0, gather_bytes = 0;
  // deferred convergence checks (build_owned): while a check is unresolved, buffers of the state it may
  // have to return to are parked instead of freed
  const std::unordered_set<void*>* keep_alive = nullptr;
  std::vector<void*>* graveyard = nullptr;
  int64_t checks_deferred = 0, rollbacks = 0, sets_migrated = 0, weight_gathers = 0;
  // A rank-local failure of the owner-sharded build (an allocation, a decode, a merge, an encode): the rank
  // goes on FOLLOWING THE EXCHANGE PROTOCOL with empty stand-ins for what it could not build -- its peers
  // are heading for exchanges it must not leave them alone in -- every all-gather carries every rank's
  // status word, and all ranks return the error together after the next one (failed_together).  A
  // failure that keeps a rank from following the protocol at all (its replica of the control loop, the
  // transport itself) ends in comm_abort instead (ksh_kss_build_owned).
  int local_rc = KSH_OK;
  std::string local_msg;
  bool failed_together = false;
  int64_t *zero_off = nullptr, *xfer_off = nullptr;  // 2^N + 1 zeros: what is sent for a set the rank does not have; where offsets arrive
  int64_t *gx_send = nullptr, *gx_recv = nullptr;    // device staging of gather_i64, allocated once: a gather cannot fail for want of memory
  size_t gx_cap = 0;                                 // int64 per rank
  struct Exchange {                                  // the two deferred check exchanges that can be in flight
    int64_t *d_send = nullptr, *d_recv = nullptr, *h_recv = nullptr;
    hipEvent_t arrived = nullptr;
  } xchg[2];
  size_t xchg_cap = 0;
};

namespace ksh {

// Wall time per phase of the build (decode of the inputs, weights, merges, encodes), for the
// shares quoted in DESIGN.md.  Every phase is closed by a stream synchronisation, which the
// loop needs at those points anyway (each ends in a read-back), so the timers cost nothing.
struct PhaseTimer {
  ksh_kss* k;
  int kind;
  std::chrono::steady_clock::time_point t0;
  PhaseTimer(ksh_kss* kss, int which) : k(kss), kind(which), t0(std::chrono::steady_clock::now()) {}
  ~PhaseTimer() {
    (void)hipStreamSynchronize(k->ctx->stream);
    k->phase_seconds[kind] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  }
};

// Lets go of a pooled buffer: parked while an unresolved check may still need it, else freed.
static void retire(ksh_kss* k, void* p) {
  if (!p) return;
  if (k->keep_alive && k->keep_alive->count(p)) k->graveyard->push_back(p);
  else pool_free(k->ctx, p);
}

static bool alive(const ksh_kss* k) { return k->local_rc == KSH_OK; }
// Records the first rank-local failure (the message of the failing call is the thread's last error).
static void poison(ksh_kss* k, int rc) {
  if (rc == KSH_OK || k->local_rc != KSH_OK) return;
  k->local_rc = rc;
  k->local_msg = ksh_last_error();
}
// A rank-local step: skipped once the rank has failed, its failure recorded instead of returned.
#define KSH_LOCAL(k, expr)             \
  do {                                 \
    if (alive(k)) poison((k), (expr)); \
  } while (0)

static void free_set(ksh_ctx* ctx, KssSet* s) {
  pool_free(ctx, s->off);
  pool_free(ctx, s->keys);
  *s = KssSet{};
}

static void free_compact(ksh_ctx* ctx, KssCompact* c) {
  if (c->owned) {
    pool_free(ctx, c->words);
    pool_free(ctx, c->lens);
  }
  *c = KssCompact{};
}

static ksh_set_view view_of(const KssSet& s) { return ksh_set_view{s.off, s.keys, s.n}; }
static ksh_spss_view view_of(const KssCompact& c) {
  return ksh_spss_view{c.words, c.lens, c.n_strings, c.n_bases};
}

static int alloc_offsets(ksh_ctx* ctx, const ksh_geom* g, KssSet* out) {
  return pool_alloc(ctx, size_t(n_buckets(g) + 1) * 8, reinterpret_cast<void**>(&out->off));
}

static int alloc_keys(ksh_ctx* ctx, const ksh_geom* g, int64_t n_keys, KssSet* out) {
  out->n = n_keys;
  return pool_alloc(ctx, std::max<size_t>(size_t(n_keys) * g->key_bytes, 16), &out->keys);
}

// (ctx may be a lane: the work runs on its stream with its scratch, the result lives in its parent's pool)
static int decode_to_set(ksh_ctx* ctx, const ksh_geom* g, const ksh_spss_view* sp, int canonical_flag,
                         KssSet* out) {
  ksh_ctx* home = result_ctx(ctx);
  KSH_TRY(alloc_offsets(home, g, out));
  int64_t n = 0;
  int rc = ksh_spss_decode_plan(ctx, g, sp, canonical_flag, out->off, &n);
  if (rc == KSH_OK) rc = alloc_keys(home, g, n, out);
  if (rc == KSH_OK) rc = ksh_spss_decode_write(ctx, g, sp, canonical_flag, out->off, out->keys, &n);
  if (rc != KSH_OK) {
    (void)hipStreamSynchronize(ctx->stream);  // (nothing of this job is still writing into what goes back to the pool)
    free_set(home, out);
    return rc;
  }
  out->n = n;
  return KSH_OK;
}

static int encode_set(ksh_ctx* ctx, const ksh_geom* g, const KssSet& s, int canonical_flag,
                      KssCompact* out) {
  ksh_ctx* home = result_ctx(ctx);
  const ksh_set_view v = view_of(s);
  int64_t ns = 0, nbases = 0;
  KSH_TRY(ksh_spss_encode_plan(ctx, g, &v, canonical_flag, 0, &ns, &nbases));
  out->owned = true;
  int rc = pool_alloc(home, std::max<size_t>(size_t((nbases + 31) / 32) * 8, 16), reinterpret_cast<void**>(&out->words));
  if (rc == KSH_OK) rc = pool_alloc(home, std::max<size_t>(size_t(ns) * 4, 16), reinterpret_cast<void**>(&out->lens));
  if (rc == KSH_OK) rc = ksh_spss_encode_write(ctx, out->words, out->lens);
  if (rc != KSH_OK) {
    (void)hipStreamSynchronize(ctx->stream);
    free_compact(home, out);
    return rc;
  }
  out->n_strings = ns;
  out->n_bases = nbases;
  out->size = s.n;
  out->valid = true;
  return KSH_OK;
}

// Independent jobs of one build on the context's lanes (ksh::run_on_lanes): largest first.
static std::vector<size_t> largest_first(const std::vector<size_t>& items, const std::function<int64_t(size_t)>& cost) {
  std::vector<size_t> order(items);
  std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return cost(a) > cost(b); });
  return order;
}

// The reference re-encodes the three nodes of every merge at once (kmer_set_set.h:345-360), but
// reads the result only through Weight() at the convergence checks (:287) and at the end.  A
// node that is merged again before the next check never needs its intermediate SPSS, so the
// encode is deferred to the points where the reference looks: same values, fewer encodes.
static int ensure_compacts(ksh_kss* k) {
  if (k->world <= 1) {
    // the stale nodes are independent (kmer_set_set.h:287,345-360; their ids were fixed when they were made): on
    // the context's lanes, several at once
    std::vector<size_t> stale;
    int64_t n_max = 0;
    for (size_t i = 0; i < k->compacts.size(); i++)
      if (!k->compacts[i].valid) {
        stale.push_back(i);
        n_max = std::max(n_max, k->sets[i].n);
      }
    const int rc = run_on_lanes(
        k->ctx, largest_first(stale, [&](size_t i) { return k->sets[i].n; }), encode_scratch_bytes(&k->g, n_max),
        [&](ksh_ctx* lane) { return encode_reserve(lane, &k->g, n_max); },
        [&](ksh_ctx* lane, size_t i) {
          KssCompact c;
          KSH_TRY(encode_set(lane, &k->g, k->sets[i], k->canonical, &c));
          k->compacts[i] = c;  // (distinct nodes: no two jobs write one element)
          return int(KSH_OK);
        });
    for (size_t i : stale)
      if (k->compacts[i].valid) {
        k->n_encodes++;
        k->n_encoded_kmers += k->sets[i].n;
      }
    return rc;
  }
  // Sharded: the stale nodes (the same list on every rank, the loop being deterministic) are
  // dealt out; a rank encodes its share and all ranks exchange what the loop reads,
  // (n_strings, n_bases) per node.
  std::vector<size_t> stale;
  for (size_t i = 0; i < k->compacts.size(); i++)
    if (!k->compacts[i].valid) stale.push_back(i);
  if (stale.empty()) return KSH_OK;
  std::vector<int64_t> send(2 * stale.size(), -1), recv(2 * stale.size() * size_t(k->world), -1);
  // longest-processing-time deal: the largest set first, each to the rank with the least k-mers
  // so far (ties: the lower rank); encode cost follows the set size.  Same answer on every rank.
  std::vector<size_t> order(stale.size());
  for (size_t q = 0; q < order.size(); q++) order[q] = q;
  std::stable_sort(order.begin(), order.end(),
                   [&](size_t a, size_t b) { return k->sets[stale[a]].n > k->sets[stale[b]].n; });
  std::vector<int64_t> load(size_t(k->world), 0);
  std::vector<int> holder_of(stale.size(), 0);
  int local_rc = KSH_OK;
  std::string local_msg;
  for (size_t q : order) {
    int best = 0;
    for (int r = 1; r < k->world; r++)
      if (load[size_t(r)] < load[size_t(best)]) best = r;
    holder_of[q] = best;
    load[size_t(best)] += k->sets[stale[q]].n + 1;
  }
  for (size_t q = 0; q < stale.size(); q++) {
    const size_t i = stale[q];
    const int holder = holder_of[q];
    KssCompact c;
    if (holder == k->rank && local_rc == KSH_OK) {
      // a failure here must not keep this rank from the all-gather the others are heading for:
      // its slots stay -1 and every rank returns an error after the exchange
      local_rc = encode_set(k->ctx, &k->g, k->sets[i], k->canonical, &c);
      if (local_rc == KSH_OK) {
        k->n_encodes++;
        k->n_encoded_kmers += k->sets[i].n;
        send[2 * q] = c.n_strings;
        send[2 * q + 1] = c.n_bases;
      } else {
        local_msg = ksh_last_error();
        c = KssCompact{};
      }
    }
    if (holder != k->rank) c.size = k->sets[i].n;
    c.holder = holder;
    k->compacts[i] = c;
  }
  if (k->gather(k->gather_user, send.data(), int64_t(send.size()), recv.data()) != 0)
    return fail(KSH_INTERNAL, "the all-gather callback of the sharded build failed");
  if (local_rc != KSH_OK) return fail(local_rc, "%s", local_msg.c_str());
  for (size_t q = 0; q < stale.size(); q++) {
    KssCompact& c = k->compacts[stale[q]];
    const int64_t* from = recv.data() + size_t(c.holder) * send.size() + 2 * q;
    if (from[0] < 0 || from[1] < 0)
      return fail(KSH_INTERNAL, "rank %d did not report node %zu", c.holder, stale[q]);
    c.n_strings = from[0];
    c.n_bases = from[1];
    c.valid = true;
  }
  return KSH_OK;
}

static int pair_weights(ksh_kss* k, const std::vector<int32_t>& ids,
                        const std::vector<std::pair<int, int>>& pairs, std::vector<int64_t>* out) {
  std::vector<ksh_set_view> views;
  for (const KssSet& s : k->sets) views.push_back(view_of(s));
  out->assign(pairs.size(), 0);
  // Sharded build: the pairs are independent (kmer_set_set.h:205-216), so every rank weighs a
  // contiguous share of the list and one all-gather of int64 puts the table on all of them.
  // Small lists are not worth the exchange.
  size_t lo = 0, hi = pairs.size();
  const bool shard = k->world > 1 && pairs.size() >= size_t(4 * k->world);
  const size_t per = (pairs.size() + size_t(k->world) - 1) / size_t(k->world);
  if (shard) {
    lo = std::min(pairs.size(), per * size_t(k->rank));
    hi = std::min(pairs.size(), lo + per);
  }
  std::vector<int32_t> flat;
  for (size_t q = lo; q < hi; q++) {
    flat.push_back(pairs[q].first);
    flat.push_back(pairs[q].second);
  }
  int local_rc = KSH_OK;
  std::string local_msg;
  if (hi > lo) {
    local_rc = ksh_pair_weights(k->ctx, &k->g, views.data(), int32_t(views.size()), ids.data(),
                                int32_t(ids.size()), flat.data(), int32_t(hi - lo), out->data() + lo);
    if (local_rc != KSH_OK) local_msg = ksh_last_error();
  }
  if (!shard) return local_rc == KSH_OK ? KSH_OK : fail(local_rc, "%s", local_msg.c_str());
  // (a rank that failed still takes part in the exchange: its slots stay -1)
  std::vector<int64_t> send(per, -1), recv(per * size_t(k->world), -1);
  if (local_rc == KSH_OK)
    for (size_t q = lo; q < hi; q++) send[q - lo] = (*out)[q];
  if (k->gather(k->gather_user, send.data(), int64_t(per), recv.data()) != 0)
    return fail(KSH_INTERNAL, "the all-gather callback of the sharded build failed");
  if (local_rc != KSH_OK) return fail(local_rc, "%s", local_msg.c_str());
  for (size_t q = 0; q < pairs.size(); q++) {
    const int64_t w = recv[(q / per) * per + q % per];
    if (w < 0) return fail(KSH_INTERNAL, "rank %zu did not report the weight of pair %zu", q / per, q);
    (*out)[q] = w;
  }
  return KSH_OK;
}

static std::string serialize_children(const std::map<int, std::vector<int>>& a);

// The weights that a merge of (j, k) into the new node n changes.  The reference weighs all three families
// again -- (j, l), (k, l) and (l, n) for every l (kmer_set_set.h:385-425) -- 3 n sampled intersections per
// iteration, which is 4 x the bytes of the iteration's full merge.  Only the last family is weighed here:
// j has become j \ n and k has become k \ n with n = j & k a subset of both, and a sampled weight is a sum
// of per-bucket intersection sizes, so |(j \ n) & l| = |j & l| - |n & l| and the same for k: the integers
// the reference counts, by subtraction from the table; j', k' and n are pairwise disjoint, weight 0.
// KSH_REWEIGH=full: all three families weighed, as the reference does (A/B runs; the tests run both).
template <typename WeighFn>
static int reweigh_after_merge(int j, int kk, int n, std::map<std::pair<int, int>, int64_t>* weights, WeighFn weigh) {
  static const bool full = [] {
    const char* e = getenv("KSH_REWEIGH");
    return e && std::string(e) == "full";
  }();
  std::vector<std::pair<int, int>> pairs;
  std::vector<int64_t> w;
  if (full) {
    for (int l = 0; l < n; l++)
      if (j != l) pairs.emplace_back(std::min(j, l), std::max(j, l));
    for (int l = 0; l < n; l++)
      if (kk != l) pairs.emplace_back(std::min(kk, l), std::max(kk, l));
    for (int l = 0; l < n; l++) pairs.emplace_back(l, n);
    KSH_TRY(weigh(pairs, &w));
    for (size_t q = 0; q < pairs.size(); q++) (*weights)[pairs[q]] = w[q];
    return KSH_OK;
  }
  for (int l = 0; l < n; l++)
    if (l != j && l != kk) pairs.emplace_back(l, n);
  KSH_TRY(weigh(pairs, &w));
  for (size_t q = 0; q < pairs.size(); q++) {
    const int l = pairs[q].first;
    (*weights)[pairs[q]] = w[q];
    (*weights)[{std::min(j, l), std::max(j, l)}] -= w[q];
    (*weights)[{std::min(kk, l), std::max(kk, l)}] -= w[q];
  }
  (*weights)[{std::min(j, kk), std::max(j, kk)}] = 0;
  (*weights)[{j, n}] = 0;
  (*weights)[{kk, n}] = 0;
  return KSH_OK;
}


// Weight() of the node's SPSS without the SPSS: the encode's plan (unitigs, path cover, string layout)
// gives n_strings and n_bases; the bases are not emitted.  For nodes the loop will merge again before it
// ends: their strings would be thrown away (the convergence checks only read Weight(), kmer_set_set.h:287).
static int weigh_set(ksh_ctx* ctx, const ksh_geom* g, const KssSet& s, int canonical_flag, KssCompact* out) {
  const ksh_set_view v = view_of(s);
  int64_t ns = 0, nbases = 0;
  KSH_TRY(ksh_spss_encode_plan(ctx, g, &v, canonical_flag, 0, &ns, &nbases));
  *out = KssCompact{};
  out->n_strings = ns;
  out->n_bases = nbases;
  out->size = s.n;
  out->valid = true;
  out->weight_only = true;
  return KSH_OK;
}

// ---------------------------------------------------------------------------------- owner-sharded
// sizes[b] = keys of bucket b if it is sampled, else 0 (the scan of it = the sample's offsets)
__global__ __launch_bounds__(256) void k_sample_sizes(const int64_t* __restrict__ off, const uint8_t* __restrict__ flag,
                                                       int64_t nb, int64_t* __restrict__ sizes) {
  const int64_t b = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (b >= nb) return;
  sizes[b] = flag[b] ? off[b + 1] - off[b] : 0;
}

// one workgroup per sampled bucket: its keys to their place in the sample
__global__ __launch_bounds__(256) void k_sample_copy(const int64_t* __restrict__ off, const int64_t* __restrict__ off_s,
                                                      const int32_t* __restrict__ ids, int key_bytes,
                                                      const char* __restrict__ keys, char* __restrict__ keys_s) {
  const int64_t b = ids[blockIdx.x];
  const int64_t bytes = (off[b + 1] - off[b]) * key_bytes;
  const char* src = keys + off[b] * key_bytes;
  char* dst = keys_s + off_s[b] * key_bytes;
  if (key_bytes >= 4) {
    for (int64_t i = int64_t(threadIdx.x) * 4; i < bytes; i += 256 * 4)
      *reinterpret_cast<uint32_t*>(dst + i) = *reinterpret_cast<const uint32_t*>(src + i);
  } else {  // 2-byte keys: a bucket's bytes need not be a multiple of four, nor its start four-byte aligned
    for (int64_t i = int64_t(threadIdx.x) * 2; i < bytes; i += 256 * 2)
      *reinterpret_cast<uint16_t*>(dst + i) = *reinterpret_cast<const uint16_t*>(src + i);
  }
}

static size_t a16(size_t x) { return (x + 15) & ~size_t(15); }

// All-gather of `count` int64 per rank between host vectors, through device staging allocated once per
// build.  Every rank's status word travels with its values: if any rank has failed, ALL ranks return an
// error from here (the failed rank its own, the others "rank r failed"), so nobody goes on alone.
static int gather_i64(ksh_kss* k, const std::vector<int64_t>& send, std::vector<int64_t>* recv) {
  ksh_ctx* ctx = k->ctx;
  const size_t count = send.size() + 1, world = size_t(k->world);
  recv->assign(send.size() * world, 0);
  if (count > k->gx_cap) {  // (sized for the build's largest gather when it starts; growing may fail: then the rank leaves, see comm_abort)
    void *a = nullptr, *b = nullptr;
    KSH_TRY(pool_alloc(ctx, count * 2 * 8, &a));
    if (const int rc = pool_alloc(ctx, count * 2 * 8 * world, &b); rc != KSH_OK) {
      pool_free(ctx, a);
      return rc;
    }
    pool_free(ctx, k->gx_send);
    pool_free(ctx, k->gx_recv);
    k->gx_send = static_cast<int64_t*>(a);
    k->gx_recv = static_cast<int64_t*>(b);
    k->gx_cap = count * 2;
  }
  std::vector<int64_t> out(send), in(count * world);
  out.push_back(int64_t(k->local_rc));
  KSH_HIP(hipMemcpyAsync(k->gx_send, out.data(), count * 8, hipMemcpyHostToDevice, ctx->stream));
  KSH_TRY(comm_allgather(k->comm, k->gx_send, k->gx_recv, count * 8));
  KSH_HIP(hipMemcpyAsync(in.data(), k->gx_recv, count * 8 * world, hipMemcpyDeviceToHost, ctx->stream));
  KSH_HIP(hipStreamSynchronize(ctx->stream));
  k->gather_bytes += int64_t(count * 8);
  int failed = -1;
  for (size_t r = 0; r < world; r++) {
    std::copy(in.begin() + r * count, in.begin() + r * count + send.size(), recv->begin() + r * send.size());
    if (in[r * count + send.size()] != 0 && failed < 0) failed = int(r);
  }
  if (failed >= 0) {
    k->failed_together = true;
    if (!alive(k)) return fail(k->local_rc, "%s", k->local_msg.c_str());
    return fail(KSH_INTERNAL, "rank %d failed (status %lld); every rank gives the build up", failed,
                (long long)in[size_t(failed) * count + send.size()]);
  }
  return KSH_OK;
}

// The sampled buckets of a resident set as a set of their own (offsets over all 2^N buckets, the
// others empty): GetSampledKmerSet (kmer_set_compact.h:120-203) as a slice.
static int extract_sample(ksh_kss* k, const KssSet& full, const uint8_t* d_flag, const int32_t* d_ids, int32_t n_ids,
                          int64_t* d_off_s, int64_t* n_keys) {
  ksh_ctx* ctx = k->ctx;
  const int64_t nb = n_buckets(&k->g);
  hipLaunchKernelGGL(k_sample_sizes, dim3(unsigned((nb + 255) / 256)), dim3(256), 0, ctx->stream, full.off, d_flag, nb,
                     d_off_s);
  KSH_TRY(scan_exclusive_i64(ctx, d_off_s, d_off_s, nb, d_off_s + nb));
  KSH_HIP(hipMemcpyAsync(ctx->h_pinned, d_off_s + nb, 8, hipMemcpyDeviceToHost, ctx->stream));
  KSH_HIP(hipStreamSynchronize(ctx->stream));
  *n_keys = ctx->h_pinned[0];
  (void)d_ids;
  (void)n_ids;
  return KSH_OK;
}

// ... as a set of its own in pooled buffers.
static int make_sample(ksh_kss* k, const KssSet& full, const uint8_t* d_flag, const int32_t* d_ids, int32_t n_ids,
                       KssSet* out) {
  ksh_ctx* ctx = k->ctx;
  KSH_TRY(alloc_offsets(ctx, &k->g, out));
  int64_t n = 0;
  KSH_TRY(extract_sample(k, full, d_flag, d_ids, n_ids, out->off, &n));
  KSH_TRY(alloc_keys(ctx, &k->g, n, out));
  if (n_ids > 0 && n > 0)
    hipLaunchKernelGGL(k_sample_copy, dim3(unsigned(n_ids)), dim3(256), 0, ctx->stream, full.off, out->off, d_ids,
                       k->g.key_bytes, static_cast<const char*>(full.keys), static_cast<char*>(out->keys));
  out->n = n;
  return KSH_OK;
}

// A set goes to `peer`: its offsets, then its keys.  A rank that does not have the set (it failed to build
// it) sends an empty one in its place.  (Two operations, not one ncclGroupStart / End group: the receiver
// learns the number of keys from the offsets and cannot post the second receive before the first has
// arrived; a step of this protocol moves data one way between two ranks -- a rank either sends or
// receives in it -- so ungrouped operations cannot wait for each other crosswise.)
static int send_set_on(ksh_kss* k, const KssSet& s, int peer, bool side) {
  const int64_t nb = n_buckets(&k->g);
  const int64_t* off = s.off ? s.off : k->zero_off;
  const int64_t n = s.off ? s.n : 0;
  if (side) {
    KSH_TRY(comm_side_send(k->comm, off, size_t(nb + 1) * 8, peer));
    KSH_TRY(comm_side_send(k->comm, s.keys, size_t(n) * k->g.key_bytes, peer));
  } else {
    KSH_TRY(comm_send(k->comm, off, size_t(nb + 1) * 8, peer));
    KSH_TRY(comm_send(k->comm, s.keys, size_t(n) * k->g.key_bytes, peer));
  }
  k->p2p_bytes_sent += (nb + 1) * 8 + n * k->g.key_bytes;
  k->p2p_sets++;
  return KSH_OK;
}
static int send_set(ksh_kss* k, const KssSet& s, int peer) { return send_set_on(k, s, peer, false); }

// A set arrives from `peer` (side: over the side channel, the context's stream picks the data up when it
// has arrived).  A rank that cannot hold it -- an allocation fails: the rank is poisoned -- still takes
// the bytes off the wire, into its encode scratch (which it will not use again), so that the sender is
// not left in its send; *out stays empty then.
static int recv_set_on(ksh_kss* k, int peer, KssSet* out, bool side) {
  ksh_ctx* ctx = k->ctx;
  const int64_t nb = n_buckets(&k->g);
  *out = KssSet{};
  if (side) {
    KSH_TRY(comm_side_recv(k->comm, k->xfer_off, size_t(nb + 1) * 8, peer));
    KSH_TRY(comm_side_join_main(k->comm));
  } else {
    KSH_TRY(comm_recv(k->comm, k->xfer_off, size_t(nb + 1) * 8, peer));
  }
  KSH_HIP(hipMemcpyAsync(ctx->h_pinned, k->xfer_off + nb, 8, hipMemcpyDeviceToHost, ctx->stream));
  KSH_HIP(hipStreamSynchronize(ctx->stream));
  const int64_t n = ctx->h_pinned[0];
  KssSet s;
  KSH_LOCAL(k, alloc_offsets(ctx, &k->g, &s));
  KSH_LOCAL(k, alloc_keys(ctx, &k->g, n, &s));
  void* dst = s.keys;
  if (!alive(k)) {
    free_set(ctx, &s);
    const size_t need = size_t(n) * k->g.key_bytes;
    dst = ctx->slot[kSlotEncode];
    if (need > ctx->slot_bytes[kSlotEncode])
      return fail(KSH_INTERNAL, "rank %d cannot take %zu bytes sent to it: %s", k->rank, need, k->local_msg.c_str());
  } else if (hipMemcpyAsync(s.off, k->xfer_off, size_t(nb + 1) * 8, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess) {
    free_set(ctx, &s);
    return fail(KSH_INTERNAL, "hipMemcpyAsync of the received offsets failed");
  }
  int rc = KSH_OK;
  if (side) {
    rc = comm_side_recv(k->comm, dst, size_t(n) * k->g.key_bytes, peer);
    if (rc == KSH_OK) rc = comm_side_join_main(k->comm);
  } else {
    rc = comm_recv(k->comm, dst, size_t(n) * k->g.key_bytes, peer);
  }
  if (rc != KSH_OK) {  // (the transport failed: the set that was to take the keys goes back to the pool)
    free_set(ctx, &s);
    return rc;
  }
  if (!alive(k)) KSH_HIP(hipStreamSynchronize(ctx->stream));  // (the scratch is free again when this returns)
  k->p2p_bytes_received += (nb + 1) * 8 + n * k->g.key_bytes;
  s.n = alive(k) ? n : 0;
  *out = s;
  return KSH_OK;
}
static int recv_set(ksh_kss* k, int peer, KssSet* out) { return recv_set_on(k, peer, out, false); }
static int recv_set_side(ksh_kss* k, int peer, KssSet* out) { return recv_set_on(k, peer, out, true); }

// Weight table over the samples (every rank computes all of it: 2 % of the data, no exchange).
static int sample_weights(ksh_kss* k, const std::vector<int32_t>& ids, const std::vector<std::pair<int, int>>& pairs,
                          std::vector<int64_t>* out) {
  std::vector<ksh_set_view> views;
  for (const KssSet& s : k->samples) views.push_back(view_of(s));
  out->assign(pairs.size(), 0);
  std::vector<int32_t> flat;
  for (const auto& p : pairs) {
    flat.push_back(p.first);
    flat.push_back(p.second);
  }
  if (!pairs.empty())
    KSH_TRY(ksh_pair_weights(k->ctx, &k->g, views.data(), int32_t(views.size()), ids.data(), int32_t(ids.size()),
                             flat.data(), int32_t(pairs.size()), out->data()));
  return KSH_OK;
}

// The same table with the pair list dealt out: the pairs are independent (kmer_set_set.h:205-216,385-425),
// every rank weighs a contiguous share on its replica of the samples and ONE all-gather of int64 per
// iteration puts the table on all of them -- the exchange north_star names (KSH_OWNED_WEIGHTS=sharded; the
// default computes all of it on every rank and exchanges nothing: at 2 % of the data the all-gather's
// latency, and the lock step it puts the ranks in, cost more than the weighing).  A rank
// whose share fails reports -1 slots and its status in the same exchange.
static int sample_weights_sharded(ksh_kss* k, const std::vector<int32_t>& ids, const std::vector<std::pair<int, int>>& pairs,
                                  std::vector<int64_t>* out) {
  out->assign(pairs.size(), 0);
  if (pairs.empty()) return KSH_OK;
  const size_t world = size_t(k->world), per = (pairs.size() + world - 1) / world;
  const size_t lo = std::min(pairs.size(), per * size_t(k->rank)), hi = std::min(pairs.size(), lo + per);
  std::vector<int64_t> send(per, -1), recv;
  if (hi > lo && alive(k)) {
    std::vector<ksh_set_view> views;
    for (const KssSet& s : k->samples) views.push_back(view_of(s));
    std::vector<int32_t> flat;
    for (size_t q = lo; q < hi; q++) {
      flat.push_back(pairs[q].first);
      flat.push_back(pairs[q].second);
    }
    std::vector<int64_t> mine(hi - lo, 0);
    poison(k, ksh_pair_weights(k->ctx, &k->g, views.data(), int32_t(views.size()), ids.data(), int32_t(ids.size()),
                               flat.data(), int32_t(hi - lo), mine.data()));
    if (alive(k)) std::copy(mine.begin(), mine.end(), send.begin());
  }
  KSH_TRY(gather_i64(k, send, &recv));
  k->weight_gathers++;
  for (size_t q = 0; q < pairs.size(); q++) {
    const int64_t w = recv[(q / per) * per + q % per];
    if (w < 0) return fail(KSH_INTERNAL, "rank %zu did not report the weight of pair %zu", q / per, q);
    (*out)[q] = w;
  }
  return KSH_OK;
}

static void free_sample(ksh_kss* k, size_t i) {
  if (k->sample_pooled[i]) {
    retire(k, k->samples[i].off);
    retire(k, k->samples[i].keys);
  }
  k->samples[i] = KssSet{};
  k->sample_pooled[i] = false;
}

// Encodes the stale nodes this rank owns; one all-gather tells every rank (n_strings, n_bases,
// size) of every stale node.
static int ensure_compacts_owned(ksh_kss* k) {
  std::vector<size_t> stale;
  for (size_t i = 0; i < k->compacts.size(); i++)
    if (!k->compacts[i].valid) stale.push_back(i);
  if (stale.empty()) return KSH_OK;
  std::vector<int64_t> send(3 * stale.size(), -1), recv;
  for (size_t q = 0; q < stale.size(); q++) {
    const size_t i = stale[q];
    KssCompact c;
    c.holder = k->owner[i];
    if (k->owner[i] == k->rank && alive(k) && k->sets[i].off) {
      // a failure must not keep this rank from the all-gather: its slots stay -1, every rank errors out after it
      poison(k, encode_set(k->ctx, &k->g, k->sets[i], k->canonical, &c));
      if (alive(k)) {
        k->n_encodes++;
        k->n_encoded_kmers += k->sets[i].n;
        send[3 * q] = c.n_strings;
        send[3 * q + 1] = c.n_bases;
        send[3 * q + 2] = c.size;
      } else {
        c = KssCompact{};
      }
      c.holder = k->rank;
    }
    k->compacts[i] = c;
  }
  KSH_TRY(gather_i64(k, send, &recv));  // (returns the error on every rank if one of them has failed)
  for (size_t q = 0; q < stale.size(); q++) {
    KssCompact& c = k->compacts[stale[q]];
    const int64_t* from = recv.data() + size_t(c.holder) * send.size() + 3 * q;
    if (from[0] < 0 || from[1] < 0 || from[2] < 0)
      return fail(KSH_INTERNAL, "rank %d did not report node %zu", c.holder, stale[q]);
    c.n_strings = from[0];
    c.n_bases = from[1];
    c.size = from[2];
    c.valid = true;
  }
  return KSH_OK;
}

static int build_owned(ksh_kss* k, const ksh_spss_view* inputs, int32_t n_inputs, const int32_t* owners,
                       const std::vector<int32_t>& ids, int32_t max_iterations) {
  ksh_ctx* ctx = k->ctx;
  const ksh_geom* g = &k->g;
  const int64_t nb = n_buckets(g);
  const int rank = k->rank, world = k->world;
  k->owned = true;
  k->owner.assign(owners, owners + n_inputs);

  // ---- what the exchange protocol itself needs, allocated before anything can fail: the staging of the
  // small all-gathers, the two deferred check exchanges, an empty set's offsets and a landing place for
  // incoming ones.  (A rank that cannot even get these leaves through comm_abort.)
  {
    k->gx_cap = std::max<size_t>(4096, 8 * size_t(n_inputs) + 64);
    k->xchg_cap = 3 * (4 * size_t(n_inputs) + 64) + 1;
    KSH_TRY(pool_alloc(ctx, k->gx_cap * 8, reinterpret_cast<void**>(&k->gx_send)));
    KSH_TRY(pool_alloc(ctx, k->gx_cap * 8 * size_t(world), reinterpret_cast<void**>(&k->gx_recv)));
    KSH_TRY(pool_alloc(ctx, size_t(nb + 1) * 8, reinterpret_cast<void**>(&k->zero_off)));
    KSH_TRY(pool_alloc(ctx, size_t(nb + 1) * 8, reinterpret_cast<void**>(&k->xfer_off)));
    KSH_HIP(hipMemsetAsync(k->zero_off, 0, size_t(nb + 1) * 8, ctx->stream));
    for (ksh_kss::Exchange& x : k->xchg) {
      KSH_TRY(pool_alloc(ctx, k->xchg_cap * 8, reinterpret_cast<void**>(&x.d_send)));
      KSH_TRY(pool_alloc(ctx, k->xchg_cap * 8 * size_t(world), reinterpret_cast<void**>(&x.d_recv)));
      KSH_HIP(hipHostMalloc(reinterpret_cast<void**>(&x.h_recv), k->xchg_cap * 8 * size_t(world), hipHostMallocDefault));
      KSH_HIP(hipEventCreateWithFlags(&x.arrived, hipEventDisableTiming));
    }
  }
  // KSH_FAIL_INJECT=rank:skip:min_bytes (tests): on that rank, once `skip` allocations of at least min_bytes
  // have succeeded, every further one fails -- a GPU that has run out of memory for large blocks
  if (const char* e = getenv("KSH_FAIL_INJECT")) {
    int r = -1;
    long long skip = 0, min_bytes = 0;
    if (sscanf(e, "%d:%lld:%lld", &r, &skip, &min_bytes) == 3 && r == rank) {
      ctx->inject_skip = skip;
      ctx->inject_min_bytes = size_t(min_bytes);
    }
  }

  // ---- inputs: decoded by their owners only, several at once on the rank's lanes (§3.7)
  {
    std::vector<size_t> mine;
    int64_t most_bases = 0;
    for (int32_t i = 0; i < n_inputs; i++) {
      KssCompact c;
      c.holder = k->owner[size_t(i)];
      k->sets.emplace_back();
      if (c.holder == rank) {
        if (!inputs[i].d_words && inputs[i].n_bases > 0)
          poison(k, fail(KSH_INVALID_ARGUMENT, "rank %d owns input %d but was not given its container", rank, i));
        c.words = const_cast<uint64_t*>(inputs[i].d_words);
        c.lens = const_cast<uint32_t*>(inputs[i].d_lens);
        c.n_strings = inputs[i].n_strings;
        c.n_bases = inputs[i].n_bases;
        c.owned = false;
        mine.push_back(size_t(i));
        most_bases = std::max(most_bases, inputs[i].n_bases);
      }
      k->compacts.push_back(c);
    }
    PhaseTimer pt(k, 0);
    if (alive(k) && !mine.empty())
      poison(k, run_on_lanes(
                    ctx, largest_first(mine, [&](size_t i) { return inputs[i].n_bases; }),
                    size_t(most_bases) * size_t(g->key_bytes + 3) + (size_t(256) << 20), [](ksh_ctx*) { return int(KSH_OK); },
                    [&](ksh_ctx* lane, size_t i) {
                      KSH_TRY(ksh_spss_size(lane, g, &inputs[i], &k->compacts[i].size));
                      return decode_to_set(lane, g, &inputs[i], k->canonical, &k->sets[i]);
                    }));
    if (!alive(k))
      for (size_t i : mine) free_set(ctx, &k->sets[i]);
  }
  {
    std::vector<int64_t> send(3 * size_t(n_inputs), -1), recv;
    for (int32_t i = 0; i < n_inputs; i++)
      if (k->owner[size_t(i)] == rank) {
        send[3 * size_t(i)] = k->compacts[size_t(i)].n_strings;
        send[3 * size_t(i) + 1] = k->compacts[size_t(i)].n_bases;
        send[3 * size_t(i) + 2] = k->compacts[size_t(i)].size;
      }
    KSH_TRY(gather_i64(k, send, &recv));
    for (int32_t i = 0; i < n_inputs; i++) {
      const int64_t* from = recv.data() + size_t(k->owner[size_t(i)]) * send.size() + 3 * size_t(i);
      k->compacts[size_t(i)].n_strings = from[0];
      k->compacts[size_t(i)].n_bases = from[1];
      k->compacts[size_t(i)].size = from[2];
    }
  }

  // ---- samples of the inputs: extracted by the owners, all-gathered once
  {
    PhaseTimer pt(k, 1);
    std::vector<uint8_t> flag(size_t(nb), 0);
    for (int32_t b : ids) flag[size_t(b)] = 1;
    uint8_t* d_flag = nullptr;
    int32_t* d_ids = nullptr;
    KSH_LOCAL(k, pool_alloc(ctx, size_t(nb), reinterpret_cast<void**>(&d_flag)));
    KSH_LOCAL(k, pool_alloc(ctx, std::max<size_t>(ids.size() * 4, 16), reinterpret_cast<void**>(&d_ids)));
    if (alive(k)) {
      KSH_HIP(hipMemcpyAsync(d_flag, flag.data(), size_t(nb), hipMemcpyHostToDevice, ctx->stream));
      KSH_HIP(hipMemcpyAsync(d_ids, ids.data(), ids.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    }
    // sample offsets of my inputs and their key counts
    std::vector<int64_t*> my_off(size_t(n_inputs), nullptr);
    std::vector<int64_t> my_n(size_t(n_inputs), -1), all_n;
    for (int32_t i = 0; i < n_inputs; i++) {
      if (k->owner[size_t(i)] != rank) continue;
      KSH_LOCAL(k, pool_alloc(ctx, size_t(nb + 1) * 8, reinterpret_cast<void**>(&my_off[size_t(i)])));
      KSH_LOCAL(k, extract_sample(k, k->sets[size_t(i)], d_flag, d_ids, int32_t(ids.size()), my_off[size_t(i)],
                                  &my_n[size_t(i)]));
    }
    KSH_TRY(gather_i64(k, my_n, &all_n));  // (every rank was alive up to here, or all of them return now)
    std::vector<int64_t> n_of(size_t(n_inputs), 0);
    for (int32_t i = 0; i < n_inputs; i++)
      n_of[size_t(i)] = all_n[size_t(k->owner[size_t(i)]) * size_t(n_inputs) + size_t(i)];
    // payload of a rank: its inputs in ascending order, each [offsets (nb + 1) x 8][keys, padded to 16]
    std::vector<size_t> payload(size_t(world), 0), at_in_payload(size_t(n_inputs), 0);
    for (int32_t i = 0; i < n_inputs; i++) {
      size_t& p = payload[size_t(k->owner[size_t(i)])];
      at_in_payload[size_t(i)] = p;
      p += a16(size_t(nb + 1) * 8) + a16(size_t(n_of[size_t(i)]) * g->key_bytes + 16);
    }
    size_t slot = 16;
    for (size_t p : payload) slot = std::max(slot, p);
    char* d_send = nullptr;
    KSH_LOCAL(k, pool_alloc(ctx, slot, reinterpret_cast<void**>(&d_send)));
    if (alive(k)) poison(k, device_alloc(ctx, slot * size_t(world), reinterpret_cast<void**>(&k->sample_block)));
    {  // the payload exchange needs its buffers on every rank: agree on that first
      std::vector<int64_t> none, unused;
      KSH_TRY(gather_i64(k, none, &unused));
    }
    for (int32_t i = 0; i < n_inputs; i++) {
      if (k->owner[size_t(i)] != rank) continue;
      char* dst = d_send + at_in_payload[size_t(i)];
      KSH_HIP(hipMemcpyAsync(dst, my_off[size_t(i)], size_t(nb + 1) * 8, hipMemcpyDeviceToDevice, ctx->stream));
      if (!ids.empty())
        hipLaunchKernelGGL(k_sample_copy, dim3(unsigned(ids.size())), dim3(256), 0, ctx->stream,
                           k->sets[size_t(i)].off, my_off[size_t(i)], d_ids, g->key_bytes,
                           static_cast<const char*>(k->sets[size_t(i)].keys), dst + a16(size_t(nb + 1) * 8));
    }
    KSH_TRY(comm_allgather(k->comm, d_send, k->sample_block, slot));
    KSH_HIP(hipStreamSynchronize(ctx->stream));
    k->gather_bytes += int64_t(slot);
    for (int32_t i = 0; i < n_inputs; i++) {
      char* base = k->sample_block + size_t(k->owner[size_t(i)]) * slot + at_in_payload[size_t(i)];
      KssSet s;
      s.off = reinterpret_cast<int64_t*>(base);
      s.keys = base + a16(size_t(nb + 1) * 8);
      s.n = n_of[size_t(i)];
      k->samples.push_back(s);
      k->sample_pooled.push_back(false);
      if (my_off[size_t(i)]) pool_free(ctx, my_off[size_t(i)]);
    }
    pool_free(ctx, d_send);
    pool_free(ctx, d_flag);
    pool_free(ctx, d_ids);
  }

  // The weight tables by pair list + ONE all-gather of int64 per iteration (the exchange north_star names;
  // lib/core/kmer_set_set.h:205-218,385-425): the default since the full-size merges are deferred to the end of
  // their interval (below) and the all-gather no longer sits behind the rank that runs an iteration's merge.
  // KSH_OWNED_WEIGHTS=replicated: every rank weighs its replica of the samples, nothing is exchanged.
  static const bool shard_weights = [] {
    const char* e = getenv("KSH_OWNED_WEIGHTS");
    return !(e && std::string(e) == "replicated");
  }();
  const auto weigh = [&](const std::vector<std::pair<int, int>>& pairs, std::vector<int64_t>* w) {
    return shard_weights && world > 1 ? sample_weights_sharded(k, ids, pairs, w) : sample_weights(k, ids, pairs, w);
  };
  std::map<std::pair<int, int>, int64_t> weights;
  {
    std::vector<std::pair<int, int>> pairs;
    for (int i = 0; i < n_inputs; i++)
      for (int j = i + 1; j < n_inputs; j++) pairs.emplace_back(i, j);
    std::vector<int64_t> w;
    {
      PhaseTimer pt(k, 1);
      KSH_TRY(weigh(pairs, &w));
    }
    for (size_t i = 0; i < pairs.size(); i++) weights[pairs[i]] = w[i];
    k->initial_weights = w;
  }

  int64_t total_size = 0;
  for (const KssCompact& c : k->compacts) total_size += c.size;
  k->initial_total_size = total_size;

  const auto total_spss_weight_now = [&](int64_t* total) {
    PhaseTimer pt(k, 3);
    KSH_TRY(ensure_compacts_owned(k));
    *total = 0;
    for (const KssCompact& c : k->compacts) *total += c.n_bases;
    return KSH_OK;
  };
  int64_t total_spss_weight = 0;
  KSH_TRY(total_spss_weight_now(&total_spss_weight));
  k->initial_spss_weight = total_spss_weight;

  const int interval = int(k->compacts.size() / 8 + 1);
  const float improvement_threshold = 0.1 * interval / k->compacts.size();

  // rows of the trace are known where the merge ran: {j, k, weight, original_size, size_diff};
  // the executing rank of every iteration is the same on all ranks
  std::vector<int64_t> my_rows;
  std::vector<int> executor;

  // ---- deferred convergence checks.  A check only decides whether the loop stops; its weights are
  // not needed to go on.  At a check every rank encodes the stale nodes it owns and, instead of
  // waiting for the others, goes on with the next interval (control, merges, and at the next check
  // its encodes); the exchange of check c happens when the ranks meet at check c + 1.  A rank with
  // little to encode at check c thus works on c + 1 while a loaded rank catches up: per pair of
  // checks the ranks wait for max(load_c + load_c+1) instead of max(load_c) + max(load_c+1)
  // (DESIGN.md 7.3).  If check c says "stop", the interval run in the meantime is undone: the state
  // at the check is kept (vectors copied, device buffers parked instead of freed) until it resolves.
  // KSH_OWNED_LOOKAHEAD=0: resolve every check on the spot (the schedule of 7.1 as first built).
  static const bool lookahead = [] {
    const char* e = getenv("KSH_OWNED_LOOKAHEAD");
    return !(e && e[0] == '0');
  }();
  struct Pending {
    bool active = false;
    int id = 0, iteration = 0;
    std::vector<size_t> stale;
    std::vector<int64_t> send;
    // the state at the check
    std::vector<KssSet> sets, samples;
    std::vector<bool> sample_pooled;
    std::vector<KssCompact> compacts;
    std::vector<int> owner, pending_of;
    std::map<int, std::vector<int>> children;
    std::map<std::pair<int, int>, int64_t> weights;
    size_t n_rows = 0, n_exec = 0;
    std::unordered_set<void*> alive;
    std::vector<void*> graveyard;
    // its exchange, started at the check, waited for at the next one: k->xchg[id & 1]
  } pend;
  std::vector<int> pending_of(k->compacts.size(), -1);  // the unresolved check that covers a node's SPSS, or -1
  int n_checks = 0;
  bool stopped = false;

  const auto collect_alive = [&](std::unordered_set<void*>* out) {
    for (const KssSet& st : k->sets) {
      if (st.off) out->insert(st.off);
      if (st.keys) out->insert(st.keys);
    }
    for (size_t q = 0; q < k->samples.size(); q++)
      if (k->sample_pooled[q]) {
        out->insert(k->samples[q].off);
        out->insert(k->samples[q].keys);
      }
    for (const KssCompact& c : k->compacts)
      if (c.owned) {
        out->insert(c.words);
        out->insert(c.lens);
      }
  };

  // ---- who encodes what at a check.  A node is encoded where it lives, unless that rank is behind: the
  // nine merges of an interval fall on eight ranks unevenly (and each leaves its large child where it
  // ran), so a check's encodes are dealt out again -- the most loaded rank hands its largest node to the
  // least loaded one while that shortens the longer of the two, a 10^8-k-mer set travels in 6 ms and
  // encodes in 26 -- and the node then lives on its encoder.  Every rank computes the same deal from
  // what all of them know: the sample sizes (2 % of the set sizes, up to noise) and, with deferred
  // checks, how far behind each rank already is (lag: its load of the last checks beyond the
  // lightest rank's).  A rank either gives or takes in one check, never both, so the transfers cannot
  // wait on each other: givers send first (on the transport's side stream, under their own encodes),
  // takers encode their own nodes first and then what arrives.  KSH_OWNED_MIGRATE=0 turns it off.
  static const bool migrate = [] {
    const char* e = getenv("KSH_OWNED_MIGRATE");
    return !(e && e[0] == '0');
  }();
  std::vector<double> lag(size_t(world), 0.0);

  // encodes the stale nodes this rank owns (after the deal); fills stale / send
  // protocol_only: the check before this one is already known to stop the loop (peek_stop below), so whatever
  // is encoded here goes with the roll-back -- the rank still takes part in every hand-over of the deal (its
  // peers may not know yet and are waiting for its sends and receives) but skips the encodes themselves
  const auto encode_stale = [&](Pending* pd, bool protocol_only) {
    PhaseTimer pt(k, 3);
    pd->stale.clear();
    for (size_t q = 0; q < k->compacts.size(); q++)
      if (!k->compacts[q].valid && pending_of[q] < 0) pd->stale.push_back(q);
    pd->send.assign(3 * pd->stale.size(), -1);
    const size_t n_tasks = pd->stale.size();
    std::vector<int> encoder(n_tasks);
    std::vector<double> cost(n_tasks);
    std::vector<double> load = lag;
    for (size_t q = 0; q < n_tasks; q++) {
      encoder[q] = k->owner[pd->stale[q]];
      cost[q] = double(k->samples[pd->stale[q]].n) + 1.0;
      load[size_t(encoder[q])] += cost[q];
    }
    if (migrate && lookahead && world > 1) {
      std::vector<int> role(size_t(world), 0);  // +1 gives, -1 takes
      for (size_t round = 0; round < n_tasks; round++) {
        int o = 0, m = 0;
        for (int r = 1; r < world; r++) {
          if (load[size_t(r)] > load[size_t(o)]) o = r;
          if (load[size_t(r)] < load[size_t(m)]) m = r;
        }
        if (o == m || role[size_t(o)] < 0 || role[size_t(m)] > 0) break;
        // the largest node of o that still shortens the longer of the two (a quarter of its cost for the trip)
        int best = -1;
        for (size_t q = 0; q < n_tasks; q++)
          if (encoder[q] == o && k->owner[pd->stale[q]] == o && load[size_t(m)] + 1.25 * cost[q] < load[size_t(o)] &&
              (best < 0 || cost[q] > cost[size_t(best)]))
            best = int(q);
        if (best < 0) break;
        encoder[size_t(best)] = m;
        load[size_t(o)] -= cost[size_t(best)];
        load[size_t(m)] += 1.25 * cost[size_t(best)];
        role[size_t(o)] = 1;
        role[size_t(m)] = -1;
      }
    }
    {
      double lightest = load[0];
      for (double l : load) lightest = std::min(lightest, l);
      for (int r = 0; r < world; r++) lag[size_t(r)] = lookahead ? load[size_t(r)] - lightest : 0.0;
    }
    // books an encoded node (c.valid: the encode ran and succeeded)
    const auto book_node = [&](size_t q, KssCompact c) {
      const size_t node = pd->stale[q];
      if (c.valid) {
        k->n_encodes++;
        k->n_encoded_kmers += k->sets[node].n;
        pd->send[3 * q] = c.n_strings;
        pd->send[3 * q + 1] = c.n_bases;
        pd->send[3 * q + 2] = c.size;
      } else {
        c = KssCompact{};
      }
      c.valid = false;  // known here, not yet everywhere
      c.holder = rank;
      k->compacts[node] = c;
    };
    const auto encode_node = [&](size_t q) {
      const size_t node = pd->stale[q];
      KssCompact c;
      c.valid = false;
      if (!protocol_only && alive(k) && k->sets[node].off) {  // (a failed rank's slots stay -1; its status travels with them)
        poison(k, encode_set(ctx, g, k->sets[node], k->canonical, &c));
        if (!alive(k)) c = KssCompact{}, c.valid = false;
      }
      book_node(q, c);
    };
    // 1. what I give away leaves first, under my own encodes
    std::vector<size_t> given;
    for (size_t q = 0; q < n_tasks; q++) {
      const size_t node = pd->stale[q];
      if (encoder[q] == k->owner[node]) continue;
      if (k->owner[node] == rank) {
        KSH_TRY(send_set_on(k, k->sets[node], encoder[q], true));
        k->sets_migrated++;
        given.push_back(node);
      }
    }
    // 2. my own nodes: independent encodes, several at once on the context's lanes
    {
      std::vector<size_t> mine;
      int64_t n_max = 0;
      for (size_t q = 0; q < n_tasks; q++)
        if (encoder[q] == rank && k->owner[pd->stale[q]] == rank) {
          mine.push_back(q);
          n_max = std::max(n_max, k->sets[pd->stale[q]].n);
        }
      std::vector<KssCompact> done(n_tasks);
      for (KssCompact& c : done) c.valid = false;
      if (!protocol_only && alive(k) && !mine.empty())
        poison(k, run_on_lanes(
                      ctx, largest_first(mine, [&](size_t q) { return k->sets[pd->stale[q]].n; }),
                      encode_scratch_bytes(g, n_max), [&](ksh_ctx* lane) { return encode_reserve(lane, g, n_max); },
                      [&](ksh_ctx* lane, size_t q) {
                        const size_t node = pd->stale[q];
                        if (!k->sets[node].off) return int(KSH_OK);
                        const int rc = encode_set(lane, g, k->sets[node], k->canonical, &done[q]);
                        if (rc != KSH_OK) done[q].valid = false;
                        return rc;
                      }));
      for (size_t q : mine) book_node(q, done[q]);
    }
    // 3. what I take, as it arrives
    for (size_t q = 0; q < n_tasks; q++) {
      const size_t node = pd->stale[q];
      if (encoder[q] == rank && k->owner[node] != rank) {
        KSH_TRY(recv_set_side(k, k->owner[node], &k->sets[node]));
        encode_node(q);
      }
    }
    if (!given.empty()) {
      KSH_TRY(comm_side_sync(k->comm));
      for (size_t node : given) {
        retire(k, k->sets[node].off);
        retire(k, k->sets[node].keys);
        k->sets[node] = KssSet{};
      }
    }
    for (size_t q = 0; q < n_tasks; q++) {
      const size_t node = pd->stale[q];
      if (encoder[q] != rank) {
        KssCompact c;
        c.valid = false;
        c.holder = encoder[q];
        k->compacts[node] = c;
      }
      k->owner[node] = encoder[q];
      pending_of[node] = pd->id;
    }
    return KSH_OK;
  };

  // a check's exchange starts right after the rank's encodes, on the transport's side channel: it
  // completes when the last rank has got that far, while this one is already in the next interval
  const auto start_exchange = [&](Pending* pd) {
    ksh_kss::Exchange& x = k->xchg[pd->id & 1];
    pd->send.push_back(int64_t(k->local_rc));  // the rank's status word, behind its values
    const size_t bytes = pd->send.size() * 8;
    if (pd->send.size() > k->xchg_cap)
      return fail(KSH_INTERNAL, "a check with %zu stale nodes: more than the exchange buffers hold", pd->stale.size());
    KSH_HIP(hipMemcpyAsync(x.d_send, pd->send.data(), bytes, hipMemcpyHostToDevice, ctx->stream));
    KSH_TRY(comm_side_allgather(k->comm, x.d_send, x.d_recv, bytes));  // (every rank gets here, whatever its encodes did)
    KSH_HIP(hipMemcpyAsync(x.h_recv, x.d_recv, bytes * size_t(world), hipMemcpyDeviceToHost, comm_side_stream(k->comm)));
    KSH_HIP(hipEventRecord(x.arrived, comm_side_stream(k->comm)));
    k->gather_bytes += int64_t(bytes);
    return KSH_OK;
  };
  // waits for a check's exchange; the values rank-major without the status words; an error on EVERY rank
  // if one of them had failed by then
  const auto finish_exchange = [&](Pending* pd, std::vector<int64_t>* recv) {
    ksh_kss::Exchange& x = k->xchg[pd->id & 1];
    const size_t count = pd->send.size(), vals = count - 1;
    recv->assign(vals * size_t(world), 0);
    KSH_HIP(hipEventSynchronize(x.arrived));
    int failed = -1;
    for (size_t r = 0; r < size_t(world); r++) {
      std::copy(x.h_recv + r * count, x.h_recv + r * count + vals, recv->begin() + r * vals);
      if (x.h_recv[r * count + vals] != 0 && failed < 0) failed = int(r);
    }
    pd->send.pop_back();
    if (failed >= 0) {
      k->failed_together = true;
      if (!alive(k)) return fail(k->local_rc, "%s", k->local_msg.c_str());
      return fail(KSH_INTERNAL, "rank %d failed; every rank gives the build up", failed);
    }
    return KSH_OK;
  };

  // Has the exchange of this unresolved check arrived already, and does it stop the loop?  (No state is touched;
  // resolve() takes the same decision from the same numbers afterwards.)  A rank that gets to the next check after
  // the slowest rank of this one -- the slowest itself, or any rank when there is only one -- sees it and spares
  // itself the next check's encodes, which the roll-back would throw away: 24 encodes, 0.04 s of the 64 x 10^8 build.
  const auto peek_stop = [&](const Pending* pd) {
    const ksh_kss::Exchange& x = k->xchg[pd->id & 1];
    if (hipEventQuery(x.arrived) != hipSuccess) {
      (void)hipGetLastError();  // (hipErrorNotReady is no error)
      return false;
    }
    const size_t count = pd->send.size(), vals = count - 1;
    for (size_t r = 0; r < size_t(world); r++)
      if (x.h_recv[r * count + vals] != 0) return false;  // a rank has failed: resolve() reports it
    const std::vector<KssCompact>& at_check = pd->compacts;
    std::vector<int64_t> bases(at_check.size(), -1);
    for (size_t q = 0; q < pd->stale.size(); q++) {
      const int64_t* from = x.h_recv + size_t(at_check[pd->stale[q]].holder) * count + 3 * q;
      if (from[0] < 0 || from[1] < 0 || from[2] < 0) return false;
      bases[pd->stale[q]] = from[1];
    }
    int64_t updated = 0;
    for (size_t q = 0; q < at_check.size(); q++) updated += bases[q] >= 0 ? bases[q] : at_check[q].n_bases;
    const float improvement = static_cast<float>(total_spss_weight - updated) / total_spss_weight;
    return improvement <= improvement_threshold;
  };

  // the decision of an unresolved check; *stop_out: the loop ends at that check
  const auto resolve = [&](Pending* pd, bool* stop_out) {
    PhaseTimer pt(k, 3);
    std::vector<int64_t> recv;
    KSH_TRY(finish_exchange(pd, &recv));
    const std::vector<KssCompact>& at_check = pd->compacts;
    std::vector<const int64_t*> from(pd->stale.size(), nullptr);
    for (size_t q = 0; q < pd->stale.size(); q++) {
      const int holder = at_check[pd->stale[q]].holder;
      from[q] = recv.data() + size_t(holder) * pd->send.size() + 3 * q;
      if (from[q][0] < 0 || from[q][1] < 0 || from[q][2] < 0)
        return fail(KSH_INTERNAL, "rank %d did not report node %zu", holder, pd->stale[q]);
    }
    int64_t updated = 0;
    {
      std::vector<int64_t> bases(at_check.size(), -1);
      for (size_t q = 0; q < pd->stale.size(); q++) bases[pd->stale[q]] = from[q][1];
      for (size_t q = 0; q < at_check.size(); q++) updated += bases[q] >= 0 ? bases[q] : at_check[q].n_bases;
    }
    const float improvement = static_cast<float>(total_spss_weight - updated) / total_spss_weight;
    const bool stop = improvement <= improvement_threshold;
    k->checkpoints.insert(k->checkpoints.end(), {int64_t(pd->iteration), total_spss_weight, updated, int64_t(stop)});
    k->improvements.push_back(improvement);
    k->keep_alive = nullptr;
    k->graveyard = nullptr;
    if (stop) {
      // back to the state at the check: whatever was made since goes, what was parked is in use again
      std::unordered_set<void*> now;
      collect_alive(&now);
      for (void* ptr : now)
        if (!pd->alive.count(ptr)) pool_free(ctx, ptr);
      k->sets = pd->sets;
      k->samples = pd->samples;
      k->sample_pooled = pd->sample_pooled;
      k->compacts = pd->compacts;
      k->owner = pd->owner;
      k->children = pd->children;
      weights = pd->weights;
      pending_of = pd->pending_of;
      if (my_rows.size() > pd->n_rows || executor.size() > pd->n_exec) k->rollbacks++;
      my_rows.resize(pd->n_rows);
      executor.resize(pd->n_exec);
    } else {
      total_spss_weight = updated;
      for (void* ptr : pd->graveyard) pool_free(ctx, ptr);
    }
    for (size_t q = 0; q < pd->stale.size(); q++) {
      const size_t node = pd->stale[q];
      if (pending_of[node] != pd->id) continue;  // merged again since: a later check covers its new version
      KssCompact& c = k->compacts[node];
      c.n_strings = from[q][0];
      c.n_bases = from[q][1];
      c.size = from[q][2];
      c.valid = true;
      pending_of[node] = -1;
    }
    pd->graveyard.clear();
    pd->alive.clear();
    pd->active = false;
    *stop_out = stop;
    return KSH_OK;
  };

  // ---- the full-size merges of an interval, deferred to its end.  Nothing the control loop decides reads the
  // full sets (the arg-max reads sample weights, the samples of a merge's results are the merge of the samples),
  // so an iteration only RECORDS its merge -- who runs it (the owner of j), who sends k -- and every rank works
  // its part of the list off, in iteration order, right before the point where the sets are read: the
  // encodes of the next check (or the end).  The order of the sends and receives between any two ranks is the
  // list's order on both, as it was when every merge ran in its iteration.  What it buys: with the weight
  // tables dealt out by pair list (one all-gather per iteration) no rank waits any more for the one that runs
  // the iteration's merge -- the merges of an interval run side by side on their executors, 2-3 per rank
  // instead of 9 one after the other with seven ranks waiting.  KSH_OWNED_MERGES=inline runs
  // every merge in its iteration, as before.
  static const bool defer_merges = [] {
    const char* e = getenv("KSH_OWNED_MERGES");
    return !(e && std::string(e) == "inline");
  }();
  struct MergeTask {
    int iteration, j, kk, n, ex, src;
    size_t row;  // its row of my_rows
  };
  std::vector<MergeTask> merge_tasks;
  const auto retire_set = [&](KssSet* st) {
    retire(k, st->off);
    retire(k, st->keys);
    *st = KssSet{};
  };
  const auto run_merge = [&](const MergeTask& t) {
    const int j = t.j, kk = t.kk, n = t.n, ex = t.ex, src = t.src;
    if (rank == ex) {
      PhaseTimer pt(k, 2);
      KssSet pulled;
      if (src != ex) KSH_TRY(recv_set(k, src, &pulled));
      const KssSet& set_k = src != ex ? pulled : k->sets[size_t(kk)];
      KssSet sn, sj, sk;
      int64_t original_size = -1;
      // rank-local from here: a failure (an allocation under memory pressure) leaves the three results
      // empty, the rank goes on with the protocol and every rank hears of it at the next exchange
      if (alive(k) && !(k->sets[size_t(j)].off && set_k.off))
        poison(k, fail(KSH_INTERNAL, "iteration %d: rank %d does not hold both sets of the pair", t.iteration, rank));
      if (alive(k)) {
        const ksh_set_view vj = view_of(k->sets[size_t(j)]), vk = view_of(set_k);
        original_size = vj.n_keys + vk.n_keys;
        int64_t totals[3] = {0, 0, 0};
        KSH_LOCAL(k, alloc_offsets(ctx, g, &sn));
        KSH_LOCAL(k, alloc_offsets(ctx, g, &sj));
        KSH_LOCAL(k, alloc_offsets(ctx, g, &sk));
        KSH_LOCAL(k, ksh_pair_plan(ctx, g, &vj, &vk, sn.off, sj.off, sk.off, totals));
        KSH_LOCAL(k, alloc_keys(ctx, g, totals[0], &sn));
        KSH_LOCAL(k, alloc_keys(ctx, g, totals[1], &sj));
        KSH_LOCAL(k, alloc_keys(ctx, g, totals[2], &sk));
        KSH_LOCAL(k, ksh_pair_write(ctx, g, &vj, &vk, sn.keys, sj.keys, sk.keys));
        if (!alive(k)) {
          free_set(ctx, &sn);
          free_set(ctx, &sj);
          free_set(ctx, &sk);
        }
      }
      if (src != ex) retire_set(&pulled); else retire_set(&k->sets[size_t(kk)]);
      retire_set(&k->sets[size_t(j)]);
      k->sets[size_t(j)] = sj;
      k->sets[size_t(kk)] = sk;
      k->sets[size_t(n)] = sn;
      int64_t* row = my_rows.data() + 5 * t.row;
      row[3] = original_size;
      row[4] = sn.n + sj.n + sk.n - original_size;
    } else if (rank == src) {
      PhaseTimer pt(k, 2);
      KSH_TRY(send_set(k, k->sets[size_t(kk)], ex));
      KSH_HIP(hipStreamSynchronize(ctx->stream));  // the keys have left before their buffer is reused
      retire_set(&k->sets[size_t(kk)]);
    }
    return int(KSH_OK);
  };
  const auto run_deferred_merges = [&]() {
    for (const MergeTask& t : merge_tasks) KSH_TRY(run_merge(t));
    merge_tasks.clear();
    return int(KSH_OK);
  };

  for (int i = 0;; i++) {
    if (max_iterations >= 0 && i >= max_iterations) break;
    if (i > 0 && i % interval == 0) {
      KSH_TRY(run_deferred_merges());  // the sets the check's encodes read
      Pending next;
      next.id = ++n_checks;
      next.iteration = i;
      KSH_TRY(encode_stale(&next, lookahead && pend.active && peek_stop(&pend)));
      KSH_TRY(start_exchange(&next));
      if (pend.active) {
        bool stop = false;
        KSH_TRY(resolve(&pend, &stop));
        if (stop) {
          stopped = true;  // (this check's encodes went with the roll-back; its exchange is in flight on every rank)
          std::vector<int64_t> unused;
          KSH_TRY(finish_exchange(&next, &unused));
          break;
        }
      }
      // this check: the state to come back to, and what may not be freed until it resolves
      next.sets = k->sets;
      next.samples = k->samples;
      next.sample_pooled = k->sample_pooled;
      next.compacts = k->compacts;
      next.owner = k->owner;
      next.pending_of = pending_of;
      next.children = k->children;
      next.weights = weights;
      next.n_rows = my_rows.size();
      next.n_exec = executor.size();
      collect_alive(&next.alive);
      next.active = true;
      pend = std::move(next);
      k->keep_alive = &pend.alive;
      k->graveyard = &pend.graveyard;
      if (!lookahead) {
        bool stop = false;
        KSH_TRY(resolve(&pend, &stop));
        if (stop) {
          stopped = true;
          break;
        }
      } else {
        k->checks_deferred++;
      }
    }
    const int n = int(k->compacts.size());
    int64_t weight = 0;
    int j = -1, kk = -1;
    for (const auto& p : weights) {
      if (p.second > weight) {
        j = p.first.first;
        kk = p.first.second;
        weight = p.second;
      }
    }
    if (weight == 0) break;

    // the merge on the samples: every rank, its own copy
    {
      PhaseTimer pt(k, 1);
      const ksh_set_view vj = view_of(k->samples[size_t(j)]), vk = view_of(k->samples[size_t(kk)]);
      KssSet sn, sj, sk;
      KSH_TRY(alloc_offsets(ctx, g, &sn));
      KSH_TRY(alloc_offsets(ctx, g, &sj));
      KSH_TRY(alloc_offsets(ctx, g, &sk));
      int64_t totals[3];
      KSH_TRY(ksh_pair_plan(ctx, g, &vj, &vk, sn.off, sj.off, sk.off, totals));
      KSH_TRY(alloc_keys(ctx, g, totals[0], &sn));
      KSH_TRY(alloc_keys(ctx, g, totals[1], &sj));
      KSH_TRY(alloc_keys(ctx, g, totals[2], &sk));
      KSH_TRY(ksh_pair_write(ctx, g, &vj, &vk, sn.keys, sj.keys, sk.keys));
      free_sample(k, size_t(j));
      free_sample(k, size_t(kk));
      k->samples[size_t(j)] = sj;
      k->samples[size_t(kk)] = sk;
      k->sample_pooled[size_t(j)] = k->sample_pooled[size_t(kk)] = true;
      k->samples.push_back(sn);
      k->sample_pooled.push_back(true);
    }

    // the merge on the sets: the owner of j, with k's keys pulled in when they live elsewhere
    const int ex = k->owner[size_t(j)], src = k->owner[size_t(kk)];
    executor.push_back(ex);
    my_rows.insert(my_rows.end(), {int64_t(j), int64_t(kk), weight, -1, 0});
    k->sets.emplace_back();
    {
      const MergeTask task{i, j, kk, n, ex, src, my_rows.size() / 5 - 1};
      if (defer_merges) merge_tasks.push_back(task);
      else KSH_TRY(run_merge(task));
    }
    for (int node : {j, kk}) {
      KssCompact& c = k->compacts[size_t(node)];
      if (c.owned) {
        retire(k, c.words);
        retire(k, c.lens);
      }
      c = KssCompact{};
      c.valid = false;
      c.holder = ex;
      pending_of[size_t(node)] = -1;
    }
    KssCompact cn;
    cn.valid = false;
    cn.holder = ex;
    k->compacts.push_back(cn);
    pending_of.push_back(-1);
    k->owner[size_t(kk)] = ex;
    k->owner.push_back(ex);
    k->children[j].push_back(n);
    k->children[kk].push_back(n);

    {
      PhaseTimer pt(k, 1);
      KSH_TRY(reweigh_after_merge(j, kk, n, &weights, weigh));
    }
  }
  if (pend.active) {  // the loop ended between checks (max_iterations, no common k-mers left): the last check still decides
    bool stop = false;
    KSH_TRY(resolve(&pend, &stop));
    stopped = stopped || stop;
    if (stop) merge_tasks.clear();  // (they belong to the interval that was just undone; the same decision on every rank)
  }
  (void)stopped;
  KSH_TRY(run_deferred_merges());
  KSH_TRY(total_spss_weight_now(&k->final_spss_weight));

  // the trace: every row from the rank that ran its merge
  {
    std::vector<int64_t> all;
    KSH_TRY(gather_i64(k, my_rows, &all));
    k->n_processed = k->initial_total_size;
    for (size_t t = 0; t < executor.size(); t++) {
      const int64_t* row = all.data() + size_t(executor[t]) * my_rows.size() + 5 * t;
      k->trace.insert(k->trace.end(), row, row + 5);
      total_size += row[4];
      k->n_processed += row[3];
    }
  }
  k->final_total_size = total_size;
  KSH_HIP(hipStreamSynchronize(ctx->stream));
  k->meta = serialize_children(k->children);
  return KSH_OK;
}

// SerializeAdjacencyList (kmer_set_set.h:45-56), keys ascending.
static std::string serialize_children(const std::map<int, std::vector<int>>& a) {
  std::string s = std::to_string(a.size());
  for (const auto& p : a) {
    s += ' ' + std::to_string(p.first);
    s += ' ' + std::to_string(p.second.size());
    for (int i : p.second) s += ' ' + std::to_string(i);
  }
  return s;
}

static int build(ksh_kss* k, const ksh_spss_view* inputs, int32_t n_inputs,
                 const std::vector<int32_t>& ids, int32_t max_iterations) {
  ksh_ctx* ctx = k->ctx;
  const ksh_geom* g = &k->g;
  // inputs: keep the caller's containers as the nodes' compacts, decode each once (the reference decodes and
  // samples them on its pool, kmer_set_set.h:138-153: independent jobs, on the context's lanes)
  for (int32_t i = 0; i < n_inputs; i++) {
    KssCompact c;
    c.words = const_cast<uint64_t*>(inputs[i].d_words);
    c.lens = const_cast<uint32_t*>(inputs[i].d_lens);
    c.n_strings = inputs[i].n_strings;
    c.n_bases = inputs[i].n_bases;
    c.owned = false;
    k->compacts.push_back(c);
    k->sets.emplace_back();
  }
  {
    PhaseTimer pt(k, 0);
    std::vector<size_t> all;
    int64_t most_bases = 0;
    for (int32_t i = 0; i < n_inputs; i++) {
      all.push_back(size_t(i));
      most_bases = std::max(most_bases, inputs[i].n_bases);
    }
    KSH_TRY(run_on_lanes(
        ctx, largest_first(all, [&](size_t i) { return inputs[i].n_bases; }),
        size_t(most_bases) * size_t(g->key_bytes + 3) + (size_t(256) << 20),  // intermediate keys + bucket ids, histograms
        [](ksh_ctx*) { return int(KSH_OK); },  // (the decode's scratch is small and grows on demand)
        [&](ksh_ctx* lane, size_t i) {
          KSH_TRY(ksh_spss_size(lane, g, &inputs[i], &k->compacts[i].size));  // KmerSetCompact::Size(): sum of len - K + 1
          return decode_to_set(lane, g, &inputs[i], k->canonical, &k->sets[i]);
        }));
  }

  std::map<std::pair<int, int>, int64_t> weights;
  {
    std::vector<std::pair<int, int>> pairs;
    for (int i = 0; i < n_inputs; i++)
      for (int j = i + 1; j < n_inputs; j++) pairs.emplace_back(i, j);
    std::vector<int64_t> w;
    {
      PhaseTimer pt(k, 1);
      KSH_TRY(pair_weights(k, ids, pairs, &w));
    }
    for (size_t i = 0; i < pairs.size(); i++) weights[pairs[i]] = w[i];
    k->initial_weights = w;
  }

  int64_t total_size = 0;
  for (const KssCompact& c : k->compacts) total_size += c.size;
  k->initial_total_size = total_size;
  k->n_processed = total_size;

  const auto total_spss_weight_now = [&](int64_t* total) {
    PhaseTimer pt(k, 3);
    KSH_TRY(ensure_compacts(k));
    *total = 0;
    for (const KssCompact& c : k->compacts) *total += c.n_bases;
    return KSH_OK;
  };
  int64_t total_spss_weight = 0;
  KSH_TRY(total_spss_weight_now(&total_spss_weight));
  k->initial_spss_weight = total_spss_weight;

  const int interval = int(k->compacts.size() / 8 + 1);
  const float improvement_threshold = 0.1 * interval / k->compacts.size();

  for (int i = 0;; i++) {
    if (max_iterations >= 0 && i >= max_iterations) break;
    if (i > 0 && i % interval == 0) {
      int64_t updated = 0;
      KSH_TRY(total_spss_weight_now(&updated));
      const float improvement = static_cast<float>(total_spss_weight - updated) / total_spss_weight;
      const bool stop = improvement <= improvement_threshold;
      k->checkpoints.insert(k->checkpoints.end(), {int64_t(i), total_spss_weight, updated, int64_t(stop)});
      k->improvements.push_back(improvement);
      if (stop) break;
      total_spss_weight = updated;
    }
    const int n = int(k->compacts.size());
    int64_t weight = 0;
    int j = -1, kk = -1;
    for (const auto& p : weights) {
      if (p.second > weight) {
        j = p.first.first;
        kk = p.first.second;
        weight = p.second;
      }
    }
    if (weight == 0) break;

    const int64_t original_size = k->compacts[j].size + k->compacts[kk].size;
    {
      const ksh_set_view vj = view_of(k->sets[j]), vk = view_of(k->sets[kk]);
      KssSet sn, sj, sk;
      KSH_TRY(alloc_offsets(ctx, g, &sn));
      KSH_TRY(alloc_offsets(ctx, g, &sj));
      KSH_TRY(alloc_offsets(ctx, g, &sk));
      int64_t totals[3];
      {
        PhaseTimer pt(k, 2);
        KSH_TRY(ksh_pair_plan(ctx, g, &vj, &vk, sn.off, sj.off, sk.off, totals));
        KSH_TRY(alloc_keys(ctx, g, totals[0], &sn));
        KSH_TRY(alloc_keys(ctx, g, totals[1], &sj));
        KSH_TRY(alloc_keys(ctx, g, totals[2], &sk));
        KSH_TRY(ksh_pair_write(ctx, g, &vj, &vk, sn.keys, sj.keys, sk.keys));
      }

      KssCompact cn, cj, ck;  // encoded lazily (ensure_compacts)
      cn.valid = cj.valid = ck.valid = false;
      cn.size = sn.n;
      cj.size = sj.n;
      ck.size = sk.n;

      k->sets.push_back(sn);
      k->compacts.push_back(cn);
      free_set(ctx, &k->sets[j]);
      free_compact(ctx, &k->compacts[j]);
      k->sets[j] = sj;
      k->compacts[j] = cj;
      free_set(ctx, &k->sets[kk]);
      free_compact(ctx, &k->compacts[kk]);
      k->sets[kk] = sk;
      k->compacts[kk] = ck;
      k->children[j].push_back(n);
      k->children[kk].push_back(n);
    }
    const int64_t size_diff =
        k->compacts[n].size + k->compacts[j].size + k->compacts[kk].size - original_size;
    total_size += size_diff;
    k->n_processed += original_size;
    k->trace.insert(k->trace.end(), {int64_t(j), int64_t(kk), weight, original_size, size_diff});

    {
      PhaseTimer pt(k, 1);
      KSH_TRY(reweigh_after_merge(j, kk, n, &weights, [&](const std::vector<std::pair<int, int>>& pairs, std::vector<int64_t>* w) {
        return pair_weights(k, ids, pairs, w);
      }));
    }
  }
  k->final_total_size = total_size;
  KSH_TRY(total_spss_weight_now(&k->final_spss_weight));
  KSH_HIP(hipStreamSynchronize(ctx->stream));  // the node views are usable when build returns
  k->meta = serialize_children(k->children);
  return KSH_OK;
}

// The single-GPU constructor, control ahead of data.  Every decision of the loop reads sampled buckets
// only (arg-max :308-316, the break on weight 0), and the samples of a merge's results are the merge of
// the samples, so the merge sequence can be computed on the 2 % samples ahead of the full-size work
// (what the loop WILL do unless a convergence check stops it; kept two intervals ahead).  The data
// plane replays it on the resident sets with the convergence checks, and knows at every check which
// stale nodes the loop merges again within that horizon: their SPSS would be thrown away, so only
// their weight is computed (the encode's plan
// without the write: 12 % of an encode, 44 % of the encoded k-mers at 64 x 10^8).  If a check stops the
// loop early, the nodes that were only weighed get their strings then.  Same trace, checkpoints and
// nodes as the statement-by-statement loop (`build`, which the replicated multi-GPU build still uses).
static int build_single(ksh_kss* k, const ksh_spss_view* inputs, int32_t n_inputs, const std::vector<int32_t>& ids,
                        int32_t max_iterations) {
  ksh_ctx* ctx = k->ctx;
  const ksh_geom* g = &k->g;
  const int64_t nb = n_buckets(g);
  for (int32_t i = 0; i < n_inputs; i++) {
    KssCompact c;
    c.words = const_cast<uint64_t*>(inputs[i].d_words);
    c.lens = const_cast<uint32_t*>(inputs[i].d_lens);
    c.n_strings = inputs[i].n_strings;
    c.n_bases = inputs[i].n_bases;
    c.owned = false;
    KSH_TRY(ksh_spss_size(ctx, g, &inputs[i], &c.size));
    k->compacts.push_back(c);
    k->sets.emplace_back();
    PhaseTimer pt(k, 0);
    KSH_TRY(decode_to_set(ctx, g, &inputs[i], k->canonical, &k->sets.back()));
  }

  // ---- the control plane: the merge sequence, on the samples, a bounded distance ahead of the sets
  // (without the convergence checks the loop would go on merging ever smaller intersections long
  // after the point where a check stops it, so the sequence is extended on demand: two intervals
  // beyond what the data plane is about to do)
  struct Step {
    int j, kk;
    int64_t weight;
  };
  std::vector<Step> seq;
  bool seq_complete = false;  // the control loop itself has ended (no common k-mers left / max_iterations)
  std::map<std::pair<int, int>, int64_t> weights;
  {
    PhaseTimer pt(k, 1);
    std::vector<uint8_t> flag(size_t(nb), 0);
    for (int32_t b : ids) flag[size_t(b)] = 1;
    uint8_t* d_flag = nullptr;
    int32_t* d_ids = nullptr;
    KSH_TRY(pool_alloc(ctx, size_t(nb), reinterpret_cast<void**>(&d_flag)));
    KSH_TRY(pool_alloc(ctx, std::max<size_t>(ids.size() * 4, 16), reinterpret_cast<void**>(&d_ids)));
    KSH_HIP(hipMemcpyAsync(d_flag, flag.data(), size_t(nb), hipMemcpyHostToDevice, ctx->stream));
    KSH_HIP(hipMemcpyAsync(d_ids, ids.data(), ids.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    for (int32_t i = 0; i < n_inputs; i++) {
      KssSet smp;
      KSH_TRY(make_sample(k, k->sets[size_t(i)], d_flag, d_ids, int32_t(ids.size()), &smp));
      k->samples.push_back(smp);
      k->sample_pooled.push_back(true);
    }
    KSH_HIP(hipStreamSynchronize(ctx->stream));
    pool_free(ctx, d_flag);
    pool_free(ctx, d_ids);
    std::vector<std::pair<int, int>> pairs;
    for (int i = 0; i < n_inputs; i++)
      for (int j = i + 1; j < n_inputs; j++) pairs.emplace_back(i, j);
    std::vector<int64_t> w;
    KSH_TRY(sample_weights(k, ids, pairs, &w));
    for (size_t i = 0; i < pairs.size(); i++) weights[pairs[i]] = w[i];
    k->initial_weights = w;
  }
  const auto extend_sequence = [&](int upto) {
    PhaseTimer pt(k, 1);
    while (!seq_complete && int(seq.size()) < upto) {
      if (max_iterations >= 0 && int(seq.size()) >= max_iterations) {
        seq_complete = true;
        break;
      }
      const int n = int(k->samples.size());
      int64_t weight = 0;
      int j = -1, kk = -1;
      for (const auto& p : weights)
        if (p.second > weight) {
          j = p.first.first;
          kk = p.first.second;
          weight = p.second;
        }
      if (weight == 0) {
        seq_complete = true;
        break;
      }
      seq.push_back({j, kk, weight});
      const ksh_set_view vj = view_of(k->samples[size_t(j)]), vk = view_of(k->samples[size_t(kk)]);
      KssSet sn, sj, sk;
      KSH_TRY(alloc_offsets(ctx, g, &sn));
      KSH_TRY(alloc_offsets(ctx, g, &sj));
      KSH_TRY(alloc_offsets(ctx, g, &sk));
      int64_t totals[3];
      KSH_TRY(ksh_pair_plan(ctx, g, &vj, &vk, sn.off, sj.off, sk.off, totals));
      KSH_TRY(alloc_keys(ctx, g, totals[0], &sn));
      KSH_TRY(alloc_keys(ctx, g, totals[1], &sj));
      KSH_TRY(alloc_keys(ctx, g, totals[2], &sk));
      KSH_TRY(ksh_pair_write(ctx, g, &vj, &vk, sn.keys, sj.keys, sk.keys));
      free_sample(k, size_t(j));
      free_sample(k, size_t(kk));
      k->samples[size_t(j)] = sj;
      k->samples[size_t(kk)] = sk;
      k->sample_pooled[size_t(j)] = k->sample_pooled[size_t(kk)] = true;
      k->samples.push_back(sn);
      k->sample_pooled.push_back(true);
      KSH_TRY(reweigh_after_merge(j, kk, n, &weights, [&](const std::vector<std::pair<int, int>>& pairs, std::vector<int64_t>* w) {
        return sample_weights(k, ids, pairs, w);
      }));
    }
    return KSH_OK;
  };
  // will the loop, as far as it is known, merge this node at iteration `from` or later?
  const auto merged_again = [&](size_t node, int from) {
    for (size_t t = size_t(from); t < seq.size(); t++)
      if (size_t(seq[t].j) == node || size_t(seq[t].kk) == node) return true;
    return false;
  };

  // ---- the data plane: the sets
  int64_t total_size = 0;
  for (const KssCompact& c : k->compacts) total_size += c.size;
  k->initial_total_size = total_size;
  k->n_processed = total_size;

  // at_iteration: the loop is about to run this iteration (stale nodes merged at or after it are only weighed)
  const auto total_spss_weight_now = [&](int at_iteration, int64_t* total) {
    PhaseTimer pt(k, 3);
    for (size_t q = 0; q < k->compacts.size(); q++) {
      if (k->compacts[q].valid) continue;
      KssCompact c;
      if (merged_again(q, at_iteration)) {
        KSH_TRY(weigh_set(ctx, g, k->sets[q], k->canonical, &c));
        k->n_weighed++;
        k->n_weighed_kmers += k->sets[q].n;
      } else {
        KSH_TRY(encode_set(ctx, g, k->sets[q], k->canonical, &c));
      }
      k->compacts[q] = c;
      k->n_encodes++;
      k->n_encoded_kmers += k->sets[q].n;
    }
    *total = 0;
    for (const KssCompact& c : k->compacts) *total += c.n_bases;
    return KSH_OK;
  };
  int64_t total_spss_weight = 0;
  KSH_TRY(total_spss_weight_now(0, &total_spss_weight));
  k->initial_spss_weight = total_spss_weight;

  const int interval = int(k->compacts.size() / 8 + 1);
  const float improvement_threshold = 0.1 * interval / k->compacts.size();
  int done = 0;  // iterations run
  for (int i = 0;; i++) {
    KSH_TRY(extend_sequence(i + 2 * interval));
    if (i >= int(seq.size())) break;  // the control loop ended here
    if (i > 0 && i % interval == 0) {
      int64_t updated = 0;
      KSH_TRY(total_spss_weight_now(i, &updated));
      const float improvement = static_cast<float>(total_spss_weight - updated) / total_spss_weight;
      const bool stop = improvement <= improvement_threshold;
      k->checkpoints.insert(k->checkpoints.end(), {int64_t(i), total_spss_weight, updated, int64_t(stop)});
      k->improvements.push_back(improvement);
      if (stop) break;
      total_spss_weight = updated;
    }
    const int n = int(k->compacts.size());
    const int j = seq[size_t(i)].j, kk = seq[size_t(i)].kk;
    const int64_t original_size = k->compacts[size_t(j)].size + k->compacts[size_t(kk)].size;
    {
      const ksh_set_view vj = view_of(k->sets[size_t(j)]), vk = view_of(k->sets[size_t(kk)]);
      KssSet sn, sj, sk;
      KSH_TRY(alloc_offsets(ctx, g, &sn));
      KSH_TRY(alloc_offsets(ctx, g, &sj));
      KSH_TRY(alloc_offsets(ctx, g, &sk));
      int64_t totals[3];
      {
        PhaseTimer pt(k, 2);
        KSH_TRY(ksh_pair_plan(ctx, g, &vj, &vk, sn.off, sj.off, sk.off, totals));
        KSH_TRY(alloc_keys(ctx, g, totals[0], &sn));
        KSH_TRY(alloc_keys(ctx, g, totals[1], &sj));
        KSH_TRY(alloc_keys(ctx, g, totals[2], &sk));
        KSH_TRY(ksh_pair_write(ctx, g, &vj, &vk, sn.keys, sj.keys, sk.keys));
      }
      KssCompact cn, cj, ck;
      cn.valid = cj.valid = ck.valid = false;
      cn.size = sn.n;
      cj.size = sj.n;
      ck.size = sk.n;
      k->sets.push_back(sn);
      k->compacts.push_back(cn);
      free_set(ctx, &k->sets[size_t(j)]);
      free_compact(ctx, &k->compacts[size_t(j)]);
      k->sets[size_t(j)] = sj;
      k->compacts[size_t(j)] = cj;
      free_set(ctx, &k->sets[size_t(kk)]);
      free_compact(ctx, &k->compacts[size_t(kk)]);
      k->sets[size_t(kk)] = sk;
      k->compacts[size_t(kk)] = ck;
      k->children[j].push_back(n);
      k->children[kk].push_back(n);
    }
    const int64_t size_diff = k->compacts[size_t(n)].size + k->compacts[size_t(j)].size +
                              k->compacts[size_t(kk)].size - original_size;
    total_size += size_diff;
    k->n_processed += original_size;
    k->trace.insert(k->trace.end(), {int64_t(j), int64_t(kk), seq[size_t(i)].weight, original_size, size_diff});
    done = i + 1;
  }
  k->final_total_size = total_size;
  for (size_t q = 0; q < k->samples.size(); q++) free_sample(k, q);
  k->samples.clear();
  k->sample_pooled.clear();
  // the loop is over: nothing is merged again.  What is stale gets its strings, and so does what was
  // only weighed (a check stopped the loop before the merge that was to come)
  {
    PhaseTimer pt(k, 3);
    for (size_t q = 0; q < k->compacts.size(); q++) {
      KssCompact& c = k->compacts[q];
      if (c.valid && !c.weight_only) continue;
      const bool was_weighed = c.valid && c.weight_only;
      const int64_t bases_before = c.n_bases;
      KssCompact full;
      KSH_TRY(encode_set(ctx, g, k->sets[q], k->canonical, &full));
      if (was_weighed && full.n_bases != bases_before)
        return fail(KSH_INTERNAL, "node %zu: weight %lld when weighed, %lld when encoded", q, (long long)bases_before,
                    (long long)full.n_bases);
      c = full;
      k->n_encodes++;
      k->n_encoded_kmers += k->sets[q].n;
    }
    k->final_spss_weight = 0;
    for (const KssCompact& c : k->compacts) k->final_spss_weight += c.n_bases;
  }
  (void)done;
  KSH_HIP(hipStreamSynchronize(ctx->stream));
  k->meta = serialize_children(k->children);
  return KSH_OK;
}

}  // namespace ksh

using namespace ksh;

extern "C" {

int ksh_kss_build(ksh_ctx* ctx, const ksh_geom* g, const ksh_spss_view* inputs, int32_t n_inputs,
                  const int32_t* bucket_ids, int32_t n_ids, int canonical_flag,
                  int32_t max_iterations, ksh_kss** out) {
  if (!ctx || !out || (n_inputs > 0 && !inputs)) return fail(KSH_INVALID_ARGUMENT, "NULL argument");
  *out = nullptr;
  KSH_TRY(check_geom(g));
  if (n_inputs < 0 || n_ids < 0) return fail(KSH_INVALID_ARGUMENT, "negative count");
  KSH_HIP(hipSetDevice(ctx->device));
  ksh_kss* k = new ksh_kss;
  k->ctx = ctx;
  k->g = *g;
  k->canonical = canonical_flag;
  std::vector<int32_t> ids(bucket_ids, bucket_ids + n_ids);
  // KSH_KSS_LOOP=ahead: the control loop run on the samples ahead of the sets, so that stale nodes the
  // loop merges again are only weighed (build_single).  Measured on 64 x 10^8 (7 checks): it skips the
  // write of 45 % of the encoded k-mers, and gives the time back when the last check stops the loop and
  // the nodes it had only weighed need their strings after all -- 1.97 s against 1.92 s; not the default.
  static const bool control_ahead = [] {
    const char* e = getenv("KSH_KSS_LOOP");
    return e && std::string(e) == "ahead";
  }();
  int rc = KSH_OK;
  if (n_inputs > 0)
    rc = control_ahead ? build_single(k, inputs, n_inputs, ids, max_iterations) : build(k, inputs, n_inputs, ids, max_iterations);
  if (rc != KSH_OK) {
    ksh_kss_destroy(k);
    return rc;
  }
  *out = k;
  return KSH_OK;
}

int ksh_kss_build_sharded(ksh_ctx* ctx, const ksh_geom* g, const ksh_spss_view* inputs, int32_t n_inputs,
                          const int32_t* bucket_ids, int32_t n_ids, int canonical_flag,
                          int32_t max_iterations, int32_t rank, int32_t world, ksh_allgather_i64 gather,
                          void* gather_user, ksh_kss** out) {
  if (!ctx || !out || (n_inputs > 0 && !inputs)) return fail(KSH_INVALID_ARGUMENT, "NULL argument");
  *out = nullptr;
  KSH_TRY(check_geom(g));
  if (n_inputs < 0 || n_ids < 0) return fail(KSH_INVALID_ARGUMENT, "negative count");
  if (world < 1 || rank < 0 || rank >= world) return fail(KSH_INVALID_ARGUMENT, "bad rank / world");
  if (world > 1 && !gather) return fail(KSH_INVALID_ARGUMENT, "a sharded build needs the all-gather callback");
  KSH_HIP(hipSetDevice(ctx->device));
  ksh_kss* k = new ksh_kss;
  k->ctx = ctx;
  k->g = *g;
  k->canonical = canonical_flag;
  k->rank = rank;
  k->world = world;
  k->gather = gather;
  k->gather_user = gather_user;
  std::vector<int32_t> ids(bucket_ids, bucket_ids + n_ids);
  int rc = KSH_OK;
  if (n_inputs > 0) rc = build(k, inputs, n_inputs, ids, max_iterations);
  if (rc != KSH_OK) {
    ksh_kss_destroy(k);
    return rc;
  }
  *out = k;
  return KSH_OK;
}

int ksh_kss_build_owned(ksh_ctx* ctx, ksh_comm* comm, const ksh_geom* g, const ksh_spss_view* inputs,
                        int32_t n_inputs, const int32_t* owners, const int32_t* bucket_ids, int32_t n_ids,
                        int canonical_flag, int32_t max_iterations, ksh_kss** out) {
  if (!ctx || !comm || !out || (n_inputs > 0 && (!inputs || !owners))) return fail(KSH_INVALID_ARGUMENT, "NULL argument");
  *out = nullptr;
  KSH_TRY(check_geom(g));
  if (n_inputs < 0 || n_ids < 0) return fail(KSH_INVALID_ARGUMENT, "negative count");
  const int world = comm_world(comm);
  for (int32_t i = 0; i < n_inputs; i++)
    if (owners[i] < 0 || owners[i] >= world) return fail(KSH_INVALID_ARGUMENT, "owners[%d] = %d is no rank", i, owners[i]);
  KSH_HIP(hipSetDevice(ctx->device));
  ksh_kss* k = new ksh_kss;
  k->ctx = ctx;
  k->g = *g;
  k->canonical = canonical_flag;
  k->rank = comm_rank(comm);
  k->world = world;
  k->comm = comm;
  std::vector<int32_t> ids(bucket_ids, bucket_ids + n_ids);
  int rc = KSH_OK;
  if (n_inputs > 0) rc = build_owned(k, inputs, n_inputs, owners, ids, max_iterations);
  ctx->inject_skip = -1;
  if (rc != KSH_OK) {
    // Either every rank returns this error together, after an exchange that carried a failed rank's status
    // (failed_together) -- or this rank alone cannot follow the protocol any longer (its replica of the control
    // loop or the transport failed): then the transport is torn down, so that its peers' pending operations
    // end in errors rather than wait for it.  Whatever the case, a non-OK return means the job is over: the
    // caller destroys the communicator and its process group.
    const std::string msg = ksh_last_error();
    if (!k->failed_together) comm_abort(comm);
    ksh_kss_destroy(k);
    return fail(rc, "%s", msg.c_str());
  }
  *out = k;
  return KSH_OK;
}

int ksh_kss_comm_stats(const ksh_kss* k, int64_t stats[8]) {
  if (!k || !stats) return fail(KSH_INVALID_ARGUMENT, "NULL argument");
  stats[0] = k->p2p_bytes_sent;
  stats[1] = k->p2p_bytes_received;
  stats[2] = k->p2p_sets;
  stats[3] = k->gather_bytes;
  stats[4] = k->checks_deferred;
  stats[5] = k->rollbacks;
  stats[6] = k->sets_migrated;
  stats[7] = k->weight_gathers;
  return KSH_OK;
}

int ksh_kss_weighed_counts(const ksh_kss* k, int64_t* n_weighed, int64_t* n_weighed_kmers) {
  if (!k || !n_weighed || !n_weighed_kmers) return fail(KSH_INVALID_ARGUMENT, "NULL argument");
  *n_weighed = k->n_weighed;
  *n_weighed_kmers = k->n_weighed_kmers;
  return KSH_OK;
}

int ksh_kss_encode_counts(const ksh_kss* k, int64_t* n_encodes, int64_t* n_encoded_kmers) {
  if (!k || !n_encodes || !n_encoded_kmers) return fail(KSH_INVALID_ARGUMENT, "NULL argument");
  *n_encodes = k->n_encodes;
  *n_encoded_kmers = k->n_encoded_kmers;
  return KSH_OK;
}

int ksh_kss_phase_seconds(const ksh_kss* k, double seconds[4]) {
  if (!k || !seconds) return fail(KSH_INVALID_ARGUMENT, "NULL argument");
  for (int i = 0; i < 4; i++) seconds[i] = k->phase_seconds[i];
  return KSH_OK;
}

int ksh_kss_node_holder(const ksh_kss* k, int32_t i, int32_t* rank) {
  if (!k || !rank) return fail(KSH_INVALID_ARGUMENT, "NULL argument");
  if (i < 0 || size_t(i) >= k->compacts.size()) return fail(KSH_INVALID_ARGUMENT, "no node %d", i);
  *rank = k->compacts[i].holder;
  return KSH_OK;
}

int ksh_kss_destroy(ksh_kss* k) {
  if (!k) return KSH_OK;
  (void)hipSetDevice(k->ctx->device);
  (void)hipStreamSynchronize(k->ctx->stream);
  for (KssSet& s : k->sets) free_set(k->ctx, &s);
  for (KssCompact& c : k->compacts) free_compact(k->ctx, &c);
  for (size_t i = 0; i < k->samples.size(); i++)
    if (k->sample_pooled[i]) {
      pool_free(k->ctx, k->samples[i].off);
      pool_free(k->ctx, k->samples[i].keys);
    }
  if (k->sample_block) (void)hipFree(k->sample_block);
  pool_free(k->ctx, k->gx_send);
  pool_free(k->ctx, k->gx_recv);
  pool_free(k->ctx, k->zero_off);
  pool_free(k->ctx, k->xfer_off);
  for (ksh_kss::Exchange& x : k->xchg) {
    pool_free(k->ctx, x.d_send);
    pool_free(k->ctx, x.d_recv);
    if (x.h_recv) (void)hipHostFree(x.h_recv);
    if (x.arrived) (void)hipEventDestroy(x.arrived);
  }
  delete k;
  return KSH_OK;
}

int ksh_kss_size(const ksh_kss* k, int32_t* n_nodes) {
  if (!k || !n_nodes) return fail(KSH_INVALID_ARGUMENT, "NULL argument");
  *n_nodes = int32_t(k->compacts.size());
  return KSH_OK;
}

int ksh_kss_node(const ksh_kss* k, int32_t i, ksh_spss_view* compact, ksh_set_view* set,
                 int64_t* size) {
  if (!k) return fail(KSH_INVALID_ARGUMENT, "NULL argument");
  if (i < 0 || size_t(i) >= k->compacts.size()) return fail(KSH_INVALID_ARGUMENT, "no node %d", i);
  if (compact) {
    const KssCompact& c = k->compacts[i];
    if (c.holder >= 0 && c.holder != k->rank)
      return fail(KSH_FAILED_PRECONDITION, "the SPSS of node %d is held by rank %d", i, c.holder);
    *compact = view_of(c);
  }
  if (set) {
    if (k->owned && k->owner[size_t(i)] != k->rank)
      return fail(KSH_FAILED_PRECONDITION, "the set of node %d lives on rank %d", i, k->owner[size_t(i)]);
    *set = view_of(k->sets[i]);
  }
  if (size) *size = k->compacts[i].size;
  return KSH_OK;
}

int ksh_kss_children(const ksh_kss* k, int32_t i, const int32_t** children, int32_t* n_children) {
  if (!k || !children || !n_children) return fail(KSH_INVALID_ARGUMENT, "NULL argument");
  auto it = k->children.find(i);
  if (it == k->children.end()) {
    *children = nullptr;
    *n_children = 0;
  } else {
    static_assert(sizeof(int) == sizeof(int32_t), "int is 32 bits");
    *children = reinterpret_cast<const int32_t*>(it->second.data());
    *n_children = int32_t(it->second.size());
  }
  return KSH_OK;
}

const char* ksh_kss_meta(const ksh_kss* k) { return k ? k->meta.c_str() : ""; }

int ksh_kss_trace(const ksh_kss* k, int64_t* n_iterations, const int64_t** rows,
                  int64_t* n_checkpoints, const int64_t** checkpoint_rows,
                  const float** improvements) {
  if (!k) return fail(KSH_INVALID_ARGUMENT, "NULL argument");
  if (n_iterations) *n_iterations = int64_t(k->trace.size() / 5);
  if (rows) *rows = k->trace.data();
  if (n_checkpoints) *n_checkpoints = int64_t(k->checkpoints.size() / 4);
  if (checkpoint_rows) *checkpoint_rows = k->checkpoints.data();
  if (improvements) *improvements = k->improvements.data();
  return KSH_OK;
}

int ksh_kss_initial_weights(const ksh_kss* k, const int64_t** weights, int64_t* n) {
  if (!k || !weights || !n) return fail(KSH_INVALID_ARGUMENT, "NULL argument");
  *weights = k->initial_weights.data();
  *n = int64_t(k->initial_weights.size());
  return KSH_OK;
}

int ksh_kss_stats(const ksh_kss* k, int64_t stats[8]) {
  if (!k || !stats) return fail(KSH_INVALID_ARGUMENT, "NULL argument");
  stats[0] = k->initial_total_size;
  stats[1] = k->final_total_size;
  stats[2] = k->initial_spss_weight;
  stats[3] = k->n_processed;
  stats[4] = k->final_spss_weight;
  int64_t bytes = 0, len_bytes = 0;
  for (const KssCompact& c : k->compacts) {
    bytes += (2 * c.n_bases + 7) / 8;
    if (c.holder >= 0 && c.holder != k->rank) continue;  // sharded build: lengths live on the holder
    int64_t b = 0;
    KSH_TRY(ksh_svb_encode_0124(k->ctx, c.lens, c.n_strings, nullptr, &b));
    len_bytes += b;
  }
  stats[5] = bytes;      // sum over nodes of ceil(2 * Weight / 8)
  stats[6] = len_bytes;  // sum over nodes of |lengths_compressed| (StreamVByte 0124)
  stats[7] = int64_t(k->compacts.size());
  return KSH_OK;
}

// KmerSetSet::Get (kmer_set_set.h:433-454): BFS over children_, union of the reached nodes.
// Returns freshly allocated device buffers (ksh_free them).
int ksh_kss_get(const ksh_kss* k, int32_t i, int64_t** d_offsets, void** d_keys, int64_t* n_keys) {
  if (!k || !d_offsets || !d_keys || !n_keys) return fail(KSH_INVALID_ARGUMENT, "NULL argument");
  if (i < 0 || size_t(i) >= k->sets.size()) return fail(KSH_INVALID_ARGUMENT, "no node %d", i);
  ksh_ctx* ctx = k->ctx;
  const ksh_geom* g = &k->g;
  KSH_HIP(hipSetDevice(ctx->device));
  const int64_t nb = n_buckets(g);
  KssSet acc;
  KSH_TRY(alloc_offsets(ctx, g, &acc));
  KSH_TRY(alloc_keys(ctx, g, 0, &acc));
  KSH_HIP(hipMemsetAsync(acc.off, 0, size_t(nb + 1) * 8, ctx->stream));
  std::queue<int> queue;
  queue.push(i);
  while (!queue.empty()) {
    const int current = queue.front();
    queue.pop();
    if (k->owned && k->owner[size_t(current)] != k->rank) {
      free_set(ctx, &acc);
      return fail(KSH_FAILED_PRECONDITION, "node %d, reachable from %d, lives on rank %d", current, i,
                  k->owner[size_t(current)]);
    }
    const ksh_set_view va = view_of(acc), vb = view_of(k->sets[current]);
    KssSet next;
    KSH_TRY(alloc_offsets(ctx, g, &next));
    int64_t total = 0;
    KSH_TRY(ksh_set_union_plan(ctx, g, &va, &vb, next.off, &total));
    KSH_TRY(alloc_keys(ctx, g, total, &next));
    KSH_TRY(ksh_set_union_write(ctx, g, &va, &vb, next.keys));
    free_set(ctx, &acc);
    acc = next;
    auto it = k->children.find(current);
    if (it != k->children.end())
      for (int child : it->second) queue.push(child);
  }
  // hand the result over in plain hipMalloc'ed buffers (the caller releases them with ksh_free)
  {
    KssSet out;
    KSH_TRY(device_alloc(ctx, size_t(nb + 1) * 8, reinterpret_cast<void**>(&out.off)));
    KSH_TRY(device_alloc(ctx, std::max<size_t>(size_t(acc.n) * g->key_bytes, 16), &out.keys));
    KSH_HIP(hipMemcpyAsync(out.off, acc.off, size_t(nb + 1) * 8, hipMemcpyDeviceToDevice, ctx->stream));
    if (acc.n)
      KSH_HIP(hipMemcpyAsync(out.keys, acc.keys, size_t(acc.n) * g->key_bytes, hipMemcpyDeviceToDevice,
                             ctx->stream));
    KSH_HIP(hipStreamSynchronize(ctx->stream));
    out.n = acc.n;
    free_set(ctx, &acc);
    acc = out;
  }
  *d_offsets = acc.off;
  *d_keys = acc.keys;
  *n_keys = acc.n;
  return KSH_OK;
}

}  // extern "C"
