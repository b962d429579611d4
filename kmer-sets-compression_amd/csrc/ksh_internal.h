// Internal declarations shared by the HIP translation units behind
// include/kmersets_hip.h.  gfx950 only.
#ifndef KSH_INTERNAL_H_
#define KSH_INTERNAL_H_

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <functional>
#include <map>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "kmersets_hip.h"

namespace ksh {

void set_error(const char* fmt, ...);
int fail(int code, const char* fmt, ...);

#define KSH_HIP(expr)                                                                   \
  do {                                                                                  \
    hipError_t err__ = (expr);                                                          \
    if (err__ != hipSuccess)                                                            \
      return ::ksh::fail(KSH_INTERNAL, "%s failed: %s (%s:%d)", #expr,                  \
                         hipGetErrorString(err__), __FILE__, __LINE__);                 \
  } while (0)

// Device key types: the reference's KeyType (lib/core/kmer_set.h:20-31) -- uint16_t for (15, 14), uint32_t for
// (19, 10) and (23, 14), uint64_t for (31, 14) -- chosen by ksh_geom::key_bytes at run time.
#define KSH_BY_KEY(key_bytes, FN, ...)                                   \
  ((key_bytes) == 2 ? FN<uint16_t>(__VA_ARGS__)                          \
                    : (key_bytes) == 4 ? FN<uint32_t>(__VA_ARGS__) : FN<uint64_t>(__VA_ARGS__))

#define KSH_TRY(expr)              \
  do {                             \
    int rc__ = (expr);             \
    if (rc__ != KSH_OK) return rc__; \
  } while (0)

constexpr int kNumTimers = 8;

}  // namespace ksh

// Scratch arena: one growing device buffer, carved by bump allocation and reset
// at the start of every public call that needs scratch.  The pair plan keeps its
// own buffers (they must survive until ksh_pair_write).
struct ksh_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;

  char* arena = nullptr;
  size_t arena_bytes = 0;
  size_t arena_used = 0;

  // pinned host staging for small read-backs
  int64_t* h_pinned = nullptr;  // 64 int64
  int64_t* h_batch = nullptr;   // pinned, grown on demand (ksh_pair_algebra_batch totals)
  size_t h_batch_count = 0;

  // persistent device buffers that must survive between two calls
  // (decode plan -> write, encode plan -> write)
  char* slot[3] = {nullptr, nullptr, nullptr};
  size_t slot_bytes[3] = {0, 0, 0};

  // decode plan state
  int64_t dec_words = 0, dec_groups = 0, dec_kmers = 0, dec_max_bucket = 0;
  const void* dec_src = nullptr;

  // encode plan state (ksh_encode.hip)
  void* enc_state = nullptr;

  // kernels of this context's device that have been granted more than 64 KB of dynamic LDS
  // (hipFuncSetAttribute acts on the current device: once per context, not once per process)
  uint32_t lds_opt_in = 0;

  // text -> SPSS plan state (ksh_text.hip), FASTA -> fragments plan state (ksh_fasta.hip);
  // both use slot kSlotText, so a plan of one kind invalidates a pending plan of the other
  void* text_plan = nullptr;
  void (*text_plan_free)(void*) = nullptr;
  void* fasta_plan = nullptr;
  void (*fasta_plan_free)(void*) = nullptr;
  int text_slot_owner = 0;  // 1: the text plan's arrays are in the slot, 2: the FASTA plan's

  // pair plan (ksh_pair_plan -> ksh_pair_write)
  char* plan = nullptr;
  size_t plan_bytes = 0;
  int64_t plan_tiles = 0;
  const void* plan_a_keys = nullptr;
  const void* plan_b_keys = nullptr;
  int64_t plan_buckets = 0;

  // chained scan (ksh_scan.h): the sums published by the workgroups of the running launch,
  // tagged with the launch's epoch
  unsigned long long* scan_state = nullptr;
  uint32_t scan_epoch = 0;

  // caching allocator for the loop's per-iteration buffers (ksh::pool_alloc / pool_free):
  // freed blocks are kept and reused, because hipMalloc / hipFree of 100 MB-scale blocks
  // cost milliseconds and hipFree synchronises the device.  Single stream, so reuse after
  // free is ordered.
  std::mutex pool_mu;  // (lane jobs allocate their results from their parent's pool: run_on_lanes)
  std::multimap<size_t, void*> pool_free_blocks;
  std::unordered_map<void*, size_t> pool_sizes;
  size_t pool_cached_bytes = 0;
  size_t pool_live_bytes = 0, pool_peak_bytes = 0;  // handed out and not yet given back; its maximum (ksh_ctx_mem_stats)
  // test hook (KSH_FAIL_INJECT, ksh_kss_build_owned): once inject_skip allocations of at least
  // inject_min_bytes have succeeded, every further one fails; -1: off
  long long inject_skip = -1;
  size_t inject_min_bytes = 0;

  // Lanes: helper contexts on the same device, each with a stream, scratch slots and arena of its own, on
  // which independent jobs of one call run side by side with the context's own (ksh::run_on_lanes: the stale
  // nodes of a convergence check are independent encodes, the inputs of a build independent decodes).
  // They live as long as the context: their scratch is mapped once.
  std::vector<ksh_ctx*> lanes;
  ksh_ctx* lane_parent = nullptr;  // set in a lane: where its jobs' results are allocated
  int lanes_wanted = -1;           // ksh_ctx_set_lanes; -1: KSH_LANES or the default
  bool lanes_busy = false;

  // kernel timers: when enabled, every launch of a timed kind gets its own event
  // pair from a pool; ksh_ctx_timing_read sums them after a stream sync.
  bool timing = false;
  int timing_stride = 1;                                 // every n-th launch of a kind is timed
  int64_t timing_seen[ksh::kNumTimers] = {};    // launches of each kind since the reset
  int64_t timing_units[ksh::kNumTimers] = {};   // units (keys / k-mers) of the timed launches of each kind
  std::vector<hipEvent_t> ev_pool;                       // all events ever created
  size_t ev_next = 0;                                    // next unused event in ev_pool
  std::vector<std::pair<size_t, size_t>> ev_spans[ksh::kNumTimers];  // (start, stop) indices
};

namespace ksh {

// hipMalloc that never lets the runtime run out of memory: when the device's free memory (hipMemGetInfo) does
// not hold `bytes` plus a margin, the context's pool (and its parent's, and idle lanes' scratch) give back what
// they only cache, and if that is not enough the call FAILS here with KSH_INTERNAL.  (ROCm 7.2's own
// out-of-memory path crashes the process: hipMalloc -> KfdDriver::AllocateMemory -> GpuAgent::Trim ->
// AqlQueue::AsyncReclaimMainScratch, SIGSEGV)
int device_alloc(ksh_ctx* ctx, size_t bytes, void** out);
int arena_reserve(ksh_ctx* ctx, size_t bytes);
inline void arena_reset(ksh_ctx* ctx) { ctx->arena_used = 0; }
// Returns nullptr when the arena is too small (callers reserve first).
void* arena_alloc(ksh_ctx* ctx, size_t bytes);
int plan_reserve(ksh_ctx* ctx, size_t bytes);
enum { kSlotDecode = 0, kSlotEncode = 1, kSlotText = 2 };
int slot_reserve(ksh_ctx* ctx, int which, size_t bytes);
// (thread-safe; a lane's jobs allocate what outlives them from lane->lane_parent's pool)
int pool_alloc(ksh_ctx* ctx, size_t bytes, void** out);
void pool_free(ksh_ctx* ctx, void* p);
void pool_trim(ksh_ctx* ctx);

// Runs job(lane, q) for every q of `order` (most expensive first) on up to lanes_wanted helper contexts at once,
// each from a host thread of its own (the calling thread works one of them); with one lane, on the context itself.
// `prepare(lane)` reserves the scratch the largest job needs and is called for a helper lane before it is
// used (a lane that cannot get its scratch -- memory -- is left out, that is no error); `need_bytes` is what
// that reservation takes, checked against the free memory first.  The context's stream is drained before
// the jobs start and every lane's stream when they end, so callers see ordinary single-stream ordering.
// Returns the first failure (its message becomes the calling thread's last error).
int run_on_lanes(ksh_ctx* ctx, const std::vector<size_t>& order, size_t need_bytes,
                 const std::function<int(ksh_ctx*)>& prepare, const std::function<int(ksh_ctx*, size_t)>& job);
inline ksh_ctx* result_ctx(ksh_ctx* lane) { return lane->lane_parent ? lane->lane_parent : lane; }

// Scratch of an SPSS encode / decode of n k-mers (slot + arena bytes): what a lane reserves (ksh_encode.hip, ksh_decode.hip)
size_t encode_scratch_bytes(const ksh_geom* g, int64_t n);
int encode_reserve(ksh_ctx* ctx, const ksh_geom* g, int64_t n);

int check_geom(const ksh_geom* g);
int check_view(const ksh_set_view* v, const char* name);  // ksh_pair.hip
inline int64_t n_buckets(const ksh_geom* g) { return int64_t(1) << g->n_bucket_bits; }
inline int key_bits(const ksh_geom* g) { return 2 * g->k - g->n_bucket_bits; }

hipEvent_t timer_event(ksh_ctx* ctx, size_t* index);
void free_plan(ksh_ctx* ctx);  // ksh_encode.hip

struct Timer {
  ksh_ctx* ctx;
  int kind;
  size_t i0 = 0;
  bool on = false;
  // units: what the launch processes (k-mers, keys), summed over the timed launches of the kind
  Timer(ksh_ctx* c, int k, int64_t units = 0) : ctx(c), kind(k) {
    on = ctx->timing && (ctx->timing_seen[kind]++ % ctx->timing_stride) == 0;
    if (on) {
      ctx->timing_units[kind] += units;
      (void)hipEventRecord(timer_event(ctx, &i0), ctx->stream);
    }
  }
  ~Timer() {
    if (on) {
      size_t i1;
      (void)hipEventRecord(timer_event(ctx, &i1), ctx->stream);
      ctx->ev_spans[kind].emplace_back(i0, i1);
    }
  }
};

// Exclusive prefix sum of n int64 values on the context's stream; `total`
// (device, one int64, may be nullptr) receives the sum.  in may equal out.
int scan_exclusive_i64(ksh_ctx* ctx, const int64_t* d_in, int64_t* d_out, int64_t n,
                       int64_t* d_total);

}  // namespace ksh

#endif
