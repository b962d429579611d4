// Context, errors, device plumbing, prefix sum and the XOR-hash reduction.
// gfx950 only; wavefront = 64.
#include "ksh_internal.h"
#include "ksh_dsu.h"
#include "ksh_kmer.h"
#include "ksh_scan.h"

#include <algorithm>
#include <atomic>
#include <cstring>
#include <iterator>
#include <thread>
#include <type_traits>

namespace ksh {

static thread_local std::string g_last_error;

void set_error(const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_last_error = buf;
}

int fail(int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_last_error = buf;
  return code;
}

int check_geom(const ksh_geom* g) {
  if (!g) return fail(KSH_INVALID_ARGUMENT, "geometry is NULL");
  if (g->k < 2 || g->k > 31) return fail(KSH_INVALID_ARGUMENT, "k = %d is outside [2, 31]", g->k);
  if (g->n_bucket_bits < 1 || g->n_bucket_bits > 24 || g->n_bucket_bits >= 2 * g->k)
    return fail(KSH_INVALID_ARGUMENT, "n_bucket_bits = %d is not usable with k = %d",
                g->n_bucket_bits, g->k);
  const int kb = 2 * g->k - g->n_bucket_bits;
  if (g->key_bytes != 2 && g->key_bytes != 4 && g->key_bytes != 8)
    return fail(KSH_INVALID_ARGUMENT, "key_bytes = %d (device keys are 2, 4 or 8 bytes)", g->key_bytes);
  if (kb > 8 * g->key_bytes)
    return fail(KSH_INVALID_ARGUMENT, "%d key bits do not fit %d key bytes", kb, g->key_bytes);
  return KSH_OK;
}

static void pool_trim_locked(ksh_ctx* ctx);
static void lanes_release_scratch(ksh_ctx* ctx);

static size_t device_total_bytes() {
  static const size_t total = [] {
    size_t f = 0, t = 0;
    return hipMemGetInfo(&f, &t) == hipSuccess ? t : size_t(288) << 30;
  }();
  return total;
}
// the margin kept free for everybody else (the caller's own allocator, the runtime's queues and code objects)
static size_t device_margin_bytes() { return std::max<size_t>(size_t(2) << 30, device_total_bytes() / 64); }

static bool device_has_room(size_t bytes) {
  size_t f = 0, t = 0;
  if (hipMemGetInfo(&f, &t) != hipSuccess) return true;  // (cannot tell: let hipMalloc answer)
  return f >= bytes + device_margin_bytes();
}

// `pool_locked`: the caller holds ctx->pool_mu (pool_alloc)
static int device_alloc_impl(ksh_ctx* ctx, size_t bytes, void** out, bool pool_locked) {
  *out = nullptr;
  if (!device_has_room(bytes)) {
    // what this context only caches
    if (pool_locked) {
      pool_trim_locked(ctx);
    } else {
      pool_trim(ctx);
    }
    // ... its parent (a lane's results live in the parent's pool: its cached blocks are nobody's)
    if (!device_has_room(bytes) && ctx->lane_parent) pool_trim(ctx->lane_parent);
    // ... and the scratch of idle lanes (mapped again when they are next used)
    if (!device_has_room(bytes) && !ctx->lanes.empty() && !ctx->lanes_busy) lanes_release_scratch(ctx);
    if (!device_has_room(bytes)) {
      size_t f = 0, t = 0;
      (void)hipMemGetInfo(&f, &t);
      return fail(KSH_INTERNAL, "out of device memory: %zu bytes asked for, %zu of %zu free (a margin of %zu is kept)", bytes,
                  f, t, device_margin_bytes());
    }
  }
  const hipError_t e = hipMalloc(out, bytes);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    *out = nullptr;
    return fail(KSH_INTERNAL, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
  }
  return KSH_OK;
}

int device_alloc(ksh_ctx* ctx, size_t bytes, void** out) { return device_alloc_impl(ctx, bytes, out, false); }

int arena_reserve(ksh_ctx* ctx, size_t bytes) {
  if (bytes <= ctx->arena_bytes) return KSH_OK;
  KSH_HIP(hipStreamSynchronize(ctx->stream));
  if (ctx->arena) KSH_HIP(hipFree(ctx->arena));
  ctx->arena = nullptr;
  ctx->arena_bytes = 0;
  size_t want = bytes + (bytes >> 2) + (1u << 20);
  KSH_TRY(device_alloc(ctx, want, reinterpret_cast<void**>(&ctx->arena)));
  ctx->arena_bytes = want;
  ctx->arena_used = 0;
  return KSH_OK;
}

void* arena_alloc(ksh_ctx* ctx, size_t bytes) {
  size_t at = (ctx->arena_used + 255) & ~size_t(255);
  if (at + bytes > ctx->arena_bytes) return nullptr;
  ctx->arena_used = at + bytes;
  return ctx->arena + at;
}

int slot_reserve(ksh_ctx* ctx, int which, size_t bytes) {
  if (bytes <= ctx->slot_bytes[which]) return KSH_OK;
  KSH_HIP(hipStreamSynchronize(ctx->stream));
  if (ctx->slot[which]) KSH_HIP(hipFree(ctx->slot[which]));
  ctx->slot[which] = nullptr;
  ctx->slot_bytes[which] = 0;
  size_t want = bytes + (bytes >> 3) + (1u << 16);
  KSH_TRY(device_alloc(ctx, want, reinterpret_cast<void**>(&ctx->slot[which])));
  ctx->slot_bytes[which] = want;
  return KSH_OK;
}

static size_t pool_round(size_t bytes) {
  if (bytes < 4096) return 4096;
  size_t gran = 4096;
  while (gran * 16 <= bytes) gran <<= 1;  // gran in (bytes/16, bytes/8]: at most 12.5 % slack
  return (bytes + gran - 1) / gran * gran;
}

int pool_alloc(ksh_ctx* ctx, size_t bytes, void** out) {
  std::lock_guard<std::mutex> lock(ctx->pool_mu);
  if (ctx->inject_skip >= 0 && bytes >= ctx->inject_min_bytes) {
    if (ctx->inject_skip == 0) {
      *out = nullptr;
      return fail(KSH_INTERNAL, "hipMalloc(%zu) failed: out of memory (injected)", bytes);
    }
    ctx->inject_skip--;
  }
  const size_t want = pool_round(bytes);
  auto it = ctx->pool_free_blocks.lower_bound(want);
  if (it != ctx->pool_free_blocks.end() && it->first <= want + want / 4) {
    *out = it->second;
    ctx->pool_cached_bytes -= it->first;
    ctx->pool_live_bytes += it->first;
    ctx->pool_peak_bytes = std::max(ctx->pool_peak_bytes, ctx->pool_live_bytes);
    ctx->pool_free_blocks.erase(it);
    return KSH_OK;
  }
  KSH_TRY(device_alloc_impl(ctx, want, out, true));  // (gives the cached blocks back first when memory is short)
  ctx->pool_sizes[*out] = want;
  ctx->pool_live_bytes += want;
  ctx->pool_peak_bytes = std::max(ctx->pool_peak_bytes, ctx->pool_live_bytes);
  return KSH_OK;
}

void pool_free(ksh_ctx* ctx, void* p) {
  if (!p) return;
  std::lock_guard<std::mutex> lock(ctx->pool_mu);
  auto it = ctx->pool_sizes.find(p);
  if (it == ctx->pool_sizes.end()) {
    (void)hipFree(p);
    return;
  }
  ctx->pool_free_blocks.emplace(it->second, p);
  ctx->pool_cached_bytes += it->second;
  ctx->pool_live_bytes -= std::min(ctx->pool_live_bytes, it->second);
  // keep the cache bounded: drop the largest blocks beyond a quarter of the device's memory (at most 96 GiB)
  const size_t cache_cap = std::min<size_t>(size_t(96) << 30, device_total_bytes() / 4);
  while (ctx->pool_cached_bytes > cache_cap && !ctx->pool_free_blocks.empty()) {
    auto last = std::prev(ctx->pool_free_blocks.end());
    ctx->pool_cached_bytes -= last->first;
    ctx->pool_sizes.erase(last->second);
    (void)hipFree(last->second);
    ctx->pool_free_blocks.erase(last);
  }
}

void pool_trim(ksh_ctx* ctx) {
  std::lock_guard<std::mutex> lock(ctx->pool_mu);
  pool_trim_locked(ctx);
}

static void pool_trim_locked(ksh_ctx* ctx) {
  (void)hipStreamSynchronize(ctx->stream);
  for (auto& kv : ctx->pool_free_blocks) {
    ctx->pool_sizes.erase(kv.second);
    (void)hipFree(kv.second);
  }
  ctx->pool_free_blocks.clear();
  ctx->pool_cached_bytes = 0;
}

// ---- lanes
static void lanes_release_scratch(ksh_ctx* ctx) {
  for (ksh_ctx* lane : ctx->lanes) {
    (void)hipStreamSynchronize(lane->stream);
    free_plan(lane);
    pool_trim(lane);
    for (int i = 0; i < 3; i++) {
      if (lane->slot[i]) (void)hipFree(lane->slot[i]);
      lane->slot[i] = nullptr;
      lane->slot_bytes[i] = 0;
    }
    if (lane->arena) (void)hipFree(lane->arena);
    lane->arena = nullptr;
    lane->arena_bytes = lane->arena_used = 0;
  }
}

static int lanes_default() {
  static const int v = [] {
    const char* e = getenv("KSH_LANES");
    const int n = e ? atoi(e) : 4;
    return n < 1 ? 1 : (n > 8 ? 8 : n);
  }();
  return v;
}

int run_on_lanes(ksh_ctx* ctx, const std::vector<size_t>& order, size_t need_bytes,
                 const std::function<int(ksh_ctx*)>& prepare, const std::function<int(ksh_ctx*, size_t)>& job) {
  if (order.empty()) return KSH_OK;
  const int wanted = ctx->lane_parent ? 1 : (ctx->lanes_wanted > 0 ? ctx->lanes_wanted : lanes_default());
  const size_t n_lanes = std::min<size_t>(size_t(wanted), order.size());
  KSH_HIP(hipStreamSynchronize(ctx->stream));  // what the jobs read was written on this stream
  // With two lanes or more EVERY job runs on a helper context, the calling thread's too: the context's own pool
  // hands out the jobs' results, from several threads, and must not at the same time serve a job's short-lived
  // scratch -- a block that a job on this context's stream gave back with kernels still queued behind it (the
  // decode's intermediate keys, an encode's unitig block: fine on ONE stream, where the next user queues up
  // behind them) would be handed to another lane as its result while those kernels run.  (Found the hard way:
  // tools/scale_sweep.py, (19, 10) sets of 2.7 x 10^7 k-mers -- oversize buckets, so the bucket sort's scratch
  // copy is the size of a set's keys -- faulted the GPU)  During a run the context's pool only
  // hands out; what it holds was last used before the drain above or on a lane drained at the end of a run.
  std::vector<ksh_ctx*> use;
  for (size_t l = 1; l <= n_lanes && n_lanes >= 2; l++) {
    if (ctx->lanes.size() < l) {
      ksh_ctx* lane = nullptr;
      // The lanes share the GPU by SPACE, two shares of the CUs, odd and even lanes (hipExtStreamCreateWithCUMask;
      // the lower and the upper half of the mask's bits: on gfx942 / gfx950 consecutive bits go round the XCDs, so
      // every XCD gives half of its CUs to either share).  Kernels of different encodes that time-share a CU get
      // in each other's way -- a CU full of walk waves keeps the probe's workgroups out until they drain, a 78 KB
      // LDS window beside 21 KB ones leaves LDS nobody fits --; on disjoint CUs two lanes per share pack well:
      // 64 x 10^8 build, one box: 3 lanes 782 ms, 4: 793, 6: 765, 8: 762; 4 lanes on 2 shares 737 (734 - 741 on three
      // boxes), 5 on 2: 752, 2 on 2: 765, 4 on 4: 762, 3 on 3 (shares that cut through XCDs unevenly): 899, 4 on 2 with
      // the shares by XCD parity: 757 (profiles/r04_lanes_occupancy_ab.txt).  KSH_LANE_CUS=1: every lane on all CUs.
      static const int split = [] {
        const char* e = getenv("KSH_LANE_CUS");
        const int n = e ? atoi(e) : 2;
        return n < 1 ? 1 : (n > 8 ? 8 : n);
      }();
      hipStream_t masked = nullptr;
      if (split > 1) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, ctx->device) == hipSuccess) {
          const int n_cu = prop.multiProcessorCount;
          std::vector<uint32_t> mask(size_t((n_cu + 31) / 32), 0u);
          const int share = int(l - 1) % split;
          for (int cu = 0; cu < n_cu; cu++)
            if (cu * split / n_cu == share) mask[size_t(cu / 32)] |= 1u << (cu % 32);
          if (hipExtStreamCreateWithCUMask(&masked, uint32_t(mask.size()), mask.data()) != hipSuccess) {
            (void)hipGetLastError();
            masked = nullptr;  // (a plain stream then)
          }
        }
      }
      if (ksh_ctx_create(ctx->device, masked, &lane) != KSH_OK) break;  // (no stream, no pinned page: go on with fewer)
      if (masked) lane->own_stream = true;
      lane->lane_parent = ctx;
      ctx->lanes.push_back(lane);
    }
    ksh_ctx* lane = ctx->lanes[l - 1];
    // scratch for the largest job, mapped here, on one thread, before anything runs -- and only when it fits with
    // room to spare: a lane is an optimisation, the memory may be owed to the sets (config 5: 215 of 288 GB)
    size_t free_b = 0, total_b = 0;
    const size_t have = lane->slot_bytes[kSlotEncode] + lane->slot_bytes[kSlotDecode] + lane->arena_bytes;
    if (have < need_bytes) {
      if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) break;
      if (free_b < (need_bytes - have) + need_bytes / 4 + (size_t(4) << 30)) break;
      // ... and the lanes together hold at most two fifths of the device: the sets, the results and the caller's
      // own allocator need the rest (four lanes of a 5 x 10^8-k-mer k = 31 encode would be 190 of 288 GB)
      size_t lanes_total = need_bytes;
      for (size_t q = 0; q + 1 < l; q++) lanes_total += std::max(need_bytes, ctx->lanes[q]->slot_bytes[kSlotEncode] + ctx->lanes[q]->slot_bytes[kSlotDecode] + ctx->lanes[q]->arena_bytes);
      if (lanes_total > total_b / 5 * 2) break;
    }
    if (prepare(lane) != KSH_OK) {
      (void)hipGetLastError();
      break;
    }
    lane->timing = ctx->timing;
    lane->timing_stride = ctx->timing_stride;
    use.push_back(lane);
  }
  if (use.size() < 2) {  // one lane (asked for, or all that fits): everything on the context's own stream
    for (size_t q : order) KSH_TRY(job(ctx, q));
    return KSH_OK;
  }
  std::atomic<size_t> next{0};
  std::atomic<int> first_rc{KSH_OK};
  std::mutex msg_mu;
  std::string first_msg;
  const auto work = [&](ksh_ctx* lane) {
    (void)hipSetDevice(lane->device);
    while (first_rc.load() == KSH_OK) {
      const size_t at = next.fetch_add(1);
      if (at >= order.size()) break;
      const int rc = job(lane, order[at]);
      if (rc != KSH_OK) {
        std::lock_guard<std::mutex> lock(msg_mu);
        if (first_rc.load() == KSH_OK) {
          first_msg = g_last_error;  // (this thread's)
          first_rc.store(rc);
        }
      }
    }
    (void)hipStreamSynchronize(lane->stream);  // what the job wrote is read on the parent's stream afterwards
  };
  ctx->lanes_busy = true;
  std::vector<std::thread> threads;
  for (size_t l = 1; l < use.size(); l++) threads.emplace_back(work, use[l]);
  work(use[0]);
  for (std::thread& t : threads) t.join();
  ctx->lanes_busy = false;
  if (first_rc.load() != KSH_OK) return fail(first_rc.load(), "%s", first_msg.c_str());
  return KSH_OK;
}

hipEvent_t timer_event(ksh_ctx* ctx, size_t* index) {
  if (ctx->ev_next == ctx->ev_pool.size()) {
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    ctx->ev_pool.push_back(e);
  }
  *index = ctx->ev_next;
  return ctx->ev_pool[ctx->ev_next++];
}

int plan_reserve(ksh_ctx* ctx, size_t bytes) {
  if (bytes <= ctx->plan_bytes) return KSH_OK;
  KSH_HIP(hipStreamSynchronize(ctx->stream));
  if (ctx->plan) KSH_HIP(hipFree(ctx->plan));
  ctx->plan = nullptr;
  ctx->plan_bytes = 0;
  size_t want = bytes + (bytes >> 2) + (1u << 16);
  KSH_TRY(device_alloc(ctx, want, reinterpret_cast<void**>(&ctx->plan)));
  ctx->plan_bytes = want;
  return KSH_OK;
}

// ------------------------------------------------------------------------------ scan
// Three-kernel exclusive scan: 256 threads x 8 items per block, block totals
// scanned recursively.  Inputs here are small (bucket and tile counts), so this
// is launch-bound, not bandwidth-bound.
constexpr int kScanThreads = 256;
constexpr int kScanItems = 8;
constexpr int kScanTile = kScanThreads * kScanItems;

__device__ inline int64_t wave_inclusive_scan(int64_t v) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    int64_t o = __shfl_up(v, d, 64);
    if (lane >= d) v += o;
  }
  return v;
}

// Exclusive scan of one value per thread across a 256-thread block.
__device__ inline int64_t block_exclusive_scan(int64_t v, int64_t* total, int64_t* lds4) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int64_t inc = wave_inclusive_scan(v);
  if (lane == 63) lds4[wave] = inc;
  __syncthreads();
  int64_t base = 0;
#pragma unroll
  for (int w = 0; w < kScanThreads / 64; w++) {
    if (w < wave) base += lds4[w];
  }
  *total = lds4[0] + lds4[1] + lds4[2] + lds4[3];
  __syncthreads();
  return base + inc - v;
}

__global__ __launch_bounds__(kScanThreads) void k_scan_tiles(const int64_t* __restrict__ in,
                                                              int64_t* __restrict__ out,
                                                              int64_t* __restrict__ block_sums,
                                                              int64_t n) {
  __shared__ int64_t lds4[4];
  const int64_t base = int64_t(blockIdx.x) * kScanTile + int64_t(threadIdx.x) * kScanItems;
  int64_t v[kScanItems];
  int64_t sum = 0;
#pragma unroll
  for (int i = 0; i < kScanItems; i++) {
    v[i] = (base + i < n) ? in[base + i] : 0;
    sum += v[i];
  }
  int64_t total;
  int64_t excl = block_exclusive_scan(sum, &total, lds4);
#pragma unroll
  for (int i = 0; i < kScanItems; i++) {
    if (base + i < n) out[base + i] = excl;
    excl += v[i];
  }
  if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

__global__ __launch_bounds__(kScanThreads) void k_scan_add(int64_t* __restrict__ out,
                                                            const int64_t* __restrict__ block_prefix,
                                                            int64_t n) {
  const int64_t base = int64_t(blockIdx.x) * kScanTile + int64_t(threadIdx.x) * kScanItems;
  const int64_t add = block_prefix[blockIdx.x];
#pragma unroll
  for (int i = 0; i < kScanItems; i++) {
    if (base + i < n) out[base + i] += add;
  }
}

// Second (last) launch of the mid-size scan: every block sums the totals of the blocks
// before it (at most kScanFixMaxBlocks values) and adds that base to its tile.
constexpr int64_t kScanFixMaxBlocks = 4096;

__global__ __launch_bounds__(kScanThreads) void k_scan_fix(int64_t* __restrict__ out,
                                                            const int64_t* __restrict__ block_sums,
                                                            int64_t n, int64_t n_blocks,
                                                            int64_t* __restrict__ total_out) {
  __shared__ int64_t lds4[4];
  int64_t mine = 0;
  const int64_t upto = total_out && blockIdx.x == 0 ? n_blocks : int64_t(blockIdx.x);
  for (int64_t b = threadIdx.x; b < upto; b += kScanThreads) mine += block_sums[b];
  // block reduce
  int64_t acc = mine;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d, 64);
  if ((threadIdx.x & 63) == 0) lds4[threadIdx.x >> 6] = acc;
  __syncthreads();
  const int64_t sum = lds4[0] + lds4[1] + lds4[2] + lds4[3];
  if (blockIdx.x == 0) {
    if (total_out && threadIdx.x == 0) *total_out = sum;  // block 0 summed every block
    return;                                                // its own base is 0
  }
  const int64_t base = int64_t(blockIdx.x) * kScanTile + int64_t(threadIdx.x) * kScanItems;
#pragma unroll
  for (int i = 0; i < kScanItems; i++) {
    if (base + i < n) out[base + i] += sum;
  }
}

// Small inputs (bucket counts, tile counts): one 1024-thread workgroup, one launch.
constexpr int64_t kScanSmallMax = 4096;  // beyond this the two-launch tiled scan is faster than one CU

__global__ __launch_bounds__(1024) void k_scan_small(const int64_t* __restrict__ in,
                                                      int64_t* __restrict__ out, int64_t n,
                                                      int64_t* __restrict__ total_out) {
  __shared__ int64_t wave_sums[16];
  const int tid = threadIdx.x;
  const int64_t per = (n + 1023) / 1024;
  const int64_t lo = min(int64_t(tid) * per, n), hi = min(lo + per, n);
  int64_t mine = 0;
  for (int64_t i = lo; i < hi; i++) mine += in[i];
  int64_t inc = wave_inclusive_scan(mine);
  const int lane = tid & 63, wave = tid >> 6;
  if (lane == 63) wave_sums[wave] = inc;
  __syncthreads();
  int64_t base = 0, total = 0;
  for (int w = 0; w < 16; w++) {
    if (w < wave) base += wave_sums[w];
    total += wave_sums[w];
  }
  int64_t run = base + inc - mine;
  for (int64_t i = lo; i < hi; i++) {
    const int64_t v = in[i];
    out[i] = run;
    run += v;
  }
  if (tid == 0 && total_out) *total_out = total;
}

int scan_exclusive_i64(ksh_ctx* ctx, const int64_t* d_in, int64_t* d_out, int64_t n,
                       int64_t* d_total) {
  if (n <= 0) {
    if (d_total) KSH_HIP(hipMemsetAsync(d_total, 0, sizeof(int64_t), ctx->stream));
    return KSH_OK;
  }
  if (n <= kScanSmallMax) {
    hipLaunchKernelGGL(k_scan_small, dim3(1), dim3(1024), 0, ctx->stream, d_in, d_out, n, d_total);
    KSH_HIP(hipGetLastError());
    return KSH_OK;
  }
  if (scan_exclusive_chained(ctx, LoadArray{d_in}, d_out, n, d_total)) {
    KSH_HIP(hipGetLastError());
    return KSH_OK;
  }
  const int64_t blocks = (n + kScanTile - 1) / kScanTile;
  if (blocks <= kScanFixMaxBlocks) {
    int64_t* bsums = static_cast<int64_t*>(arena_alloc(ctx, size_t(blocks) * sizeof(int64_t)));
    if (!bsums) return fail(KSH_INTERNAL, "scan: scratch arena too small");
    hipLaunchKernelGGL(k_scan_tiles, dim3(unsigned(blocks)), dim3(kScanThreads), 0, ctx->stream, d_in,
                       d_out, bsums, n);
    hipLaunchKernelGGL(k_scan_fix, dim3(unsigned(blocks)), dim3(kScanThreads), 0, ctx->stream, d_out,
                       bsums, n, blocks, d_total);
    KSH_HIP(hipGetLastError());
    return KSH_OK;
  }
  int64_t* sums = static_cast<int64_t*>(arena_alloc(ctx, size_t(blocks) * sizeof(int64_t)));
  int64_t* sums_total = static_cast<int64_t*>(arena_alloc(ctx, sizeof(int64_t)));
  if (!sums || !sums_total) return fail(KSH_INTERNAL, "scan: scratch arena too small");
  hipLaunchKernelGGL(k_scan_tiles, dim3(unsigned(blocks)), dim3(kScanThreads), 0, ctx->stream, d_in,
                     d_out, sums, n);
  if (blocks > 1) {
    // sums -> exclusive prefix (in place), recursively
    KSH_TRY(scan_exclusive_i64(ctx, sums, sums, blocks, sums_total));
    hipLaunchKernelGGL(k_scan_add, dim3(unsigned(blocks)), dim3(kScanThreads), 0, ctx->stream, d_out,
                       sums, n);
    if (d_total)
      KSH_HIP(hipMemcpyAsync(d_total, sums_total, sizeof(int64_t), hipMemcpyDeviceToDevice,
                             ctx->stream));
  } else if (d_total) {
    KSH_HIP(hipMemcpyAsync(d_total, sums, sizeof(int64_t), hipMemcpyDeviceToDevice, ctx->stream));
  }
  KSH_HIP(hipGetLastError());
  return KSH_OK;
}

// ------------------------------------------------------------------------------ hash
// KmerSet::Hash (lib/core/kmer_set.h:224-244) = XOR of every k-mer's bit pattern
// = XOR of all keys  ^  XOR over buckets with an odd key count of (b << key_bits).
// The first term is a streaming reduction over the key array (16 B per lane).
template <typename KeyT>
__global__ __launch_bounds__(256) void k_xor_keys(const KeyT* __restrict__ keys, int64_t n,
                                                   unsigned long long* __restrict__ out) {
  constexpr int kPer = 16 / sizeof(KeyT);
  using Vec = typename std::conditional<sizeof(KeyT) <= 4, uint4, ulonglong2>::type;
  unsigned long long acc = 0;
  const int64_t n_vec = n / kPer;
  const int64_t stride = int64_t(gridDim.x) * blockDim.x;
  const Vec* vp = reinterpret_cast<const Vec*>(keys);
  for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n_vec; i += stride) {
    Vec v = vp[i];
    if constexpr (sizeof(KeyT) == 2) {
      const uint32_t w = v.x ^ v.y ^ v.z ^ v.w;  // eight 16-bit keys, two to a word
      acc ^= (unsigned long long)((w & 0xFFFFu) ^ (w >> 16));
    } else if constexpr (sizeof(KeyT) == 4) {
      acc ^= (unsigned long long)(v.x ^ v.y ^ v.z ^ v.w);
    } else {
      acc ^= v.x ^ v.y;
    }
  }
  for (int64_t i = n_vec * kPer + int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride)
    acc ^= (unsigned long long)keys[i];
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) acc ^= __shfl_xor(acc, d, 64);
  if ((threadIdx.x & 63) == 0 && acc) atomicXor(out, acc);
}

__global__ __launch_bounds__(256) void k_xor_buckets(const int64_t* __restrict__ offsets,
                                                      int64_t n_buckets, int key_bits,
                                                      unsigned long long* __restrict__ out) {
  unsigned long long acc = 0;
  for (int64_t b = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; b < n_buckets;
       b += int64_t(gridDim.x) * blockDim.x) {
    if ((offsets[b + 1] - offsets[b]) & 1) acc ^= (unsigned long long)b << key_bits;
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) acc ^= __shfl_xor(acc, d, 64);
  if ((threadIdx.x & 63) == 0 && acc) atomicXor(out, acc);
}

}  // namespace ksh

using namespace ksh;

namespace ksh {

// KmerSet::Contains (kmer_set.h:99-105), batched: one thread per query, a search in its bucket.
template <typename KeyT>
__global__ __launch_bounds__(256) void k_contains(DevSet<KeyT> set, const uint64_t* __restrict__ kmers, int64_t n,
                                                   uint8_t* __restrict__ found) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t z = kmers[i];
  found[i] = (z >> set.key_bits) < uint64_t(set.n_buckets) && set.find(z) >= 0 ? 1 : 0;
}

__global__ __launch_bounds__(256) void k_dsu_reset(unsigned long long* __restrict__ a, int64_t n) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) a[i] = (unsigned long long)i;
}
__global__ __launch_bounds__(256) void k_dsu_unite_pairs(DevDsu dsu, const int32_t* __restrict__ x,
                                                          const int32_t* __restrict__ y, int64_t m, int64_t n,
                                                          int* __restrict__ bad) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= m) return;
  const uint32_t a = uint32_t(x[i]), b = uint32_t(y[i]);
  if (a >= uint64_t(n) || b >= uint64_t(n)) {  // (a negative id is a huge unsigned one)
    *bad = 1;
    return;
  }
  dsu.unite(a, b);
}
__global__ __launch_bounds__(256) void k_dsu_roots(DevDsu dsu, int64_t n, int32_t* __restrict__ root) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) root[i] = int32_t(dsu.find(uint32_t(i)));
}

// KmerSet::Find(n_workers) (kmer_set.h:157-161): every k-mer as its full 2K-bit pattern, ascending.
template <typename KeyT>
__global__ __launch_bounds__(256) void k_expand_kmers(DevSet<KeyT> set, uint64_t* __restrict__ out) {
  __shared__ int64_t s_bucket[2];
  const int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  const uint64_t x = set.kmer_in_block(t, s_bucket);
  if (t < set.n) out[t] = x;
}

template <typename KeyT>
int launch_contains(ksh_ctx* ctx, const ksh_geom* g, const ksh_set_view* s, const uint64_t* d_kmers, int64_t n,
                    uint8_t* d_found) {
  DevSet<KeyT> set{s->d_offsets, static_cast<const KeyT*>(s->d_keys), n_buckets(g), s->n_keys, g->k, key_bits(g)};
  hipLaunchKernelGGL(k_contains<KeyT>, dim3(unsigned((n + 255) / 256)), dim3(256), 0, ctx->stream, set, d_kmers, n, d_found);
  KSH_HIP(hipGetLastError());
  return KSH_OK;
}

template <typename KeyT>
int launch_expand(ksh_ctx* ctx, const ksh_geom* g, const ksh_set_view* s, uint64_t* d_kmers) {
  DevSet<KeyT> set{s->d_offsets, static_cast<const KeyT*>(s->d_keys), n_buckets(g), s->n_keys, g->k, key_bits(g)};
  hipLaunchKernelGGL(k_expand_kmers<KeyT>, dim3(unsigned((s->n_keys + 255) / 256)), dim3(256), 0, ctx->stream, set, d_kmers);
  KSH_HIP(hipGetLastError());
  return KSH_OK;
}

template <typename KeyT>
int launch_xor_keys(ksh_ctx* ctx, const ksh_set_view* s, unsigned blocks, unsigned long long* d_acc) {
  hipLaunchKernelGGL(k_xor_keys<KeyT>, dim3(blocks), dim3(256), 0, ctx->stream, static_cast<const KeyT*>(s->d_keys),
                     s->n_keys, d_acc);
  return KSH_OK;
}

}  // namespace ksh

extern "C" {

int ksh_version(void) { return 2; }  // 2: ksh_comm_fns::struct_size, lanes, encode routes

const char* ksh_last_error(void) { return g_last_error.c_str(); }

int ksh_device_count(int* count) {
  if (!count) return fail(KSH_INVALID_ARGUMENT, "count is NULL");
  *count = 0;
  hipError_t e = hipGetDeviceCount(count);
  if (e != hipSuccess) {
    *count = 0;
    return fail(KSH_INTERNAL, "hipGetDeviceCount failed: %s", hipGetErrorString(e));
  }
  return KSH_OK;
}

int ksh_malloc(int device, size_t bytes, void** d_ptr) {
  if (!d_ptr) return fail(KSH_INVALID_ARGUMENT, "d_ptr is NULL");
  *d_ptr = nullptr;
  KSH_HIP(hipSetDevice(device));
  if (bytes == 0) bytes = 16;
  // (no context here to give cached memory back: refuse rather than let the runtime run out, see device_alloc)
  if (!device_has_room(bytes)) {
    size_t f = 0, t = 0;
    (void)hipMemGetInfo(&f, &t);
    return fail(KSH_INTERNAL, "out of device memory: %zu bytes asked for, %zu of %zu free", bytes, f, t);
  }
  KSH_HIP(hipMalloc(d_ptr, bytes));
  return KSH_OK;
}

int ksh_free(int device, void* d_ptr) {
  if (!d_ptr) return KSH_OK;
  KSH_HIP(hipSetDevice(device));
  KSH_HIP(hipFree(d_ptr));
  return KSH_OK;
}

int ksh_memcpy_h2d(int device, void* d_dst, const void* src, size_t bytes) {
  KSH_HIP(hipSetDevice(device));
  if (bytes) {
    KSH_HIP(hipDeviceSynchronize());  // whatever stream produced / still reads the buffer: a context's stream need not be a blocking one
    KSH_HIP(hipMemcpy(d_dst, src, bytes, hipMemcpyHostToDevice));
  }
  return KSH_OK;
}

int ksh_memcpy_d2h(int device, void* dst, const void* d_src, size_t bytes) {
  KSH_HIP(hipSetDevice(device));
  if (bytes) {
    KSH_HIP(hipDeviceSynchronize());  // whatever stream produced / still reads the buffer: a context's stream need not be a blocking one
    KSH_HIP(hipMemcpy(dst, d_src, bytes, hipMemcpyDeviceToHost));
  }
  return KSH_OK;
}

int ksh_memcpy_d2d(int device, void* d_dst, const void* d_src, size_t bytes) {
  KSH_HIP(hipSetDevice(device));
  if (bytes) {
    KSH_HIP(hipDeviceSynchronize());  // whatever stream produced / still reads the buffer: a context's stream need not be a blocking one
    KSH_HIP(hipMemcpy(d_dst, d_src, bytes, hipMemcpyDeviceToDevice));
  }
  return KSH_OK;
}

// The same three copies ordered after ONE context's stream only (other contexts' work goes on):
// what a host thread with a context of its own uses for buffers it produced itself, or that were
// complete before they were handed to it.
int ksh_ctx_memcpy_h2d(ksh_ctx* ctx, void* d_dst, const void* src, size_t bytes) {
  if (!ctx) return fail(KSH_INVALID_ARGUMENT, "ctx is NULL");
  KSH_HIP(hipSetDevice(ctx->device));
  if (bytes) {
    KSH_HIP(hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    KSH_HIP(hipStreamSynchronize(ctx->stream));
  }
  return KSH_OK;
}

int ksh_ctx_memcpy_d2h(ksh_ctx* ctx, void* dst, const void* d_src, size_t bytes) {
  if (!ctx) return fail(KSH_INVALID_ARGUMENT, "ctx is NULL");
  KSH_HIP(hipSetDevice(ctx->device));
  if (bytes) {
    KSH_HIP(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    KSH_HIP(hipStreamSynchronize(ctx->stream));
  }
  return KSH_OK;
}

int ksh_ctx_memcpy_d2d(ksh_ctx* ctx, void* d_dst, const void* d_src, size_t bytes) {
  if (!ctx) return fail(KSH_INVALID_ARGUMENT, "ctx is NULL");
  KSH_HIP(hipSetDevice(ctx->device));
  if (bytes) {
    KSH_HIP(hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    KSH_HIP(hipStreamSynchronize(ctx->stream));
  }
  return KSH_OK;
}

int ksh_ctx_create(int device, void* stream, ksh_ctx** out) {
  if (!out) return fail(KSH_INVALID_ARGUMENT, "out is NULL");
  *out = nullptr;
  int count = 0;
  KSH_TRY(ksh_device_count(&count));
  if (device < 0 || device >= count)
    return fail(KSH_INTERNAL, "no HIP device %d (%d visible); this library has no CPU path", device,
                count);
  KSH_HIP(hipSetDevice(device));
  ksh_ctx* ctx = new ksh_ctx;
  ctx->device = device;
  if (stream) {
    ctx->stream = static_cast<hipStream_t>(stream);
  } else {
    // a blocking stream: the plain hipMemcpy calls behind ksh_memcpy_* (legacy default stream)
    // then order themselves after the kernels enqueued here
    hipError_t e = hipStreamCreateWithFlags(&ctx->stream, hipStreamDefault);
    if (e != hipSuccess) {
      delete ctx;
      return fail(KSH_INTERNAL, "hipStreamCreate failed: %s", hipGetErrorString(e));
    }
    ctx->own_stream = true;
  }
  hipError_t e = hipHostMalloc(reinterpret_cast<void**>(&ctx->h_pinned), 64 * sizeof(int64_t));
  if (e != hipSuccess) {
    delete ctx;
    return fail(KSH_INTERNAL, "hipHostMalloc failed: %s", hipGetErrorString(e));
  }
  int rc = arena_reserve(ctx, size_t(8) << 20);
  if (rc != KSH_OK) {
    delete ctx;
    return rc;
  }
  const size_t state_bytes = size_t(2 * kChainMaxBlocks) * sizeof(unsigned long long);
  if (hipMalloc(reinterpret_cast<void**>(&ctx->scan_state), state_bytes) != hipSuccess ||
      hipMemset(ctx->scan_state, 0, state_bytes) != hipSuccess) {
    ksh_ctx_destroy(ctx);
    return fail(KSH_INTERNAL, "hipMalloc of the scan state failed");
  }
  *out = ctx;
  return KSH_OK;
}

int ksh_ctx_destroy(ksh_ctx* ctx) {
  if (!ctx) return KSH_OK;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  for (ksh_ctx* lane : ctx->lanes) ksh_ctx_destroy(lane);
  ctx->lanes.clear();
  free_plan(ctx);
  pool_trim(ctx);
  if (ctx->arena) (void)hipFree(ctx->arena);
  if (ctx->scan_state) (void)hipFree(ctx->scan_state);
  if (ctx->plan) (void)hipFree(ctx->plan);
  for (int i = 0; i < 3; i++)
    if (ctx->slot[i]) (void)hipFree(ctx->slot[i]);
  if (ctx->text_plan && ctx->text_plan_free) ctx->text_plan_free(ctx->text_plan);
  if (ctx->fasta_plan && ctx->fasta_plan_free) ctx->fasta_plan_free(ctx->fasta_plan);
  if (ctx->h_pinned) (void)hipHostFree(ctx->h_pinned);
  if (ctx->h_batch) (void)hipHostFree(ctx->h_batch);
  for (hipEvent_t ev : ctx->ev_pool)
    if (ev) (void)hipEventDestroy(ev);
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
  return KSH_OK;
}

int ksh_ctx_sync(ksh_ctx* ctx) {
  if (!ctx) return fail(KSH_INVALID_ARGUMENT, "ctx is NULL");
  KSH_HIP(hipStreamSynchronize(ctx->stream));
  return KSH_OK;
}

int ksh_ctx_reserve(ksh_ctx* ctx, size_t bytes) {
  if (!ctx) return fail(KSH_INVALID_ARGUMENT, "ctx is NULL");
  KSH_HIP(hipSetDevice(ctx->device));
  return arena_reserve(ctx, bytes);
}

int ksh_ctx_enable_timing(ksh_ctx* ctx, int enable) {
  if (!ctx) return fail(KSH_INVALID_ARGUMENT, "ctx is NULL");
  ctx->timing = enable > 0;
  ctx->timing_stride = enable > 0 ? enable : 1;
  return KSH_OK;  // (the lanes take the setting over whenever they are used)
}

static size_t scratch_bytes_of(const ksh_ctx* c) {
  return c->slot_bytes[0] + c->slot_bytes[1] + c->slot_bytes[2] + c->arena_bytes + c->plan_bytes;
}

int ksh_ctx_mem_stats(ksh_ctx* ctx, int64_t stats[6], int reset_peak) {
  if (!ctx || !stats) return fail(KSH_INVALID_ARGUMENT, "NULL argument");
  std::lock_guard<std::mutex> lock(ctx->pool_mu);
  stats[0] = int64_t(ctx->pool_live_bytes);
  stats[1] = int64_t(ctx->pool_peak_bytes);
  stats[2] = int64_t(ctx->pool_cached_bytes);
  stats[3] = int64_t(scratch_bytes_of(ctx));
  stats[4] = 0;
  for (const ksh_ctx* lane : ctx->lanes) stats[4] += int64_t(scratch_bytes_of(lane) + lane->pool_live_bytes + lane->pool_cached_bytes);
  stats[5] = int64_t(ctx->lanes.size());
  if (reset_peak) ctx->pool_peak_bytes = ctx->pool_live_bytes;
  return KSH_OK;
}

int ksh_ctx_set_lanes(ksh_ctx* ctx, int n_lanes) {
  if (!ctx) return fail(KSH_INVALID_ARGUMENT, "ctx is NULL");
  if (n_lanes > 8) return fail(KSH_INVALID_ARGUMENT, "at most 8 lanes");
  ctx->lanes_wanted = n_lanes < 1 ? -1 : n_lanes;
  return KSH_OK;
}

int ksh_ctx_timing_reset(ksh_ctx* ctx) {
  if (!ctx) return fail(KSH_INVALID_ARGUMENT, "ctx is NULL");
  KSH_HIP(hipStreamSynchronize(ctx->stream));
  ctx->ev_next = 0;
  for (int i = 0; i < kNumTimers; i++) {
    ctx->ev_spans[i].clear();
    ctx->timing_seen[i] = 0;
    ctx->timing_units[i] = 0;
  }
  for (ksh_ctx* lane : ctx->lanes) KSH_TRY(ksh_ctx_timing_reset(lane));
  return KSH_OK;
}

int ksh_ctx_timing_read(ksh_ctx* ctx, int kind, float* total_ms, int64_t* launches) {
  if (!ctx || !total_ms || !launches) return fail(KSH_INVALID_ARGUMENT, "NULL argument");
  if (kind < 0 || kind >= kNumTimers) return fail(KSH_INVALID_ARGUMENT, "no timer kind %d", kind);
  KSH_HIP(hipStreamSynchronize(ctx->stream));
  double sum = 0;
  for (const auto& span : ctx->ev_spans[kind]) {
    float ms = 0;
    KSH_HIP(hipEventElapsedTime(&ms, ctx->ev_pool[span.first], ctx->ev_pool[span.second]));
    sum += ms;
  }
  *total_ms = float(sum);
  *launches = int64_t(ctx->ev_spans[kind].size());
  for (ksh_ctx* lane : ctx->lanes) {  // (stream time: spans of different lanes overlap; ksh_ctx_timing_wall for the union)
    float ms = 0;
    int64_t n = 0;
    KSH_TRY(ksh_ctx_timing_read(lane, kind, &ms, &n));
    *total_ms += ms;
    *launches += n;
  }
  return KSH_OK;
}

int ksh_ctx_timing_wall(ksh_ctx* ctx, int kind, float* wall_ms) {
  if (!ctx || !wall_ms) return fail(KSH_INVALID_ARGUMENT, "NULL argument");
  if (kind < 0 || kind >= kNumTimers) return fail(KSH_INVALID_ARGUMENT, "no timer kind %d", kind);
  std::vector<ksh_ctx*> all{ctx};
  all.insert(all.end(), ctx->lanes.begin(), ctx->lanes.end());
  hipEvent_t ref = nullptr;
  std::vector<std::pair<float, float>> iv;
  for (ksh_ctx* c : all) {
    KSH_HIP(hipStreamSynchronize(c->stream));
    for (const auto& span : c->ev_spans[kind]) {
      if (!ref) ref = c->ev_pool[span.first];
      float t0 = 0, t1 = 0;
      // (all on one device: the events share a clock; a span that began before the reference has t0 < 0)
      KSH_HIP(hipEventElapsedTime(&t0, ref, c->ev_pool[span.first]));
      KSH_HIP(hipEventElapsedTime(&t1, ref, c->ev_pool[span.second]));
      iv.emplace_back(t0, t1);
    }
  }
  std::sort(iv.begin(), iv.end());
  double sum = 0;
  float lo = 0, hi = 0;
  bool open = false;
  for (const auto& x : iv) {
    if (open && x.first <= hi) {
      hi = std::max(hi, x.second);
    } else {
      if (open) sum += hi - lo;
      lo = x.first;
      hi = x.second;
      open = true;
    }
  }
  if (open) sum += hi - lo;
  *wall_ms = float(sum);
  return KSH_OK;
}

int ksh_dsu_components(ksh_ctx* ctx, int64_t n, const int32_t* d_x, const int32_t* d_y, int64_t m, int32_t* d_root) {
  if (!ctx || (n > 0 && !d_root) || (m > 0 && (!d_x || !d_y))) return fail(KSH_INVALID_ARGUMENT, "NULL argument");
  if (n < 0 || m < 0 || n > int64_t(0x7FFFFFF0)) return fail(KSH_INVALID_ARGUMENT, "bad node / pair count");
  if (n == 0) return KSH_OK;
  KSH_HIP(hipSetDevice(ctx->device));
  void* words = nullptr;
  KSH_TRY(pool_alloc(ctx, size_t(n) * 8, &words));
  DevDsu dsu{static_cast<unsigned long long*>(words)};
  arena_reset(ctx);
  int* d_bad = static_cast<int*>(arena_alloc(ctx, sizeof(int)));
  if (!d_bad) {
    pool_free(ctx, words);
    return fail(KSH_INTERNAL, "scratch arena too small");
  }
  KSH_HIP(hipMemsetAsync(d_bad, 0, sizeof(int), ctx->stream));
  hipLaunchKernelGGL(k_dsu_reset, dim3(unsigned((n + 255) / 256)), dim3(256), 0, ctx->stream, dsu.a, n);
  if (m > 0)
    hipLaunchKernelGGL(k_dsu_unite_pairs, dim3(unsigned((m + 255) / 256)), dim3(256), 0, ctx->stream, dsu, d_x, d_y, m,
                       n, d_bad);
  hipLaunchKernelGGL(k_dsu_roots, dim3(unsigned((n + 255) / 256)), dim3(256), 0, ctx->stream, dsu, n, d_root);
  KSH_HIP(hipGetLastError());
  KSH_HIP(hipMemcpyAsync(ctx->h_pinned, d_bad, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  KSH_HIP(hipStreamSynchronize(ctx->stream));
  pool_free(ctx, words);
  if (*reinterpret_cast<int*>(ctx->h_pinned))
    return fail(KSH_INVALID_ARGUMENT, "ksh_dsu_components: a pair names a node outside [0, n)");
  return KSH_OK;
}

int ksh_set_contains(ksh_ctx* ctx, const ksh_geom* g, const ksh_set_view* s, const uint64_t* d_kmers, int64_t n,
                     uint8_t* d_found) {
  if (!ctx || !s || (n > 0 && (!d_kmers || !d_found))) return fail(KSH_INVALID_ARGUMENT, "NULL argument");
  KSH_TRY(check_geom(g));
  KSH_TRY(check_view(s, "s"));
  if (n <= 0) return KSH_OK;
  KSH_HIP(hipSetDevice(ctx->device));
  return KSH_BY_KEY(g->key_bytes, launch_contains, ctx, g, s, d_kmers, n, d_found);
}

int ksh_set_kmers(ksh_ctx* ctx, const ksh_geom* g, const ksh_set_view* s, uint64_t* d_kmers) {
  if (!ctx || !s) return fail(KSH_INVALID_ARGUMENT, "NULL argument");
  KSH_TRY(check_geom(g));
  KSH_TRY(check_view(s, "s"));
  if (s->n_keys == 0) return KSH_OK;
  if (!d_kmers) return fail(KSH_INVALID_ARGUMENT, "NULL output");
  KSH_HIP(hipSetDevice(ctx->device));
  return KSH_BY_KEY(g->key_bytes, launch_expand, ctx, g, s, d_kmers);
}

int ksh_ctx_timing_units(ksh_ctx* ctx, int kind, int64_t* units) {
  if (!ctx || !units) return fail(KSH_INVALID_ARGUMENT, "NULL argument");
  if (kind < 0 || kind >= kNumTimers) return fail(KSH_INVALID_ARGUMENT, "no timer kind %d", kind);
  *units = ctx->timing_units[kind];
  for (ksh_ctx* lane : ctx->lanes) *units += lane->timing_units[kind];
  return KSH_OK;
}

int ksh_set_hash(ksh_ctx* ctx, const ksh_geom* g, const ksh_set_view* s, uint64_t* hash) {
  if (!ctx || !s || !hash) return fail(KSH_INVALID_ARGUMENT, "NULL argument");
  KSH_TRY(check_geom(g));
  KSH_TRY(check_view(s, "s"));  // the key reduction reads 16-byte vectors
  KSH_HIP(hipSetDevice(ctx->device));
  arena_reset(ctx);
  unsigned long long* d_acc = static_cast<unsigned long long*>(arena_alloc(ctx, 8));
  if (!d_acc) return fail(KSH_INTERNAL, "scratch arena too small");
  KSH_HIP(hipMemsetAsync(d_acc, 0, 8, ctx->stream));
  const int64_t nb = n_buckets(g);
  if (s->n_keys > 0) {
    const int64_t per_block = 256 * 16;
    unsigned blocks = unsigned(std::min<int64_t>((s->n_keys + per_block - 1) / per_block, 2048));
    (void)KSH_BY_KEY(g->key_bytes, launch_xor_keys, ctx, s, blocks, d_acc);
  }
  hipLaunchKernelGGL(k_xor_buckets, dim3(unsigned(std::min<int64_t>((nb + 255) / 256, 1024))),
                     dim3(256), 0, ctx->stream, s->d_offsets, nb, key_bits(g), d_acc);
  KSH_HIP(hipGetLastError());
  KSH_HIP(hipMemcpyAsync(ctx->h_pinned, d_acc, 8, hipMemcpyDeviceToHost, ctx->stream));
  KSH_HIP(hipStreamSynchronize(ctx->stream));
  *hash = static_cast<uint64_t>(ctx->h_pinned[0]);
  return KSH_OK;
}

}  // extern "C"
