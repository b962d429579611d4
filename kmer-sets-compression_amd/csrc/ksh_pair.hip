// Pair kernels over bucketed sorted key arrays: intersection / both differences
// (KmerSet::Sub / Intersection, lib/core/kmer_set.h:164-187,286-305), the
// symmetric-difference count (KmerSet::Diff, :191-219) and the sampled-bucket
// pair weights (GetEdgeWeight, lib/core/kmer_set_set.h:158-184).
//
// A set is one ascending array of k-mers with a bucket index, so every one of
// these is a merge of two sorted ranges per bucket.  The work is cut into
// *segments* (one per bucket of a pair, or one per (pair, sampled bucket)) and each
// segment into *tiles* of at most kTileCap (1024) merged keys by merge-path; one
// wavefront per tile stages the tile's two key ranges in LDS with 16-byte coalesced
// loads, every lane merges kVT (16) keys from LDS, and the three result streams are
// compacted in LDS and written back coalesced.  HBM-bound integer work; no MFMA.
//
// Tie rule: on equal keys the A key is merged first, so a common key shows up as
// "a, b" adjacent in the merged order.  The tile split never separates such a
// pair (the split kernel moves the B boundary by one when it would), which makes
// every tile self-contained: matches are decided from LDS only.
#include "ksh_internal.h"

#include <algorithm>
#include <type_traits>
#include <vector>

namespace ksh {

// One wavefront per tile: 64 lanes x 16 keys.  Tiles are bounded by buckets (about 1.2 k
// merged keys per bucket at 10^7 keys per set), so the kernel is latency-bound per tile;
// single-wave workgroups put 4x more tiles in flight per CU than 256-thread ones and need
// no cross-wave scan.
constexpr int kThreads = 64;
constexpr int kVT = 16;                    // merged keys per lane
constexpr int kTile = kThreads * kVT - 1;  // diagonal spacing (the +1 is the tie fix-up)
constexpr int kTileCap = kThreads * kVT;   // LDS capacity in keys

// ---- segment sources -------------------------------------------------------------
// One segment per bucket of a pair of sets.
template <typename KeyT>
struct BucketSegs {
  const KeyT* a_keys;
  const int64_t* a_off;
  const KeyT* b_keys;
  const int64_t* b_off;
  __device__ void get(int64_t s, const KeyT*& a, int64_t& a_lo, int64_t& a_hi, const KeyT*& b,
                      int64_t& b_lo, int64_t& b_hi) const {
    a = a_keys;
    b = b_keys;
    a_lo = a_off[s];
    a_hi = a_off[s + 1];
    b_lo = b_off[s];
    b_hi = b_off[s + 1];
  }
};

struct SetPtrs {
  const void* keys;
  const int64_t* off;
};

// One segment per (pair p, sampled bucket i): s = p * n_ids + i.
template <typename KeyT>
struct PairSegs {
  const SetPtrs* sets;
  const int32_t* bucket_ids;
  const int32_t* pairs;
  int32_t n_ids;
  __device__ void get(int64_t s, const KeyT*& a, int64_t& a_lo, int64_t& a_hi, const KeyT*& b,
                      int64_t& b_lo, int64_t& b_hi) const {
    const int64_t p = s / n_ids;
    const int32_t bucket = bucket_ids[s - p * n_ids];
    const SetPtrs sa = sets[pairs[2 * p]];
    const SetPtrs sb = sets[pairs[2 * p + 1]];
    a = static_cast<const KeyT*>(sa.keys);
    b = static_cast<const KeyT*>(sb.keys);
    a_lo = sa.off[bucket];
    a_hi = sa.off[bucket + 1];
    b_lo = sb.off[bucket];
    b_hi = sb.off[bucket + 1];
  }
};

// Everything a tile needs, in one 40-byte record (one dependent load in the merge kernel).
struct TileDesc {
  const void* pa;  // first A key of the tile
  const void* pb;  // first B key of the tile
  int64_t a0;      // its index in A's key array
  int64_t b0;
  int32_t ca, cb;  // keys of A / B in the tile
};

// ---- tile plan --------------------------------------------------------------------------
template <typename KeyT, typename Segs>
__global__ __launch_bounds__(256) void k_seg_tiles(Segs segs, int64_t n_segs,
                                                    int64_t* __restrict__ tiles_per_seg) {
  const int64_t s = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (s >= n_segs) return;
  const KeyT *a, *b;
  int64_t a_lo, a_hi, b_lo, b_hi;
  segs.get(s, a, a_lo, a_hi, b, b_lo, b_hi);
  const int64_t len = (a_hi - a_lo) + (b_hi - b_lo);
  tiles_per_seg[s] = (len + kTile - 1) / kTile;
}

// Merge-path split of (a[0, na), b[0, nb)) at `diag` (A first on ties); a common key's
// "a, b" pair is never separated: the B side moves by one when the split falls inside it.
template <typename KeyT>
__device__ __forceinline__ void merge_path_split(const KeyT* __restrict__ a, int64_t na,
                                                 const KeyT* __restrict__ b, int64_t nb,
                                                 int64_t diag, int64_t* i_out, int64_t* j_out) {
  int64_t lo = diag > nb ? diag - nb : 0;
  int64_t hi = diag < na ? diag : na;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (a[mid] <= b[diag - 1 - mid]) lo = mid + 1; else hi = mid;
  }
  int64_t i = lo, j = diag - lo;
  if (i > 0 && j < nb && a[i - 1] == b[j]) j += 1;
  *i_out = i;
  *j_out = j;
}

// One thread per tile: which segment, where the tile starts and ends in A and in B.
template <typename KeyT, typename Segs>
__global__ __launch_bounds__(256) void k_tile_split(Segs segs, int64_t n_segs,
                                                     const int64_t* __restrict__ tile_base,
                                                     TileDesc* __restrict__ desc) {
  const int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  const int64_t total = tile_base[n_segs];
  if (t >= total) return;
  // last s with tile_base[s] <= t
  int64_t lo = 0, hi = n_segs;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (tile_base[mid] <= t) lo = mid + 1; else hi = mid;
  }
  const int64_t s = lo - 1;
  const KeyT *a, *b;
  int64_t a_lo, a_hi, b_lo, b_hi;
  segs.get(s, a, a_lo, a_hi, b, b_lo, b_hi);
  const int64_t na = a_hi - a_lo, nb = b_hi - b_lo;
  const int64_t diag = (t - tile_base[s]) * kTile;
  int64_t i0, j0, i1 = na, j1 = nb;
  merge_path_split(a + a_lo, na, b + b_lo, nb, diag, &i0, &j0);
  if (diag + kTile < na + nb) merge_path_split(a + a_lo, na, b + b_lo, nb, diag + kTile, &i1, &j1);
  TileDesc d;
  d.pa = a + a_lo + i0;
  d.pb = b + b_lo + j0;
  d.a0 = a_lo + i0;
  d.b0 = b_lo + j0;
  d.ca = int32_t(i1 - i0);
  d.cb = int32_t(j1 - j0);
  desc[t] = d;
}

// Coalesced staging of src[0, cnt) into LDS by one wave: 16-byte global loads on the
// aligned interior, scalar loads on the ragged ends.
template <typename KeyT>
__device__ __forceinline__ void stage(const KeyT* __restrict__ src, int cnt, KeyT* __restrict__ dst) {
  constexpr int kPer = 16 / int(sizeof(KeyT));
  const int mis = int((reinterpret_cast<uintptr_t>(src) / sizeof(KeyT)) & (kPer - 1));
  const int head = (kPer - mis) & (kPer - 1);
  const int h = head < cnt ? head : cnt;
  if (int(threadIdx.x) < h) dst[threadIdx.x] = src[threadIdx.x];
  const int n_vec = (cnt - h) / kPer;
  using Vec = typename std::conditional<sizeof(KeyT) == 4, uint4, ulonglong2>::type;
  const Vec* vp = reinterpret_cast<const Vec*>(src + h);
  for (int v = threadIdx.x; v < n_vec; v += kThreads) {
    const Vec x = vp[v];
    KeyT* d = dst + h + v * kPer;
    if constexpr (sizeof(KeyT) == 4) {
      d[0] = x.x;
      d[1] = x.y;
      d[2] = x.z;
      d[3] = x.w;
    } else {
      d[0] = x.x;
      d[1] = x.y;
    }
  }
  const int done = h + n_vec * kPer;
  if (int(threadIdx.x) < cnt - done) dst[done + threadIdx.x] = src[done + threadIdx.x];
}

// ---- the merge kernel ------------------------------------------------------------------------
// kMode == 0: tile_m[t] = number of common keys in tile t.
// kMode == 1: compacts the tile's A&B / A\B / B\A keys and writes them at
//             tile_ioff[t], a0 - tile_ioff[t], b0 - tile_ioff[t].
// kMode == 2: writes the tile's A|B keys (merged order, common keys once) to out_i at
//             a0 + b0 - tile_ioff[t]  (KmerSet::Add, kmer_set.h:164-174).
template <typename KeyT, int kMode>
__global__ __launch_bounds__(kThreads) void k_tile_merge(
    const TileDesc* __restrict__ desc, const int64_t* __restrict__ total_tiles,
    int64_t* __restrict__ tile_m, const int64_t* __restrict__ tile_ioff, KeyT* __restrict__ out_i,
    KeyT* __restrict__ out_amb, KeyT* __restrict__ out_bma) {
  __shared__ KeyT lds[kTileCap + 1];  // + 1: a clamped read past an empty B range stays inside

  const int64_t t = blockIdx.x;
  constexpr bool kWrite = kMode != 0;
  if (t >= *total_tiles) {
    if (!kWrite && threadIdx.x == 0) tile_m[t] = 0;
    return;
  }
  const TileDesc d = desc[t];
  const int ca = d.ca, cb = d.cb;
  KeyT* sa = lds;
  KeyT* sb = lds + ca;
  stage(static_cast<const KeyT*>(d.pa), ca, sa);
  stage(static_cast<const KeyT*>(d.pb), cb, sb);
  __syncthreads();

  const int n = ca + cb;
  const int d0 = min(int(threadIdx.x) * kVT, n);
  const int d1 = min(d0 + kVT, n);
  // merge-path split of this lane's diagonal (A first on ties)
  int lo = max(0, d0 - cb), hi = min(d0, ca);
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (sa[mid] <= sb[d0 - 1 - mid]) lo = mid + 1; else hi = mid;
  }
  int i = lo, j = d0 - lo;

  // Branch-free merge: both heads are (re)loaded every step with clamped indices and the
  // lane advances one of them; divergent lanes would otherwise execute both sides of every
  // branch.  last_a is the A key merged just before (possibly by the previous lane): a B
  // key equal to it is the second half of a common pair.
  KeyT vals[kVT];
  uint32_t cls_bits = 0;  // 2 bits per step: 0 = A&B, 1 = A\B, 2 = B\A, 3 = nothing
  int n_i = 0, n_a = 0, n_b = 0;
  bool has_last = i > 0;
  KeyT last_a = sa[has_last ? i - 1 : 0];
#pragma unroll
  for (int step = 0; step < kVT; step++) {
    const bool active = d0 + step < d1;
    const bool has_a = i < ca, has_b = j < cb;
    const KeyT av = sa[has_a ? i : 0];
    const KeyT bv = sb[has_b ? j : 0];
    const bool take_a = has_a && (!has_b || av <= bv);
    const bool eq_ab = has_a && has_b && av == bv;
    const bool eq_prev = has_last && last_a == bv;
    uint32_t c = take_a ? (eq_ab ? 0u : 1u) : (eq_prev ? 3u : 2u);
    c = active ? c : 3u;
    vals[step] = take_a ? av : bv;
    cls_bits |= c << (2 * step);
    n_i += c == 0;
    n_a += c == 1;
    n_b += c == 2;
    const bool adv_a = take_a && active;
    last_a = adv_a ? av : last_a;
    has_last = has_last || adv_a;
    i += adv_a ? 1 : 0;
    j += (!take_a && active) ? 1 : 0;
  }

  // wave scan of the three counts, packed into one word
  const uint64_t packed = uint64_t(n_i) | (uint64_t(n_a) << 20) | (uint64_t(n_b) << 40);
  uint64_t inc = packed;
  const int lane = threadIdx.x;
#pragma unroll
  for (int dd = 1; dd < 64; dd <<= 1) {
    const uint64_t o = __shfl_up(inc, dd, 64);
    if (lane >= dd) inc += o;
  }
  const uint64_t tot = __shfl(inc, 63, 64);
  const uint64_t excl = inc - packed;
  const int tot_i = int(tot & 0xFFFFF), tot_a = int((tot >> 20) & 0xFFFFF), tot_b = int(tot >> 40);

  if (!kWrite) {
    if (lane == 0) tile_m[t] = tot_i;
    return;
  }

  __syncthreads();  // every lane is done reading sa / sb: reuse the LDS for compaction
  const int64_t ioff = tile_ioff[t];
  if (kMode == 2) {
    int p_u = int(excl & 0xFFFFF) + int((excl >> 20) & 0xFFFFF) + int(excl >> 40);
#pragma unroll
    for (int step = 0; step < kVT; step++)
      if (((cls_bits >> (2 * step)) & 3) != 3) lds[p_u++] = vals[step];
    __syncthreads();
    const int tot_u = tot_i + tot_a + tot_b;
    KeyT* o = out_i + (d.a0 + d.b0 - ioff);
    for (int x = lane; x < tot_u; x += kThreads) o[x] = lds[x];
    return;
  }
  int p_i = int(excl & 0xFFFFF);
  int p_a = tot_i + int((excl >> 20) & 0xFFFFF);
  int p_b = tot_i + tot_a + int(excl >> 40);
#pragma unroll
  for (int step = 0; step < kVT; step++) {
    const uint32_t c = (cls_bits >> (2 * step)) & 3;
    const int pos = c == 0 ? p_i : (c == 1 ? p_a : p_b);
    if (c != 3) lds[pos] = vals[step];
    p_i += c == 0;
    p_a += c == 1;
    p_b += c == 2;
  }
  __syncthreads();
  if (out_i) {
    KeyT* o = out_i + ioff;
    for (int x = lane; x < tot_i; x += kThreads) o[x] = lds[x];
  }
  if (out_amb) {
    KeyT* o = out_amb + (d.a0 - ioff);
    for (int x = lane; x < tot_a; x += kThreads) o[x] = lds[tot_i + x];
  }
  if (out_bma) {
    KeyT* o = out_bma + (d.b0 - ioff);
    for (int x = lane; x < tot_b; x += kThreads) o[x] = lds[tot_i + tot_a + x];
  }
}

// Bucket offsets of the three results from the per-tile prefix of common keys.
__global__ __launch_bounds__(256) void k_result_offsets(
    const int64_t* __restrict__ a_off, const int64_t* __restrict__ b_off,
    const int64_t* __restrict__ tile_base, const int64_t* __restrict__ tile_ioff,
    const int64_t* __restrict__ total_m, int64_t n_buckets, int64_t* __restrict__ off_i,
    int64_t* __restrict__ off_amb, int64_t* __restrict__ off_bma, int64_t* __restrict__ totals3) {
  const int64_t s = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (s > n_buckets) return;
  const int64_t total_tiles = tile_base[n_buckets];
  const int64_t first = tile_base[s];
  const int64_t m = first < total_tiles ? tile_ioff[first] : *total_m;
  off_i[s] = m;
  off_amb[s] = a_off[s] - m;
  off_bma[s] = b_off[s] - m;
  if (s == n_buckets) {
    totals3[0] = m;
    totals3[1] = a_off[s] - m;
    totals3[2] = b_off[s] - m;
  }
}

// weights[p] = common keys over the segments [p * n_ids, (p + 1) * n_ids).
__global__ __launch_bounds__(256) void k_pair_weight_gather(
    const int64_t* __restrict__ tile_base, const int64_t* __restrict__ tile_ioff,
    const int64_t* __restrict__ total_m, int64_t n_segs, int32_t n_ids, int32_t n_pairs,
    int64_t* __restrict__ weights) {
  const int32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_pairs) return;
  const int64_t total_tiles = tile_base[n_segs];
  const int64_t t0 = tile_base[int64_t(p) * n_ids];
  const int64_t t1 = tile_base[int64_t(p + 1) * n_ids];
  const int64_t m0 = t0 < total_tiles ? tile_ioff[t0] : *total_m;
  const int64_t m1 = t1 < total_tiles ? tile_ioff[t1] : *total_m;
  weights[p] = m1 - m0;
}

// ---- host-side plan ------------------------------------------------------------------------------
struct Plan {
  int64_t n_segs = 0;
  int64_t max_tiles = 0;
  int64_t* tile_base = nullptr;  // n_segs + 1
  TileDesc* desc = nullptr;      // max_tiles
  int64_t* tile_ioff = nullptr;  // max_tiles (count, then exclusive prefix in place)
  int64_t* total_m = nullptr;    // 1
};

inline size_t align256(size_t x) { return (x + 255) & ~size_t(255); }

inline size_t plan_bytes(int64_t n_segs, int64_t max_tiles) {
  return align256(size_t(n_segs + 1) * 8) + align256(size_t(max_tiles) * sizeof(TileDesc)) +
         align256(size_t(max_tiles) * 8) + 256;
}

inline void plan_carve(char* base, int64_t n_segs, int64_t max_tiles, Plan* p) {
  p->n_segs = n_segs;
  p->max_tiles = max_tiles;
  char* at = base;
  p->tile_base = reinterpret_cast<int64_t*>(at);
  at += align256(size_t(n_segs + 1) * 8);
  p->desc = reinterpret_cast<TileDesc*>(at);
  at += align256(size_t(max_tiles) * sizeof(TileDesc));
  p->tile_ioff = reinterpret_cast<int64_t*>(at);
  at += align256(size_t(max_tiles) * 8);
  p->total_m = reinterpret_cast<int64_t*>(at);
}

inline unsigned blocks_for(int64_t n, int per) { return unsigned(std::max<int64_t>(1, (n + per - 1) / per)); }

// Tiles per segment -> tile_base (exclusive prefix, total at [n_segs]).
template <typename KeyT, typename Segs>
int plan_tile_base(ksh_ctx* ctx, const Segs& segs, int64_t n_segs, int64_t* tile_base) {
  hipLaunchKernelGGL((k_seg_tiles<KeyT, Segs>), dim3(blocks_for(n_segs, 256)), dim3(256), 0,
                     ctx->stream, segs, n_segs, tile_base);
  KSH_TRY(scan_exclusive_i64(ctx, tile_base, tile_base, n_segs, tile_base + n_segs));
  return KSH_OK;
}

// Splits + count pass + prefix of the per-tile counts.
template <typename KeyT, typename Segs>
int plan_count(ksh_ctx* ctx, const Segs& segs, const Plan& p, int timer_kind) {
  hipLaunchKernelGGL((k_tile_split<KeyT, Segs>), dim3(blocks_for(p.max_tiles, 256)), dim3(256), 0,
                     ctx->stream, segs, p.n_segs, p.tile_base, p.desc);
  {
    Timer timer(ctx, timer_kind);
    hipLaunchKernelGGL((k_tile_merge<KeyT, 0>), dim3(unsigned(p.max_tiles)), dim3(kThreads), 0,
                       ctx->stream, p.desc, p.tile_base + p.n_segs, p.tile_ioff, nullptr,
                       static_cast<KeyT*>(nullptr), static_cast<KeyT*>(nullptr),
                       static_cast<KeyT*>(nullptr));
  }
  KSH_TRY(scan_exclusive_i64(ctx, p.tile_ioff, p.tile_ioff, p.max_tiles, p.total_m));
  KSH_HIP(hipGetLastError());
  return KSH_OK;
}

int check_view(const ksh_set_view* v, const char* name) {
  if (!v) return fail(KSH_INVALID_ARGUMENT, "%s is NULL", name);
  if (!v->d_offsets) return fail(KSH_INVALID_ARGUMENT, "%s.d_offsets is NULL", name);
  if (v->n_keys < 0) return fail(KSH_INVALID_ARGUMENT, "%s.n_keys < 0", name);
  if (v->n_keys > 0 && !v->d_keys) return fail(KSH_INVALID_ARGUMENT, "%s.d_keys is NULL", name);
  if (reinterpret_cast<uintptr_t>(v->d_keys) & 15)
    return fail(KSH_INVALID_ARGUMENT, "%s.d_keys must be 16-byte aligned", name);
  return KSH_OK;
}

template <typename KeyT>
int pair_plan_t(ksh_ctx* ctx, const ksh_geom* g, const ksh_set_view* a, const ksh_set_view* b,
                int64_t* d_off_i, int64_t* d_off_amb, int64_t* d_off_bma, int64_t totals[3]) {
  const int64_t nb = n_buckets(g);
  const int64_t max_tiles = nb + (a->n_keys + b->n_keys) / kTile + 1;
  if (max_tiles > int64_t(0x7FFFFFF0)) return fail(KSH_INVALID_ARGUMENT, "pair too large for one launch");
  KSH_TRY(plan_reserve(ctx, plan_bytes(nb, max_tiles)));
  KSH_TRY(arena_reserve(ctx, size_t(max_tiles / 256 + 4096) * 8 + (1u << 16)));
  arena_reset(ctx);
  Plan p;
  plan_carve(ctx->plan, nb, max_tiles, &p);
  BucketSegs<KeyT> segs{static_cast<const KeyT*>(a->d_keys), a->d_offsets,
                        static_cast<const KeyT*>(b->d_keys), b->d_offsets};
  KSH_TRY((plan_tile_base<KeyT>(ctx, segs, nb, p.tile_base)));
  KSH_TRY((plan_count<KeyT>(ctx, segs, p, 1)));
  int64_t* d_totals = static_cast<int64_t*>(arena_alloc(ctx, 3 * sizeof(int64_t)));
  if (!d_totals) return fail(KSH_INTERNAL, "scratch arena too small");
  hipLaunchKernelGGL(k_result_offsets, dim3(blocks_for(nb + 1, 256)), dim3(256), 0, ctx->stream,
                     a->d_offsets, b->d_offsets, p.tile_base, p.tile_ioff, p.total_m, nb, d_off_i,
                     d_off_amb, d_off_bma, d_totals);
  KSH_HIP(hipGetLastError());
  KSH_HIP(hipMemcpyAsync(ctx->h_pinned, d_totals, 3 * sizeof(int64_t), hipMemcpyDeviceToHost,
                         ctx->stream));
  KSH_HIP(hipStreamSynchronize(ctx->stream));
  totals[0] = ctx->h_pinned[0];
  totals[1] = ctx->h_pinned[1];
  totals[2] = ctx->h_pinned[2];
  ctx->plan_tiles = max_tiles;
  ctx->plan_buckets = nb;
  ctx->plan_a_keys = a->d_keys;
  ctx->plan_b_keys = b->d_keys;
  return KSH_OK;
}

template <typename KeyT>
int pair_write_t(ksh_ctx* ctx, const ksh_geom* g, const ksh_set_view* a, const ksh_set_view* b,
                 void* d_keys_i, void* d_keys_amb, void* d_keys_bma) {
  const int64_t nb = n_buckets(g);
  if (!ctx->plan || ctx->plan_buckets != nb || ctx->plan_a_keys != a->d_keys ||
      ctx->plan_b_keys != b->d_keys)
    return fail(KSH_FAILED_PRECONDITION, "ksh_pair_write without a matching ksh_pair_plan");
  Plan p;
  plan_carve(ctx->plan, nb, ctx->plan_tiles, &p);
  {
    Timer timer(ctx, 0);
    hipLaunchKernelGGL((k_tile_merge<KeyT, 1>), dim3(unsigned(p.max_tiles)), dim3(kThreads), 0,
                       ctx->stream, p.desc, p.tile_base + p.n_segs, nullptr, p.tile_ioff,
                       static_cast<KeyT*>(d_keys_i), static_cast<KeyT*>(d_keys_amb),
                       static_cast<KeyT*>(d_keys_bma));
  }
  KSH_HIP(hipGetLastError());
  return KSH_OK;
}

// Plan + write back to back with caller-provided upper-bound buffers: one stream sync
// (for the totals) instead of two and no allocation between the passes.
inline size_t pair_scratch_bytes(int64_t nb, int64_t max_tiles) {
  return plan_bytes(nb, max_tiles) + size_t(max_tiles / 256 + 4096) * 8 + (1u << 16);
}

// Enqueues both passes of one pair; the three totals land in d_totals (device).  The scratch
// arena must already be large enough (reuse across consecutive pairs is stream-ordered).
template <typename KeyT>
int pair_algebra_enqueue(ksh_ctx* ctx, const ksh_geom* g, const ksh_set_view* a, const ksh_set_view* b,
                         int64_t* d_off_i, int64_t* d_off_amb, int64_t* d_off_bma, void* d_keys_i,
                         void* d_keys_amb, void* d_keys_bma, int64_t* d_totals) {
  const int64_t nb = n_buckets(g);
  const int64_t max_tiles = nb + (a->n_keys + b->n_keys) / kTile + 1;
  arena_reset(ctx);
  char* base = static_cast<char*>(arena_alloc(ctx, plan_bytes(nb, max_tiles)));
  Plan p;
  plan_carve(base, nb, max_tiles, &p);
  BucketSegs<KeyT> segs{static_cast<const KeyT*>(a->d_keys), a->d_offsets,
                        static_cast<const KeyT*>(b->d_keys), b->d_offsets};
  KSH_TRY((plan_tile_base<KeyT>(ctx, segs, nb, p.tile_base)));
  KSH_TRY((plan_count<KeyT>(ctx, segs, p, 1)));
  hipLaunchKernelGGL(k_result_offsets, dim3(blocks_for(nb + 1, 256)), dim3(256), 0, ctx->stream,
                     a->d_offsets, b->d_offsets, p.tile_base, p.tile_ioff, p.total_m, nb, d_off_i,
                     d_off_amb, d_off_bma, d_totals);
  {
    Timer timer(ctx, 0);
    hipLaunchKernelGGL((k_tile_merge<KeyT, 1>), dim3(unsigned(p.max_tiles)), dim3(kThreads), 0,
                       ctx->stream, p.desc, p.tile_base + p.n_segs, nullptr, p.tile_ioff,
                       static_cast<KeyT*>(d_keys_i), static_cast<KeyT*>(d_keys_amb),
                       static_cast<KeyT*>(d_keys_bma));
  }
  KSH_HIP(hipGetLastError());
  return KSH_OK;
}

// One pair or many: all pairs are enqueued back to back and ONE stream synchronisation brings
// every pair's totals to the host (the pairs of a batch are independent, as in the reference's
// pooled loops over pairs, kmer_set_set.h:205-216).
template <typename KeyT>
int pair_algebra_batch_t(ksh_ctx* ctx, const ksh_geom* g, ksh_pair_job* jobs, int32_t n_jobs) {
  const int64_t nb = n_buckets(g);
  size_t need = 0;
  for (int32_t i = 0; i < n_jobs; i++) {
    const int64_t max_tiles = nb + (jobs[i].a.n_keys + jobs[i].b.n_keys) / kTile + 1;
    if (max_tiles > int64_t(0x7FFFFFF0)) return fail(KSH_INVALID_ARGUMENT, "pair too large for one launch");
    need = std::max(need, pair_scratch_bytes(nb, max_tiles));
  }
  KSH_TRY(arena_reserve(ctx, need));
  void* tot = nullptr;
  KSH_TRY(pool_alloc(ctx, size_t(n_jobs) * 3 * sizeof(int64_t), &tot));
  int64_t* d_totals = static_cast<int64_t*>(tot);
  int rc = KSH_OK;
  for (int32_t i = 0; i < n_jobs && rc == KSH_OK; i++) {
    ksh_pair_job& j = jobs[i];
    rc = pair_algebra_enqueue<KeyT>(ctx, g, &j.a, &j.b, j.d_off_i, j.d_off_amb, j.d_off_bma, j.d_keys_i,
                                    j.d_keys_amb, j.d_keys_bma, d_totals + 3 * i);
  }
  const size_t n_vals = static_cast<size_t>(n_jobs) * 3;
  int64_t* h = ctx->h_pinned;  // 64 values
  if (n_vals > 64) {
    if (ctx->h_batch_count < n_vals) {
      if (ctx->h_batch) (void)hipHostFree(ctx->h_batch);
      ctx->h_batch = nullptr;
      ctx->h_batch_count = 0;
      if (hipHostMalloc(reinterpret_cast<void**>(&ctx->h_batch), n_vals * 2 * sizeof(int64_t)) != hipSuccess) {
        pool_free(ctx, tot);
        return fail(KSH_INTERNAL, "hipHostMalloc failed");
      }
      ctx->h_batch_count = n_vals * 2;
    }
    h = ctx->h_batch;
  }
  if (rc == KSH_OK) {
    hipError_t e = hipMemcpyAsync(h, d_totals, n_vals * sizeof(int64_t), hipMemcpyDeviceToHost,
                                  ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) rc = fail(KSH_INTERNAL, "totals read-back failed: %s", hipGetErrorString(e));
  }
  pool_free(ctx, tot);
  if (rc != KSH_OK) return rc;
  for (int32_t i = 0; i < n_jobs; i++)
    for (int q = 0; q < 3; q++) jobs[i].totals[q] = h[size_t(3 * i + q)];
  return KSH_OK;
}

// A | B in two calls like the pair plan: offsets + total first, keys second.
__global__ __launch_bounds__(256) void k_union_offsets(
    const int64_t* __restrict__ a_off, const int64_t* __restrict__ b_off,
    const int64_t* __restrict__ tile_base, const int64_t* __restrict__ tile_ioff,
    const int64_t* __restrict__ total_m, int64_t n_buckets, int64_t* __restrict__ off_u,
    int64_t* __restrict__ total_u) {
  const int64_t s = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (s > n_buckets) return;
  const int64_t total_tiles = tile_base[n_buckets];
  const int64_t first = tile_base[s];
  const int64_t m = first < total_tiles ? tile_ioff[first] : *total_m;
  off_u[s] = a_off[s] + b_off[s] - m;
  if (s == n_buckets) *total_u = a_off[s] + b_off[s] - m;
}

template <typename KeyT>
int union_plan_t(ksh_ctx* ctx, const ksh_geom* g, const ksh_set_view* a, const ksh_set_view* b,
                 int64_t* d_off_u, int64_t* total) {
  const int64_t nb = n_buckets(g);
  const int64_t max_tiles = nb + (a->n_keys + b->n_keys) / kTile + 1;
  if (max_tiles > int64_t(0x7FFFFFF0)) return fail(KSH_INVALID_ARGUMENT, "pair too large for one launch");
  KSH_TRY(plan_reserve(ctx, plan_bytes(nb, max_tiles)));
  KSH_TRY(arena_reserve(ctx, size_t(max_tiles / 256 + 4096) * 8 + (1u << 16)));
  arena_reset(ctx);
  Plan p;
  plan_carve(ctx->plan, nb, max_tiles, &p);
  BucketSegs<KeyT> segs{static_cast<const KeyT*>(a->d_keys), a->d_offsets,
                        static_cast<const KeyT*>(b->d_keys), b->d_offsets};
  KSH_TRY((plan_tile_base<KeyT>(ctx, segs, nb, p.tile_base)));
  KSH_TRY((plan_count<KeyT>(ctx, segs, p, 1)));
  int64_t* d_total = static_cast<int64_t*>(arena_alloc(ctx, sizeof(int64_t)));
  if (!d_total) return fail(KSH_INTERNAL, "scratch arena too small");
  hipLaunchKernelGGL(k_union_offsets, dim3(blocks_for(nb + 1, 256)), dim3(256), 0, ctx->stream,
                     a->d_offsets, b->d_offsets, p.tile_base, p.tile_ioff, p.total_m, nb, d_off_u,
                     d_total);
  KSH_HIP(hipGetLastError());
  KSH_HIP(hipMemcpyAsync(ctx->h_pinned, d_total, sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
  KSH_HIP(hipStreamSynchronize(ctx->stream));
  *total = ctx->h_pinned[0];
  ctx->plan_tiles = max_tiles;
  ctx->plan_buckets = nb;
  ctx->plan_a_keys = a->d_keys;
  ctx->plan_b_keys = b->d_keys;
  return KSH_OK;
}

template <typename KeyT>
int union_write_t(ksh_ctx* ctx, const ksh_geom* g, const ksh_set_view* a, const ksh_set_view* b,
                  void* d_keys_u) {
  const int64_t nb = n_buckets(g);
  if (!ctx->plan || ctx->plan_buckets != nb || ctx->plan_a_keys != a->d_keys ||
      ctx->plan_b_keys != b->d_keys)
    return fail(KSH_FAILED_PRECONDITION, "ksh_set_union_write without a matching ksh_set_union_plan");
  Plan p;
  plan_carve(ctx->plan, nb, ctx->plan_tiles, &p);
  hipLaunchKernelGGL((k_tile_merge<KeyT, 2>), dim3(unsigned(p.max_tiles)), dim3(kThreads), 0,
                     ctx->stream, p.desc, p.tile_base + p.n_segs, nullptr, p.tile_ioff,
                     static_cast<KeyT*>(d_keys_u), static_cast<KeyT*>(nullptr),
                     static_cast<KeyT*>(nullptr));
  KSH_HIP(hipGetLastError());
  return KSH_OK;
}

template <typename KeyT>
int set_diff_t(ksh_ctx* ctx, const ksh_geom* g, const ksh_set_view* a, const ksh_set_view* b,
               int64_t* diff) {
  const int64_t nb = n_buckets(g);
  const int64_t max_tiles = nb + (a->n_keys + b->n_keys) / kTile + 1;
  if (max_tiles > int64_t(0x7FFFFFF0)) return fail(KSH_INVALID_ARGUMENT, "pair too large for one launch");
  KSH_TRY(arena_reserve(ctx, plan_bytes(nb, max_tiles) + size_t(max_tiles / 256 + 4096) * 8 + (1u << 16)));
  arena_reset(ctx);
  char* base = static_cast<char*>(arena_alloc(ctx, plan_bytes(nb, max_tiles)));
  Plan p;
  plan_carve(base, nb, max_tiles, &p);
  BucketSegs<KeyT> segs{static_cast<const KeyT*>(a->d_keys), a->d_offsets,
                        static_cast<const KeyT*>(b->d_keys), b->d_offsets};
  KSH_TRY((plan_tile_base<KeyT>(ctx, segs, nb, p.tile_base)));
  KSH_TRY((plan_count<KeyT>(ctx, segs, p, 1)));
  KSH_HIP(hipMemcpyAsync(ctx->h_pinned, p.total_m, sizeof(int64_t), hipMemcpyDeviceToHost,
                         ctx->stream));
  KSH_HIP(hipStreamSynchronize(ctx->stream));
  *diff = a->n_keys + b->n_keys - 2 * ctx->h_pinned[0];
  return KSH_OK;
}

template <typename KeyT>
int pair_weights_t(ksh_ctx* ctx, const ksh_geom* g, const ksh_set_view* sets, int32_t n_sets,
                   const int32_t* bucket_ids, int32_t n_ids, const int32_t* pairs, int32_t n_pairs,
                   int64_t* weights) {
  (void)g;
  const int64_t n_segs = int64_t(n_pairs) * n_ids;
  // stage the descriptors
  const size_t desc_bytes = align256(size_t(n_sets) * sizeof(SetPtrs)) +
                            align256(size_t(n_ids) * 4) + align256(size_t(n_pairs) * 8) +
                            align256(size_t(n_pairs) * 8);
  const size_t base_bytes = align256(size_t(n_segs + 1) * 8);
  KSH_TRY(arena_reserve(ctx, desc_bytes + base_bytes + size_t(n_segs / 256 + 4096) * 8 + (1u << 16)));
  arena_reset(ctx);
  SetPtrs* d_sets = static_cast<SetPtrs*>(arena_alloc(ctx, size_t(n_sets) * sizeof(SetPtrs)));
  int32_t* d_ids = static_cast<int32_t*>(arena_alloc(ctx, size_t(n_ids) * 4));
  int32_t* d_pairs = static_cast<int32_t*>(arena_alloc(ctx, size_t(n_pairs) * 8));
  int64_t* d_weights = static_cast<int64_t*>(arena_alloc(ctx, size_t(n_pairs) * 8));
  int64_t* tile_base = static_cast<int64_t*>(arena_alloc(ctx, size_t(n_segs + 1) * 8));
  if (!d_sets || !d_ids || !d_pairs || !d_weights || !tile_base)
    return fail(KSH_INTERNAL, "scratch arena too small");
  std::vector<SetPtrs> h_sets(static_cast<size_t>(n_sets));
  for (int32_t i = 0; i < n_sets; i++) h_sets[i] = SetPtrs{sets[i].d_keys, sets[i].d_offsets};
  KSH_HIP(hipMemcpyAsync(d_sets, h_sets.data(), size_t(n_sets) * sizeof(SetPtrs),
                         hipMemcpyHostToDevice, ctx->stream));
  KSH_HIP(hipMemcpyAsync(d_ids, bucket_ids, size_t(n_ids) * 4, hipMemcpyHostToDevice, ctx->stream));
  KSH_HIP(hipMemcpyAsync(d_pairs, pairs, size_t(n_pairs) * 8, hipMemcpyHostToDevice, ctx->stream));
  PairSegs<KeyT> segs{d_sets, d_ids, d_pairs, n_ids};
  KSH_TRY((plan_tile_base<KeyT>(ctx, segs, n_segs, tile_base)));
  // exact tile count (the sampled slices are ~2 % of each set; no useful bound without it)
  KSH_HIP(hipMemcpyAsync(ctx->h_pinned, tile_base + n_segs, sizeof(int64_t), hipMemcpyDeviceToHost,
                         ctx->stream));
  KSH_HIP(hipStreamSynchronize(ctx->stream));  // also keeps h_sets alive long enough
  const int64_t n_tiles = std::max<int64_t>(ctx->h_pinned[0], 1);
  if (n_tiles > int64_t(0x7FFFFFF0)) return fail(KSH_INVALID_ARGUMENT, "too many tiles for one launch");
  // the remaining plan arrays: keep what is already in the arena, so grow by allocating a second
  // block if needed
  const size_t rest = plan_bytes(0, n_tiles) + size_t(n_tiles / 256 + 4096) * 8 + (1u << 16);
  if (ctx->arena_used + rest > ctx->arena_bytes) {
    // re-run with a bigger arena (descriptors are re-staged); rare: first call at a new size
    KSH_TRY(arena_reserve(ctx, ctx->arena_used + rest + desc_bytes + base_bytes));
    return pair_weights_t<KeyT>(ctx, g, sets, n_sets, bucket_ids, n_ids, pairs, n_pairs, weights);
  }
  char* base = static_cast<char*>(arena_alloc(ctx, plan_bytes(0, n_tiles)));
  Plan p;
  plan_carve(base, 0, n_tiles, &p);
  p.n_segs = n_segs;
  p.tile_base = tile_base;
  KSH_TRY((plan_count<KeyT>(ctx, segs, p, 2)));
  hipLaunchKernelGGL(k_pair_weight_gather, dim3(blocks_for(n_pairs, 256)), dim3(256), 0,
                     ctx->stream, p.tile_base, p.tile_ioff, p.total_m, n_segs, n_ids, n_pairs,
                     d_weights);
  KSH_HIP(hipGetLastError());
  KSH_HIP(hipMemcpyAsync(weights, d_weights, size_t(n_pairs) * 8, hipMemcpyDeviceToHost,
                         ctx->stream));
  KSH_HIP(hipStreamSynchronize(ctx->stream));
  return KSH_OK;
}

}  // namespace ksh

using namespace ksh;

extern "C" {

int ksh_pair_plan(ksh_ctx* ctx, const ksh_geom* g, const ksh_set_view* a, const ksh_set_view* b,
                  int64_t* d_off_i, int64_t* d_off_amb, int64_t* d_off_bma, int64_t totals[3]) {
  if (!ctx || !d_off_i || !d_off_amb || !d_off_bma || !totals)
    return fail(KSH_INVALID_ARGUMENT, "NULL argument");
  KSH_TRY(check_geom(g));
  KSH_TRY(check_view(a, "a"));
  KSH_TRY(check_view(b, "b"));
  KSH_HIP(hipSetDevice(ctx->device));
  return g->key_bytes == 4 ? pair_plan_t<uint32_t>(ctx, g, a, b, d_off_i, d_off_amb, d_off_bma, totals)
                           : pair_plan_t<uint64_t>(ctx, g, a, b, d_off_i, d_off_amb, d_off_bma, totals);
}

int ksh_pair_write(ksh_ctx* ctx, const ksh_geom* g, const ksh_set_view* a, const ksh_set_view* b,
                   void* d_keys_i, void* d_keys_amb, void* d_keys_bma) {
  if (!ctx) return fail(KSH_INVALID_ARGUMENT, "ctx is NULL");
  KSH_TRY(check_geom(g));
  KSH_TRY(check_view(a, "a"));
  KSH_TRY(check_view(b, "b"));
  KSH_HIP(hipSetDevice(ctx->device));
  return g->key_bytes == 4 ? pair_write_t<uint32_t>(ctx, g, a, b, d_keys_i, d_keys_amb, d_keys_bma)
                           : pair_write_t<uint64_t>(ctx, g, a, b, d_keys_i, d_keys_amb, d_keys_bma);
}

int ksh_pair_algebra(ksh_ctx* ctx, const ksh_geom* g, const ksh_set_view* a, const ksh_set_view* b,
                     int64_t* d_off_i, int64_t* d_off_amb, int64_t* d_off_bma, void* d_keys_i,
                     void* d_keys_amb, void* d_keys_bma, int64_t totals[3]) {
  if (!ctx || !d_off_i || !d_off_amb || !d_off_bma || !totals || !d_keys_i || !d_keys_amb ||
      !d_keys_bma)
    return fail(KSH_INVALID_ARGUMENT, "NULL argument");
  KSH_TRY(check_geom(g));
  KSH_TRY(check_view(a, "a"));
  KSH_TRY(check_view(b, "b"));
  KSH_HIP(hipSetDevice(ctx->device));
  ksh_pair_job job{*a, *b, d_off_i, d_off_amb, d_off_bma, d_keys_i, d_keys_amb, d_keys_bma, {0, 0, 0}};
  const int rc = g->key_bytes == 4 ? pair_algebra_batch_t<uint32_t>(ctx, g, &job, 1)
                                   : pair_algebra_batch_t<uint64_t>(ctx, g, &job, 1);
  if (rc != KSH_OK) return rc;
  totals[0] = job.totals[0];
  totals[1] = job.totals[1];
  totals[2] = job.totals[2];
  return KSH_OK;
}

int ksh_pair_algebra_batch(ksh_ctx* ctx, const ksh_geom* g, ksh_pair_job* jobs, int32_t n_jobs) {
  if (!ctx || (n_jobs > 0 && !jobs)) return fail(KSH_INVALID_ARGUMENT, "NULL argument");
  KSH_TRY(check_geom(g));
  if (n_jobs <= 0) return KSH_OK;
  for (int32_t i = 0; i < n_jobs; i++) {
    KSH_TRY(check_view(&jobs[i].a, "jobs[i].a"));
    KSH_TRY(check_view(&jobs[i].b, "jobs[i].b"));
    if (!jobs[i].d_off_i || !jobs[i].d_off_amb || !jobs[i].d_off_bma || !jobs[i].d_keys_i ||
        !jobs[i].d_keys_amb || !jobs[i].d_keys_bma)
      return fail(KSH_INVALID_ARGUMENT, "jobs[%d] has a NULL output", i);
  }
  KSH_HIP(hipSetDevice(ctx->device));
  return g->key_bytes == 4 ? pair_algebra_batch_t<uint32_t>(ctx, g, jobs, n_jobs)
                           : pair_algebra_batch_t<uint64_t>(ctx, g, jobs, n_jobs);
}

int ksh_set_union_plan(ksh_ctx* ctx, const ksh_geom* g, const ksh_set_view* a,
                       const ksh_set_view* b, int64_t* d_off_u, int64_t* total) {
  if (!ctx || !d_off_u || !total) return fail(KSH_INVALID_ARGUMENT, "NULL argument");
  KSH_TRY(check_geom(g));
  KSH_TRY(check_view(a, "a"));
  KSH_TRY(check_view(b, "b"));
  KSH_HIP(hipSetDevice(ctx->device));
  return g->key_bytes == 4 ? union_plan_t<uint32_t>(ctx, g, a, b, d_off_u, total)
                           : union_plan_t<uint64_t>(ctx, g, a, b, d_off_u, total);
}

int ksh_set_union_write(ksh_ctx* ctx, const ksh_geom* g, const ksh_set_view* a,
                        const ksh_set_view* b, void* d_keys_u) {
  if (!ctx) return fail(KSH_INVALID_ARGUMENT, "ctx is NULL");
  KSH_TRY(check_geom(g));
  KSH_TRY(check_view(a, "a"));
  KSH_TRY(check_view(b, "b"));
  KSH_HIP(hipSetDevice(ctx->device));
  return g->key_bytes == 4 ? union_write_t<uint32_t>(ctx, g, a, b, d_keys_u)
                           : union_write_t<uint64_t>(ctx, g, a, b, d_keys_u);
}

int ksh_set_diff(ksh_ctx* ctx, const ksh_geom* g, const ksh_set_view* a, const ksh_set_view* b,
                 int64_t* diff) {
  if (!ctx || !diff) return fail(KSH_INVALID_ARGUMENT, "NULL argument");
  KSH_TRY(check_geom(g));
  KSH_TRY(check_view(a, "a"));
  KSH_TRY(check_view(b, "b"));
  KSH_HIP(hipSetDevice(ctx->device));
  return g->key_bytes == 4 ? set_diff_t<uint32_t>(ctx, g, a, b, diff)
                           : set_diff_t<uint64_t>(ctx, g, a, b, diff);
}

int ksh_pair_weights(ksh_ctx* ctx, const ksh_geom* g, const ksh_set_view* sets, int32_t n_sets,
                     const int32_t* bucket_ids, int32_t n_ids, const int32_t* pairs,
                     int32_t n_pairs, int64_t* weights) {
  if (!ctx || !sets || !weights) return fail(KSH_INVALID_ARGUMENT, "NULL argument");
  KSH_TRY(check_geom(g));
  if (n_pairs <= 0) return KSH_OK;
  if (n_ids <= 0) {
    for (int32_t p = 0; p < n_pairs; p++) weights[p] = 0;
    return KSH_OK;
  }
  if (!bucket_ids || !pairs) return fail(KSH_INVALID_ARGUMENT, "NULL argument");
  const int64_t nb = n_buckets(g);
  for (int32_t i = 0; i < n_ids; i++)
    if (bucket_ids[i] < 0 || bucket_ids[i] >= nb)
      return fail(KSH_INVALID_ARGUMENT, "bucket_ids[%d] = %d is outside [0, 2^N)", i, bucket_ids[i]);
  for (int32_t p = 0; p < 2 * n_pairs; p++)
    if (pairs[p] < 0 || pairs[p] >= n_sets)
      return fail(KSH_INVALID_ARGUMENT, "pairs[%d] = %d is outside [0, n_sets)", p, pairs[p]);
  for (int32_t i = 0; i < n_sets; i++) KSH_TRY(check_view(&sets[i], "sets[i]"));
  KSH_HIP(hipSetDevice(ctx->device));
  return g->key_bytes == 4
             ? pair_weights_t<uint32_t>(ctx, g, sets, n_sets, bucket_ids, n_ids, pairs, n_pairs, weights)
             : pair_weights_t<uint64_t>(ctx, g, sets, n_sets, bucket_ids, n_ids, pairs, n_pairs, weights);
}

}  // extern "C"
