// Pair kernels over bucketed sorted key arrays: intersection / both differences
// (KmerSet::Sub / Intersection, lib/core/kmer_set.h:164-187,286-305), the
// symmetric-difference count (KmerSet::Diff, :191-219) and the sampled-bucket
// pair weights (GetEdgeWeight, lib/core/kmer_set_set.h:158-184).
//
// A set is one ascending array of k-mers with a bucket index, so every one of
// these is a merge of two sorted ranges per bucket.  The work is cut into
// *segments* (one per bucket of a pair, or one per (pair, sampled bucket)) and each
// segment into *tiles* of a bounded number of merged keys (TileCfg) by merge-path; one
// wavefront per tile stages the tile's two key ranges in LDS with 16-byte coalesced
// loads, every lane merges up to kVT keys from LDS, and the three result streams are
// compacted in LDS and written back coalesced.  HBM-bound integer work; no MFMA.
//
// Tie rule: on equal keys the A key is merged first, so a common key shows up as
// "a, b" adjacent in the merged order.  The tile split never separates such a
// pair (the split kernel moves the B boundary by one when it would), which makes
// every tile self-contained: matches are decided from LDS only.
#include "ksh_internal.h"
#include "ksh_scan.h"

#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include <vector>

namespace ksh {

// One wavefront per tile: 64 lanes x up to kVT keys each.  A bucket of a pair holds about
// 1.2 k merged keys at 10^7 keys per set and is cut into two balanced tiles of about 610; the
// per-tile fixed cost (descriptor and key loads, the split search, the wave scan) is what a small
// tile pays for (kVT = 15 and 13 measure the same on config 2, 11 and 9 are 6 % and 11 % slower).
// Single-wave workgroups need no cross-wave scan and no barrier that waits for another wave.
// The LDS footprint (4 KB) lets the hardware's 32 waves per CU be resident; 8-byte keys use
// smaller tiles to stay at that footprint.
constexpr int kThreads = 64;        // lanes per tile (one wavefront)
// Wavefronts per workgroup of k_tile_merge.  The waves of a workgroup work on different tiles and
// never talk to each other.  The dispatcher launches about 4.4 workgroups per ns whatever their
// size (tools/launch_rate.hip: 32768 empty single-wave workgroups take 8 us), but packing four
// waves into a workgroup measured the same kernel times on config 2 (dispatch overlaps the
// waves already running), so workgroups stay single waves: their LDS is released wave by wave.
constexpr int kGroupWaves = 1;

template <typename KeyT>
struct TileCfg {
#ifndef KSH_VT32
#define KSH_VT32 15
#endif
  static constexpr int kVT = sizeof(KeyT) <= 4 ? KSH_VT32 : 11;  // merged keys per lane, at most (odd)
  static constexpr int kTile = kThreads * kVT - 1;          // largest diagonal span (+1: tie fix-up)
  static constexpr int kCap = kThreads * kVT;               // keys in a tile, at most
  static constexpr int kPer = 16 / int(sizeof(KeyT));       // keys per 16-byte vector
  // A and B each keep their 16-byte phase (up to kPer - 1 slots before each, and the last
  // vector of each side may run kPer - 1 past it); kVT + 1 slots after B are only read.
  static constexpr int kLds = (kCap + 4 * (kPer - 1) + kVT + 1 + 63) & ~63;
  // 16-byte vectors that cover a tile's two ranges, per lane (both ranges rounded outwards)
  static constexpr int kVecs = (kCap / kPer + 2 + kThreads - 1) / kThreads;
};

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

// One segment per (sampled bucket i, pair p): s = i * n_pairs + p -- bucket-major, so that the workgroups in
// flight at any time work on the same few sampled buckets of all sets (64 sets x 6 100 keys x 4 B = 1.5 MB per
// bucket of a 10^8-k-mer family: L2-resident) instead of streaming every set's whole 8 MB sample once per pair
// it takes part in (2 016 pairs of 64 sets: 63 times).
template <typename KeyT>
struct PairSegs {
  const SetPtrs* sets;
  const int32_t* bucket_ids;
  const int32_t* pairs;
  int32_t n_ids;
  int32_t n_pairs;
  __device__ void get(int64_t s, const KeyT*& a, int64_t& a_lo, int64_t& a_hi, const KeyT*& b,
                      int64_t& b_lo, int64_t& b_hi) const {
    const int64_t i = s / n_pairs;
    const int64_t p = s - i * n_pairs;
    const int32_t bucket = bucket_ids[i];
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

// One pair of a batch (ksh_pair_algebra_batch): its two sets, where its three results go, and
// the common-key prefix at its first tile (filled in on the device once the counts are scanned).
struct BatchPair {
  const void* a_keys;
  const int64_t* a_off;
  const void* b_keys;
  const int64_t* b_off;
  int64_t *off_i, *off_amb, *off_bma;
  void *out_i, *out_amb, *out_bma;
  int64_t ioff_base;
};

// One segment per (pair p of the batch, bucket b): s = p * 2^bucket_bits + b.
template <typename KeyT>
struct BatchSegs {
  const BatchPair* pairs;
  int bucket_bits;
  __device__ void get(int64_t s, const KeyT*& a, int64_t& a_lo, int64_t& a_hi, const KeyT*& b,
                      int64_t& b_lo, int64_t& b_hi) const {
    const BatchPair& p = pairs[s >> bucket_bits];
    const int64_t bucket = s & ((int64_t(1) << bucket_bits) - 1);
    a = static_cast<const KeyT*>(p.a_keys);
    b = static_cast<const KeyT*>(p.b_keys);
    a_lo = p.a_off[bucket];
    a_hi = p.a_off[bucket + 1];
    b_lo = p.b_off[bucket];
    b_hi = p.b_off[bucket + 1];
  }
};

// Everything a tile needs, in one 48-byte record (one dependent load in the merge kernel):
// the two key arrays and the tile's index ranges [a0, a1) and [b0, b1) in them.
struct TileDesc {
  const void* a_keys;
  const void* b_keys;
  int64_t a0, b0;
  int64_t a1, b1;
};

// ---- tile plan --------------------------------------------------------------------------
template <typename KeyT, typename Segs>
struct LoadSegTiles {
  Segs segs;
  __device__ __forceinline__ int64_t operator()(int64_t s) const {
    const KeyT *a, *b;
    int64_t a_lo, a_hi, b_lo, b_hi;
    segs.get(s, a, a_lo, a_hi, b, b_lo, b_hi);
    const int64_t len = (a_hi - a_lo) + (b_hi - b_lo);
    return (len + TileCfg<KeyT>::kTile - 1) / TileCfg<KeyT>::kTile;
  }
};

// Which segment a tile belongs to, which of the segment's tiles it is and how many there are.
struct TileOwner {
  int32_t seg, q, n_tiles, pad;
};

// Scan epilogue: segment s with n_tiles tiles starting at tile `first` labels them.
struct EmitTileOwners {
  TileOwner* owner;
  __device__ __forceinline__ void operator()(int64_t s, int64_t first, int64_t n_tiles) const {
    for (int64_t q = 0; q < n_tiles; q++) owner[first + q] = TileOwner{int32_t(s), int32_t(q), int32_t(n_tiles), 0};
  }
};

__global__ __launch_bounds__(256) void k_tile_owners(const int64_t* __restrict__ tile_base, int64_t n_segs,
                                                      TileOwner* __restrict__ owner) {
  const int64_t s = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (s >= n_segs) return;
  EmitTileOwners{owner}(s, tile_base[s], tile_base[s + 1] - tile_base[s]);
}

template <typename KeyT, typename Segs>
__global__ __launch_bounds__(256) void k_seg_tiles(Segs segs, int64_t n_segs,
                                                    int64_t* __restrict__ tiles_per_seg) {
  const int64_t s = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (s >= n_segs) return;
  tiles_per_seg[s] = LoadSegTiles<KeyT, Segs>{segs}(s);
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

// One thread per tile: which segment it belongs to and where it starts in A and in B.  The
// segment's tiles share its keys evenly (each at most TileCfg::kTile): the merge kernel's
// cost follows the keys in a tile, not the tile count.  A tile ends where the next tile of
// its segment starts, so every boundary inside a segment is searched once, by the thread of
// the tile that starts there, which also closes the tile before it; a segment's first tile
// starts at the segment's start and its last tile ends at the segment's end without a search.
template <typename KeyT, typename Segs>
__global__ __launch_bounds__(256) void k_tile_split(Segs segs, int64_t n_segs,
                                                     const int64_t* __restrict__ tile_base,
                                                     const TileOwner* __restrict__ owner,
                                                     TileDesc* __restrict__ desc) {
  const int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  const int64_t total = tile_base[n_segs];
  if (t >= total) return;
  const TileOwner o = owner[t];
  const KeyT *a, *b;
  int64_t a_lo, a_hi, b_lo, b_hi;
  segs.get(o.seg, a, a_lo, a_hi, b, b_lo, b_hi);
  const int64_t na = a_hi - a_lo, nb = b_hi - b_lo;
  int64_t i0 = 0, j0 = 0;
  if (o.q > 0) {
    merge_path_split(a + a_lo, na, b + b_lo, nb, int64_t(o.q) * (na + nb) / o.n_tiles, &i0, &j0);
    desc[t - 1].a1 = a_lo + i0;
    desc[t - 1].b1 = b_lo + j0;
  }
  desc[t].a_keys = a;
  desc[t].b_keys = b;
  desc[t].a0 = a_lo + i0;
  desc[t].b0 = b_lo + j0;
  if (o.q + 1 == o.n_tiles) {
    desc[t].a1 = a_hi;
    desc[t].b1 = b_hi;
  }
}

// A tile's keys travel global -> registers -> LDS as whole 16-byte vectors, A's vectors
// first and B's after them: vector v of the wave is A's v-th vector if v < va, else B's
// (v - va)-th.  The vector holding a range's first key goes to a 16-byte aligned LDS slot, so
// key k of the range lands `mis` slots after it (mis = the range's position inside its
// 16-byte chunk in global memory), global loads are 16 B per lane and LDS writes are
// ds_write_b128.  The first and last vector of a range may carry up to kPer - 1 keys that
// are not part of it; they belong to the same aligned 16-byte chunk of the key array as a
// key that is, so the loads stay inside the array's memory.  Keys live in device memory:
// address space 1 keeps the loads global_load instead of flat_load.
template <typename KeyT>
struct TileStage {
  using Cfg = TileCfg<KeyT>;
  static constexpr int kPer = Cfg::kPer;
  typedef KeyT Vec __attribute__((ext_vector_type(kPer)));
  typedef const Vec __attribute__((address_space(1))) * GlobalVecs;

  int a_lo, a_hi, b_lo, b_hi;  // LDS key indices of the two ranges
  uint32_t va, vb, b_vec0;     // vectors of A, of B; LDS vector index of B's first vector
  GlobalVecs ga, gb;

  int ca, cb;                  // keys of A / B in the tile

  __device__ __forceinline__ void init(const TileDesc& d) {
    ca = int(d.a1 - d.a0);
    cb = int(d.b1 - d.b0);
    const uintptr_t pa = reinterpret_cast<uintptr_t>(d.a_keys) + uintptr_t(d.a0) * sizeof(KeyT);
    const uintptr_t pb = reinterpret_cast<uintptr_t>(d.b_keys) + uintptr_t(d.b0) * sizeof(KeyT);
    const int mis_a = int((pa / sizeof(KeyT)) & (kPer - 1)), mis_b = int((pb / sizeof(KeyT)) & (kPer - 1));
    ga = (GlobalVecs)(pa & ~uintptr_t(15));
    gb = (GlobalVecs)(pb & ~uintptr_t(15));
    va = ca > 0 ? uint32_t(mis_a + ca + kPer - 1) / kPer : 0u;
    vb = cb > 0 ? uint32_t(mis_b + cb + kPer - 1) / kPer : 0u;
    a_lo = mis_a;
    a_hi = a_lo + ca;
    b_vec0 = uint32_t(a_hi + kPer - 1) / kPer;
    b_lo = int(b_vec0) * kPer + mis_b;
    b_hi = b_lo + cb;
  }
  // global -> LDS: every load of both ranges is issued before the first one is waited for
  // (a loop that loads and stores one vector per trip pays one memory round trip per trip).
  // The two ranges are one sequence of va + vb vectors dealt out 64 at a time, so a lane asks
  // for kVecs vectors in all, not kVecs per side.
  __device__ __forceinline__ void copy(KeyT* lds) const {
    Vec* l = reinterpret_cast<Vec*>(lds);
    const uint32_t lane = threadIdx.x & (kThreads - 1);
    Vec r[Cfg::kVecs];
    // Straight-line, unconditional loads: lanes past the end re-read the last vector (one
    // coalesced request; a tile is never empty on both sides); only the stores are predicated.
    // Conditional loads would make the compiler wait at every join.
    const uint32_t n_vecs = va + vb;
#pragma unroll
    for (int j = 0; j < Cfg::kVecs; j++) {
      const uint32_t v = min(lane + j * kThreads, n_vecs - 1);
      r[j] = *(v < va ? ga + v : gb + (v - va));
    }
#pragma unroll
    for (int j = 0; j < Cfg::kVecs; j++) {
      const uint32_t v = lane + j * kThreads;
      if (uint32_t(j * kThreads) < n_vecs && v < n_vecs) l[v < va ? v : b_vec0 + (v - va)] = r[j];
    }
  }
};

// Copies cnt keys from LDS (starting at key index lds_first) to out[0, cnt), one wave,
// coalesced.  Offsets are 32-bit byte offsets from a wave-uniform base, which lets the
// stores use the scalar-base addressing form.
template <typename KeyT>
__device__ __forceinline__ void copy_out(const KeyT* __restrict__ lds, uint32_t lds_first,
                                         uint32_t cnt, KeyT* out) {
  typedef char __attribute__((address_space(1))) * GlobalBytes;
  typedef KeyT __attribute__((address_space(1))) * GlobalKeys;
  GlobalBytes o = (GlobalBytes)out;
  const char* l = reinterpret_cast<const char*>(lds + lds_first);
  const uint32_t end = cnt * uint32_t(sizeof(KeyT));
#pragma unroll 1
  for (uint32_t x = (threadIdx.x & (kThreads - 1)) * uint32_t(sizeof(KeyT)); x < end; x += kThreads * uint32_t(sizeof(KeyT)))
    *(GlobalKeys)(o + x) = *reinterpret_cast<const KeyT*>(l + x);
}

// Inclusive prefix sum across the wave's 64 lanes on the DPP cross-lane path (no LDS):
// four shifts inside each row of 16 lanes, then the last lane of row 0 / 2 is added to the
// row after it and the last lane of row 1 to rows 2 and 3.
__device__ __forceinline__ uint32_t wave_inclusive_scan_u32(uint32_t v) {
  constexpr int kRowShr1 = 0x111, kRowShr2 = 0x112, kRowShr4 = 0x114, kRowShr8 = 0x118;
  constexpr int kRowBcast15 = 0x142, kRowBcast31 = 0x143;
  v += uint32_t(__builtin_amdgcn_update_dpp(0, int(v), kRowShr1, 0xf, 0xf, false));
  v += uint32_t(__builtin_amdgcn_update_dpp(0, int(v), kRowShr2, 0xf, 0xf, false));
  v += uint32_t(__builtin_amdgcn_update_dpp(0, int(v), kRowShr4, 0xf, 0xf, false));
  v += uint32_t(__builtin_amdgcn_update_dpp(0, int(v), kRowShr8, 0xf, 0xf, false));
  v += uint32_t(__builtin_amdgcn_update_dpp(0, int(v), kRowBcast15, 0xa, 0xf, false));
  v += uint32_t(__builtin_amdgcn_update_dpp(0, int(v), kRowBcast31, 0xc, 0xf, false));
  return v;
}

// Orders the LDS accesses of one wave: a wave's LDS instructions execute in issue order, so a
// read placed after this sees every write placed before it, whichever lane made it.
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Build with -DKSH_TRACE to record s_memtime at the phases of every tile (debugging aid for
// the latency breakdown in DESIGN.md; compiled out otherwise).
#ifdef KSH_TRACE
__device__ unsigned long long* g_tile_trace = nullptr;
__device__ int g_tile_stop_after = 0;  // 1: return once the tile's keys are in LDS
#define KSH_MARK(k, tile)                                                                      \
  do {                                                                                   \
    if (g_tile_trace && (threadIdx.x & 63) == 0)                                                \
      g_tile_trace[(int64_t(kMode != 0) * max_tiles + (tile)) * 8 + (k)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define KSH_MARK(k, tile) do { } while (0)
#endif

template <typename KeyT, int kMode>
__global__ __launch_bounds__(kThreads * kGroupWaves) void k_tile_merge(
    const TileDesc* __restrict__ desc, const int64_t* __restrict__ total_tiles, int64_t max_tiles,
    int64_t* __restrict__ tile_m, const int64_t* __restrict__ tile_ioff,
    uint16_t* __restrict__ split, KeyT* __restrict__ out_i, KeyT* __restrict__ out_amb,
    KeyT* __restrict__ out_bma, const BatchPair* __restrict__ batch, const TileOwner* __restrict__ owner,
    int batch_bucket_bits) {
  using Cfg = TileCfg<KeyT>;
  using Stage = TileStage<KeyT>;
  constexpr int kVT = Cfg::kVT;
  constexpr bool kWrite = kMode != 0;
  __shared__ __attribute__((aligned(16))) KeyT lds_all[kGroupWaves][Cfg::kLds];

  // This wave owns one tile and one slice of the workgroup's LDS.  Nothing below synchronises
  // across waves: wave_sync() orders this wave's own LDS traffic.  (Several consecutive tiles
  // per wave, with or without the next tile's keys prefetched into registers, measured slower at
  // every count: DESIGN.md 3.1.)
  const int lane = threadIdx.x & (kThreads - 1);
  const int wave = kGroupWaves == 1 ? 0 : __builtin_amdgcn_readfirstlane(threadIdx.x / kThreads);  // uniform: stays scalar
  KeyT* const lds = lds_all[wave];
  const int64_t t = int64_t(blockIdx.x) * kGroupWaves + wave;

  // Everything a tile needs besides its keys: the descriptor, and for the write pass the lane's
  // saved split, the tile's prefix of common keys and (batch launches) its pair's outputs.  All
  // of it is requested in one go, before anything waits: a wave's life is mostly memory round
  // trips, and these used to be five of them in a row (tile count, descriptor, keys, split,
  // prefix / owner / pair).
  struct TileMeta {
    int64_t ioff;
    TileOwner own;
  };
  auto load_meta = [&](int64_t t) {
    TileMeta m{};
    if constexpr (kWrite) {
      m.ioff = tile_ioff[t];
      // (unconditional, so that it travels with the others: without a batch the value is unused
      // and the descriptor array stands in as readable memory)
      m.own = *(batch ? owner + t : reinterpret_cast<const TileOwner*>(desc + t));
    }
    return m;
  };
  // The requests go out before the tile count is known; tile max_tiles - 1 is allocated
  // whatever the count turns out to be (launches have max_tiles >= 1).
  const int64_t t_safe = min(t, max_tiles - 1);
  const TileDesc d = desc[t_safe];
  const TileMeta meta = load_meta(t_safe);
  int64_t n_tiles = *total_tiles;
  // The scalar requests above have to be out before the early exit below needs the tile count
  // (left alone, the compiler sinks them past that branch, one more round trip each).  The
  // statement ties them to the count by data flow only: it is not volatile, because a volatile
  // asm counts as a possible store and every load after it would stop being a scalar load.
  asm("" : "+s"(n_tiles) : "s"(d.a_keys), "s"(d.b_keys), "s"(d.a0), "s"(d.a1), "s"(d.b0), "s"(d.b1),
      "s"(meta.ioff), "s"(meta.own.seg));
  if (t >= n_tiles) {
    // the prefix scan over tile_m runs to max_tiles: tiles that do not exist count zero
    if (!kWrite && lane == 0 && t < max_tiles) tile_m[t] = 0;
    return;
  }

  KSH_MARK(0, t);
  // Everything after a tile's keys are in LDS (and a barrier has passed).
  auto process = [&](const TileDesc& d, const Stage& st, int64_t t, const TileMeta& meta,
                     uint32_t saved_split) {
    KSH_MARK(2, t);
#ifdef KSH_TRACE
    if (g_tile_stop_after == 1) {
      if (!kWrite && lane == 0) tile_m[t] = lds[st.a_lo] == lds[st.b_lo];
      wave_sync();
      return;
    }
#endif
    const int ca = st.ca, cb = st.cb;
    const int a_lo = st.a_lo, a_hi = st.a_hi, b_lo = st.b_lo, b_hi = st.b_hi;

    const int n = ca + cb;
    // keys per lane for this tile: enough for n keys, odd (bank spread), at most kVT
    const int n_steps = min(kVT, ((n + kThreads - 1) / kThreads) | 1);
    const int d0 = min(lane * n_steps, n);
    // Merge-path split of this lane's diagonal (A first on ties), in LDS indices: a[m] is
    // lds[a_lo + m] and b[d0 - 1 - m] is lds[s0 - 1 - (a_lo + m)].  The count pass searches
    // and, when a write pass follows, leaves every lane's split in `split`; the write pass
    // reads it back.
    const int s0 = a_lo + b_lo + d0;  // i + (LDS index of B's head) at step 0
    int i;                            // LDS index of A's head
    if constexpr (kWrite) {
      i = a_lo + int(saved_split);
    } else {
      // The answer is the number of m in [m_lo, m_hi) with a[m] <= b[d0 - 1 - m] (true for a
      // prefix).  Bit-by-bit descent with a wave-uniform trip count: `pos` is the last index
      // known to satisfy it; a candidate beyond the lane's own range fails on the range
      // test.  Ranges are at most min(ca, cb) long, so a candidate is never more than that
      // past the range: the A-side read stays below B's end and the B-side read at or above
      // A's start.
      const int m_hi = a_lo + min(d0, ca);
      int pos = a_lo + max(0, d0 - cb) - 1;
      const int range = min(ca, cb);
      for (int stride = range > 0 ? (1 << (31 - __builtin_clz(range))) : 0; stride > 0; stride >>= 1) {
        const int cand = pos + stride;
        const bool ok = (cand < m_hi) & (lds[cand] <= lds[s0 - 1 - cand]);
        pos = ok ? cand : pos;
      }
      i = pos + 1;
      if (split) split[t * kThreads + lane] = uint16_t(i - a_lo);
    }
    KSH_MARK(3, t);
    // the key merged just before this lane's first one is a common A key (its B half is ours)
    const bool straddle = i > a_lo && s0 - i < b_hi && lds[i - 1] == lds[s0 - i];
    bool prev_common = straddle;

    KeyT vals[kVT];
    uint32_t m_a = 0, m_x = 0;  // bit (kVT - 1 - step) belongs to `step`: from A / the other class of its side
    uint32_t m_c = 0;           // the same as two-bit class codes, for the compaction
    int n_i = 0, n_a = 0;
    // The heads as byte offsets into the wave's LDS slice: an LDS instruction takes a byte address
    // plus an immediate, so the A head needs no arithmetic and the B head one subtraction.
    constexpr uint32_t kKey = uint32_t(sizeof(KeyT));
    const char* const lds_b = reinterpret_cast<const char*>(lds);
    uint32_t ia = uint32_t(i) * kKey;
    const uint32_t sb = uint32_t(s0) * kKey, a_end = uint32_t(a_hi) * kKey, b_end = uint32_t(b_hi) * kKey;
#pragma unroll
    for (int step = 0; step < kVT; step++) {
      if (step < n_steps) {  // wave-uniform
        const uint32_t jb0 = sb - ia;  // the B head is at jb0 + step keys: the step goes into the immediate
        const KeyT av = *reinterpret_cast<const KeyT*>(lds_b + ia);
        const KeyT bv = *reinterpret_cast<const KeyT*>(lds_b + jb0 + uint32_t(step) * kKey);
        // (signed: a tiny tile's B end can lie below step keys; every offset is far below 2^31)
        const bool has_a = ia < a_end, has_b = int(jb0) < int(b_end) - step * int(kKey);
        const bool take_a = has_a & (!has_b | (av <= bv));
        const bool common = take_a & has_b & (av == bv);
        const uint32_t took = uint32_t(take_a);
        if constexpr (kWrite) {
          const bool keep_b = !take_a & has_b & !prev_common;
          const bool x = (take_a & !common) | (!take_a & !keep_b);
          vals[step] = take_a ? av : bv;
          m_a = m_a + m_a + took;
          m_x = m_x + m_x + uint32_t(x);
          prev_common = common;
        } else {
          n_i += int(common);
        }
        ia += took * kKey;
      }
    }
    // (batch launches) where this tile's pair wants its results: asked for here, behind the
    // merge, so that nothing waits for it -- ahead of the key loads it held them back by one
    // more scalar round trip
    BatchPair bp{};
    if constexpr (kWrite) {
      if (batch) bp = batch[meta.own.seg >> batch_bucket_bits];
    }
    if constexpr (kWrite) {
      m_a <<= kVT - n_steps;  // step's bit is bit (kVT - 1 - step) whatever n_steps is
      m_x = (m_x << (kVT - n_steps)) | ((1u << (kVT - n_steps)) - 1);  // steps not run: dropped
      n_i = __popc(m_a & ~m_x);
      n_a = __popc(m_a & m_x);
      // class code of a step = (not from A) << 1 | x: 0 = A&B, 1 = A\B, 2 = B\A, 3 = dropped; its
      // two bits sit at 2 (kVT - 1 - step)
      auto spread = [](uint32_t v) {  // bit b -> bit 2 b (16 bits in)
        v = (v | (v << 8)) & 0x00FF00FFu;
        v = (v | (v << 4)) & 0x0F0F0F0Fu;
        v = (v | (v << 2)) & 0x33333333u;
        return (v | (v << 1)) & 0x55555555u;
      };
      m_c = (spread(~m_a & ((1u << kVT) - 1)) << 1) | spread(m_x);
    }
    KSH_MARK(4, t);

    // Wave scan of the A&B and A\B counts, packed into one word.  The B\A prefix follows
    // from them: the lanes before this one merged d0 keys, each common pair among those is
    // one A&B key plus one dropped B key, except that the pair straddling into this lane
    // has not dropped its B key yet.
    static_assert(Cfg::kCap < 65536, "packed 16-bit counters");
    const uint32_t packed = uint32_t(n_i) | (uint32_t(n_a) << 16);
    const uint32_t inc = wave_inclusive_scan_u32(packed);
    const uint32_t tot = __builtin_amdgcn_readlane(inc, 63);
    const uint32_t excl = inc - packed;
    const uint32_t tot_i = tot & 0xFFFF, tot_a = tot >> 16, tot_b = uint32_t(n) - 2 * tot_i - tot_a;

    if constexpr (!kWrite) {
      if (lane == 0) tile_m[t] = tot_i;
      KSH_MARK(5, t);
    } else {
      KSH_MARK(5, t);
      const uint32_t excl_i = excl & 0xFFFF, excl_a = excl >> 16;
      const uint32_t excl_b = uint32_t(d0) - 2 * excl_i - excl_a + uint32_t(straddle);
      wave_sync();  // every lane is done reading the inputs: reuse the LDS for compaction
      int64_t ioff = meta.ioff;
      if (batch) {  // a batch launch: this tile's pair says where its results go
        out_i = static_cast<KeyT*>(bp.out_i);
        out_amb = static_cast<KeyT*>(bp.out_amb);
        out_bma = static_cast<KeyT*>(bp.out_bma);
        ioff -= bp.ioff_base;
      }
      // A&B keys go to [0, tot_i), A\B keys to [tot_i, tot_i + tot_a), B\A keys after them
      // (union: every kept key, in merged order).  While a tile holds fewer than 1024 keys the
      // three write positions travel in one register, 10 bits each.
      if constexpr (kMode == 2) {
        uint32_t p_u = excl_i + excl_a + excl_b;
#pragma unroll
        for (int step = 0; step < kVT; step++) {
          if (step < n_steps) {
            const bool keep = ((m_c >> (2 * (kVT - 1 - step))) & 3) != 3;
            if (keep) lds[p_u] = vals[step];
            p_u += uint32_t(keep);
          }
        }
      } else if constexpr (Cfg::kCap < 1024) {
        uint32_t pos3 = excl_i + ((tot_i + excl_a) << 10) + ((tot_i + tot_a + excl_b) << 20);
#pragma unroll
        for (int step = 0; step < kVT; step++) {
          if (step < n_steps) {
            const uint32_t c = (m_c >> (2 * (kVT - 1 - step))) & 3;
            const bool keep = c != 3;
            const uint32_t sh = c * 10;  // the class's field of pos3; a dropped key counts in the spare
            if (keep) lds[(pos3 >> sh) & 1023] = vals[step];  // bits 30-31, which nothing reads
            pos3 += 1u << sh;
          }
        }
      } else {
        uint32_t p_i = excl_i, p_a = tot_i + excl_a, p_b = tot_i + tot_a + excl_b;
#pragma unroll
        for (int step = 0; step < kVT; step++) {
          if (step < n_steps) {
            const uint32_t c = (m_c >> (2 * (kVT - 1 - step))) & 3;
            const uint32_t pos = c == 0 ? p_i : (c == 1 ? p_a : p_b);
            if (c != 3) lds[pos] = vals[step];
            p_i += uint32_t(c == 0);
            p_a += uint32_t(c == 1);
            p_b += uint32_t(c == 2);
          }
        }
      }
      wave_sync();
      KSH_MARK(6, t);
      if (kMode == 2) {
        copy_out(lds, 0, tot_i + tot_a + tot_b, out_i + (d.a0 + d.b0 - ioff));
      } else {
        if (out_i) copy_out(lds, 0, tot_i, out_i + ioff);
        if (out_amb) copy_out(lds, tot_i, tot_a, out_amb + (d.a0 - ioff));
        if (out_bma) copy_out(lds, tot_i + tot_a, tot_b, out_bma + (d.b0 - ioff));
      }
      KSH_MARK(7, t);
    }
  };

  uint32_t saved_split = 0;
  if constexpr (kWrite) saved_split = split[t * kThreads + lane];  // ahead of the keys, not after them
  Stage st;
  st.init(d);
  KSH_MARK(1, t);
  st.copy(lds);
  wave_sync();
  process(d, st, t, meta, saved_split);
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

// weights[p] = common keys over the segments i * n_pairs + p, i = 0 .. n_ids - 1 (one wave per pair: the lanes
// take the sampled buckets in turn, a segment's count is the difference of the common-key prefixes at its
// first tile and at the next segment's).
__global__ __launch_bounds__(256) void k_pair_weight_gather(
    const int64_t* __restrict__ tile_base, const int64_t* __restrict__ tile_ioff,
    const int64_t* __restrict__ total_m, int64_t n_segs, int32_t n_ids, int32_t n_pairs,
    int64_t* __restrict__ weights) {
  const int32_t p = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (p >= n_pairs) return;
  const int64_t total_tiles = tile_base[n_segs];
  int64_t sum = 0;
  for (int32_t i = lane; i < n_ids; i += 64) {
    const int64_t s = int64_t(i) * n_pairs + p;
    const int64_t t0 = tile_base[s], t1 = tile_base[s + 1];
    if (t1 == t0) continue;  // (an empty segment has no tile)
    const int64_t m0 = t0 < total_tiles ? tile_ioff[t0] : *total_m;
    const int64_t m1 = t1 < total_tiles ? tile_ioff[t1] : *total_m;
    sum += m1 - m0;
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) sum += __shfl_xor(sum, d, 64);
  if (lane == 0) weights[p] = sum;
}

// The same for every pair of a batch: pair p owns segments [p * nb, (p + 1) * nb]; its prefixes
// are taken relative to the prefix at its first tile, which is also left in pairs[p].ioff_base
// for the write pass.  totals3[3 p ..] = |A&B|, |A\\B|, |B\\A|.
__global__ __launch_bounds__(256) void k_batch_offsets(BatchPair* __restrict__ pairs, int32_t n_pairs,
                                                        const int64_t* __restrict__ tile_base,
                                                        const int64_t* __restrict__ tile_ioff,
                                                        const int64_t* __restrict__ total_m, int64_t n_buckets,
                                                        int64_t n_segs, int64_t* __restrict__ totals3) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  const int64_t per = n_buckets + 1;
  if (i >= per * n_pairs) return;
  const int32_t p = int32_t(i / per);
  const int64_t s = i - int64_t(p) * per;
  const int64_t total_tiles = tile_base[n_segs];
  const int64_t first_of_pair = tile_base[int64_t(p) * n_buckets];
  const int64_t first = tile_base[int64_t(p) * n_buckets + s];
  const int64_t base = first_of_pair < total_tiles ? tile_ioff[first_of_pair] : *total_m;
  const int64_t m = (first < total_tiles ? tile_ioff[first] : *total_m) - base;
  const BatchPair bp = pairs[p];
  bp.off_i[s] = m;
  bp.off_amb[s] = bp.a_off[s] - m;
  bp.off_bma[s] = bp.b_off[s] - m;
  if (s == 0) pairs[p].ioff_base = base;
  if (s == n_buckets) {
    totals3[3 * p + 0] = m;
    totals3[3 * p + 1] = bp.a_off[s] - m;
    totals3[3 * p + 2] = bp.b_off[s] - m;
  }
}

// ---- host-side plan ------------------------------------------------------------------------------
constexpr int kSplitPerTile = kThreads;

struct Plan {
  int64_t n_segs = 0;
  int64_t max_tiles = 0;
  int64_t* tile_base = nullptr;  // n_segs + 1
  TileDesc* desc = nullptr;      // max_tiles
  int64_t* tile_ioff = nullptr;  // max_tiles (count, then exclusive prefix in place)
  int64_t* total_m = nullptr;    // 1
  TileOwner* owner = nullptr;    // max_tiles
  uint16_t* split = nullptr;     // max_tiles * kSplitPerTile: every chain's merge-path split (count pass -> write pass)
};

inline size_t align256(size_t x) { return (x + 255) & ~size_t(255); }

inline size_t plan_bytes(int64_t n_segs, int64_t max_tiles) {
  return align256(size_t(n_segs + 1) * 8) + align256(size_t(max_tiles) * sizeof(TileDesc)) +
         align256(size_t(max_tiles) * 8) + 256 + align256(size_t(max_tiles) * kSplitPerTile * 2) +
         align256(size_t(max_tiles) * sizeof(TileOwner));
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
  at += 256;
  p->split = reinterpret_cast<uint16_t*>(at);
  at += align256(size_t(max_tiles) * kSplitPerTile * 2);
  p->owner = reinterpret_cast<TileOwner*>(at);
}

template <typename KeyT, int kMode>
void launch_tile_merge(ksh_ctx* ctx, const Plan& p, int64_t* tile_m, const int64_t* tile_ioff,
                       uint16_t* split, KeyT* out_i, KeyT* out_amb, KeyT* out_bma,
                       const BatchPair* batch = nullptr, int batch_bucket_bits = 0) {
  const int64_t groups = (p.max_tiles + kGroupWaves - 1) / kGroupWaves;
  hipLaunchKernelGGL((k_tile_merge<KeyT, kMode>), dim3(unsigned(groups)), dim3(kThreads * kGroupWaves), 0, ctx->stream,
                     p.desc, p.tile_base + p.n_segs, p.max_tiles, tile_m, tile_ioff, split, out_i,
                     out_amb, out_bma, batch, p.owner, batch_bucket_bits);
}

inline unsigned blocks_for(int64_t n, int per) { return unsigned(std::max<int64_t>(1, (n + per - 1) / per)); }

// Tiles per segment -> tile_base (exclusive prefix, total at [n_segs]).
// owner == nullptr: the caller does not know the tile count yet and labels the tiles later
// (label_tiles).
template <typename KeyT, typename Segs>
int plan_tile_base(ksh_ctx* ctx, const Segs& segs, int64_t n_segs, int64_t* tile_base, TileOwner* owner) {
  // one launch when the chained scan takes it: the per-segment tile counts are computed
  // while they are scanned, and every segment labels its tiles on the way out
  const LoadSegTiles<KeyT, Segs> load{segs};
  const bool chained = owner ? scan_exclusive_chained(ctx, load, tile_base, n_segs, tile_base + n_segs,
                                                      EmitTileOwners{owner})
                             : scan_exclusive_chained(ctx, load, tile_base, n_segs, tile_base + n_segs);
  if (!chained) {
    hipLaunchKernelGGL((k_seg_tiles<KeyT, Segs>), dim3(blocks_for(n_segs, 256)), dim3(256), 0,
                       ctx->stream, segs, n_segs, tile_base);
    KSH_TRY(scan_exclusive_i64(ctx, tile_base, tile_base, n_segs, tile_base + n_segs));
    if (owner)
      hipLaunchKernelGGL(k_tile_owners, dim3(blocks_for(n_segs, 256)), dim3(256), 0, ctx->stream, tile_base,
                         n_segs, owner);
  }
  KSH_HIP(hipGetLastError());
  return KSH_OK;
}

inline int label_tiles(ksh_ctx* ctx, const int64_t* tile_base, int64_t n_segs, TileOwner* owner) {
  hipLaunchKernelGGL(k_tile_owners, dim3(blocks_for(n_segs, 256)), dim3(256), 0, ctx->stream, tile_base,
                     n_segs, owner);
  KSH_HIP(hipGetLastError());
  return KSH_OK;
}

// Splits + count pass + prefix of the per-tile counts.
template <typename KeyT, typename Segs>
int plan_count(ksh_ctx* ctx, const Segs& segs, const Plan& p, int timer_kind, bool save_split) {
  hipLaunchKernelGGL((k_tile_split<KeyT, Segs>), dim3(blocks_for(p.max_tiles, 256)), dim3(256), 0,
                     ctx->stream, segs, p.n_segs, p.tile_base, p.owner, p.desc);
  {
    Timer timer(ctx, timer_kind);
    launch_tile_merge<KeyT, 0>(ctx, p, p.tile_ioff, nullptr, save_split ? p.split : nullptr,
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
  const int64_t max_tiles = nb + (a->n_keys + b->n_keys) / TileCfg<KeyT>::kTile + 1;
  if (max_tiles > int64_t(0x7FFFFFF0)) return fail(KSH_INVALID_ARGUMENT, "pair too large for one launch");
  KSH_TRY(plan_reserve(ctx, plan_bytes(nb, max_tiles)));
  KSH_TRY(arena_reserve(ctx, size_t(max_tiles / 256 + 4096) * 8 + (1u << 16)));
  arena_reset(ctx);
  Plan p;
  plan_carve(ctx->plan, nb, max_tiles, &p);
  BucketSegs<KeyT> segs{static_cast<const KeyT*>(a->d_keys), a->d_offsets,
                        static_cast<const KeyT*>(b->d_keys), b->d_offsets};
  KSH_TRY((plan_tile_base<KeyT>(ctx, segs, nb, p.tile_base, p.owner)));
  KSH_TRY((plan_count<KeyT>(ctx, segs, p, 1, true)));
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
    launch_tile_merge<KeyT, 1>(ctx, p, nullptr, p.tile_ioff, p.split, static_cast<KeyT*>(d_keys_i),
                               static_cast<KeyT*>(d_keys_amb), static_cast<KeyT*>(d_keys_bma));
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
  const int64_t max_tiles = nb + (a->n_keys + b->n_keys) / TileCfg<KeyT>::kTile + 1;
  arena_reset(ctx);
  char* base = static_cast<char*>(arena_alloc(ctx, plan_bytes(nb, max_tiles)));
  Plan p;
  plan_carve(base, nb, max_tiles, &p);
  BucketSegs<KeyT> segs{static_cast<const KeyT*>(a->d_keys), a->d_offsets,
                        static_cast<const KeyT*>(b->d_keys), b->d_offsets};
  KSH_TRY((plan_tile_base<KeyT>(ctx, segs, nb, p.tile_base, p.owner)));
  KSH_TRY((plan_count<KeyT>(ctx, segs, p, 1, true)));
  hipLaunchKernelGGL(k_result_offsets, dim3(blocks_for(nb + 1, 256)), dim3(256), 0, ctx->stream,
                     a->d_offsets, b->d_offsets, p.tile_base, p.tile_ioff, p.total_m, nb, d_off_i,
                     d_off_amb, d_off_bma, d_totals);
  {
    Timer timer(ctx, 0);
    launch_tile_merge<KeyT, 1>(ctx, p, nullptr, p.tile_ioff, p.split, static_cast<KeyT*>(d_keys_i),
                               static_cast<KeyT*>(d_keys_amb), static_cast<KeyT*>(d_keys_bma));
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
  // One plan, one count launch and one write launch for the whole batch: the segments of all
  // pairs are tiled together, so the launches are n_jobs times longer and their ramp-up, tail
  // and launch gaps are paid once.
  int64_t max_tiles = 0;
  for (int32_t i = 0; i < n_jobs; i++)
    max_tiles += nb + (jobs[i].a.n_keys + jobs[i].b.n_keys) / TileCfg<KeyT>::kTile + 1;
  const int64_t n_segs = int64_t(n_jobs) * nb;
  if (max_tiles > int64_t(0x7FFFFFF0) || n_segs > int64_t(0x7FFFFFF0))
    return fail(KSH_INVALID_ARGUMENT, "batch too large for one launch");
  const size_t pairs_bytes = align256(size_t(n_jobs) * sizeof(BatchPair));
  KSH_TRY(arena_reserve(ctx, pairs_bytes + align256(size_t(n_jobs) * 24) + plan_bytes(n_segs, max_tiles) +
                                 size_t(std::max(max_tiles, n_segs) / 256 + 4096) * 8 + (1u << 16)));
  arena_reset(ctx);
  BatchPair* d_pairs = static_cast<BatchPair*>(arena_alloc(ctx, size_t(n_jobs) * sizeof(BatchPair)));
  int64_t* d_totals = static_cast<int64_t*>(arena_alloc(ctx, size_t(n_jobs) * 24));
  char* base = static_cast<char*>(arena_alloc(ctx, plan_bytes(n_segs, max_tiles)));
  if (!d_pairs || !d_totals || !base) return fail(KSH_INTERNAL, "scratch arena too small");
  // the pair records and, later, the totals travel through one pinned buffer (a pageable source
  // would be staged by the runtime first); the stream is synchronised before this call returns,
  // so the buffer is free again by then
  const size_t n_vals = static_cast<size_t>(n_jobs) * 3;
  static_assert(sizeof(BatchPair) % 8 == 0, "pair records are whole int64 words");
  const size_t pair_words = size_t(n_jobs) * sizeof(BatchPair) / 8;
  const size_t pinned_words = pair_words + n_vals;  // records, then totals
  if (ctx->h_batch_count < pinned_words) {
    if (ctx->h_batch) (void)hipHostFree(ctx->h_batch);
    ctx->h_batch = nullptr;
    ctx->h_batch_count = 0;
    if (hipHostMalloc(reinterpret_cast<void**>(&ctx->h_batch), pinned_words * 2 * sizeof(int64_t)) != hipSuccess)
      return fail(KSH_INTERNAL, "hipHostMalloc failed");
    ctx->h_batch_count = pinned_words * 2;
  }
  BatchPair* h_pairs = reinterpret_cast<BatchPair*>(ctx->h_batch);
  for (int32_t i = 0; i < n_jobs; i++) {
    const ksh_pair_job& j = jobs[i];
    h_pairs[i] = BatchPair{j.a.d_keys, j.a.d_offsets, j.b.d_keys, j.b.d_offsets, j.d_off_i, j.d_off_amb,
                           j.d_off_bma, j.d_keys_i, j.d_keys_amb, j.d_keys_bma, 0};
  }
  KSH_HIP(hipMemcpyAsync(d_pairs, h_pairs, size_t(n_jobs) * sizeof(BatchPair), hipMemcpyHostToDevice,
                         ctx->stream));
  Plan p;
  plan_carve(base, n_segs, max_tiles, &p);
  const BatchSegs<KeyT> segs{d_pairs, g->n_bucket_bits};
  KSH_TRY((plan_tile_base<KeyT>(ctx, segs, n_segs, p.tile_base, p.owner)));
  KSH_TRY((plan_count<KeyT>(ctx, segs, p, 1, true)));
  hipLaunchKernelGGL(k_batch_offsets, dim3(blocks_for((nb + 1) * n_jobs, 256)), dim3(256), 0, ctx->stream,
                     d_pairs, n_jobs, p.tile_base, p.tile_ioff, p.total_m, nb, n_segs, d_totals);
  {
    Timer timer(ctx, 0);
    launch_tile_merge<KeyT, 1>(ctx, p, nullptr, p.tile_ioff, p.split, static_cast<KeyT*>(nullptr),
                               static_cast<KeyT*>(nullptr), static_cast<KeyT*>(nullptr), d_pairs,
                               g->n_bucket_bits);
  }
  KSH_HIP(hipGetLastError());
  // the totals land behind the pair records (still being read by the copy above until the
  // stream gets there: stream order keeps the two apart)
  int64_t* h = ctx->h_batch + pair_words;
  KSH_HIP(hipMemcpyAsync(h, d_totals, n_vals * sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
  KSH_HIP(hipStreamSynchronize(ctx->stream));
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
  const int64_t max_tiles = nb + (a->n_keys + b->n_keys) / TileCfg<KeyT>::kTile + 1;
  if (max_tiles > int64_t(0x7FFFFFF0)) return fail(KSH_INVALID_ARGUMENT, "pair too large for one launch");
  KSH_TRY(plan_reserve(ctx, plan_bytes(nb, max_tiles)));
  KSH_TRY(arena_reserve(ctx, size_t(max_tiles / 256 + 4096) * 8 + (1u << 16)));
  arena_reset(ctx);
  Plan p;
  plan_carve(ctx->plan, nb, max_tiles, &p);
  BucketSegs<KeyT> segs{static_cast<const KeyT*>(a->d_keys), a->d_offsets,
                        static_cast<const KeyT*>(b->d_keys), b->d_offsets};
  KSH_TRY((plan_tile_base<KeyT>(ctx, segs, nb, p.tile_base, p.owner)));
  KSH_TRY((plan_count<KeyT>(ctx, segs, p, 1, true)));
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
  launch_tile_merge<KeyT, 2>(ctx, p, nullptr, p.tile_ioff, p.split, static_cast<KeyT*>(d_keys_u),
                             static_cast<KeyT*>(nullptr), static_cast<KeyT*>(nullptr));
  KSH_HIP(hipGetLastError());
  return KSH_OK;
}

template <typename KeyT>
int set_diff_t(ksh_ctx* ctx, const ksh_geom* g, const ksh_set_view* a, const ksh_set_view* b,
               int64_t* diff) {
  const int64_t nb = n_buckets(g);
  const int64_t max_tiles = nb + (a->n_keys + b->n_keys) / TileCfg<KeyT>::kTile + 1;
  if (max_tiles > int64_t(0x7FFFFFF0)) return fail(KSH_INVALID_ARGUMENT, "pair too large for one launch");
  KSH_TRY(arena_reserve(ctx, plan_bytes(nb, max_tiles) + size_t(max_tiles / 256 + 4096) * 8 + (1u << 16)));
  arena_reset(ctx);
  char* base = static_cast<char*>(arena_alloc(ctx, plan_bytes(nb, max_tiles)));
  Plan p;
  plan_carve(base, nb, max_tiles, &p);
  BucketSegs<KeyT> segs{static_cast<const KeyT*>(a->d_keys), a->d_offsets,
                        static_cast<const KeyT*>(b->d_keys), b->d_offsets};
  KSH_TRY((plan_tile_base<KeyT>(ctx, segs, nb, p.tile_base, p.owner)));
  KSH_TRY((plan_count<KeyT>(ctx, segs, p, 1, false)));
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
  PairSegs<KeyT> segs{d_sets, d_ids, d_pairs, n_ids, n_pairs};
  KSH_TRY((plan_tile_base<KeyT>(ctx, segs, n_segs, tile_base, nullptr)));
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
  KSH_TRY(label_tiles(ctx, tile_base, n_segs, p.owner));
  KSH_TRY((plan_count<KeyT>(ctx, segs, p, 2, false)));
  hipLaunchKernelGGL(k_pair_weight_gather, dim3(blocks_for(int64_t(n_pairs) * 64, 256)), dim3(256), 0,
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

#ifdef KSH_TRACE
extern "C" int ksh_debug_set_tile_trace(void* d_buf) {
  unsigned long long* p = static_cast<unsigned long long*>(d_buf);
  return hipMemcpyToSymbol(HIP_SYMBOL(ksh::g_tile_trace), &p, sizeof(p)) == hipSuccess ? 0 : 13;
}
extern "C" int ksh_debug_set_tile_stop(int stop_after) {
  return hipMemcpyToSymbol(HIP_SYMBOL(ksh::g_tile_stop_after), &stop_after, sizeof(int)) == hipSuccess ? 0 : 13;
}
#endif

extern "C" {

int ksh_pair_plan(ksh_ctx* ctx, const ksh_geom* g, const ksh_set_view* a, const ksh_set_view* b,
                  int64_t* d_off_i, int64_t* d_off_amb, int64_t* d_off_bma, int64_t totals[3]) {
  if (!ctx || !d_off_i || !d_off_amb || !d_off_bma || !totals)
    return fail(KSH_INVALID_ARGUMENT, "NULL argument");
  KSH_TRY(check_geom(g));
  KSH_TRY(check_view(a, "a"));
  KSH_TRY(check_view(b, "b"));
  KSH_HIP(hipSetDevice(ctx->device));
  return KSH_BY_KEY(g->key_bytes, pair_plan_t, ctx, g, a, b, d_off_i, d_off_amb, d_off_bma, totals);
}

int ksh_pair_write(ksh_ctx* ctx, const ksh_geom* g, const ksh_set_view* a, const ksh_set_view* b,
                   void* d_keys_i, void* d_keys_amb, void* d_keys_bma) {
  if (!ctx) return fail(KSH_INVALID_ARGUMENT, "ctx is NULL");
  KSH_TRY(check_geom(g));
  KSH_TRY(check_view(a, "a"));
  KSH_TRY(check_view(b, "b"));
  KSH_HIP(hipSetDevice(ctx->device));
  return KSH_BY_KEY(g->key_bytes, pair_write_t, ctx, g, a, b, d_keys_i, d_keys_amb, d_keys_bma);
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
  const int rc = KSH_BY_KEY(g->key_bytes, pair_algebra_batch_t, ctx, g, &job, 1);
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
  return KSH_BY_KEY(g->key_bytes, pair_algebra_batch_t, ctx, g, jobs, n_jobs);
}

int ksh_set_union_plan(ksh_ctx* ctx, const ksh_geom* g, const ksh_set_view* a,
                       const ksh_set_view* b, int64_t* d_off_u, int64_t* total) {
  if (!ctx || !d_off_u || !total) return fail(KSH_INVALID_ARGUMENT, "NULL argument");
  KSH_TRY(check_geom(g));
  KSH_TRY(check_view(a, "a"));
  KSH_TRY(check_view(b, "b"));
  KSH_HIP(hipSetDevice(ctx->device));
  return KSH_BY_KEY(g->key_bytes, union_plan_t, ctx, g, a, b, d_off_u, total);
}

int ksh_set_union_write(ksh_ctx* ctx, const ksh_geom* g, const ksh_set_view* a,
                        const ksh_set_view* b, void* d_keys_u) {
  if (!ctx) return fail(KSH_INVALID_ARGUMENT, "ctx is NULL");
  KSH_TRY(check_geom(g));
  KSH_TRY(check_view(a, "a"));
  KSH_TRY(check_view(b, "b"));
  KSH_HIP(hipSetDevice(ctx->device));
  return KSH_BY_KEY(g->key_bytes, union_write_t, ctx, g, a, b, d_keys_u);
}

int ksh_set_diff(ksh_ctx* ctx, const ksh_geom* g, const ksh_set_view* a, const ksh_set_view* b,
                 int64_t* diff) {
  if (!ctx || !diff) return fail(KSH_INVALID_ARGUMENT, "NULL argument");
  KSH_TRY(check_geom(g));
  KSH_TRY(check_view(a, "a"));
  KSH_TRY(check_view(b, "b"));
  KSH_HIP(hipSetDevice(ctx->device));
  return KSH_BY_KEY(g->key_bytes, set_diff_t, ctx, g, a, b, diff);
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
  return KSH_BY_KEY(g->key_bytes, pair_weights_t, ctx, g, sets, n_sets, bucket_ids, n_ids, pairs, n_pairs, weights);
}

}  // extern "C"
