// SPSS encode on device: bucketed sorted key set -> unitigs -> path cover -> packed strings.
//
// Replaces KmerSetCompact::FromKmerSet = GetSPSSCanonical(kmer_set, fast = true) +
// the 2-bit packing constructor (lib/core/kmer_set_compact.h:36-47,206-266;
// lib/core/spss.h:230-615, :619-695, :1039-1206, :1358-1858), with the
// reference's n_workers == 1 results and the oracle's ordering rules:
// the output strings are equal to the oracle's, string for string and in order.
// tests/model_encode.py is the array-level model of exactly this file.
//
// Vocabulary.  A k-mer is its index t in the set's ascending order.  A *state*
// s = 2t + d walks k-mer t forward (d = 0: enters through its left side, leaves
// through its right side, spelled as is) or reversed (d = 1).  side index: 0 = left,
// 1 = right, so state s enters through side (s & 1) and leaves through side (s & 1) ^ 1,
// i.e. through link[s ^ 1].  A side has a link when it has exactly one neighbour
// whose facing side also has exactly one (spss.h:276-313).
//
//   neighbour probe: per side none / the single neighbour / many, from the 8 candidates of a k-mer
//                 (4 Next, 4 Prev, forward or reverse complement).  Canonical sets: k_rc_* deal the
//                 k-mers out (as records) by the bucket their reverse complement's successors lie in;
//                 k_adj_rc1 chains a group's records in LDS and lets the k-mers of the group's ranges
//                 look for them (k_adj_rc, for groups too large for that: the records look for the
//                 k-mers in staged windows and mark them); k_adj_fwd_targets does the forward half with
//                 one search per k-mer, the Prev side read off the marks the Next probes leave at their
//                 targets (k_adj_fwd_staged: five searches in five windows); k_adjacency: all of it by
//                 searches in global memory (non-canonical sets, geometries outside the staged kernels'
//                 range)
//   k_link_cut    mutual singles: the k-mers with several neighbours on a side cut the facing entries
//   k_end_*       the end k-mers (a side without a link), compacted once
//   k_rank_walk / k_rank_heads / k_ruler_jump / k_l2_*   chains of states ranked through a sparse ruler
//                 set: every ruler and chain start learns (end, distance to end); no per-k-mer records
//   k_choose_ends per end k-mer: the chain that starts at the larger end (spss.h:511,555) -> the
//                 unitigs' heads, lengths, last states
//   k_ruler_walk / k_ruler_heads / k_choose / k_loops / k_emit   the same with a record per k-mer:
//                 sets with a non-branching loop (spelled from its smallest k-mer, spss.h:585-610)
//   k_head_counts / scans / k_unitig_fill   unitig ids in the reference's push order
//   k_edges       <= 4 edges per unitig side, in the reference's enumeration order
//   k_match_*     lexicographically-first maximal matching by rounds of mutual minima
//                 == the sequential greedy sweep of spss.h:1445-1499
//   k_dsu_* / k_loop_cut    loops of the path cover: components by parallel union-find, a loop cut
//                 where the reference's union-by-rank root says (spss.h:1541-1647)
//   k_walk_* / k_string_*    walks over the path cover ranked by pointer jumping; stitch order and
//                 orientation (spss.h:1649-1829)
//   k_emit_log_* / k_pack   the strings' bases from the logs of the ranking walks -> 2-bit words, len - K per string
//
// All of it is integer gather/scatter work bounded by HBM random-access rate; no MFMA.
#include "ksh_internal.h"
#include "ksh_dsu.h"
#include "ksh_kmer.h"

#include <algorithm>
#include <cstddef>
#include <cstdlib>
#include <string>

namespace ksh {

// Sampled rulers: both states of every kRulerEvery-th k-mer (E2).  (Every 32nd, measured again with the one-launch
// ranking walks of round 3: every 16th, an encode of 10^8 k-mers 10.0 ms and the build 0.94 s; every 64th, 11.8 ms
// -- the emit from the logs triples, its stretches outgrow the logged k-mers -- and 0.91 s; every 32nd 9.7 ms, 0.89 s.)
#ifndef KSH_RULER_SHIFT
#define KSH_RULER_SHIFT 5
#endif
constexpr int kRulerShift = KSH_RULER_SHIFT;
constexpr uint32_t kRulerEvery = 1u << kRulerShift;

constexpr uint32_t kNone = 0xFFFFFFFFu;
constexpr uint32_t kMulti = 0xFFFFFFFEu;

// Hops per launch of the pointer-jumping rounds (k_ruler_jump, k_l2_jump, k_walk_jump): a record's reach
// grows (kJumpHops + 1)-fold per launch whatever snapshots of its successors it reads, so that many fewer
// launches end every chain; the rounds are launch-bound on all but the largest sets.
#ifndef KSH_JUMP_HOPS
#define KSH_JUMP_HOPS 4
#endif
constexpr int kJumpHops = KSH_JUMP_HOPS;

// Build with -DKSH_TRACE (make BUILD=build_trace OUT=libkmersets_hip_trace.so EXTRA=-DKSH_TRACE) to record
// s_memtime at the phases of the two probe kernels' workgroups (tools/probe_trace.py); compiled out otherwise.
#ifdef KSH_TRACE
__device__ unsigned long long* g_probe_trace = nullptr;  // [which kernel][workgroup][16]
__device__ long long g_probe_trace_rows = 0;
#define KSH_PMARK(which, m)                                                                              \
  do {                                                                                                   \
    if (g_probe_trace && threadIdx.x == 0 && int64_t(blockIdx.x) < g_probe_trace_rows)                    \
      g_probe_trace[((which) * g_probe_trace_rows + blockIdx.x) * 16 + (m)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
// the same by the first lane of every wave: slot = the wave's number
#define KSH_PMARK_WAVE(which)                                                                            \
  do {                                                                                                   \
    if (g_probe_trace && threadIdx.x % 64 == 0 && threadIdx.x / 64 < 16 && int64_t(blockIdx.x) < g_probe_trace_rows) \
      g_probe_trace[((which) * g_probe_trace_rows + blockIdx.x) * 16 + threadIdx.x / 64] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define KSH_PMARK(which, m) do { } while (0)
#define KSH_PMARK_WAVE(which) do { } while (0)
#endif

// ---------------------------------------------------------------------------------- E1
// Fine index of a set: one workgroup per bucket walks its sorted keys once and records
// where the top fine_bits key bits change.  The bucket's row of the index (2^fine_bits entries, about
// two thirds of an entry per key) is put together in LDS and written out in whole lines: written from
// the key loop, a wave's stores fell a line and a half apart and the kernel ran at a third of the
// memory rate (0.25 ms for the 0.67 GB of a 10^8-k-mer set).
template <typename KeyT>
__global__ __launch_bounds__(256) void k_fine_index(const int64_t* __restrict__ off,
                                                     const KeyT* __restrict__ keys, int key_bits,
                                                     int fine_bits, uint32_t* __restrict__ fine,
                                                     int64_t n_buckets) {
  extern __shared__ uint32_t s_fine[];  // 2^fine_bits
  const int64_t b = blockIdx.x;
  const int64_t lo = off[b], hi = off[b + 1];
  const int kSlices = 1 << fine_bits;
  const int sh = key_bits - fine_bits;
  uint32_t* f = fine + (b << fine_bits);
  if (lo == hi) {
    for (int sub = threadIdx.x; sub < kSlices; sub += 256) f[sub] = uint32_t(lo);
  } else {
    // four rounds of loads in flight per thread
    constexpr int kAhead = 4;
    for (int64_t base = lo; base < hi; base += 256 * kAhead) {
      KeyT cur_key[kAhead], prev_key[kAhead];
#pragma unroll
      for (int u = 0; u < kAhead; u++) {
        const int64_t i = base + u * 256 + threadIdx.x;
        if (i < hi) {
          cur_key[u] = keys[i];
          prev_key[u] = i > lo ? keys[i - 1] : KeyT(0);
        }
      }
#pragma unroll
      for (int u = 0; u < kAhead; u++) {
        const int64_t i = base + u * 256 + threadIdx.x;
        if (i < hi) {
          const int cur = int(uint64_t(cur_key[u]) >> sh);
          const int prev = i > lo ? int(uint64_t(prev_key[u]) >> sh) : -1;
          for (int sub = prev + 1; sub <= cur; sub++) s_fine[sub] = uint32_t(i);
        }
      }
    }
    const int last = int(uint64_t(keys[hi - 1]) >> sh);
    for (int sub = last + 1 + int(threadIdx.x); sub < kSlices; sub += 256) s_fine[sub] = uint32_t(hi);
    __syncthreads();
    for (int sub = threadIdx.x; sub < kSlices; sub += 256) f[sub] = s_fine[sub];
  }
  if (b == n_buckets - 1 && threadIdx.x == 0) fine[n_buckets << fine_bits] = uint32_t(hi);
}

// DevSet::coarse: the bucket of every 256th index.
template <typename KeyT>
__global__ __launch_bounds__(256) void k_coarse_index(DevSet<KeyT> set, int64_t n_entries, uint32_t* __restrict__ coarse) {
  const int64_t e = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (e >= n_entries) return;
  const int64_t t = e << kCoarseShift;
  coarse[e] = uint32_t(set.bucket_search(t < set.n ? t : set.n - 1));
}

// The neighbours of one side of k-mer x (index t, rx = rc(x)): f(index, same) for each, where
// same = 1 when the edge joins the same side of both k-mers (the neighbour is reached through a
// reverse complement).
//   Canonical sets: the 8 candidates of the reference (Next / Prev of x, each as is or
//   reverse-complemented, spss.h:238-273) in two kinds.  The four Next(x, c) are consecutive
//   values, and so are the four Next(rc(x), c) = rc(Prev(x, 3 - c)): one bounded search each
//   finds all of their members that are in the set (only a canonical k-mer can be).  The other
//   candidates, Prev(x, c) and Prev(rc(x), c) = rc(Next(x, 3 - c)), sit in four different buckets
//   and are probed one by one, and only when they are the canonical form.
//   Non-canonical sets (GetUnitigs, spss.h:76-96): k-mers as they are, side 1 = outgoing edges
//   Next(x, .) (one bounded search), side 0 = incoming edges Prev(x, .) (four buckets, probed one
//   by one); no edge flips an orientation.
template <typename KeyT, bool kDirected, typename F>
__device__ __forceinline__ void for_side_neighbours(const DevSet<KeyT>& set, uint64_t x, uint64_t rx,
                                                    int64_t t, int side, F f) {
  const int k = set.k;
  if (kDirected) {
    if (side) {
      set.for_group4(kmer_next(x, k, 0), [&](int64_t idx) {
        if (idx != t) f(idx, 0u);  // next != kmer (spss.h:81)
      });
    } else {
#pragma unroll
      for (int c = 0; c < 4; c++) {
        const uint64_t z = kmer_prev(x, k, c);
        if (z == x) continue;  // prev != kmer (spss.h:91)
        const int64_t idx = set.find(z);
        if (idx >= 0) f(idx, 0u);
      }
    }
    return;
  }
  // group: side 1 -> Next(x, .) (neighbour as is);  side 0 -> Next(rc(x), .) (neighbour
  // reverse-complemented, i.e. a same-side edge)
  set.for_group4(kmer_next(side ? x : rx, k, 0), [&](int64_t idx) {
    if (idx != t) f(idx, side ? 0u : 1u);  // kmer != next (spss.h:242,248)
  });
  // singles: side 1 -> Prev(rc(x), c) = rc(Next(x, 3 - c));  side 0 -> Prev(x, c) as is
  const uint64_t base = side ? rx : x;
#pragma unroll
  for (int c = 0; c < 4; c++) {
    const uint64_t z = kmer_prev(base, k, c);
    if (revcomp(z, k) < z) continue;  // not canonical: cannot be in the set
    if (z == x) continue;
    const int64_t idx = set.find(z);
    if (idx >= 0) f(idx, side ? 1u : 0u);
  }
}

// *self_rc is raised when a canonical set holds a k-mer that is its own reverse complement
// (possible for even k only): its two sides coincide, which the edge table of the path cover (one
// slot per base) does not model, and no instantiation of the reference has an even K.
template <typename KeyT, bool kDirected>
__global__ __launch_bounds__(256) void k_adjacency(DevSet<KeyT> set, uint32_t* __restrict__ nbr,
                                                    int* __restrict__ self_rc) {
  __shared__ int64_t s_bucket[2];
  const int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  const uint64_t x = set.kmer_in_block(t, s_bucket);
  if (t >= set.n) return;
  const uint64_t rx = kDirected ? 0 : revcomp(x, set.k);
  if (!kDirected && rx == x) *self_rc = 1;
  int cnt[2] = {0, 0};
  uint32_t single[2] = {kNone, kNone};
#pragma unroll
  for (int side = 0; side < 2; side++)
    for_side_neighbours<KeyT, kDirected>(set, x, rx, t, side, [&](int64_t idx, uint32_t same) {
      cnt[side]++;
      single[side] = (uint32_t(idx) << 1) | same;
    });
  nbr[2 * t] = cnt[0] == 0 ? kNone : (cnt[0] == 1 ? single[0] : kMulti);
  nbr[2 * t + 1] = cnt[1] == 0 ? kNone : (cnt[1] == 1 ? single[1] : kMulti);
}

// A link is an edge that is the only one on its side at BOTH ends (spss.h:275-313: a side is
// terminal unless it has exactly one neighbour whose facing side has exactly one).  Instead of
// every k-mer reading its neighbour's facing side (two cache-missing reads per k-mer), the rare
// k-mers with several neighbours on a side look those neighbours up again and cut the facing
// entries: nbr turns into the link table in place (several -> none on the way).  Entries only
// ever change to kNone, and a
// k-mer decides from its own two entries alone whether it cuts (`several` is never overwritten
// by another k-mer), so the order of the threads does not matter.
template <typename KeyT, bool kDirected>
__global__ __launch_bounds__(256) void k_link_cut(DevSet<KeyT> set, uint32_t* __restrict__ nbr) {
  const int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= set.n) return;
  const uint2 mine = reinterpret_cast<const uint2*>(nbr)[t];
  if (mine.x != kMulti && mine.y != kMulti) return;
  const uint64_t x = set.kmer(t);
  const uint64_t rx = kDirected ? 0 : revcomp(x, set.k);
  for (int side = 0; side < 2; side++) {
    if ((side ? mine.y : mine.x) != kMulti) continue;
    for_side_neighbours<KeyT, kDirected>(set, x, rx, t, side, [&](int64_t idx, uint32_t same) {
      uint32_t* facing = nbr + 2 * idx + (same ? side : side ^ 1);
      if (*reinterpret_cast<volatile uint32_t*>(facing) < kMulti) *reinterpret_cast<volatile uint32_t*>(facing) = kNone;
    });
    nbr[2 * t + side] = kNone;
  }
}

// ---------------------------------------------------------------------------------- E1b
// The neighbour probe of a canonical set, LDS-staged.  Of the 8 candidates of a k-mer x, the
// forward ones (Next(x, .), Prev(x, .)) sit at addresses that ascend with x and are probed in
// place (k_adj_fwd).  The other half goes through rx = rc(x) and lands anywhere in the set:
// 3 cache-missing probes (6 dependent reads) per k-mer in k_adjacency, 64-byte lines fetched
// for 4-byte answers.  Here those probes are made local instead:
//   * every edge through a reverse complement is seen from both of its ends (x finds y iff y
//     finds x, on the same side of both), so a k-mer does not need its OWN probes answered: it
//     is enough that every probe leaves its mark at the TARGET, which sits in a sorted range
//     that can be staged in LDS and written back coalesced;
//   * the targets of x are fixed by rx: Next(rx, .) lies in bucket G = bits [2K-3, 2K-2-N] of
//     rx, and the four Prev(rx, c) in the 16 sub-ranges of the set with the (N + 4)-bit
//     prefixes [c][top base of rx][G].  So the k-mers are first dealt into 2^N groups by G
//     (k_rc_hist / k_rc_scatter: LDS histograms, one scattered record per k-mer, the same shape
//     as the SPSS decode's bucket scatter), and one workgroup per group (k_adj_rc) stages bucket
//     G (then the 16 sub-ranges) with a slice index over it, lets the group's records look
//     their targets up in LDS, marks the hits with compare-and-swap (none -> the single
//     neighbour -> many) and stores the marks of the whole range: rc0[i] / rc1[i] = what
//     reaches side 0 / side 1 of k-mer i through a reverse complement.  Ranges larger than the
//     LDS window are staged in several batches.
// record of x: key = [top base of rx][low 2K-2-N bits of rx] (2K - N bits, a KeyT), t = index of x.
// G only depends on the low N + 2 bits of x (whole bases), i.e. on the key alone when the key has
// that many bits: the histogram pass streams the keys without knowing their buckets.
template <typename KeyT>
struct __attribute__((aligned(8))) RcRecord {
  KeyT key;
  uint32_t t;
};

template <typename KeyT>
__global__ __launch_bounds__(1024) void k_rc_hist(const KeyT* __restrict__ keys, int64_t n, int k, int key_bits,
                                                   int nbits, int64_t per_row,
                                                   uint32_t* __restrict__ hist_matrix) {
  // (nbits = bits of the group id: the bucket bits N, or N + 1 when a group is half a bucket)
  extern __shared__ uint32_t lds_hist[];
  const int nb = 1 << nbits;
  for (int b = threadIdx.x; b < nb; b += blockDim.x) lds_hist[b] = 0u;
  __syncthreads();
  const int64_t t_begin = int64_t(blockIdx.x) * per_row;
  const int64_t t_end = min(t_begin + per_row, n);
  const int low_bits = 2 * k - 2 - nbits;
  for (int64_t t = t_begin + threadIdx.x; t < t_end; t += blockDim.x) {
    const uint64_t rx = revcomp(uint64_t(keys[t]), k);  // its top nbits + 2 bits are those of rc(x)
    atomicAdd(&lds_hist[uint32_t(rx >> low_bits) & uint32_t(nb - 1)], 1u);
  }
  __syncthreads();
  uint32_t* my_row = hist_matrix + int64_t(blockIdx.x) * nb;
  for (int b = threadIdx.x; b < nb; b += blockDim.x) my_row[b] = lds_hist[b];
}

// One thread per group: exclusive running sum down the workgroups' rows; totals[G] = column sum.
// (A workgroup takes 64 groups; its 16 teams of 64 threads take a sixteenth of the rows each: the
// running sum down a column is 1/16 as long a chain of dependent reads, stitched through LDS.)
constexpr int kColTeams = 16;
__global__ __launch_bounds__(64 * kColTeams) void k_rc_columns(uint32_t* __restrict__ hist_matrix, int64_t n_rows,
                                                                int nb, int64_t* __restrict__ totals) {
  __shared__ uint32_t team_total[kColTeams][64];
  const int lane = threadIdx.x & 63, team = threadIdx.x >> 6;
  const int b = blockIdx.x * 64 + lane;
  const int64_t per_team = (n_rows + kColTeams - 1) / kColTeams;
  const int64_t g0 = min(int64_t(team) * per_team, n_rows), g1 = min(g0 + per_team, n_rows);
  uint32_t run = 0;
  if (b < nb)
    for (int64_t g = g0; g < g1; g++) run += hist_matrix[g * nb + b];
  team_total[team][lane] = run;
  __syncthreads();
  if (b >= nb) return;
  uint32_t before = 0, all = 0;
#pragma unroll
  for (int t2 = 0; t2 < kColTeams; t2++) {
    const uint32_t v = team_total[t2][lane];
    if (t2 < team) before += v;
    all += v;
  }
  run = before;
  for (int64_t g = g0; g < g1; g++) {
    uint32_t* cell = hist_matrix + g * nb + b;
    const uint32_t c = *cell;
    *cell = run;
    run += c;
  }
  if (team == 0) totals[b] = all;
}

// hist_matrix holds each row's exclusive base inside a group, goff the groups' starts.
// The records of a 10^8-k-mer set are 800 MB written at random; scattered writes are absorbed by
// the Infinity Cache only while their destination fits it, so the scatter runs in 2^round_bits
// rounds, round r writing the groups whose top round_bits bits are r (a contiguous share of the
// record array) and streaming the keys once more for it.
template <typename KeyT>
__global__ __launch_bounds__(1024) void k_rc_scatter(DevSet<KeyT> set, int nbits, int round_bits, int round,
                                                      int64_t per_row,
                                                      const uint32_t* __restrict__ hist_matrix,
                                                      const int64_t* __restrict__ goff,
                                                      RcRecord<KeyT>* __restrict__ rec) {
  extern __shared__ uint32_t lds_hist[];  // all of the 64 KB a workgroup may have at N = 14: no static LDS here
  const int nb = 1 << nbits;
  const uint32_t* my_row = hist_matrix + int64_t(blockIdx.x) * nb;
  for (int b = threadIdx.x; b < nb; b += blockDim.x) lds_hist[b] = uint32_t(goff[b]) + my_row[b];
  const int64_t t_begin = int64_t(blockIdx.x) * per_row;
  const int64_t t_end = min(t_begin + per_row, set.n);
  __syncthreads();
  if (t_begin >= t_end) return;
  // the buckets this row's k-mers lie in (uniform addresses: every thread searches for itself)
  const int64_t b_first = set.bucket_of(t_begin), b_last = set.bucket_of(t_end - 1);
  const int k = set.k, low_bits = 2 * set.k - 2 - nbits;
  const uint64_t low_mask = (uint64_t(1) << low_bits) - 1;
  for (int64_t t = t_begin + threadIdx.x; t < t_end; t += blockDim.x) {
    const uint64_t key = uint64_t(set.keys[t]);
    // the group follows from the key alone (k_rc_hist): keys of other rounds stop here
    const uint32_t grp = uint32_t(revcomp(key, k) >> low_bits) & uint32_t(nb - 1);
    if (int(grp >> (nbits - round_bits)) != round) continue;
    int64_t lo = b_first, hi = b_last;  // largest b in the row's span with off[b] <= t
    while (lo < hi) {
      const int64_t mid = (lo + hi + 1) >> 1;
      if (set.off[mid] <= t) lo = mid; else hi = mid - 1;
    }
    const uint64_t rx = revcomp((uint64_t(lo) << set.key_bits) | key, k);
    const uint32_t dst = atomicAdd(&lds_hist[grp], 1u);
    RcRecord<KeyT> r;
    r.key = KeyT(((rx >> (2 * k - 2)) << low_bits) | (rx & low_mask));
    r.t = uint32_t(t);
    rec[dst] = r;
  }
}

// ---- the scatter in two levels, for large sets --------------------------------------------------
// k_rc_scatter writes one record per k-mer to a random one of 2^N groups: 10^8 scattered 8-byte
// stores, 2.9 ms, as slow as the memory system is at it.  Here every store instruction writes
// runs instead.  Level 1 (k_rc_scatter_l1, the same persistent rows and histogram rows): a row
// takes its k-mers tile by tile, sorts a tile by SUPER-GROUP (the top kSgBits of G) in LDS and
// appends each super-group's run to the row's share of that super-group -- known exactly from the
// histogram rows, so no atomics and the same order every run.  Level 2 (k_rc_scatter_l2): a tile
// of the intermediate array lies in one super-group (two at a seam), is sorted by group in LDS, a
// group's run is appended where one atomicAdd per (tile, group) says.  The order of the records
// inside a group is not deterministic; nothing downstream depends on it.
constexpr int kSgBits = 7;
constexpr int kL1Threads = 1024;
constexpr int kL2Threads = 256;
constexpr int kL2Per = 8;
constexpr int kL2Tile = kL2Threads * kL2Per;

template <typename KeyT, int kPer>
__global__ __launch_bounds__(kL1Threads) void k_rc_scatter_l1(DevSet<KeyT> set, int nbits, int64_t per_row,
                                                               const uint32_t* __restrict__ hist_matrix,
                                                               const int64_t* __restrict__ goff,
                                                               RcRecord<KeyT>* __restrict__ tmp_rec,
                                                               uint16_t* __restrict__ tmp_g) {
  constexpr int kTile = kL1Threads * kPer;
  constexpr int kSg = 1 << kSgBits;
  __shared__ RcRecord<KeyT> s_rec[kTile];
  __shared__ uint16_t s_g[kTile];
  __shared__ uint32_t s_cur[kSg], s_cnt[kSg], s_lbase[kSg + 1];
  // the offsets of the buckets the row's k-mers lie in: a k-mer's bucket is a search in LDS, not a chain of
  // dependent global loads per k-mer (round 2: 8 loads per k-mer, 77 % of the kernel's wave cycles waiting)
  constexpr int kOffCap = 1024;
  __shared__ int64_t s_off[kOffCap];
  const int tid = threadIdx.x;
  const int nb = 1 << nbits, lo_bits = nbits - kSgBits;
  const uint32_t* my_row = hist_matrix + int64_t(blockIdx.x) * nb;
  if (tid < kSg) {
    // where this row's share of super-group tid starts: the groups before it inside the super-group are
    // full, plus what the rows before this one put into each of its groups
    uint32_t at = uint32_t(goff[int64_t(tid) << lo_bits]);
    for (int gi = 0; gi < (1 << lo_bits); gi++) at += my_row[(tid << lo_bits) + gi];
    // (my_row holds, per group, the records of the rows before this one; the records of EARLIER groups of
    // the super-group that belong to LATER rows do not precede us in the intermediate array's layout
    // [super-group][row][tile order], so the share's start is the sum above)
    s_cur[tid] = at;
    s_cnt[tid] = 0;
  }
  const int64_t t_begin = int64_t(blockIdx.x) * per_row;
  const int64_t t_end = min(t_begin + per_row, set.n);
  __syncthreads();
  if (t_begin >= t_end) return;
  const int64_t b_first = set.bucket_of(t_begin), b_last = set.bucket_of(t_end - 1);
  const int k = set.k, low_bits = 2 * set.k - 2 - nbits;
  const uint64_t low_mask = (uint64_t(1) << low_bits) - 1;
  const int n_span = int(min<int64_t>(b_last - b_first + 1, kOffCap + 1));
  const bool off_staged = n_span <= kOffCap;
  if (off_staged)
    for (int j = tid; j < n_span; j += kL1Threads) s_off[j] = set.off[b_first + j];
  __syncthreads();
  for (int64_t t0 = t_begin; t0 < t_end; t0 += kTile) {
    const int tile_n = int(min<int64_t>(kTile, t_end - t0));
    RcRecord<KeyT> mine[kPer];
    uint32_t grp[kPer], rank[kPer];
#pragma unroll
    for (int j = 0; j < kPer; j++) {
      const int64_t t = t0 + tid + j * kL1Threads;
      grp[j] = 0xFFFFFFFFu;
      if (t >= t_end) continue;
      const uint64_t key = uint64_t(set.keys[t]);
      int64_t lo = b_first, hi = b_last;  // largest b in the row's span with off[b] <= t
      if (off_staged) {
        int l2 = 0, h2 = n_span - 1;
        while (l2 < h2) {
          const int mid = (l2 + h2 + 1) >> 1;
          if (s_off[mid] <= t) l2 = mid; else h2 = mid - 1;
        }
        lo = b_first + l2;
      } else {
        while (lo < hi) {
          const int64_t mid = (lo + hi + 1) >> 1;
          if (set.off[mid] <= t) lo = mid; else hi = mid - 1;
        }
      }
      const uint64_t rx = revcomp((uint64_t(lo) << set.key_bits) | key, k);
      grp[j] = uint32_t(rx >> low_bits) & uint32_t(nb - 1);
      mine[j].key = KeyT(((rx >> (2 * k - 2)) << low_bits) | (rx & low_mask));
      mine[j].t = uint32_t(t);
      rank[j] = atomicAdd(&s_cnt[grp[j] >> lo_bits], 1u);
    }
    __syncthreads();
    if (tid < 64) {
      // exclusive scan of the kSg counts by the first wave, two counts per lane (it used to be 128 threads adding up
      // to 127 counts each, one after the other, while fourteen waves waited at the barrier)
      static_assert(kSg == 128, "two super-groups per lane of one wave");
      const uint32_t c0 = s_cnt[2 * tid], c1 = s_cnt[2 * tid + 1];
      uint32_t inc = c0 + c1;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(inc, d, 64);
        if (tid >= d) inc += o;
      }
      s_lbase[2 * tid] = inc - c0 - c1;
      s_lbase[2 * tid + 1] = inc - c1;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kPer; j++) {
      if (grp[j] == 0xFFFFFFFFu) continue;
      const uint32_t pos = s_lbase[grp[j] >> lo_bits] + rank[j];
      s_rec[pos] = mine[j];
      s_g[pos] = uint16_t(grp[j]);
    }
    __syncthreads();
    for (int i = tid; i < tile_n; i += kL1Threads) {
      const uint32_t sg = uint32_t(s_g[i]) >> lo_bits;
      const uint32_t dst = s_cur[sg] + (uint32_t(i) - s_lbase[sg]);
      tmp_rec[dst] = s_rec[i];
      tmp_g[dst] = s_g[i];
    }
    __syncthreads();
    if (tid < kSg) {
      s_cur[tid] += s_cnt[tid];
      s_cnt[tid] = 0;
    }
    __syncthreads();
  }
}

template <typename KeyT>
__global__ __launch_bounds__(kL2Threads) void k_rc_scatter_l2(int64_t n, int nbits, const int64_t* __restrict__ goff,
                                                               const RcRecord<KeyT>* __restrict__ tmp_rec,
                                                               const uint16_t* __restrict__ tmp_g,
                                                               uint32_t* __restrict__ cursor,
                                                               RcRecord<KeyT>* __restrict__ rec) {
  constexpr int kBins = 512;  // two super-groups' worth of groups: a super-group has at most 2^(15 - kSgBits) of them
  __shared__ RcRecord<KeyT> s_rec[kL2Tile];
  __shared__ uint16_t s_bin[kL2Tile];
  __shared__ uint32_t s_cnt[kBins], s_lbase[kBins], s_gbase[kBins];
  __shared__ uint32_t s_first;
  const int tid = threadIdx.x;
  const int lo_bits = nbits - kSgBits;
  const int64_t p0 = int64_t(blockIdx.x) * kL2Tile;
  const int tile_n = int(min<int64_t>(kL2Tile, n - p0));
  for (int b = tid; b < kBins; b += kL2Threads) s_cnt[b] = 0;
  if (tid == 0) s_first = (uint32_t(tmp_g[p0]) >> lo_bits) << lo_bits;  // first group of the tile's first super-group
  __syncthreads();
  const uint32_t base_g = s_first;
  const uint32_t n_bins = min<uint32_t>(uint32_t(kBins), (2u << lo_bits));
  RcRecord<KeyT> mine[kL2Per];
  uint32_t bin[kL2Per], rank[kL2Per];
#pragma unroll
  for (int j = 0; j < kL2Per; j++) {
    const int i = tid + j * kL2Threads;
    bin[j] = 0xFFFFFFFFu;
    if (i >= tile_n) continue;
    mine[j] = tmp_rec[p0 + i];
    const uint32_t g = tmp_g[p0 + i];
    if (g - base_g < n_bins) {
      bin[j] = g - base_g;
      rank[j] = atomicAdd(&s_cnt[bin[j]], 1u);
    } else {
      // a tile over three or more super-groups (tiny or very skewed sets): one record at a time
      rec[uint32_t(goff[g]) + atomicAdd(&cursor[g], 1u)] = mine[j];
    }
  }
  __syncthreads();
  {
    // exclusive scan of the bin counts: thread t holds bins 2t, 2t + 1 (kBins == 2 * kL2Threads)
    static_assert(kBins == 2 * kL2Threads, "two bins per thread");
    __shared__ uint32_t s_wave[kL2Threads / 64];
    const uint32_t c0 = s_cnt[2 * tid], c1 = s_cnt[2 * tid + 1];
    uint32_t inc = c0 + c1;
    const int lane = tid & 63, wv = tid >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t o = __shfl_up(inc, d, 64);
      if (lane >= d) inc += o;
    }
    if (lane == 63) s_wave[wv] = inc;
    __syncthreads();
    uint32_t before = inc - (c0 + c1);
    for (int w = 0; w < wv; w++) before += s_wave[w];
    s_lbase[2 * tid] = before;
    s_lbase[2 * tid + 1] = before + c0;
    s_gbase[2 * tid] = c0 ? uint32_t(goff[base_g + 2 * tid]) + atomicAdd(&cursor[base_g + 2 * tid], c0) : 0u;
    s_gbase[2 * tid + 1] = c1 ? uint32_t(goff[base_g + 2 * tid + 1]) + atomicAdd(&cursor[base_g + 2 * tid + 1], c1) : 0u;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < kL2Per; j++) {
    if (bin[j] == 0xFFFFFFFFu) continue;
    const uint32_t pos = s_lbase[bin[j]] + rank[j];
    s_rec[pos] = mine[j];
    s_bin[pos] = uint16_t(bin[j]);
  }
  __syncthreads();
  uint32_t staged = 0;
  for (int b = 0; b < kBins; b++) staged += s_cnt[b];  // (uniform; the far records were written directly)
  for (uint32_t i = tid; i < staged; i += kL2Threads) {
    const uint32_t b = s_bin[i];
    rec[s_gbase[b] + (i - s_lbase[b])] = s_rec[i];
  }
}

// First index of the set whose k-mer is >= value (value may be 4^K: the set's end).
template <typename KeyT>
__device__ int64_t lower_bound_kmer(const DevSet<KeyT>& set, uint64_t value) {
  const int64_t b = int64_t(value >> set.key_bits);
  if (b >= set.n_buckets) return set.n;
  const KeyT key = KeyT(value & set.key_mask());
  int64_t lo, hi;
  set.probe_range(b, key, &lo, &hi);  // the key's slice of the fine index (the first key >= it lies in it or right after it)
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (set.keys[mid] < key) lo = mid + 1; else hi = mid;
  }
  return lo;
}

__device__ __forceinline__ void mark_hit(uint32_t* slot, uint32_t v) {
  if (atomicCAS(slot, kNone, v) != kNone) *reinterpret_cast<volatile uint32_t*>(slot) = kMulti;
}

constexpr int kRcSegs = 16;

// pb[32 * G + 2 * seg + which]: the index range of the set with the (N + 4)-bit prefix
// [c][tb][G] (seg = 4 c + tb), for every group G: one thread per bound, all searches in flight
// together instead of a chain of dependent loads at the head of every k_adj_rc workgroup.
// (nbits = bits of the group id.  pb0[2 * G + which], when the groups are finer than buckets: the range with
// the nbits-bit prefix G, where pass 0 looks; with whole buckets that is the bucket's offsets.)
template <typename KeyT>
__global__ __launch_bounds__(256) void k_rc_bounds(DevSet<KeyT> set, int nbits, int64_t* __restrict__ pb,
                                                    int64_t* __restrict__ pb0) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  const int64_t n_pb = int64_t(2 * kRcSegs) << nbits;
  if (i < n_pb) {
    const uint64_t grp = uint64_t(i) >> 5, seg = (uint64_t(i) >> 1) & 15, which = uint64_t(i) & 1;
    const uint64_t prefix = (seg << nbits) | grp;
    pb[i] = lower_bound_kmer(set, (prefix + which) << (2 * set.k - 4 - nbits));
  } else if (pb0 && i - n_pb < (int64_t(2) << nbits)) {
    const uint64_t j = uint64_t(i - n_pb);
    pb0[j] = lower_bound_kmer(set, ((j >> 1) + (j & 1)) << (2 * set.k - nbits));
  }
}
struct RcBatch {
  int64_t seg_lo[kRcSegs], seg_hi[kRcSegs];  // the pass's target ranges (indices of the set)
  int64_t win_lo[kRcSegs];                   // the part of each that is staged in this batch ...
  int win_len[kRcSegs];
  int win_off[kRcSegs];                      // ... where it sits in the LDS window ...
  int idx_off[kRcSegs];                      // ... and where its slice index sits
  int idx_shift[kRcSegs];                    // slice of a key = (key & seg_mask) >> idx_shift
  unsigned long long packed[kRcSegs];        // win_len | win_off << 16 | idx_off << 32 | idx_shift << 48: one read per look-up
  int used;                                  // keys staged
  int next_seg;                              // cursor: first range not yet fully staged ...
  int64_t next_pos;                          // ... and how far it got
};

// One workgroup per group G (a bucket, or half a bucket: gbits = N + extra bits of rx below its top base).
// Pass 0: range = the keys with the gbits-bit prefix G, a record looks for the members of Next(rx, .)
// and marks their side 0.  Pass 1: 16 ranges [c][tb][G], a record looks for its canonical Prev(rx, c)
// in range (c, its tb) and marks side 1.
// LDS: cap keys | cap marks | cap + 2 * kRcSegs slice bounds (u16).  A range's slice index has
// the largest power of two <= its length many slices over the key bits the range's keys differ in.
// A wave's life here is memory round trips and barriers, so each is taken once: the next record is
// requested before this record's look-ups, the range bounds of both passes arrive with the group's
// record range, every staging loop issues all its loads before the first LDS store, in pass 1
// team w (a sixteenth of the workgroup) stages, indexes and stores range w, and a pass whose ranges
// fit the window together -- all but the densest groups -- is laid out by sixteen lanes at once and
// ends after its one batch (round 2: one thread planned every batch and every pass took a second
// trip through the planner to learn that nothing was left).  kRcThreads follows the group size.
template <typename KeyT, int kRcThreads>
__global__ __launch_bounds__(kRcThreads, (kRcThreads >= 512 ? 8 : (kRcThreads >= 128 ? kRcThreads / 128 : 1))) void k_adj_rc(
    DevSet<KeyT> set, int gbits, const int64_t* __restrict__ goff, const RcRecord<KeyT>* __restrict__ rec,
    const int64_t* __restrict__ pb, const int64_t* __restrict__ pb0, int cap, uint32_t* __restrict__ rc0,
    uint32_t* __restrict__ rc1, int* __restrict__ batched, int pass1_cap, int* __restrict__ taken) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  KeyT* skeys = reinterpret_cast<KeyT*>(lds_raw);
  uint32_t* slots = reinterpret_cast<uint32_t*>(lds_raw + size_t(cap) * sizeof(KeyT));
  uint16_t* sidx = reinterpret_cast<uint16_t*>(lds_raw + size_t(cap) * (sizeof(KeyT) + 4));
  __shared__ RcBatch bt;
  __shared__ int64_t prev_bounds[2 * kRcSegs];
  static_assert(kRcThreads % kRcSegs == 0, "pass 1 deals the ranges out to sixteen teams");
  constexpr int kTeam = kRcThreads / kRcSegs;
  const int tid = threadIdx.x, lane = tid % kTeam, wave = tid / kTeam;
  const int64_t grp = blockIdx.x;
  const int k = set.k, low_bits = 2 * set.k - 2 - gbits;
  const int extra = gbits - int(2 * set.k - set.key_bits);  // group bits beyond the bucket bits
  const uint64_t low_mask = (uint64_t(1) << low_bits) - 1;
  const uint64_t kmask = kmer_mask(k);
  const int64_t r0 = goff[grp], r1 = goff[grp + 1];
  if (r1 - r0 <= int64_t(pass1_cap)) return;  // a group of at most that many records is k_adj_rc1's, both passes
  if (pass1_cap >= 0 && threadIdx.x == 0) *taken = 1;  // (ksh_spss_encode_routes: some group did not fit k_adj_rc1)
  // pass 0's range: the bucket's offsets, or the searched bounds of a finer group
  const int64_t p0_lo = pb0 ? pb0[2 * grp] : set.off[grp], p0_hi = pb0 ? pb0[2 * grp + 1] : set.off[grp + 1];
  // the bits of a bucket key that a pass-0 target of this group starts with (the group bits below the bucket's)
  const uint64_t gkey_top = extra > 0 ? (uint64_t(grp) & ((uint64_t(1) << extra) - 1)) << (set.key_bits - extra) : 0;
  KSH_PMARK_WAVE(2);  // when each wave of the workgroup started
  if (tid < 2 * kRcSegs) prev_bounds[tid] = pb[2 * kRcSegs * grp + tid];
  KSH_PMARK(1, 0);
  for (int pass = 0; pass < 2; pass++) {
    const int n_seg = pass == 0 ? 1 : kRcSegs;
    // key bits that vary inside a range: all below the group's in its own range, four fewer in [c][tb][G]
    const int seg_bits = (pass == 0 ? set.key_bits : set.key_bits - 4) - extra;
    const uint64_t seg_mask = seg_bits >= 64 ? ~uint64_t(0) : ((uint64_t(1) << seg_bits) - 1);
    uint32_t* out = pass == 0 ? rc0 : rc1;
    __syncthreads();
    // slices of a range of `take` keys: the largest power of two <= take, each at least 4 values wide
    const auto slice_lg = [&](int take) {
      int lg = take >= 2 ? 31 - __builtin_clz(unsigned(take)) : 0;
      if (lg > seg_bits - 2) lg = seg_bits - 2 > 0 ? seg_bits - 2 : 0;
      return lg;
    };
    // one thread lays out the next batch from the cursor (ranges that do not fit the window together)
    const auto plan_serial = [&]() {
      int used = 0, iused = 0, seg = bt.next_seg;
      int64_t pos = bt.next_pos;
      for (int s2 = 0; s2 < n_seg; s2++) {
        bt.win_len[s2] = 0;
        bt.packed[s2] = 0;
      }
      while (seg < n_seg) {
        const int64_t avail = bt.seg_hi[seg] - pos;
        const int take = int(avail < int64_t(cap - used) ? avail : int64_t(cap - used));
        const int lg = slice_lg(take);
        bt.win_lo[seg] = pos;
        bt.win_len[seg] = take;
        bt.win_off[seg] = used;
        bt.idx_off[seg] = iused;
        bt.idx_shift[seg] = seg_bits - lg;
        bt.packed[seg] = (unsigned long long)(take) | ((unsigned long long)(used) << 16) |
                         ((unsigned long long)(iused) << 32) | ((unsigned long long)(seg_bits - lg) << 48);
        iused += (1 << lg) + 1;
        used += take;
        pos += take;
        if (pos < bt.seg_hi[seg]) break;  // the window is full
        seg++;
        if (seg < n_seg) pos = bt.seg_lo[seg];
      }
      bt.used = used;
      bt.next_seg = seg;
      bt.next_pos = pos;
    };
    // the pass's first batch: the first wave, lane s for range s; everything at once when it fits
    if (tid < 64) {
      const int s2 = tid;
      int64_t lo = 0, hi = 0;
      if (s2 < n_seg) {
        lo = pass == 0 ? p0_lo : prev_bounds[2 * s2];
        hi = pass == 0 ? p0_hi : prev_bounds[2 * s2 + 1];
        bt.seg_lo[s2] = lo;
        bt.seg_hi[s2] = hi;
      }
      const int64_t len64 = hi - lo;
      const int len = int(len64 < int64_t(cap) + 1 ? len64 : int64_t(cap) + 1);
      const int lg = slice_lg(len);
      int inc = len, iinc = s2 < n_seg ? (1 << lg) + 1 : 0;
#pragma unroll
      for (int d = 1; d < kRcSegs; d <<= 1) {
        const int o = __shfl_up(inc, d, 64), io = __shfl_up(iinc, d, 64);
        if (s2 >= d) {
          inc += o;
          iinc += io;
        }
      }
      const int total = __shfl(inc, kRcSegs - 1, 64);  // (lanes past n_seg hold empty ranges)
      if (total <= cap) {
        if (s2 < n_seg) {
          const int used = inc - len, iused = iinc - ((1 << lg) + 1);
          bt.win_lo[s2] = lo;
          bt.win_len[s2] = len;
          bt.win_off[s2] = used;
          bt.idx_off[s2] = iused;
          bt.idx_shift[s2] = seg_bits - lg;
          bt.packed[s2] = (unsigned long long)(len) | ((unsigned long long)(used) << 16) |
                          ((unsigned long long)(iused) << 32) | ((unsigned long long)(seg_bits - lg) << 48);
        }
        if (s2 == 0) {
          bt.used = total;
          bt.next_seg = n_seg;
          bt.next_pos = 0;
        }
      } else if (s2 == 0) {
        bt.next_seg = 0;
        bt.next_pos = lo;
        *batched = 1;  // (ksh_spss_encode_routes: some group's ranges did not fit the window together)
      }
      // (the serial planner reads seg_lo / seg_hi written by the other lanes of this wave: same wave, in order)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      if (total > cap && s2 == 0) plan_serial();
    }
    while (true) {
      if (pass == 0) KSH_PMARK_WAVE(3);  // when each wave reached the barrier behind the planner (its last visit)
      __syncthreads();
      KSH_PMARK(1, 1 + 7 * pass);  // planned (the group's record range and the bounds have arrived)
      if (bt.used == 0) break;  // an empty pass
      // who stages what: pass 0, the whole workgroup its one range; pass 1, team w range w
      const int my_seg = pass == 0 ? 0 : wave;
      const int my_id = pass == 0 ? tid : lane, my_step = pass == 0 ? kRcThreads : kTeam;
      const int64_t my_lo = bt.win_lo[my_seg];
      const int my_len = bt.win_len[my_seg], my_off = bt.win_off[my_seg];
      const bool last_batch = bt.next_seg >= n_seg;
      // the first record of this thread is requested ahead of the staging loads, the next one ahead of each
      // record's look-ups.  (Three turns ahead measured the same look-up phase -- 18.6 k against 18.2 k cycles
      // for a thread's six records in pass 0, tools/probe_trace.py -- and a slower kernel, 1.73 against 1.62 ms:
      // the look-ups do not wait for their records but for the LDS, which 32 waves per CU keep busy with
      // reads at random addresses, some 500 cycles per LDS instruction and wave.)
      RcRecord<KeyT> nxt;
      nxt.t = kNone;
      if (r0 + tid < r1) nxt = rec[r0 + tid];
      for (int base = 0; base < my_len; base += 4 * my_step) {
        KeyT v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int i = base + u * my_step + my_id;
          if (i < my_len) v[u] = set.keys[my_lo + i];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int i = base + u * my_step + my_id;
          if (i < my_len) {
            skeys[my_off + i] = v[u];
            slots[my_off + i] = kNone;
          }
        }
      }
      KSH_PMARK(1, 2 + 7 * pass);  // this wave's staging loads are in LDS
      __syncthreads();
      KSH_PMARK(1, 3 + 7 * pass);
      // slice index: sidx[idx_off + j] = first position of the window whose slice is >= j
      if (my_len > 0) {
        const int ioff = bt.idx_off[my_seg], sh = bt.idx_shift[my_seg];
        const int n_slices = 1 << (seg_bits - sh);
        for (int i = my_id; i < my_len; i += my_step) {
          const int cur = int((uint64_t(skeys[my_off + i]) & seg_mask) >> sh);
          const int prev = i > 0 ? int((uint64_t(skeys[my_off + i - 1]) & seg_mask) >> sh) : -1;
          for (int j = prev + 1; j <= cur; j++) sidx[ioff + j] = uint16_t(i);
          if (i == my_len - 1)
            for (int j = cur + 1; j <= n_slices; j++) sidx[ioff + j] = uint16_t(my_len);
        }
      }
      __syncthreads();
      KSH_PMARK(1, 4 + 7 * pass);  // slice index built
      // (what every record of the pass reads from the batch: once, not per record -- the compiler cannot hoist an
      // LDS read over the compare-and-swaps on `slots`)
      const int p0_shift = bt.idx_shift[0];
      const uint32_t p0_win_lo = uint32_t(bt.win_lo[0]);
#pragma unroll 1
      for (int64_t r = r0 + tid; r < r1; r += kRcThreads) {
        const RcRecord<KeyT> rr = nxt;
        if (r + kRcThreads < r1) nxt = rec[r + kRcThreads];
        const uint64_t low = uint64_t(rr.key) & low_mask;
        const uint32_t mark = (rr.t << 1) | 1u;
        if (pass == 0) {
          const uint64_t gkey = gkey_top | (low << 2);  // the bucket key of Next(rx, A)
          const int sl = int((gkey & seg_mask) >> p0_shift);
          for (int i = sidx[sl], end = sidx[sl + 1]; i < end; i++) {
            const uint64_t d = uint64_t(skeys[i]) - gkey;  // members: gkey .. gkey + 3
            if (d < 4 && p0_win_lo + uint32_t(i) != rr.t) mark_hit(&slots[i], mark);
          }
        } else {
          const uint64_t tb = uint64_t(rr.key) >> low_bits;
          const uint64_t rx = (tb << (2 * k - 2)) | (uint64_t(grp) << low_bits) | low;
          const uint64_t x = revcomp(rx, k);
#pragma unroll
          for (int c = 0; c < 4; c++) {
            const unsigned long long pk = bt.packed[4 * c + int(tb)];
            const int wlen = int(pk & 0xFFFF);
            const uint64_t z = (uint64_t(c) << (2 * k - 2)) | (rx >> 2);  // Prev(rx, c)
            const uint64_t rz = ((x << 2) & kmask) | uint64_t(3 - c);     // its reverse complement
            if (wlen == 0 || rz < z || z == x) continue;  // not staged now; not canonical (cannot be in the set); x itself
            const KeyT zkey = KeyT(z & set.key_mask());
            const int off = int((pk >> 16) & 0xFFFF);
            const uint16_t* ix = sidx + int((pk >> 32) & 0xFFFF) + int((uint64_t(zkey) & seg_mask) >> int(pk >> 48));
            for (int i = ix[0], end = ix[1]; i < end; i++)
              if (skeys[off + i] == zkey) mark_hit(&slots[off + i], mark);
          }
        }
      }
      KSH_PMARK(1, 5 + 7 * pass);  // this wave's records looked up
      __syncthreads();
      KSH_PMARK(1, 6 + 7 * pass);  // everybody's
      for (int i = my_id; i < my_len; i += my_step) out[my_lo + i] = slots[my_off + i];
      KSH_PMARK(1, 7 + 7 * pass);  // marks stored
      if (last_batch) break;  // (the usual case: the pass was one batch)
      __syncthreads();
      if (tid == 0) plan_serial();
    }
  }
}

// ---------------------------------------------------------------------------------- E1b turned round
// k_adj_rc asks, per record x, for the members of Next(rx, .) in the group's own range (pass 0) and for the four
// Prev(rx, c) in sixteen staged and indexed ranges (pass 1), and leaves marks at what it finds.  Turned round
// (round 4, after k_adj_fwd_targets): every edge through a reverse complement is seen from both of its ends, so
// instead of the records looking for k-mers, the k-mers of those ranges look for records:
//   pass 1: z is reached on its side 1 by the records with rx = Next(z, c') -- the four record keys that differ in
//           their last base only -- and the z that group G's records can reach are the k-mers of G's sixteen
//           ranges [c][tb][G];
//   pass 0: y of the group's own range is reached on its side 0 by the records with rx = Prev(y, a) -- the four
//           record keys that differ in their top base only.
// So a workgroup puts the group's RECORDS in LDS, chained by the bits of the key between its top and its last base
// (what both passes know of the records they look for; arrival order, one exchange per record, no sort), streams
// the group's range and the sixteen ranges once each, coalesced, and every k-mer walks one chain and counts what
// it finds: its own verdict, written in place.  No staging or slice index of the ranges, no marks, no canonical
// test (a streamed k-mer IS in the set), no batches for the dense A-led buckets (the streams have no capacity;
// what has to fit is the group's records, and those are spread evenly: +-19 % by the first base of G).
// A chain entry of a 2- or 4-byte key is ONE 8-byte word: the key without the bits the chain implies | t | next.
// Groups whose records do not fit (10^8 k-mers in 2^10 buckets, 5 x 10^8 in 2^14) stay with k_adj_rc.
constexpr int kRc1SlicesMax = 4096;
constexpr int kRc1LdsBytes = 81408;  // two workgroups and their static LDS in a CU's 160 KB
template <typename KeyT>
struct Rc1Cfg {
  static constexpr bool kPacked = sizeof(KeyT) <= 4;
  // heads (4 bytes a slice) | entries: one word, or key | t | next (2 bytes)
  static constexpr int kEntryBytes = kPacked ? 8 : int(sizeof(KeyT)) + 6;
  static constexpr int kCap = int((kRc1LdsBytes - kRc1SlicesMax * 4) / kEntryBytes) & ~7;
  static_assert(kCap < 2 * kRc1SlicesMax - 1, "chain links have one bit more than the slices");
};

template <typename KeyT, int kThreads>
__global__ __launch_bounds__(kThreads) void k_adj_rc1(DevSet<KeyT> set, int gbits, const int64_t* __restrict__ goff,
                                                      const RcRecord<KeyT>* __restrict__ rec,
                                                      const int64_t* __restrict__ pb, const int64_t* __restrict__ pb0,
                                                      int cap, int sbits, int batched, uint32_t* __restrict__ rc0,
                                                      uint32_t* __restrict__ rc1, int* __restrict__ took_batches) {
  constexpr bool kPacked = Rc1Cfg<KeyT>::kPacked;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  uint32_t* head = reinterpret_cast<uint32_t*>(lds_raw);
  const int n_slices = 1 << sbits;
  // packed: entry[i]; else rt[i], rkey[i], nxt[i]
  unsigned long long* entry = reinterpret_cast<unsigned long long*>(head + n_slices);
  uint32_t* rt = head + n_slices;
  KeyT* rkey = reinterpret_cast<KeyT*>(rt + cap);
  uint16_t* nxt = reinterpret_cast<uint16_t*>(rkey + cap);
  __shared__ int64_t s_lo[kRcSegs];
  __shared__ __attribute__((aligned(16))) int s_cum[kRcSegs + 4];
  const int tid = threadIdx.x;
  const int64_t grp = blockIdx.x;
  const int64_t r0 = goff[grp], r1 = goff[grp + 1];
  if (!batched && r1 - r0 > int64_t(cap)) return;  // k_adj_rc has this group
  // (KSH_RC1=batched) a group of more records than fit is taken in batches of `cap`: the streams run once per batch, a
  // k-mer's verdict of the batches before is read back and merged (none -> the single neighbour -> several)
  const int n_rec = int(r1 - r0 < int64_t(cap) ? r1 - r0 : int64_t(cap));
  if (r1 - r0 > int64_t(cap) && threadIdx.x == 0) *took_batches = 1;  // (ksh_spss_encode_routes)
  const int rest_bits = 2 * set.k - 4 - gbits;  // bits of a record key between its top base and its last base
  const uint64_t rest_mask = (uint64_t(1) << rest_bits) - 1;
  const uint64_t low_mask = (uint64_t(1) << (rest_bits + 2)) - 1;  // a record key without its top base
  // the chain of a record: the low e_sbits of those bits
  const int e_sbits = sbits < rest_bits ? sbits : rest_bits;
  const uint32_t slice_mask = (1u << e_sbits) - 1u;
  // a packed entry: [t : 31][next : sbits + 1][the key without the chain's bits: its last base below the rest]
  const int kf_bits = 2 * set.k - gbits - e_sbits;  // (key bits 2 + rest_bits + 2, less the chain's: <= 32 - sbits when they matter)
  const int t_shift = kf_bits + sbits + 1;
  const uint32_t none = (2u << sbits) - 1u;  // "no next": all ones in sbits + 1 bits (cap < that)
  const uint64_t kf_mask = (uint64_t(1) << kf_bits) - 1;
  // pass 0's range: the bucket's offsets, or the searched bounds of a finer group
  const int64_t p0_lo = pb0 ? pb0[2 * grp] : set.off[grp], p0_hi = pb0 ? pb0[2 * grp + 1] : set.off[grp + 1];
  for (int sl = tid; sl < n_slices; sl += kThreads) head[sl] = none;
  if (tid < 64) {
    int64_t lo = 0, hi = 0;
    if (tid < kRcSegs) {
      lo = pb[2 * kRcSegs * grp + 2 * tid];
      hi = pb[2 * kRcSegs * grp + 2 * tid + 1];
      s_lo[tid] = lo;
    }
    int inc = int(hi - lo);  // (a range of the set: < 2^31)
#pragma unroll
    for (int d = 1; d < kRcSegs; d <<= 1) {
      const int o = __shfl_up(inc, d, 64);
      if (tid >= d) inc += o;
    }
    if (tid < kRcSegs) s_cum[tid + 1] = inc;
    if (tid == 0) s_cum[0] = 0;
  }
  // four records of a thread are requested together, then chained in; the first four before the barrier
  RcRecord<KeyT> r[4];
#pragma unroll
  for (int u = 0; u < 4; u++) {
    const int i = u * kThreads + tid;
    if (i < n_rec) r[u] = rec[r0 + i];
  }
  __syncthreads();
  // (the thread's first k-mers of both streams are requested before the records are chained in: a round trip less
  // on the workgroup's path)
  const int total = s_cum[kRcSegs];
  const int c4 = s_cum[4], c8 = s_cum[8], c12 = s_cum[12];
  // position idx of pass 1's stream -> index of the set, top base tb of its range: by c (three compares), then by
  // tb within the four ranges of that c (one 16-byte read)
  const auto place = [&](int idx, int* tb) {
    const int c = (idx >= c4 ? 1 : 0) + (idx >= c8 ? 1 : 0) + (idx >= c12 ? 1 : 0);
    const int4 cq = *reinterpret_cast<const int4*>(&s_cum[4 * c]);
    *tb = (idx >= cq.y ? 1 : 0) + (idx >= cq.z ? 1 : 0) + (idx >= cq.w ? 1 : 0);
    const int before = *tb == 0 ? cq.x : (*tb == 1 ? cq.y : (*tb == 2 ? cq.z : cq.w));
    return s_lo[4 * c + *tb] + (idx - before);
  };
  KeyT y_next = 0;
  if (p0_lo + tid < p0_hi) y_next = set.keys[p0_lo + tid];
  int tb_next = 0;
  int64_t i_next = 0;
  KeyT key_next = 0;
  if (tid < total) {
    i_next = place(tid, &tb_next);
    key_next = set.keys[i_next];
  }
  const auto chain_in = [&](int i, const RcRecord<KeyT>& rr) {
    const uint64_t key = uint64_t(rr.key);
    const uint32_t sl = uint32_t(key >> 2) & slice_mask;
    const uint32_t before = atomicExch(&head[sl], uint32_t(i));  // (none = all ones in sbits + 1 bits)
    if (kPacked) {
      const uint64_t kf = ((key >> (2 + e_sbits)) << 2) | (key & 3);
      entry[i] = (uint64_t(rr.t) << t_shift) | (uint64_t(before) << kf_bits) | kf;
    } else {
      rkey[i] = rr.key;
      rt[i] = rr.t;
      nxt[i] = uint16_t(before);
    }
  };
  for (int base = 0; base < n_rec; base += 4 * kThreads) {
    if (base > 0) {
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int i = base + u * kThreads + tid;
        if (i < n_rec) r[u] = rec[r0 + i];
      }
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int i = base + u * kThreads + tid;
      if (i < n_rec) chain_in(i, r[u]);
    }
  }
  __syncthreads();
  // One chain walk: f(key without the chain's bits [the rest above e_sbits, tb on top][last base], t) per entry.
  const auto walk = [&](uint32_t sl, auto f) {
    uint32_t j = head[sl];
    while (j != none) {
      if (kPacked) {
        const uint64_t e = entry[j];
        f(e & kf_mask, uint32_t(e >> t_shift));
        j = uint32_t(e >> kf_bits) & none;
      } else {
        const uint64_t key = uint64_t(rkey[j]);
        const uint32_t nj = nxt[j];
        f(((key >> (2 + e_sbits)) << 2) | (key & 3), rt[j]);
        j = uint32_t(nj);  // (none fits 16 bits)
      }
    }
  };
  const uint64_t hi_mask = (uint64_t(1) << (rest_bits - e_sbits)) - 1;  // (the rest above the chain's bits, without tb)
  // pass 0: y (index i, key ykey) of the group's own range against the records [a][y without its last base], whatever a
  const auto look0 = [&](int64_t i, KeyT ykey) {
    const uint64_t low = (uint64_t(ykey) >> 2) & low_mask;  // the records' key below their top base
    const uint64_t want = ((low >> (2 + e_sbits)) << 2) | (low & 3);
    int cnt = 0;
    uint32_t single = kNone;
    walk(uint32_t(low >> 2) & slice_mask, [&](uint64_t kf, uint32_t t) {
      if ((((kf >> 2) & hi_mask) << 2 | (kf & 3)) == want && t != uint32_t(i)) {  // (t == i: y = Next(rc(y), c), the k-mer itself)
        cnt++;
        single = (t << 1) | 1u;
      }
    });
    return cnt == 0 ? kNone : (cnt == 1 ? single : kMulti);
  };
  // pass 1: z (index i, key) of range [c][tb][G] against the records [tb][z's rest][c'], whatever c'
  const auto look1 = [&](int64_t i, int tb, KeyT key) {
    const uint64_t q = (uint64_t(tb) << rest_bits) | (uint64_t(key) & rest_mask);  // Next(z, .) >> 2, as a record key
    const uint64_t want = q >> e_sbits;
    int cnt = 0;
    uint32_t single = kNone;
    walk(uint32_t(q) & slice_mask, [&](uint64_t kf, uint32_t t) {
      if ((kf >> 2) == want && t != uint32_t(i)) {  // (t == i: rc(z) = Next(z, c'), the k-mer itself)
        cnt++;
        single = (t << 1) | 1u;
      }
    });
    return cnt == 0 ? kNone : (cnt == 1 ? single : kMulti);
  };
  // what a batch found, added to what the batches before it found
  const auto merged = [](uint32_t now, uint32_t before) {
    const int total = (now == kNone ? 0 : (now == kMulti ? 2 : 1)) + (before == kNone ? 0 : (before == kMulti ? 2 : 1));
    return total == 0 ? kNone : (total > 1 ? kMulti : (now != kNone ? now : before));
  };
#pragma unroll 1
  for (int64_t i = p0_lo + tid; i < p0_hi; i += kThreads) {
    const KeyT ykey = y_next;
    if (i + kThreads < p0_hi) y_next = set.keys[i + kThreads];
    rc0[i] = look0(i, ykey);
  }
#pragma unroll 1
  for (int idx = tid; idx < total; idx += kThreads) {
    const int tb = tb_next;
    const int64_t i = i_next;
    const KeyT key = key_next;
    if (idx + kThreads < total) {
      i_next = place(idx + kThreads, &tb_next);
      key_next = set.keys[i_next];
    }
    rc1[i] = look1(i, tb, key);
  }
  // the batches after the first (groups of 30 000 records: 5 x 10^8 k-mers in 2^14 buckets, 10^8 in 2^10)
#pragma unroll 1
  for (int64_t rb = r0 + cap; rb < r1; rb += cap) {
    const int n_b = int(r1 - rb < int64_t(cap) ? r1 - rb : int64_t(cap));
    __syncthreads();  // everybody is through with the chains of the batch before
    for (int sl = tid; sl < n_slices; sl += kThreads) head[sl] = none;
    __syncthreads();
    for (int i = tid; i < n_b; i += kThreads) chain_in(i, rec[rb + i]);
    __syncthreads();
    for (int64_t i = p0_lo + tid; i < p0_hi; i += kThreads) {
      const uint32_t now = look0(i, set.keys[i]);
      if (now != kNone) rc0[i] = merged(now, rc0[i]);
    }
    for (int idx = tid; idx < total; idx += kThreads) {
      int tb;
      const int64_t i = place(idx, &tb);
      const uint32_t now = look1(i, tb, set.keys[i]);
      if (now != kNone) rc1[i] = merged(now, rc1[i]);
    }
  }
}

// ---------------------------------------------------------------------------------- E1c
// The forward half, LDS-staged.  A workgroup owns kFwdChunk consecutive k-mers.  Their Next(x, .)
// are ascending with x (same top base: one contiguous range of the set, about four times as many
// keys as the chunk has k-mers) and so is each of their Prev(x, c) (a range of about a quarter of
// the chunk): five ranges, staged with whole-line loads, searched in LDS.  k_fwd_bounds finds the
// ten range bounds of every chunk beforehand, one thread per chunk boundary (the searches of all
// chunks in flight together instead of ten dependent round trips at the head of every workgroup),
// and leaves the chunk's first k-mer beside them: the buckets of the chunk's own k-mers and of
// its five ranges follow from it, and their offsets are read into LDS together with the keys.
// A chunk whose k-mers span two top bases, or whose range does not fit the window or spans too
// many buckets, probes that part in global memory as k_adj_fwd does.
constexpr int kFwdChunk = 512;
// The ranges are "about" 4 x and 1/4 x the chunk only where the set is equally dense at the chunk and at
// its targets, and a canonical set is not: a k-mer is the smaller of itself and its reverse complement,
// so k-mers starting with A, C, G, T are kept in the proportions 7 : 5 : 3 : 1, and a chunk of T... k-mers
// whose successors start with A finds 7 x as many keys in its Next range.  The windows below hold a
// ratio of 1.5; 15 % of the k-mers of a random genome exceed it and probe in global memory (a build with
// -DKSH_FWD_DEBUG counts them).  Measured: windows for a ratio of 4 cut that to 3 % and made the kernel
// SLOWER (2.0 -> 2.2 ms per 10^8: 48 KB of LDS, three workgroups per CU instead of four) -- the kernel is
// bound by how many workgroups are in flight, not by the fall-backs.
template <typename KeyT>
struct FwdCfg {
  static constexpr int kCapNext = 3072;  // keys: 4 x chunk with half to spare
  static constexpr int kCapPrev = 512;   // keys per c (a whole number of rounds of the workgroup)
  static_assert(kCapPrev % kFwdChunk == 0 && kCapNext % kFwdChunk == 0, "staging shape");
};
constexpr int kFwdSpan = 16;       // bucket offsets kept per range (and for the chunk itself)

// bounds[kFwdBounds * c + ..] for chunk boundary t = c * kFwdChunk (c = 0 .. n_chunks):
//   [0]     first index >= Next(x_t, 0)            (start of chunk c's Next range)
//   [1..4]  first index >= Prev(x_t, cc)           (start of chunk c's Prev ranges)
//   [5]     first index >  Next(x_{t-1}, 3)        (end of chunk c-1's Next range)
//   [6..9]  first index >  Prev(x_{t-1}, cc)       (end of chunk c-1's Prev ranges)
//   [10]    x_t, [11] x_{t-1}: Next(x, .) drops the top base, so the Next range of a chunk is one
//           range only when its first and last k-mer agree on it
constexpr int kFwdBounds = 12;

template <typename KeyT>
__global__ __launch_bounds__(256) void k_fwd_bounds(DevSet<KeyT> set, int64_t n_chunks,
                                                     int64_t* __restrict__ bounds) {
  // a thread per search (ten per boundary): the searches are chains of dependent reads, and a thread per
  // boundary put two hundred thousand of them on a GPU that holds half a million threads
  const int64_t id = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  const int64_t c = id / 10;
  const int j = int(id - c * 10);
  if (c > n_chunks) return;
  const int k = set.k;
  const int64_t t = c * kFwdChunk;
  int64_t* out = bounds + kFwdBounds * c;
  if (j < 5) {
    if (t < set.n) {
      const uint64_t x = set.kmer(t);
      out[j] = lower_bound_kmer(set, j == 0 ? kmer_next(x, k, 0) : kmer_prev(x, k, j - 1));
      if (j == 0) out[10] = int64_t(x);
    }
  } else if (t > 0) {
    const uint64_t y = set.kmer((t < set.n ? t : set.n) - 1);
    out[j] = lower_bound_kmer(set, (j == 5 ? kmer_next(y, k, 3) : kmer_prev(y, k, j - 6)) + 1);
    if (j == 5) out[11] = int64_t(y);
  }
}

// First position of [lo, hi) whose key is >= key (hi when there is none).  Halving form: the trip count depends on
// the range's length alone -- the lanes of a wave that search the same bucket's part of a window leave the loop
// together -- and a step is a read, a compare and a select.
template <typename KeyT>
__device__ __forceinline__ int lds_lower_bound(const KeyT* a, int lo, int hi, KeyT key) {
  int len = hi - lo;
  if (len <= 0) return lo;
  while (len > 1) {
    const int half = len >> 1;
    lo = a[lo + half - 1] < key ? lo + half : lo;
    len -= half;
  }
  return lo + (a[lo] < key ? 1 : 0);
}
// The position of `key` in [lo, hi), or -1: the same search; the range's last key decides.
template <typename KeyT>
__device__ __forceinline__ int lds_find(const KeyT* a, int lo, int hi, KeyT key) {
  int len = hi - lo;
  if (len <= 0) return -1;
  while (len > 1) {
    const int half = len >> 1;
    lo = a[lo + half - 1] < key ? lo + half : lo;
    len -= half;
  }
  const KeyT at = a[lo];
  if (at == key) return lo;
  return (at < key && lo + 1 < hi && a[lo + 1] == key) ? lo + 1 : -1;
}

// KSH_FWD_DEBUG=1: how often the staged forward probe falls back to global probes (per k-mer and cause)
__device__ unsigned long long g_fwd_dbg[8];


// (Measured and dropped, round 4: the five searches of a k-mer IN LOCKSTEP -- branch-free halving of all open ranges
// per step, the LDS reads of a step issued together and waited for together, so that the longest dependent chain is
// the Next search's 12 reads instead of 11 + 4 x 7 one after the other; 64 vector registers, no spills, the oracle's
// strings: 2.30 ms per 10^8 against 1.85, 213 us against 176 on the difference set.  With 32 waves per CU the
// LDS pipeline is kept busy by OTHER waves while one waits; what a wave's own reads in flight together add is
// predication and state (five ranges, five probes per step), not throughput.
// gpurun_out/r04/encs_lock_kernel_stats.csv)
// (Measured and dropped, round 4: a hybrid -- the Next window staged and searched in LDS, the four Prev(x, c) probed in
// global memory through the fine index (their targets are as local as the windows: neighbouring threads hit the
// same lines), so that the LDS pipeline carries 11 of the 39 dependent reads and the vector-memory pipeline the
// rest: 2.60 ms per 10^8 against 1.85.)
template <typename KeyT, int kFwdCapNext = FwdCfg<KeyT>::kCapNext>
__global__ __launch_bounds__(kFwdChunk, (sizeof(KeyT) <= 4 ? 8 : 4)) void k_adj_fwd_staged(DevSet<KeyT> set, const int64_t* __restrict__ bounds,
                                                              const uint32_t* __restrict__ rc0,
                                                              const uint32_t* __restrict__ rc1,
                                                              uint32_t* __restrict__ nbr,
                                                              int* __restrict__ self_rc) {
  constexpr int kFwdCapPrev = FwdCfg<KeyT>::kCapPrev;
  static_assert(kFwdCapNext % kFwdChunk == 0, "staging shape");
  __shared__ KeyT s_next[kFwdCapNext];
  __shared__ KeyT s_prev[4][kFwdCapPrev];
  __shared__ int64_t s_b[2 * kFwdBounds];     // this boundary's and the next one's records
  // where the buckets begin, in window coordinates (32-bit: the searches' bounds need no 64-bit clamps): for range r
  // < 5 the position in its window of the first key of bucket fb[r] + j, clamped to [0, len[r]]; [5]: the chunk's own
  // buckets relative to its first index
  __shared__ int s_rel[6][kFwdSpan + 1];
  const int tid = threadIdx.x;
  const int64_t chunk = blockIdx.x;
  const int64_t t = chunk * kFwdChunk + tid;
  const int k = set.k;
  KSH_PMARK(0, 0);
  if (tid < 2 * kFwdBounds) s_b[tid] = bounds[kFwdBounds * chunk + tid];
  KeyT my_key = 0;
  uint2 rc = make_uint2(kNone, kNone);
  if (t < set.n) {
    my_key = set.keys[t];
    rc = make_uint2(rc0[t], rc1[t]);
  }
  __syncthreads();
  KSH_PMARK(0, 1);  // the bound record has arrived
  const uint64_t x_first = uint64_t(s_b[10]), x_last = uint64_t(s_b[kFwdBounds + 11]);
  // first bucket of: the five ranges, the chunk itself
  int fb[6];
  fb[0] = int(kmer_next(x_first, k, 0) >> set.key_bits);
#pragma unroll
  for (int cc = 0; cc < 4; cc++) fb[1 + cc] = int(kmer_prev(x_first, k, cc) >> set.key_bits);
  fb[5] = int(x_first >> set.key_bits);
  // usable[r]: range r is staged (a range that wraps around a top base has hi < lo)
  bool usable[5];
  int len[5];
  int64_t lo_of[5];
#pragma unroll
  for (int r = 0; r < 5; r++) {
    lo_of[r] = s_b[r];
    const int64_t l = s_b[kFwdBounds + 5 + r] - lo_of[r];
    usable[r] = l >= 0 && l <= (r == 0 ? kFwdCapNext : kFwdCapPrev);
    len[r] = usable[r] ? int(l) : 0;
  }
  if ((x_first >> (2 * k - 2)) != (x_last >> (2 * k - 2))) {
    usable[0] = false;
    len[0] = 0;
  }
  if (tid < 6 * (kFwdSpan + 1)) {
    const int r = tid / (kFwdSpan + 1), j = tid % (kFwdSpan + 1);
    const int64_t b = int64_t(fb[r]) + j;
    const int64_t at = b <= set.n_buckets ? set.off[b] : set.n;
    int64_t rel;
    if (r < 5) {
      // (dynamic indexing of lo_of / len would put them in scratch: pick by comparison)
      int64_t lo_r = lo_of[0];
      int len_r = len[0];
#pragma unroll
      for (int q = 1; q < 5; q++)
        if (r == q) {
          lo_r = lo_of[q];
          len_r = len[q];
        }
      rel = at - lo_r;
      rel = rel < 0 ? 0 : (rel > len_r ? len_r : rel);
    } else {
      rel = at - chunk * kFwdChunk;
      rel = rel < -(int64_t(1) << 30) ? -(int64_t(1) << 30) : (rel > (int64_t(1) << 30) ? (int64_t(1) << 30) : rel);
    }
    s_rel[r][j] = int(rel);
  }
  {
    // every load of the staging before the first LDS store: one round trip, not ten
    constexpr int kPer = kFwdCapNext / kFwdChunk, kPerPrev = kFwdCapPrev / kFwdChunk;
    KeyT vn[kPer], vp[4][kPerPrev];
#pragma unroll
    for (int u = 0; u < kPer; u++)
      if (tid + u * kFwdChunk < len[0]) vn[u] = set.keys[lo_of[0] + tid + u * kFwdChunk];
#pragma unroll
    for (int cc = 0; cc < 4; cc++)
#pragma unroll
      for (int u = 0; u < kPerPrev; u++)
        if (tid + u * kFwdChunk < len[1 + cc]) vp[cc][u] = set.keys[lo_of[1 + cc] + tid + u * kFwdChunk];
#pragma unroll
    for (int u = 0; u < kPer; u++)
      if (tid + u * kFwdChunk < len[0]) s_next[tid + u * kFwdChunk] = vn[u];
#pragma unroll
    for (int cc = 0; cc < 4; cc++)
#pragma unroll
      for (int u = 0; u < kPerPrev; u++)
        if (tid + u * kFwdChunk < len[1 + cc]) s_prev[cc][tid + u * kFwdChunk] = vp[cc][u];
    KSH_PMARK(0, 2);  // this wave's window loads have arrived and are in LDS
  }
  __syncthreads();
  KSH_PMARK(0, 3);    // everybody's are
  if (t >= set.n) return;
  // my bucket: the chunk's first bucket, or one of the next few
  int64_t my_b = fb[5];
  {
    int j = 0;
    while (j < kFwdSpan && s_rel[5][j + 1] <= tid) j++;
    my_b += j;
    if (j == kFwdSpan)
      while (set.off[my_b + 1] <= t) my_b++;
  }
  const uint64_t x = (uint64_t(my_b) << set.key_bits) | uint64_t(my_key);
  // one reverse complement per k-mer: rc(Prev(x, c)) = Next(rc(x), 3 - c) is a shift and an or (the canonical tests of
  // the four candidates used to be four more bit reversals of 64-bit words, a fifth of the thread's vector ALU work)
  const uint64_t rx = revcomp(x, k);
  const uint64_t rx_next = (rx << 2) & kmer_mask(k);
  if (rx == x) *self_rc = 1;
  int cnt[2] = {0, 0};
  uint32_t single[2] = {kNone, kNone};
  // the part [lo, hi) of window r that bucket b's keys take; false when b is beyond the buckets kept for r
  const auto bucket_range = [&](int r, int b, int* lo, int* hi) {
    const int j = b - fb[r];
    if (j < 0 || j >= kFwdSpan) return false;
    *lo = s_rel[r][j];
    *hi = s_rel[r][j + 1];
    return true;
  };
  const uint32_t t32 = uint32_t(t);
  uint32_t lo32[5];
#pragma unroll
  for (int r = 0; r < 5; r++) lo32[r] = uint32_t(lo_of[r]);
  // candidates whose range is not staged (a range that wraps a top base, overflows its window or spans too many
  // buckets: about a seventh of the k-mers of a canonical set) are probed in global memory AFTER the staged
  // searches, all of a wave's pending Prev candidates in the same trips of one loop -- at their five separate
  // places a wave ran each fall-back whenever any of its lanes needed it
  uint32_t pend = 0;  // bit c: Prev(x, c); bit 4: the Next group
  const uint64_t g0 = kmer_next(x, k, 0);
  // side 1: Next(x, .), neighbour as is
  {
    int lo, hi;
#ifdef KSH_FWD_DEBUG
    atomicAdd(&g_fwd_dbg[0], 1ull);
    if (!usable[0]) atomicAdd(&g_fwd_dbg[1], 1ull);
    else if (!bucket_range(0, int(g0 >> set.key_bits), &lo, &hi)) atomicAdd(&g_fwd_dbg[2], 1ull);
    if (s_b[kFwdBounds + 5] - s_b[0] > kFwdCapNext) atomicAdd(&g_fwd_dbg[5], 1ull);
    if ((x_first >> (2 * k - 2)) != (x_last >> (2 * k - 2))) atomicAdd(&g_fwd_dbg[6], 1ull);
#endif
    if (usable[0] && bucket_range(0, int(g0 >> set.key_bits), &lo, &hi)) {
      const KeyT gkey = KeyT(g0 & set.key_mask());
      if (lo < hi) {
        int i = lds_lower_bound(s_next, lo, hi, gkey);
        for (; i < hi && KeyT(s_next[i] - gkey) < KeyT(4); i++) {  // (gkey ends in base A: no wrap below it)
          const uint32_t idx = lo32[0] + uint32_t(i);
          if (idx == t32) continue;
          cnt[1]++;
          single[1] = idx << 1;
        }
      }
    } else {
      pend |= 16u;
    }
  }
  KSH_PMARK(0, 4);    // Next side done
  // side 0: Prev(x, c), neighbour as is
#pragma unroll
  for (int c = 0; c < 4; c++) {
    const uint64_t z = kmer_prev(x, k, c);
    if ((rx_next | uint64_t(3 - c)) < z) continue;  // rc(z) < z: not canonical, cannot be in the set
    if (z == x) continue;
    int lo, hi;
#ifdef KSH_FWD_DEBUG
    if (!usable[1 + c]) atomicAdd(&g_fwd_dbg[3], 1ull);
    else if (!bucket_range(1 + c, int(z >> set.key_bits), &lo, &hi)) atomicAdd(&g_fwd_dbg[4], 1ull);
#endif
    if (usable[1 + c] && bucket_range(1 + c, int(z >> set.key_bits), &lo, &hi)) {
      const KeyT zkey = KeyT(z & set.key_mask());
      const int i = lds_find(s_prev[c], lo, hi, zkey);
      if (i >= 0) {
        cnt[0]++;
        single[0] = (lo32[1 + c] + uint32_t(i)) << 1;
      }
    } else {
      pend |= 1u << c;
    }
  }
  if (pend & 16u)
    set.for_group4(g0, [&](int64_t idx) {
      if (idx == t) return;
      cnt[1]++;
      single[1] = uint32_t(idx) << 1;
    });
  pend &= 15u;
#pragma unroll 1
  while (pend) {
    const int c = __ffs(int(pend)) - 1;
    pend &= pend - 1;
    const int64_t idx = set.find(kmer_prev(x, k, c));
    if (idx >= 0) {
      cnt[0]++;
      single[0] = uint32_t(idx) << 1;
    }
  }
  KSH_PMARK(0, 5);    // Prev side done
  const uint32_t r[2] = {rc.x, rc.y};
  uint32_t out[2];
#pragma unroll
  for (int side = 0; side < 2; side++) {
    const int total = cnt[side] + (r[side] == kNone ? 0 : (r[side] == kMulti ? 2 : 1));
    out[side] = total == 0 ? kNone : (total > 1 ? kMulti : (cnt[side] == 1 ? single[side] : r[side]));
  }
  reinterpret_cast<uint2*>(nbr)[t] = make_uint2(out[0], out[1]);
  KSH_PMARK(0, 6);    // searched and stored (the first wave)
}

// ---------------------------------------------------------------------------------- E1d
// The forward half with ONE search per k-mer (round 4).  An edge "as is" is seen from both of its ends like an
// edge through a reverse complement: y = Next(x, c) iff x = Prev(y, x's top base).  So the four Prev(x, c)
// searches of k_adj_fwd_staged find nothing that the Next searches of other k-mers do not find as well -- it is
// enough that every Next probe leaves its mark at the TARGET, as k_adj_rc's probes do.  The cut of the work that
// makes the targets local: a workgroup owns a window W of kTgtChunk consecutive k-mers of the set, cut where
// the (K-1)-base prefix changes, i.e. all k-mers whose prefix w lies in [v, v'), and it streams the k-mers Q
// whose SUFFIX lies in [v, v'): four ranges of the set, one per top base a, [a v, a v').  Every Next(q, .) of a
// q in Q is in W or nowhere, and nobody else's is.  W is staged in LDS with a mark per key (8 KB at 4-byte keys
// against the 20 KB of five windows); every q searches once, counts what it finds (its side 1) and marks the
// found (their side 0: none -> q -> several); the marks are combined with rc0 and stored by the window's owner.
// Per k-mer: one staged key and one streamed key instead of six staged keys, one search instead of five, no
// reverse complement (a target found in the set IS canonical), no range that overflows its window.  Only the
// counts of Q vary: a canonical set keeps a k-mer that starts with T^j only if it ends in A^j, so the window
// over the last prefixes of the set spans a wide range of them and its stream is long (measured: 1.8 x 10^5 k-mers
// for the last window of a 10^8-k-mer genome set, 473 us of one workgroup at the end of a 1.37 ms kernel; a
// median window streams 800).  A window whose stream is longer than kTgtStreamMax is therefore cut again, at the
// quantiles of its longest stream range: k_tgt_split counts the parts and k_tgt_subcuts searches their bounds, both
// on the device; the parts run as workgroups behind the windows'.
#ifndef KSH_TGT_CHUNK
#define KSH_TGT_CHUNK 2048
#endif
#ifndef KSH_TGT_CHUNK64
#define KSH_TGT_CHUNK64 2048
#endif
// (window keys per workgroup; a stream longer than twice that is cut.  8 bytes of LDS per 4-byte key, 12 per 8-byte
// key -- 20 and 30 KB per workgroup: four workgroups per CU, as many as its 2048 threads allow, either way.  Measured
// at 4-byte keys, 512 threads, 10^8-k-mer genome set / 9 x 10^6 difference set: 1024 keys 1038 / 85 us, 1536 keys
// 896 / 76, 2048 keys 865 / 75; 2048 keys and 256 threads 925 / 80, 1024 keys and 128 threads 955 / 82.)
template <typename KeyT>
struct TgtCfg {
  static constexpr int kChunk = sizeof(KeyT) == 8 ? KSH_TGT_CHUNK64 : KSH_TGT_CHUNK;
  static constexpr int kStreamMax = 2 * kChunk;
};
#ifndef KSH_TGT_THREADS
#define KSH_TGT_THREADS 512
#endif
constexpr int kTgtThreads = KSH_TGT_THREADS;
constexpr int kTgtSpan = 16;    // bucket offsets kept per range
constexpr int kTgtBounds = 6;   // per cut: [0] index b of the cut, [1] v = the (K-1)-base prefix there, [2 + a] first index >= a v
struct TgtTask {
  int64_t c;     // the window
  int j, m;      // part j of m (j = 1 .. m - 1; part 0 is the window's own workgroup)
};

// Eight lanes per cut c (c = 0 .. n_chunks): lanes 0..4 read the k-mers around index c * kTgtChunk and agree on
// the cut (the first index >= c * kTgtChunk where the prefix changes: at most 3 further, a prefix has four
// k-mers), lanes 0..3 then search the four stream bounds.
template <typename KeyT>
__global__ __launch_bounds__(256) void k_tgt_bounds(DevSet<KeyT> set, int64_t n_chunks, int64_t* __restrict__ bounds) {
  const int64_t id = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  const int64_t c = id >> 3;
  const int j = int(id & 7);
  const bool live = c <= n_chunks;
  const int k = set.k;
  const int64_t base = (live ? c : 0) * TgtCfg<KeyT>::kChunk;
  // prefix of the k-mer at base - 1 + j (j = 0 .. 4); past the end: none
  const int64_t p = base - 1 + j;
  uint64_t pref = ~uint64_t(0);
  if (live && j < 5 && p >= 0 && p < set.n) pref = set.kmer(p) >> 2;
  const int lane0 = int(threadIdx.x & 63) & ~7;
  const uint64_t before = __shfl(pref, lane0, 64);  // the k-mer in front of the nominal cut
  // (lanes 1..4: index base .. base + 3) the cut is the first of them that is past the end or starts a new prefix
  const bool breaks = j >= 1 && j < 5 && (p >= set.n || pref != before);
  const unsigned long long m = __ballot(breaks) >> lane0;
  const int first = __ffsll((long long)(m & 0x1Eull)) - 1;  // 1 .. 4 (a prefix has at most four k-mers)
  int64_t b = base - 1 + first;
  uint64_t v = __shfl(pref, lane0 + (first > 0 ? first : 0), 64);
  if (c == 0) {
    b = 0;
    v = 0;
  } else if (b >= set.n) {
    b = set.n;
    v = uint64_t(1) << (2 * k - 2);
  }
  if (!live) return;
  if (j < 4) bounds[kTgtBounds * c + 2 + j] = lower_bound_kmer(set, (uint64_t(j) << (2 * k - 2)) + v);
  if (j == 4) {
    bounds[kTgtBounds * c] = b;
    bounds[kTgtBounds * c + 1] = int64_t(v);
  }
}

// One thread per window: rec[c] = its two cuts when its stream is short; else part 0 keeps the first cut, the
// other parts become tasks (their workgroups: n_chunks + e, e from the counter) and k_tgt_subcuts fills the rest.
__global__ __launch_bounds__(256) void k_tgt_split(const int64_t* __restrict__ cuts, int64_t n_chunks, int64_t cap,
                                                    int stream_max, int64_t* __restrict__ rec,
                                                    TgtTask* __restrict__ task, int* __restrict__ n_extra) {
  const int64_t c = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (c >= n_chunks) return;
  const int64_t* lo = cuts + kTgtBounds * c;
  const int64_t* hi = lo + kTgtBounds;
  int64_t stream = 0;
#pragma unroll
  for (int a = 0; a < 4; a++) stream += hi[2 + a] - lo[2 + a];
  int64_t* r = rec + 2 * kTgtBounds * c;
#pragma unroll
  for (int i = 0; i < kTgtBounds; i++) r[i] = lo[i];
  const int64_t m = (stream + stream_max - 1) / stream_max;
  int64_t e = 0;
  if (m > 1) {
    e = int64_t(atomicAdd(n_extra, int(m - 1)));
    if (e + m - 1 > cap) {  // (cannot happen: the streams add up to n, the parts beyond the first to less than n / stream_max)
      atomicSub(n_extra, int(m - 1));
      e = -1;
    }
  }
  if (m <= 1 || e < 0) {
#pragma unroll
    for (int i = 0; i < kTgtBounds; i++) r[kTgtBounds + i] = hi[i];
    return;
  }
  for (int64_t j = 1; j < m; j++) task[e + j - 1] = TgtTask{c, int(j), int(m)};
}

// Eight lanes per task: the cut of part j of m -- five searches -- is where part j begins and part j - 1 ends.
template <typename KeyT>
__global__ __launch_bounds__(256) void k_tgt_subcuts(DevSet<KeyT> set, const int64_t* __restrict__ cuts, int64_t n_chunks,
                                                      int64_t cap, const TgtTask* __restrict__ task,
                                                      const int* __restrict__ n_extra, int64_t* __restrict__ rec) {
  const int64_t id = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  const int64_t e = id >> 3;
  const int lane = int(id & 7);
  const int64_t count = *n_extra < cap ? int64_t(*n_extra) : cap;
  if (e >= count || lane > 4) return;
  const TgtTask t = task[e];
  const int k = set.k;
  // the cut: the suffix of the k-mer a j-th of the way through the LONGEST of the window's four stream ranges -- that
  // range is cut into equal parts whatever its k-mers' spread (a range ascends in its suffixes: the cuts ascend with
  // j), the other three where their suffixes fall.  (The first version cut the prefix range [v_c, v_c+1) evenly:
  // right for k-mers spread like a genome's, one part for a stream clustered in a corner of a mostly empty range.)
  const int64_t* lo_c = cuts + kTgtBounds * t.c;
  const int64_t* hi_c = lo_c + kTgtBounds;
  int best = 0;
  int64_t best_len = hi_c[2] - lo_c[2];
#pragma unroll
  for (int a = 1; a < 4; a++) {
    const int64_t len = hi_c[2 + a] - lo_c[2 + a];
    if (len > best_len) {
      best_len = len;
      best = a;
    }
  }
  const int64_t pos = lo_c[2 + best] + int64_t((uint64_t(best_len) * uint64_t(t.j)) / uint64_t(t.m));  // (< 2^31 * 2^31)
  const uint64_t v = set.kmer(pos) & ((uint64_t(1) << (2 * k - 2)) - 1);
  const int64_t at = lane < 4 ? lower_bound_kmer(set, (uint64_t(lane) << (2 * k - 2)) + v) : lower_bound_kmer(set, v << 2);
  const int slot = lane < 4 ? 2 + lane : 0;
  int64_t* mine = rec + 2 * kTgtBounds * (n_chunks + e);                                  // part j begins here
  int64_t* before = rec + 2 * kTgtBounds * (t.j == 1 ? t.c : n_chunks + e - 1) + kTgtBounds;  // part j - 1 ends here
  mine[slot] = at;
  before[slot] = at;
  if (lane == 4) {
    mine[1] = int64_t(v);
    before[1] = int64_t(v);
  }
  if (t.j == t.m - 1) {  // the last part ends where the window does
    const int64_t* hi = cuts + kTgtBounds * (t.c + 1);
    mine[kTgtBounds + slot] = hi[slot];
    if (lane == 4) mine[kTgtBounds + 1] = hi[1];
  }
}

__device__ __forceinline__ uint32_t side_verdict(uint32_t direct, uint32_t through_rc) {
  const int total = (direct == kNone ? 0 : (direct == kMulti ? 2 : 1)) + (through_rc == kNone ? 0 : (through_rc == kMulti ? 2 : 1));
  return total == 0 ? kNone : (total > 1 ? kMulti : (direct != kNone ? direct : through_rc));
}

template <typename KeyT>
__global__ __launch_bounds__(kTgtThreads, 8) void k_adj_fwd_targets(DevSet<KeyT> set, const int64_t* __restrict__ rec,
                                                                   int64_t n_chunks, const int* __restrict__ n_extra,
                                                                   const uint32_t* __restrict__ rc0,
                                                                   const uint32_t* __restrict__ rc1,
                                                                   uint32_t* __restrict__ nbr, int* __restrict__ self_rc) {
  constexpr int kRounds = (TgtCfg<KeyT>::kChunk + 3 + kTgtThreads - 1) / kTgtThreads;  // staging rounds of the window
  __shared__ KeyT s_keys[kRounds * kTgtThreads];
  __shared__ uint32_t s_slot[kRounds * kTgtThreads];
  __shared__ int64_t s_b[2 * kTgtBounds];
  // where the buckets begin, relative to the range: [a] for the stream of top base a, [4] for the window
  __shared__ int s_rel[5 * (kTgtSpan + 1)];  // row r at r * (kTgtSpan + 1)
  const int tid = threadIdx.x;
  const int64_t chunk = blockIdx.x;
  const int k = set.k;
  KSH_PMARK(0, 0);
  if (chunk >= n_chunks && chunk - n_chunks >= int64_t(*n_extra)) return;  // a part nobody needed (uniform: no barrier yet)
  if (tid < 2 * kTgtBounds) s_b[tid] = rec[2 * kTgtBounds * chunk + tid];
  __syncthreads();
  KSH_PMARK(0, 1);  // the cuts have arrived
  const int64_t b = s_b[0];
  const int len_w = int(s_b[kTgtBounds] - b);
  const uint64_t v = uint64_t(s_b[1]);
  // first bucket of: the stream of top base a (where a v lies), the window (where v A lies)
  const auto first_bucket = [&](int r) {
    return int((r == 4 ? (v << 2) : ((uint64_t(r) << (2 * k - 2)) | v)) >> set.key_bits);
  };
  const int fb_w = first_bucket(4);
  if (tid < 5 * (kTgtSpan + 1)) {
    const int r = tid / (kTgtSpan + 1), j = tid % (kTgtSpan + 1);
    const int64_t bb = int64_t(first_bucket(r)) + j;
    const int64_t at = bb <= set.n_buckets ? set.off[bb] : set.n;
    const int64_t lo_r = r == 4 ? b : s_b[2 + r];
    const int64_t len_r = r == 4 ? int64_t(len_w) : s_b[kTgtBounds + 2 + r] - lo_r;
    int64_t rel = at - lo_r;
    rel = rel < 0 ? 0 : (rel > len_r ? len_r : rel);
    s_rel[r * (kTgtSpan + 1) + j] = int(rel);
  }
  // the window, and what reaches its k-mers' side 0 through a reverse complement: all loads before the first store
  KeyT wk[kRounds];
  uint32_t wr[kRounds];
#pragma unroll
  for (int u = 0; u < kRounds; u++) {
    const int i = tid + u * kTgtThreads;
    wr[u] = kNone;
    if (i < len_w) {
      wk[u] = set.keys[b + i];
      wr[u] = rc0[b + i];
    }
  }
#pragma unroll
  for (int u = 0; u < kRounds; u++) {
    const int i = tid + u * kTgtThreads;
    if (i < len_w) s_keys[i] = wk[u];
    s_slot[i] = wr[u];  // a mark starts as what reaches the side through a reverse complement: none -> the one -> several
  }
  KSH_PMARK(0, 2);  // this wave's window loads have arrived and are in LDS
  // the stream: four ranges one behind the other; position idx of the stream is index s_qbase[a] + idx of the set
  // and the (idx - s_cum[a])-th k-mer of range a.  (Per-range values are read from LDS by a: selected among four
  // registers they cost a nest of branches per k-mer.)
  __shared__ uint32_t s_qbase[4];
  __shared__ int s_cum[4], s_fb[4];
  const int c1 = int(s_b[kTgtBounds + 2] - s_b[2]), c2 = c1 + int(s_b[kTgtBounds + 3] - s_b[3]),
            c3 = c2 + int(s_b[kTgtBounds + 4] - s_b[4]);
  const int total = c3 + int(s_b[kTgtBounds + 5] - s_b[5]);
  if (tid < 4) {
    const int cum = tid == 0 ? 0 : (tid == 1 ? c1 : (tid == 2 ? c2 : c3));
    s_cum[tid] = cum;
    s_qbase[tid] = uint32_t(s_b[2 + tid]) - uint32_t(cum);  // (>= 0: the ranges ascend in the set)
    s_fb[tid] = first_bucket(tid);
  }
  const uint32_t b32 = uint32_t(b);
  const bool even_k = (k & 1) == 0;
  __syncthreads();
  KSH_PMARK(0, 3);
  const auto range_of = [&](int idx) { return (idx >= c1 ? 1 : 0) + (idx >= c2 ? 1 : 0) + (idx >= c3 ? 1 : 0); };
  // the next k-mer of this thread is requested before this one's search.  (Measured and dropped: TWO k-mers of the
  // stream per thread and turn, their searches step by step side by side -- the waves of this kernel wait 75 % of their
  // cycles and issue 16 %, `tools/pmc_stalls.sh` -- 891 / 78 / 794 us on the genome / difference / intersection sets
  // against 895 / 70 / 815: what the waves wait for is not the one search's chain of LDS reads alone.)
  KeyT key_next = 0;
  uint32_t rc_next = kNone;
  if (tid < total) {
    const uint32_t p = s_qbase[range_of(tid)] + uint32_t(tid);
    key_next = set.keys[p];
    rc_next = rc1[p];
  }
#pragma unroll 1
  for (int idx = tid; idx < total; idx += kTgtThreads) {
    const int a = range_of(idx);
    const int r = idx - s_cum[a];
    const uint32_t p32 = s_qbase[a] + uint32_t(idx);
    const KeyT key = key_next;
    const uint32_t through_rc = rc_next;
    if (idx + kTgtThreads < total) {
      const uint32_t p2 = s_qbase[range_of(idx + kTgtThreads)] + uint32_t(idx + kTgtThreads);
      key_next = set.keys[p2];
      rc_next = rc1[p2];
    }
    // its bucket: the range's first, or one of the next few
    int64_t my_b;
    {
      const int row = a * (kTgtSpan + 1);
      // (a step-halving search over the row -- four dependent reads of s_rel[row + j + step] -- does not get through
      // this compiler: "Illegal instruction detected: V_CMP_NE_U32_e32 0, $src_shared_base")
      int j = 0;
      while (j < kTgtSpan && s_rel[row + j + 1] <= r) j++;
      my_b = j < kTgtSpan ? int64_t(s_fb[a] + j) : set.bucket_of(int64_t(p32));
    }
    const uint64_t x = (uint64_t(my_b) << set.key_bits) | uint64_t(key);
    if (even_k && revcomp(x, k) == x) *self_rc = 1;
    const uint64_t g0 = kmer_next(x, k, 0);
    int cnt = 0;
    uint32_t single = kNone;
    const int jw = int(g0 >> set.key_bits) - fb_w;
    if (jw >= 0 && jw < kTgtSpan) {
      const int lo = s_rel[4 * (kTgtSpan + 1) + jw], hi = s_rel[4 * (kTgtSpan + 1) + jw + 1];
      const KeyT gkey = KeyT(g0 & set.key_mask());
      if (lo < hi) {
        int i = lds_lower_bound(s_keys, lo, hi, gkey);
        for (; i < hi && KeyT(s_keys[i] - gkey) < KeyT(4); i++) {  // (gkey ends in base A: no wrap below it)
          const uint32_t at = b32 + uint32_t(i);
          if (at == p32) continue;  // Next(x, c) == x
          cnt++;
          single = at << 1;
          mark_hit(&s_slot[i], p32 << 1);
        }
      }
    } else {
      // the window spans more buckets than the table holds (a sparse stretch of the set): found in global memory,
      // marked in the window all the same
      set.for_group4(g0, [&](int64_t at) {
        if (uint32_t(at) == p32) return;
        cnt++;
        single = uint32_t(at) << 1;
        const int64_t i = at - b;
        if (i >= 0 && i < len_w) mark_hit(&s_slot[i], p32 << 1);
      });
    }
    const uint32_t direct = cnt == 0 ? kNone : (cnt == 1 ? single : kMulti);
    nbr[2 * size_t(p32) + 1] = side_verdict(direct, through_rc);
  }
  KSH_PMARK(0, 4);  // the first wave's share of the stream
  __syncthreads();
  KSH_PMARK(0, 5);  // everybody's
#pragma unroll
  for (int u = 0; u < kRounds; u++) {
    const int i = tid + u * kTgtThreads;
    if (i < len_w) nbr[2 * (b + i)] = s_slot[i];
  }
  KSH_PMARK(0, 6);
#ifdef KSH_TRACE
  if (g_probe_trace && tid == 0 && chunk < g_probe_trace_rows) g_probe_trace[chunk * 16 + 7] = (unsigned long long)(total);
#endif
}

// (Measured and dropped, round 3: workgroups that STAY and take chunks in turn with only the next chunk's bound
// record on its way -- the light form of the software pipeline below, to take the first of a chunk's two
// round trips off its path: the loop's state does not fit the 64 vector registers that four workgroups per
// CU allow (168 bytes of scratch per lane), 5.1 ms per 10^8 against 1.86.)
// (Measured and dropped, round 3: a RUN of consecutive k-mers per thread -- their queries ascend in every window,
// so only a thread's first k-mer searches and the following ones step on from where the one before ended: a
// third of the LDS reads per k-mer with runs of four, half with runs of two.  The windows' LDS per k-mer stays
// what it is, so the threads per CU fall with the run: runs of four (128 threads per chunk, 14 waves per CU)
// 3.53 ms per 10^8 against 1.86, runs of two (28 waves) 2.33.  And the other way round, THREE PIVOTS per step
// read together -- half the dependent steps of a binary search for 1.5 x the reads: 2.04 ms.  One k-mer per
// thread, binary searches, four workgroups per CU is where this kernel's minimum is: neither the count of its
// LDS reads nor the length of their dependency chains alone is what it waits for.)
// (Measured and dropped, round 3: SLICE INDEXES over the five windows, as k_adj_rc has them -- extended keys
// (bucket within the range << key bits | key) cut into about one slice per key over the interval the chunk's
// queries can take, a query reads its slice's two bounds and a key or two instead of searching 9 .. 12 steps:
// a third of the LDS reads at random addresses, the strings the oracle's -- and 2.86 ms per 10^8 against 1.86.
// Putting the index together costs more than the searches it saves: every position fills the slices between
// its predecessor's and its own, a loop whose trip count is the longest gap among 64 lanes, ten times per
// thread, and a second barrier.)
// (Measured and dropped, round 3: ranges longer than their windows staged and searched in PARTS, one after
// the other, instead of their k-mers probing in global memory -- no lane of a wavefront waits for global
// round trips any more, a T-led chunk takes up to five staging rounds: 2.01 ms per 10^8 against 1.87, 197 us
// against 176 on a 9 x 10^6 difference set.  The fall-back probes are not what this kernel waits for
// either way.  profiles/r03_probe_stage_ab.txt)
// (Measured and dropped, round 3: the same kernel software-pipelined -- persistent workgroups that take
// chunks blockIdx.x, + gridDim.x, ..., with the windows, keys and marks of the next chunk on their way into
// registers and the bound record of the one after it behind them while this chunk is searched, and a Next
// window twice as large.  The data in flight cost 22 more vector registers (80: three workgroups per CU
// instead of four, and a few spills), and the kernel took 2.34 ms per 10^8 k-mers against 1.93, 207 us
// against 181 on a 9 x 10^6 difference set: four short-lived workgroups per CU in different phases overlap
// their round trips better than three long-lived ones that prefetch.  profiles/r03_fwd_pipe_ab.txt)
// The forward half of the probe, in place, and the verdict per side: rc0 / rc1 are what k_adj_rc
// found to reach the k-mer's side 0 / side 1 through a reverse complement.
template <typename KeyT>
__global__ __launch_bounds__(256) void k_adj_fwd(DevSet<KeyT> set, const uint32_t* __restrict__ rc0,
                                                  const uint32_t* __restrict__ rc1,
                                                  uint32_t* __restrict__ nbr, int* __restrict__ self_rc) {
  __shared__ int64_t s_bucket[2];
  const int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  const uint64_t x = set.kmer_in_block(t, s_bucket);
  if (t >= set.n) return;
  const int k = set.k;
  if (revcomp(x, k) == x) *self_rc = 1;
  int cnt[2] = {0, 0};
  uint32_t single[2] = {kNone, kNone};
  // side 1: Next(x, .), neighbour as is
  set.for_group4(kmer_next(x, k, 0), [&](int64_t idx) {
    if (idx == t) return;
    cnt[1]++;
    single[1] = uint32_t(idx) << 1;
  });
  // side 0: Prev(x, c), neighbour as is
#pragma unroll
  for (int c = 0; c < 4; c++) {
    const uint64_t z = kmer_prev(x, k, c);
    if (revcomp(z, k) < z) continue;
    if (z == x) continue;
    const int64_t idx = set.find(z);
    if (idx < 0) continue;
    cnt[0]++;
    single[0] = uint32_t(idx) << 1;
  }
  const uint32_t r[2] = {rc0[t], rc1[t]};
  uint32_t out[2];
#pragma unroll
  for (int side = 0; side < 2; side++) {
    const int total = cnt[side] + (r[side] == kNone ? 0 : (r[side] == kMulti ? 2 : 1));
    out[side] = total == 0 ? kNone : (total > 1 ? kMulti : (cnt[side] == 1 ? single[side] : r[side]));
  }
  reinterpret_cast<uint2*>(nbr)[t] = make_uint2(out[0], out[1]);
}

// ---------------------------------------------------------------------------------- E2
// Chains of states are ranked with a sparse ruler set instead of one serial walk per
// chain (a 10^7-k-mer unitig would otherwise be a 10^7-step dependent walk):
//   rulers = both states of every 32nd k-mer (kRulerEvery; index order is unrelated to chain order);
//   k_ruler_walk  every sampled ruler walks to the next one (about 32 steps), stamping the
//                 states it passes with (ruler, offset);  k_ruler_heads does the same for the
//                 head segment of a chain that starts between two sampled k-mers;
//   k_ruler_jump  pointer jumping over the dense ruler array only (1/32 of the states), until
//                 each ruler holds (chain end, distance to it);
//   k_choose      resolves every state through its ruler: (end state, distance to end).
// Rulers on a non-branching loop never reach an end; they and their segments stay unset
// and k_loops handles the loop.
// Both links of the k-mer of state s in one 8-byte load: .x = link[2t], .y = link[2t+1].
__device__ __forceinline__ uint2 link_pair(const uint32_t* __restrict__ link, uint32_t s) {
  return reinterpret_cast<const uint2*>(link)[s >> 1];
}
__device__ __forceinline__ uint32_t leave_link(uint2 pr, uint32_t s) { return (s & 1) ? pr.x : pr.y; }
__device__ __forceinline__ uint32_t enter_link(uint2 pr, uint32_t s) { return (s & 1) ? pr.y : pr.x; }
__device__ __forceinline__ uint32_t step_to(uint32_t s, uint32_t lk) {
  return ((lk >> 1) << 1) | ((s & 1) ^ (lk & 1));
}

// Sampled rulers are both states of every 32nd k-mer (index order is unrelated to chain
// order, so this is as good as a hash), which makes them enumerable without compaction:
// dense thread i <-> state 64 * (i >> 1) + (i & 1).  The other rulers are chain starts and
// chain ends; "is a ruler" needs only the state's own link pair, which the walk loads anyway.
__device__ __forceinline__ bool sampled_ruler(uint32_t s) { return (s & (2u * kRulerEvery - 2u)) == 0; }

// Chain-rank records.
//   rinfo[i] (one per sampled ruler, dense index i): end_flag:1 | dist:31 | next:32 -- the next
//            ruler (or, once end_flag is set, the chain's end state) and the distance to it.
//   rec[s]   (one per state; only the d == 0 state of a k-mer is written, the other one follows
//            from it, see mirror_rec and k_ruler_heads): kind:2 | off:30 | ref:32
//            kind 0: s lies `off` steps after sampled ruler `ref` (dense index)
//            kind 1: s lies `off` steps before sampled ruler `ref` (only as the mirror of kind 0)
//            kind 2: s is a chain of its own (one state), written for both states of the k-mer
//            kind 3: s lies `off` steps after the chain start `ref` (a state); what lies ahead of
//                    that start is in chain_info (k_ruler_heads)
// The dense ruler array is 1/32 of the states (L2/MALL resident at 10^8 k-mers), so pointer
// jumping and the final lookups stay on-chip; every state's link pair is read once and its
// record written once.
constexpr uint64_t kRecUnset = ~uint64_t(0);
constexpr uint64_t kEndFlag = uint64_t(1) << 63;

__device__ __forceinline__ uint32_t dense_index(uint32_t s) { return ((s >> (kRulerShift + 1)) << 1) | (s & 1); }
__device__ __forceinline__ uint64_t make_rec(uint32_t kind, uint32_t off, uint32_t ref) {
  return (uint64_t(kind) << 62) | (uint64_t(off & 0x3FFFFFFFu) << 32) | ref;
}
__device__ __forceinline__ uint64_t make_rinfo(bool end, uint32_t dist, uint32_t nx) {
  return (end ? kEndFlag : 0) | (uint64_t(dist & 0x7FFFFFFFu) << 32) | nx;
}

// One thread per sampled ruler (dense index i <-> state 64 * (i >> 1) + (i & 1)).
__global__ __launch_bounds__(256) void k_ruler_walk(const uint32_t* __restrict__ link,
                                                     int64_t n_states, int64_t n_dense,
                                                     unsigned long long* __restrict__ rinfo,
                                                     unsigned long long* __restrict__ rec) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n_dense) return;
  const int64_t s64 = 2 * int64_t(kRulerEvery) * (i >> 1) + (i & 1);
  if (s64 >= n_states) {
    rinfo[i] = make_rinfo(true, 0, 0);
    return;
  }
  const uint32_t r = uint32_t(s64);
  uint32_t lk = leave_link(link_pair(link, r), r);
  if ((r & 1) == 0) rec[r] = make_rec(0, 0, uint32_t(i));
  if (lk == kNone) {
    rinfo[i] = make_rinfo(true, 0, r);
    return;
  }
  uint32_t cur = r, steps = 0;
  while (true) {
    cur = step_to(cur, lk);
    steps++;
    if (sampled_ruler(cur) || steps >= 0x3FFFFFFFu) {
      rinfo[i] = make_rinfo(false, steps, cur);
      return;
    }
    if ((cur & 1) == 0) rec[cur] = make_rec(0, steps, uint32_t(i));  // see mirror_rec
    lk = leave_link(link_pair(link, cur), cur);
    if (lk == kNone) {
      rinfo[i] = make_rinfo(true, steps, cur);
      return;
    }
  }
}

// One thread per end k-mer; acts on those of its two states that start a chain without being a
// sampled ruler (nothing links into them).  One walk from the start S to the first sampled ruler ahead,
// or to the chain's end: the d == 0 states it passes are stamped "off steps after start S" (kind
// 3), and what lies ahead goes into chain_info at S's k-mer (a k-mer starts at most one chain of
// two or more states: its other state then enters through the side that has the link).  Both
// states of the stamped k-mer follow from that record: its forward state is off steps into the
// chain, its mirror state off steps before the mirror chain's end S ^ 1.
//   chain_info: ahead:1 (0 = the chain's end state, 1 = a sampled ruler, dense index) | steps:31 | ref:32
__device__ __forceinline__ uint64_t make_chain_info(bool ruler_ahead, uint32_t steps, uint32_t ref) {
  return (ruler_ahead ? kEndFlag : 0) | (uint64_t(steps & 0x7FFFFFFFu) << 32) | ref;
}

__global__ __launch_bounds__(256) void k_ruler_heads(const uint32_t* __restrict__ link,
                                                      const uint32_t* __restrict__ ends, int64_t n_ends,
                                                      unsigned long long* __restrict__ rec,
                                                      unsigned long long* __restrict__ chain_info) {
  const int64_t e = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (e >= n_ends) return;
  const int64_t t = ends[e];
  if ((t & (kRulerEvery - 1)) == 0) return;  // a sampled ruler: k_ruler_walk
  // state 2t enters through side 0, state 2t + 1 through side 1: a start when nothing links in
  const uint2 own = reinterpret_cast<const uint2*>(link)[t];
  const uint32_t flags = (own.x == kNone ? 1u : 0u) | (own.y == kNone ? 2u : 0u);
  if (!flags) return;
  for (uint32_t d = 0; d < 2; d++) {
    if (!(flags & (1u << d))) continue;
    const uint32_t s0 = uint32_t(2 * t) | d;
    uint32_t lk = leave_link(link_pair(link, s0), s0);
    if (lk == kNone) {
      rec[s0] = make_rec(2, 0, s0);  // a one-state chain: its own end, in both orientations
      continue;
    }
    uint32_t cur = s0, off = 0;
    while (true) {
      if ((cur & 1) == 0) rec[cur] = make_rec(3, off, s0);
      cur = step_to(cur, lk);
      off++;
      if (sampled_ruler(cur)) {
        chain_info[t] = make_chain_info(true, off, dense_index(cur));
        break;
      }
      lk = leave_link(link_pair(link, cur), cur);
      if (lk == kNone || off >= 0x3FFFFFFFu) {
        if ((cur & 1) == 0) rec[cur] = make_rec(3, off, s0);
        chain_info[t] = make_chain_info(false, off, cur);
        break;
      }
    }
  }
}

// ---- the same walks without stamps (see k_choose_ends), each stretch walked once where it can be.
// A stretch between two stops of a chain (sampled rulers, or the chain's ends) is the business of
// two walks, one from either side, that pass the same k-mers and count the same steps: the walk
// from u that arrives at v is the mirror image of the walk from v ^ 1 that arrives at u ^ 1.  With
// nothing to stamp, one of them is enough: a walk writes its own record and the record of its mirror
// image.  Neither end knows the other beforehand, so the walks go in two phases: half of the
// starts (by a hash bit) in the first, and in the second only those whose record no mirror image
// has filled in -- three walks for two stretches on average instead of four.
//   Records start unset (all ones: no record looks like that).  Equal values may be written twice.
//   Within a phase a walk that starts after its mirror image has arrived finds its record filled in and
//   does not start (round 3): the walkers of a phase start over the kernel's whole duration, a walk lasts
//   a few tens of microseconds, so of two mirror images in the same phase the later one nearly always
//   sees the earlier one's record -- one walk per stretch and a few per cent, not three for two.  The
//   records of mirror images are stored and looked at with device scope (another XCD's L2 may hold the
//   line from before the store); a stale look costs a walk, not a wrong record.
__device__ __forceinline__ unsigned long long rec_look(const unsigned long long* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void rec_post(unsigned long long* p, unsigned long long v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool first_phase(uint32_t s) {
  return (((s >> 1) * 0x9E3779B1u) >> 13 ^ s) & 1u;
}

// What a ranking walk leaves behind for the writing of the strings (k_emit_log_*): where it arrived
// after how many steps, and the states it passed K, 2K, ... steps in.  A k-mer spells the K bases
// from its slot on, so the k-mers at steps 0, K, 2K, ... and the one at the arrival spell the
// whole stretch between them (in either orientation of the string: a k-mer covers the same K
// slots read forwards or as its reverse complement) -- no second walk, a handful of key reads per
// stretch instead of one link read per k-mer.  One walk per stretch is enough here too (whichever
// of the two mirror images ran).
struct WalkLog {
  unsigned long long* hdr;  // per walker: steps:32 | arrival state:32; all ones = it did not walk.  NULL: no logs
  uint32_t* mid;            // mid[j * n_walkers + w] = the state (j + 1) * K steps into walk w
  int64_t n_walkers;
  int n_mid;                // entries kept per walker; a longer stretch is walked again when it is written
  int k;
  // the stretches longer than that (one in a few hundred): ruler walkers by dense index, head
  // walkers by end-list index | long_tag
  uint32_t* long_walkers;
  unsigned int* long_count;
  uint32_t long_tag;
  __device__ __forceinline__ bool is_long(uint32_t steps) const { return steps > uint32_t(n_mid + 1) * uint32_t(k); }
  __device__ __forceinline__ void arrived(int64_t w, uint32_t steps, uint32_t state) const {
    if (!hdr) return;
    hdr[w] = (uint64_t(steps) << 32) | state;
    if (is_long(steps)) long_walkers[atomicAdd(long_count, 1u)] = uint32_t(w) | long_tag;
  }
  __device__ __forceinline__ void passed(int64_t w, int j, uint32_t state) const {
    if (hdr && j < n_mid) mid[int64_t(j) * n_walkers + w] = state;
  }
};
constexpr int kLogMidRulers = 7, kLogMidHeads = 7;

template <int kPhase>
__global__ __launch_bounds__(256) void k_rank_walk(const uint32_t* __restrict__ link, int64_t n_states,
                                                    int64_t n_dense, unsigned long long* __restrict__ rinfo,
                                                    unsigned long long* __restrict__ chain_info, WalkLog log,
                                                    unsigned long long* __restrict__ unset_a,
                                                    unsigned long long* __restrict__ unset_b) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n_dense) return;
  if (unset_a) {  // the two-level jumping's per-ruler arrays start unset (a fill of their own is a launch each)
    unset_a[i] = ~0ull;
    unset_b[i] = ~0ull;
  }
  const int64_t s64 = 2 * int64_t(kRulerEvery) * (i >> 1) + (i & 1);
  if (s64 >= n_states) {
    rinfo[i] = make_rinfo(true, 0, 0);
    return;
  }
  const uint32_t r = uint32_t(s64);
  if (kPhase == 1 && !first_phase(r)) return;
  if (rec_look(rinfo + i) != kRecUnset) return;
  const uint2 own = link_pair(link, r);
  uint32_t lk = leave_link(own, r);
  if (lk == kNone) {
    rec_post(rinfo + i, make_rinfo(true, 0, r));
    if (enter_link(own, r) == kNone) log.arrived(i, 0, r);  // a k-mer on its own: no walk arrives at it
    return;
  }
  uint32_t cur = r, steps = 0;
  int since = 0, n_passed = 0;
  while (true) {
    cur = step_to(cur, lk);
    steps++;
    if (sampled_ruler(cur) || steps >= 0x3FFFFFFFu) {
      // (its own record too with an agent-scope store: its mirror image posts the same word in the same launch,
      // and other walkers look at it)
      rec_post(rinfo + i, make_rinfo(false, steps, cur));
      rec_post(rinfo + dense_index(cur ^ 1), make_rinfo(false, steps, r ^ 1));
      log.arrived(i, steps, cur);
      return;
    }
    if (++since == log.k) {
      since = 0;
      log.passed(i, n_passed++, cur);
    }
    // (measured and dropped: another look at the own record every eighth step, to give up a walk whose mirror
    // image arrives while it is under way -- 2.40 against 2.41 ms)
    lk = leave_link(link_pair(link, cur), cur);
    if (lk == kNone) {
      rec_post(rinfo + i, make_rinfo(true, steps, cur));
      // the chain that starts at cur ^ 1 (an unsampled k-mer) has ruler r ^ 1 ahead of it
      rec_post(chain_info + (cur >> 1), make_chain_info(true, steps, dense_index(r ^ 1)));
      log.arrived(i, steps, cur);
      return;
    }
  }
}

// (Measured and dropped, round 3: the one-launch walk by waves that stay -- a wave owns a contiguous slice of the
// walkers and a lane that has arrived takes the slice's next one, so that a wave's reads per round trip stay 64
// instead of falling off towards its longest walker: 3.02 ms per 10^8 k-mers against 2.41 for a thread per
// walker, 2.72 with a quarter of the waves; the emit after it took 0.83 against 0.71 ms, i.e. more mirror
// images had both walked: a turn of a full wave waits for the slowest of 64 random reads, every walk takes
// longer, and more of them overlap their mirror image.)
template <int kPhase>
__global__ __launch_bounds__(256) void k_rank_heads(const uint32_t* __restrict__ link,
                                                     const uint32_t* __restrict__ ends, int64_t n_ends,
                                                     unsigned long long* __restrict__ rinfo,
                                                     unsigned long long* __restrict__ chain_info, WalkLog log) {
  const int64_t e = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (e >= n_ends) return;
  const uint32_t t = ends[e];
  if ((t & (kRulerEvery - 1)) == 0) return;  // a sampled ruler: k_rank_walk
  const uint2 own = reinterpret_cast<const uint2*>(link)[t];
  // the one state of t that starts a chain of two or more states, if any
  uint32_t s0;
  if (own.x == kNone && own.y != kNone) {
    s0 = 2 * t;
  } else if (own.y == kNone && own.x != kNone) {
    s0 = 2 * t + 1;
  } else {
    log.arrived(e, 0, 2 * t);  // a k-mer on its own: no walk arrives at it (either phase: the first may be left out)
    return;
  }
  if (kPhase == 1 && !first_phase(s0)) return;
  if (rec_look(chain_info + t) != kRecUnset) return;
  uint32_t lk = leave_link(own, s0);
  uint32_t cur = s0, off = 0;
  int since = 0, n_passed = 0;
  while (true) {
    cur = step_to(cur, lk);
    off++;
    if (sampled_ruler(cur)) {
      rec_post(chain_info + t, make_chain_info(true, off, dense_index(cur)));
      rec_post(rinfo + dense_index(cur ^ 1), make_rinfo(true, off, s0 ^ 1));  // its walk ends at the chain end s0 ^ 1
      log.arrived(e, off, cur);
      return;
    }
    if (++since == log.k) {
      since = 0;
      log.passed(e, n_passed++, cur);
    }
    lk = leave_link(link_pair(link, cur), cur);
    if (lk == kNone || off >= 0x3FFFFFFFu) {
      rec_post(chain_info + t, make_chain_info(false, off, cur));
      rec_post(chain_info + (cur >> 1), make_chain_info(false, off, s0 ^ 1));  // the mirror chain, from cur ^ 1 to s0 ^ 1
      log.arrived(e, off, cur);
      return;
    }
  }
}

// The records of the chain starts unset (k_rank_heads, phase 2).
__global__ __launch_bounds__(256) void k_rank_unset(const uint32_t* __restrict__ ends, int64_t n_ends,
                                                     unsigned long long* __restrict__ chain_info) {
  const int64_t e = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (e < n_ends) chain_info[ends[e]] = kRecUnset;
}

// Pointer jumping over the dense ruler array until every ruler on a path points at its end.
// (Rounds are launched back to back without a host round trip: a round whose predecessor changed
// nothing -- *prev == 0, one scalar load -- returns at once and leaves its own flag clear, so all
// later rounds do too.)
__global__ __launch_bounds__(256) void k_ruler_jump(int64_t n_dense,
                                                     unsigned long long* __restrict__ rinfo,
                                                     const int* __restrict__ prev,
                                                     int* __restrict__ changed) {
  if (prev && *prev == 0) return;
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n_dense) return;
  uint64_t mine = rinfo[i];
  if (mine & kEndFlag) return;
  // kJumpHops hops per launch (the rounds are launch-bound: a record's reach grows at least (kJumpHops + 1)-fold
  // per launch, so log5 instead of log2 of the longest chain many launches); any snapshot of a record keeps
  // the invariant
#pragma unroll
  for (int hop = 0; hop < kJumpHops && !(mine & kEndFlag); hop++) {
    const uint64_t theirs = rinfo[dense_index(uint32_t(mine))];
    const uint32_t dist = uint32_t((mine >> 32) & 0x7FFFFFFFu) + uint32_t((theirs >> 32) & 0x7FFFFFFFu);
    mine = (theirs & kEndFlag) | (uint64_t(dist & 0x7FFFFFFFu) << 32) | uint32_t(theirs);
  }
  rinfo[i] = mine;
  *changed = 1;
}

// ---- the same result in two levels (ranking without stamps): pointer jumping costs a cache-missing
// read per ruler per round (the ruler records of a 10^8-k-mer set are 50 MB) for log2(rulers per
// chain) rounds -- 24 for a genome.  The ruler records form linked lists themselves, so they are
// ranked the way the k-mers are: every 16th sampled k-mer is a level-2 ruler, the level-2 rulers
// and the first ruler of every chain (the one whose mirror image reaches a chain end without
// meeting a ruler: nothing comes before it) walk along the records to the next level-2 ruler or
// the end, stamping what they pass with (walker, distance so far); pointer jumping then runs over
// the level-2 rulers only (1/16 of the records, L2-resident), and k_l2_resolve gives every record
// its end and distance through its stamp.  Records on a loop of rulers stay without an end.
// (every 16th: measured per 10^8-k-mer genome / 9 x 10^7 intersection set, the four k_l2_* kernels together: every
// 32nd 462 / 371 us -- 2 x 10^5 walkers of 32 dependent steps do not fill the GPU --, every 16th 392 / 353, every 8th
// 393 / 340: the jumping rounds over more level-2 records take back what the shorter walks give)
#ifndef KSH_L2_SHIFT
#define KSH_L2_SHIFT 4
#endif
constexpr int kL2Shift = KSH_L2_SHIFT;
__device__ __forceinline__ bool is_level2(int64_t i) { return ((i >> 1) & ((1 << kL2Shift) - 1)) == 0; }
__device__ __forceinline__ int64_t level2_index(int64_t i) { return ((i >> (kL2Shift + 1)) << 1) | (i & 1); }
__device__ __forceinline__ int64_t level2_entry(int64_t j) { return ((j >> 1) << (kL2Shift + 1)) | (j & 1); }

// One thread per ruler record; the level-2 rulers and the chain heads walk.
//   r2[j]    (level-2 ruler j): end:1 | dist:31 | ref:32 -- the next level-2 ruler (record index) or the end state
//   head[i]  (chain head i, not level 2): the same
//   stamp[e] (everything they pass): dist:32 | walker record:32
// (two launches: the level-2 rulers, one thread each -- they are every 16th pair of records and all of
// them walk some 16 steps -- and the chain heads, found by a thread per record)
template <bool kLevel2>
__global__ __launch_bounds__(256) void k_l2_walk(const unsigned long long* __restrict__ rinfo, int64_t n_dense,
                                                  unsigned long long* __restrict__ r2,
                                                  unsigned long long* __restrict__ head,
                                                  unsigned long long* __restrict__ stamp) {
  const int64_t at = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  const int64_t i = kLevel2 ? level2_entry(at) : at;
  if (i >= n_dense) return;
  const bool l2 = kLevel2;
  if (!kLevel2 && (is_level2(i) || !(rinfo[i ^ 1] & kEndFlag))) return;  // level 2, or something comes before it: that walker passes it
  uint64_t dist = 0;
  int64_t cur = i;
  uint64_t out;
  for (int64_t steps = 0;; steps++) {
    const uint64_t ri = rinfo[cur];
    dist += (ri >> 32) & 0x7FFFFFFFu;
    if (ri & kEndFlag) {
      out = kEndFlag | ((dist & 0x7FFFFFFFu) << 32) | uint32_t(ri);
      break;
    }
    cur = dense_index(uint32_t(ri));
    if (is_level2(cur) || steps > n_dense) {
      out = ((dist & 0x7FFFFFFFu) << 32) | uint32_t(cur);
      break;
    }
    stamp[cur] = (dist << 32) | uint32_t(i);
  }
  if (l2) r2[level2_index(i)] = out; else head[i] = out;
}

// Pointer jumping over the level-2 rulers.
__global__ __launch_bounds__(256) void k_l2_jump(int64_t n_l2, unsigned long long* __restrict__ r2,
                                                  const int* __restrict__ prev, int* __restrict__ changed) {
  if (prev && *prev == 0) return;
  const int64_t j = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (j >= n_l2) return;
  uint64_t mine = r2[j];
  if (mine & kEndFlag) return;
#pragma unroll
  for (int hop = 0; hop < kJumpHops && !(mine & kEndFlag); hop++) {  // (kJumpHops hops per launch, see k_ruler_jump)
    const uint64_t theirs = r2[level2_index(int64_t(uint32_t(mine)))];
    const uint32_t dist = uint32_t((mine >> 32) & 0x7FFFFFFFu) + uint32_t((theirs >> 32) & 0x7FFFFFFFu);
    mine = (theirs & kEndFlag) | (uint64_t(dist & 0x7FFFFFFFu) << 32) | uint32_t(theirs);
  }
  r2[j] = mine;
  *changed = 1;
}

// (end, distance) of a walker of k_l2_walk once the level-2 rulers have theirs; no end flag on a loop.
__device__ __forceinline__ uint64_t l2_walker_end(int64_t w, const unsigned long long* __restrict__ r2,
                                                  const unsigned long long* __restrict__ head) {
  if (is_level2(w)) return r2[level2_index(w)];
  const uint64_t h = head[w];
  if (h & kEndFlag) return h;
  const uint64_t ahead = r2[level2_index(int64_t(uint32_t(h)))];
  const uint32_t dist = uint32_t((h >> 32) & 0x7FFFFFFFu) + uint32_t((ahead >> 32) & 0x7FFFFFFFu);
  return (ahead & kEndFlag) | (uint64_t(dist & 0x7FFFFFFFu) << 32) | uint32_t(ahead);
}

__global__ __launch_bounds__(256) void k_l2_resolve(int64_t n_dense, const unsigned long long* __restrict__ r2,
                                                     const unsigned long long* __restrict__ head,
                                                     const unsigned long long* __restrict__ stamp,
                                                     unsigned long long* __restrict__ rinfo) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n_dense) return;
  const uint64_t mine = rinfo[i];
  if (mine & kEndFlag) return;  // reaches its end without meeting a ruler
  if (is_level2(i) || head[i] != kRecUnset) {  // (head and stamp start unset)
    rinfo[i] = l2_walker_end(i, r2, head);
    return;
  }
  const uint64_t st = stamp[i];
  if (st == kRecUnset) return;  // a loop of rulers without a level-2 ruler: stays without an end
  const uint64_t we = l2_walker_end(int64_t(uint32_t(st)), r2, head);
  const uint32_t dist = uint32_t((we >> 32) & 0x7FFFFFFFu) - uint32_t(st >> 32);
  rinfo[i] = (we & kEndFlag) | (uint64_t(dist & 0x7FFFFFFFu) << 32) | uint32_t(we);
}

// A ruler segment is walked twice, once in each direction (the two states of a k-mer lie on
// mirror-image chains), and the random 8-byte record writes of those walks are what bounds them
// (31 G random writes/s against 55 G reads/s).  So a walk only stamps the states with d == 0:
// every k-mer is stamped once, by whichever walk passes its state 2t, and the record of state
// 2t + 1 follows from it -- `off` steps after ruler state R means `off` steps before R's mirror
// R ^ 1 (the two states of a sampled k-mer are neighbours in the dense ruler array), and the
// other way round.  Head segments (kind 3, k_ruler_heads) follow the same rule.
__device__ __forceinline__ uint64_t mirror_rec(uint64_t r) { return r ^ ((uint64_t(1) << 62) | 1u); }

// (end state, distance to it) of a state from its record; false on a non-branching loop.
__device__ __forceinline__ bool resolve_rec(uint64_t r, const unsigned long long* __restrict__ rinfo,
                                            uint32_t* end, uint32_t* dist) {
  if (r == kRecUnset) return false;
  const uint32_t kind = uint32_t(r >> 62), off = uint32_t((r >> 32) & 0x3FFFFFFFu), ref = uint32_t(r);
  if (kind == 2) {
    *end = ref;
    *dist = off;
    return true;
  }
  const uint64_t ri = rinfo[ref];
  if (!(ri & kEndFlag)) return false;  // the ruler never reached an end: it is on a loop
  const uint32_t rd = uint32_t((ri >> 32) & 0x7FFFFFFFu);
  *end = uint32_t(ri);
  *dist = kind == 0 ? rd - off : rd + off;
  return true;
}

// For k-mer t the chain of (t, 0) ends at E0 and the chain of (t, 1) ends at E1, i.e. the
// forward chain runs from k-mer E1 >> 1 to k-mer E0 >> 1; the spelling starts at the larger
// end (spss.h:511,555).  directed (non-canonical sets): every chain is spelled forward from its
// start k-mer, and all heads are one class, in start-k-mer order (spss.h:159-199).
constexpr int kLenSums = 32;  // partial sums of k_choose_ends
struct Chosen {
  uint32_t head_state;  // first state of the unitig in head-first order
  uint32_t p;           // the k-mer's position in that order
  uint32_t len;         // k-mers in the unitig
  uint32_t d;           // the k-mer is spelled reverse-complemented in that order
  uint32_t last;        // last state in that order
  uint8_t cls;          // class of the head (k_head_block_counts)
};
__device__ __forceinline__ Chosen choose_from_ends(uint32_t e0, uint32_t d0, uint32_t e1, uint32_t d1,
                                                   bool directed) {
  const uint32_t fwd_start = e1 >> 1, fwd_end = e0 >> 1;
  Chosen c;
  c.d = (directed || fwd_start >= fwd_end) ? 0u : 1u;
  c.head_state = (c.d ? e0 : e1) ^ 1;
  c.p = c.d ? d0 : d1;
  c.len = d0 + d1 + 1;
  c.last = c.d ? e1 : e0;
  c.cls = (directed || fwd_start == fwd_end) ? uint8_t(0) : uint8_t((c.head_state & 1) == 0 ? 1 : 2);
  return c;
}

// ori[t]: bit 0 = the k-mer is spelled reverse-complemented in its unitig's head-first order; bits
// 1..4 = the low four bits of its distance to the unitig's far end (len - 1 - pos), which is its
// position when the path cover traverses the unitig the other way round: k_emit picks its writers
// from pos and these bits without looking the unitig up.
__device__ __forceinline__ uint8_t ori_byte(uint32_t d, uint32_t to_far_end) {
  return uint8_t((d & 1u) | ((to_far_end & 15u) << 1));
}

// For k-mer t the chain of (t, 0) ends at E0 and the chain of (t, 1) ends at E1, i.e. the
// forward chain runs from k-mer E1 >> 1 to k-mer E0 >> 1; the spelling starts at the larger
// end (spss.h:511,555).  hcls: 0xFE marks a k-mer on a non-branching loop (k_loops fills it in).
// directed (non-canonical sets): every chain is spelled forward from its start k-mer, and all
// heads are one class, in start-k-mer order (spss.h:159-199).
__global__ __launch_bounds__(256) void k_choose(const unsigned long long* __restrict__ rec,
                                                 const unsigned long long* __restrict__ rinfo,
                                                 const unsigned long long* __restrict__ chain_info,
                                                 int64_t n, bool directed, uint32_t* __restrict__ head,
                                                 uint32_t* __restrict__ pos,
                                                 uint8_t* __restrict__ ori,
                                                 uint8_t* __restrict__ hcls,
                                                 uint32_t* __restrict__ hlen,
                                                 uint32_t* __restrict__ hlast) {
  const int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= n) return;
  ulonglong2 both = reinterpret_cast<const ulonglong2*>(rec)[t];
  uint32_t e0, d0, e1, d1;
  bool ok;
  if (both.x != kRecUnset && (both.x >> 62) == 3) {
    // off steps after chain start S: the forward state goes on to what lies ahead of S's walk,
    // the mirror state is off steps before the mirror chain's end S ^ 1
    const uint32_t off = uint32_t((both.x >> 32) & 0x3FFFFFFFu), start = uint32_t(both.x);
    const uint64_t ci = chain_info[start >> 1];
    const uint32_t steps = uint32_t((ci >> 32) & 0x7FFFFFFFu);
    ok = true;
    if (ci & kEndFlag) {  // a sampled ruler ahead
      const uint64_t ri = rinfo[uint32_t(ci)];
      ok = (ri & kEndFlag) != 0;  // a chain with a start is no loop: always set
      e0 = uint32_t(ri);
      d0 = steps - off + uint32_t((ri >> 32) & 0x7FFFFFFFu);
    } else {
      e0 = uint32_t(ci);
      d0 = steps - off;
    }
    e1 = start ^ 1;
    d1 = off;
  } else {
    if (both.y == kRecUnset && both.x != kRecUnset && (both.x >> 62) < 2) both.y = mirror_rec(both.x);
    ok = resolve_rec(both.x, rinfo, &e0, &d0) && resolve_rec(both.y, rinfo, &e1, &d1);
  }
  if (!ok) {
    head[t] = kNone;
    hcls[t] = 0xFE;
    return;
  }
  const Chosen c = choose_from_ends(e0, d0, e1, d1, directed);
  head[t] = c.head_state >> 1;
  pos[t] = c.p;
  ori[t] = ori_byte(c.d, c.len - 1 - c.p);
  hcls[t] = c.p == 0 ? c.cls : uint8_t(0xFF);
  if (c.p == 0) {
    hlen[t] = c.len;
    hlast[t] = c.last;
  }
}

// ---- ranking without stamps.  Of the per-k-mer results of k_choose only those of the unitigs'
// end k-mers are read before the strings are written (ids and lengths of the unitigs, the k-mers
// at their ends for the edges of the path cover), and the writing itself can walk the chains once
// more, this time in string order only (k_emit_rulers / k_emit_heads).  So the ranking walks need
// not stamp the states they pass -- the random 8-byte record write per k-mer that costs more than
// the two link reads -- and k_choose_ends looks at the k-mers with a missing link alone: such a
// k-mer ends one chain and starts its mirror image, so both of its states resolve from the
// chain-start record or the ruler record of the k-mer itself.
//   A non-branching loop has no end k-mer.  Loops with a sampled ruler show as rulers that never
// reach an end (*loop_flag), loops without one as k-mers that no unitig accounts for (*total_len
// != n): either way the encoder falls back to the stamping walks and k_loops.
__device__ __forceinline__ bool chain_end_of(uint32_t s, uint2 own, const unsigned long long* __restrict__ rinfo,
                                             const unsigned long long* __restrict__ chain_info,
                                             uint32_t* end, uint32_t* dist) {
  if (leave_link(own, s) == kNone) {
    *end = s;
    *dist = 0;
    return true;
  }
  if (sampled_ruler(s)) {
    const uint64_t ri = rinfo[dense_index(s)];
    *end = uint32_t(ri);
    *dist = uint32_t((ri >> 32) & 0x7FFFFFFFu);
    return (ri & kEndFlag) != 0;
  }
  const uint64_t ci = chain_info[s >> 1];  // s starts its chain (the callers ask for nothing else)
  const uint32_t steps = uint32_t((ci >> 32) & 0x7FFFFFFFu);
  if (!(ci & kEndFlag)) {
    *end = uint32_t(ci);
    *dist = steps;
    return true;
  }
  const uint64_t ri = rinfo[uint32_t(ci)];
  *end = uint32_t(ri);
  *dist = steps + uint32_t((ri >> 32) & 0x7FFFFFFFu);
  return (ri & kEndFlag) != 0;
}

// Both chain ends of a k-mer that is sampled or has a missing link; false on a loop.
__device__ __forceinline__ bool choose_at(uint32_t t, uint2 own, const unsigned long long* __restrict__ rinfo,
                                          const unsigned long long* __restrict__ chain_info, bool directed,
                                          Chosen* c) {
  uint32_t e0, d0, e1, d1;
  const bool ok0 = chain_end_of(2 * t, own, rinfo, chain_info, &e0, &d0);
  const bool ok1 = chain_end_of(2 * t + 1, own, rinfo, chain_info, &e1, &d1);
  if (!(ok0 && ok1)) return false;
  *c = choose_from_ends(e0, d0, e1, d1, directed);
  return true;
}

__global__ __launch_bounds__(256) void k_choose_ends(const uint32_t* __restrict__ link,
                                                      const uint32_t* __restrict__ ends, int64_t n_ends,
                                                      const unsigned long long* __restrict__ rinfo,
                                                      const unsigned long long* __restrict__ chain_info,
                                                      bool directed, uint32_t* __restrict__ head,
                                                      uint8_t* __restrict__ ori, uint8_t* __restrict__ hcls,
                                                      uint32_t* __restrict__ hlen, uint32_t* __restrict__ hlast,
                                                      unsigned long long* __restrict__ len_sums,
                                                      int* __restrict__ loop_flag) {
  __shared__ unsigned long long s_len;
  if (threadIdx.x == 0) s_len = 0;
  __syncthreads();
  const int64_t e = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (e < n_ends) {
    const uint32_t t = ends[e];
    const uint2 own = reinterpret_cast<const uint2*>(link)[t];
    Chosen c;
    if (!choose_at(t, own, rinfo, chain_info, directed, &c)) {
      *loop_flag = 1;  // a chain with an end is no loop: cannot happen
    } else {
      head[t] = c.head_state >> 1;
      if (c.p == 0) {  // (the class bytes of all other k-mers are preset to 0xFF)
        hcls[t] = c.cls;
        hlen[t] = c.len;
        hlast[t] = c.last;
        ori[t] = uint8_t(c.d);
        atomicAdd(&s_len, static_cast<unsigned long long>(c.len));
      }
    }
  }
  __syncthreads();
  // (many workgroups adding to one word would queue up behind each other)
  if (threadIdx.x == 0 && s_len) atomicAdd(len_sums + (blockIdx.x & (kLenSums - 1)), s_len);
}

// Every sampled ruler on a chain with ends has reached one after the pointer jumping.
__global__ __launch_bounds__(256) void k_rulers_done(const unsigned long long* __restrict__ rinfo, int64_t n_dense,
                                                      int* __restrict__ loop_flag) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n_dense && !(rinfo[i] & kEndFlag)) *loop_flag = 1;
}

__global__ __launch_bounds__(256) void k_loops(const uint32_t* __restrict__ link,
                                                int64_t n, uint32_t* __restrict__ head,
                                                uint32_t* __restrict__ pos,
                                                uint8_t* __restrict__ ori,
                                                uint8_t* __restrict__ hcls,
                                                uint32_t* __restrict__ hlen,
                                                uint32_t* __restrict__ hlast) {
  const int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= n) return;
  if (hcls[t] != 0xFE) return;  // only its own flag: the loop's smallest k-mer rewrites just its own
  const uint32_t start = uint32_t(2 * t);
  uint32_t s = start;
  int64_t steps = 0;
  do {
    const uint32_t lk = link[s ^ 1];
    if (lk == kNone) return;  // cannot happen on a loop; leave the k-mer unassigned
    s = ((lk >> 1) << 1) | ((s & 1) ^ (lk & 1));
    if ((s >> 1) < uint32_t(t)) return;  // a smaller k-mer owns this loop
    steps++;
  } while (s != start && steps <= 2 * n);
  if (s != start) return;
  const uint32_t len = uint32_t(steps);
  uint32_t p = 0, last = start;
  s = start;
  do {
    const uint32_t y = s >> 1;
    head[y] = uint32_t(t);
    pos[y] = p;
    ori[y] = ori_byte(s & 1, len - 1 - p);
    last = s;
    const uint32_t lk = link[s ^ 1];
    s = ((lk >> 1) << 1) | ((s & 1) ^ (lk & 1));
    p++;
  } while (s != start);
  hcls[t] = 3;
  hlen[t] = p;
  hlast[t] = last;
}

// ---------------------------------------------------------------------------------- E3
// Unitig ids = rank of (class, head k-mer index): the reference's push order at n_workers == 1.
// The ranks are not materialised per k-mer (two 8-byte prefix arrays over all k-mers would be
// written, scanned and read back for the one k-mer in tens or thousands that heads a unitig):
// a workgroup counts the heads of each class among its 2048 k-mers, the per-workgroup counts
// are scanned (a few ten thousand values), and k_unitig_fill recomputes the ranks inside its
// workgroup from the class bytes.
constexpr int kHeadItems = 8;
constexpr int kHeadSpan = 256 * kHeadItems;

// Heads of classes 0..3 among kmers [t0, t0 + 8), as four 16-bit counters in one word.
__device__ __forceinline__ uint64_t head_counts8(const uint8_t* __restrict__ hcls, int64_t t0, int64_t n,
                                                 uint8_t (&c)[kHeadItems]) {
  uint64_t packed = 0;
  if (t0 + kHeadItems <= n) {
    const uint2 v = *reinterpret_cast<const uint2*>(hcls + t0);  // t0 is a multiple of 8
#pragma unroll
    for (int i = 0; i < kHeadItems; i++) c[i] = uint8_t(((i < 4 ? v.x : v.y) >> (8 * (i & 3))) & 0xFF);
  } else {
#pragma unroll
    for (int i = 0; i < kHeadItems; i++) c[i] = t0 + i < n ? hcls[t0 + i] : uint8_t(0xFF);
  }
#pragma unroll
  for (int i = 0; i < kHeadItems; i++)
    if (c[i] <= 3) packed += uint64_t(1) << (16 * c[i]);
  return packed;
}

// Inclusive scan of the four packed counters over the 256 threads of the workgroup; *total gets
// the workgroup's sum.  Fields stay below 2^16 (at most 2048 heads per workgroup).
__device__ __forceinline__ uint64_t block_scan_packed(uint64_t v, uint64_t* lds4, uint64_t* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint64_t inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint64_t o = __shfl_up(inc, d, 64);
    if (lane >= d) inc += o;
  }
  if (lane == 63) lds4[wave] = inc;
  __syncthreads();
  uint64_t before = 0;
#pragma unroll
  for (int w = 0; w < 4; w++)
    if (w < wave) before += lds4[w];
  *total = lds4[0] + lds4[1] + lds4[2] + lds4[3];
  return before + inc;
}

// The end k-mers (a side without a link: they end one chain and start its mirror image), in
// ascending order: the same count / scan / fill over workgroups of 2048 k-mers.  The kernels that
// only have work at chain ends run one thread per entry of this list (k_ruler_heads,
// k_choose_ends, k_emit_heads) -- in a set of short unitigs a tenth of the k-mers are ends, and a
// thread per k-mer would leave the wavefronts a tenth full while they wait on their walks.
__device__ __forceinline__ uint32_t end_flags8(const uint32_t* __restrict__ link, int64_t t0, int64_t n) {
  uint32_t flags = 0;
#pragma unroll
  for (int i = 0; i < kHeadItems; i += 2) {
    if (t0 + i + 1 < n) {
      const uint4 v = *reinterpret_cast<const uint4*>(link + 2 * (t0 + i));  // two k-mers' link pairs
      if (v.x == kNone || v.y == kNone) flags |= 1u << i;
      if (v.z == kNone || v.w == kNone) flags |= 1u << (i + 1);
    } else if (t0 + i < n) {
      const uint2 v = *reinterpret_cast<const uint2*>(link + 2 * (t0 + i));
      if (v.x == kNone || v.y == kNone) flags |= 1u << i;
    }
  }
  return flags;
}

// (the flags of a thread's eight k-mers are kept, a byte per thread, for k_end_fill: it reads n / 8 bytes
// instead of the link table once more)
__global__ __launch_bounds__(256) void k_end_counts(const uint32_t* __restrict__ link, int64_t n,
                                                     int64_t* __restrict__ counts, uint8_t* __restrict__ flags8) {
  __shared__ uint64_t lds4[4];
  const int64_t t0 = (int64_t(blockIdx.x) * 256 + threadIdx.x) * kHeadItems;
  const uint32_t flags = end_flags8(link, t0, n);
  flags8[int64_t(blockIdx.x) * 256 + threadIdx.x] = uint8_t(flags);
  uint64_t total;
  (void)block_scan_packed(uint64_t(__popc(flags)), lds4, &total);
  if (threadIdx.x == 0) counts[blockIdx.x] = int64_t(total);
}

__global__ __launch_bounds__(256) void k_end_fill(const uint8_t* __restrict__ flags8, int64_t n,
                                                   const int64_t* __restrict__ before,
                                                   uint32_t* __restrict__ ends, uint8_t* __restrict__ hcls) {
  __shared__ uint64_t lds4[4];
  const int64_t t0 = (int64_t(blockIdx.x) * 256 + threadIdx.x) * kHeadItems;
  // no k-mer is a head yet (k_choose_ends marks the heads; the array is this thread's eight k-mers wide here)
  static_assert(kHeadItems == 8, "one 8-byte store per thread");
  if (t0 + kHeadItems <= n) {
    *reinterpret_cast<uint64_t*>(hcls + t0) = ~uint64_t(0);
  } else {
    for (int64_t t = t0; t < n; t++) hcls[t] = 0xFF;
  }
  const uint32_t flags = flags8[int64_t(blockIdx.x) * 256 + threadIdx.x];
  const uint64_t mine = uint64_t(__popc(flags));
  uint64_t total;
  int64_t at = before[blockIdx.x] + int64_t(block_scan_packed(mine, lds4, &total) - mine);
#pragma unroll
  for (int i = 0; i < kHeadItems; i++)
    if (flags & (1u << i)) ends[at++] = uint32_t(t0 + i);
}

// b01[b] = heads of class 0 | class 1 << 32 in workgroup b's k-mers, b23[b] the same for 2, 3.
__global__ __launch_bounds__(256) void k_head_block_counts(const uint8_t* __restrict__ hcls, int64_t n,
                                                            int64_t* __restrict__ b01,
                                                            int64_t* __restrict__ b23) {
  __shared__ uint64_t lds4[4];
  const int64_t t0 = (int64_t(blockIdx.x) * 256 + threadIdx.x) * kHeadItems;
  uint8_t c[kHeadItems];
  uint64_t total;
  (void)block_scan_packed(head_counts8(hcls, t0, n, c), lds4, &total);
  if (threadIdx.x == 0) {
    b01[blockIdx.x] = int64_t(total & 0xFFFF) | (int64_t((total >> 16) & 0xFFFF) << 32);
    b23[blockIdx.x] = int64_t((total >> 32) & 0xFFFF) | (int64_t(total >> 48) << 32);
  }
}

// b01 / b23: exclusive prefixes of the per-workgroup counts.
__global__ __launch_bounds__(256) void k_unitig_fill(
    const uint8_t* __restrict__ hcls, const int64_t* __restrict__ b01,
    const int64_t* __restrict__ b23, int64_t n, int64_t base1, int64_t base2, int64_t base3,
    const uint8_t* __restrict__ ori, const uint32_t* __restrict__ hlen,
    const uint32_t* __restrict__ hlast, uint32_t* __restrict__ uid, uint32_t* __restrict__ u_head,
    uint32_t* __restrict__ u_first, uint32_t* __restrict__ u_last, uint32_t* __restrict__ u_len) {
  __shared__ uint64_t lds4[4];
  const int64_t t0 = (int64_t(blockIdx.x) * 256 + threadIdx.x) * kHeadItems;
  uint8_t c[kHeadItems];
  const uint64_t mine = head_counts8(hcls, t0, n, c);
  uint64_t total;
  const uint64_t excl = block_scan_packed(mine, lds4, &total) - mine;
  if (mine == 0) return;
  const int64_t w01 = b01[blockIdx.x], w23 = b23[blockIdx.x];
  int64_t next[4] = {(w01 & 0xFFFFFFFF) + int64_t(excl & 0xFFFF),
                     base1 + (w01 >> 32) + int64_t((excl >> 16) & 0xFFFF),
                     base2 + (w23 & 0xFFFFFFFF) + int64_t((excl >> 32) & 0xFFFF),
                     base3 + (w23 >> 32) + int64_t(excl >> 48)};
#pragma unroll
  for (int i = 0; i < kHeadItems; i++) {
    if (c[i] > 3) continue;
    const int64_t t = t0 + i;
    const int64_t u = c[i] == 0 ? next[0]++ : c[i] == 1 ? next[1]++ : c[i] == 2 ? next[2]++ : next[3]++;
    uid[t] = uint32_t(u);
    u_head[u] = uint32_t(t);
    u_first[u] = uint32_t(2 * t) | (ori[t] & 1u);
    u_last[u] = hlast[t];
    u_len[u] = hlen[t];
  }
}

// ---------------------------------------------------------------------------------- E4
// vertex v = 2u + side (0 = left end, 1 = right end); edges[4v + c] = other vertex or kNone.
// directed (spss.h:706-726): the right end is the unitig's outgoing port, the left end its
// incoming one; an edge joins the last k-mer of u to a first k-mer Next(last, c) of another
// unitig, k-mers as they are.  (A k-mer with an edge from another unitig is an end of its own:
// inside a unitig every k-mer has exactly one neighbour on that side.)
template <typename KeyT>
__global__ __launch_bounds__(256) void k_edges(DevSet<KeyT> set, int64_t n_vertices, bool directed,
                                                const uint32_t* __restrict__ u_first,
                                                const uint32_t* __restrict__ u_last,
                                                const uint32_t* __restrict__ head,
                                                const uint32_t* __restrict__ uid,
                                                uint32_t* __restrict__ edges, uint32_t* __restrict__ mate) {
  const int64_t v = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (v >= n_vertices) return;
  mate[v] = kNone;  // nobody is matched yet
  const uint32_t u = uint32_t(v >> 1), side = uint32_t(v & 1);
  const uint32_t st = side ? u_last[u] : u_first[u];
  const int k = set.k;
  const uint64_t x = set.kmer(st >> 1);
  const uint64_t o = (st & 1) ? revcomp(x, k) : x;
  // The four candidates of an end in two families: the ones that are consecutive values -- Next(o, .) as they
  // are, or the reverse complements of Prev(o, .), which are Next(rc(o), .) -- are found by ONE bounded search
  // (DevSet::for_group4); the others sit in four different buckets and are probed one by one, and in a
  // canonical set only where that form is the canonical one.  (Four probes per end before; about three now.)
  uint32_t out[4] = {kNone, kNone, kNone, kNone};
  const auto take = [&](int c, int64_t i, bool as_is) {
    const uint32_t u2 = uid[head[i]];
    if (u2 == u) return;
    if (directed) {
      out[c] = 2 * u2 + (side ? 0u : 1u);
      return;
    }
    // side of the k-mer found that this edge touches
    const uint32_t f = side ? (as_is ? 0u : 1u) : (as_is ? 1u : 0u);
    const uint32_t fs = u_first[u2];
    const uint32_t side2 = ((fs >> 1) == uint32_t(i) && (fs & 1) == f) ? 0u : 1u;
    out[c] = 2 * u2 + side2;
  };
  if (side) {
    set.for_group4(kmer_next(o, k, 0), [&](int64_t i) { take(int(uint64_t(set.keys[i]) & 3), i, true); });
    if (!directed) {
#pragma unroll
      for (int c = 0; c < 4; c++) {
        const uint64_t y = kmer_next(o, k, c), r = revcomp(y, k);
        if (r < y) {
          const int64_t i = set.find(r);
          if (i >= 0) take(c, i, false);
        }
      }
    }
  } else {
#pragma unroll
    for (int c = 0; c < 4; c++) {
      const uint64_t y = kmer_prev(o, k, c);
      if (directed || y <= revcomp(y, k)) {
        const int64_t i = set.find(y);
        if (i >= 0) take(c, i, true);
      }
    }
    if (!directed) {
      // rc(Prev(o, c)) = Next(rc(o), 3 - c); a member is the canonical form of its candidate only when it is the smaller
      const uint64_t ro = revcomp(o, k);
      set.for_group4(kmer_next(ro, k, 0), [&](int64_t i) {
        const int c = 3 - int(uint64_t(set.keys[i]) & 3);
        const uint64_t y = kmer_prev(o, k, c);
        if (revcomp(y, k) < y) take(c, i, false);
      });
    }
  }
  *reinterpret_cast<uint4*>(edges + 4 * v) = make_uint4(out[0], out[1], out[2], out[3]);
}

// ---------------------------------------------------------------------------------- E5
// When the reference's sweep first considers the edge in slot c of vertex v: node by node, the
// right side's edges before the left side's (spss.h:1445-1499).  directed: only the outgoing
// port enumerates (spss.h:797-815), an incoming port takes the priority of the other end.
__device__ __forceinline__ uint64_t slot_priority(uint32_t v, int c, bool directed) {
  if (directed && !(v & 1)) return ~uint64_t(0);
  return uint64_t(v >> 1) * 8 + ((v & 1) ? 0 : 4) + uint64_t(c);
}

__global__ __launch_bounds__(256) void k_match_best(const uint32_t* __restrict__ edges,
                                                     const uint32_t* __restrict__ mate,
                                                     int64_t n_vertices, bool directed,
                                                     unsigned long long* __restrict__ best_prio,
                                                     uint32_t* __restrict__ best_w,
                                                     const int* __restrict__ prev,
                                                     int* __restrict__ any_live) {
  if (prev && *prev == 0) return;  // the round before found no live edge: the matching is complete
  const int64_t v = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (v >= n_vertices) return;
  uint32_t bw = kNone;
  uint64_t bp = ~uint64_t(0);
  if (mate[v] == kNone) {
#pragma unroll
    for (int c = 0; c < 4; c++) {
      const uint32_t w = edges[4 * v + c];
      if (w == kNone || mate[w] != kNone) continue;
      uint64_t pr = slot_priority(uint32_t(v), c, directed);
#pragma unroll
      for (int c2 = 0; c2 < 4; c2++) {
        if (edges[4 * int64_t(w) + c2] == uint32_t(v)) {
          const uint64_t p2 = slot_priority(w, c2, directed);
          pr = p2 < pr ? p2 : pr;
        }
      }
      if (pr < bp) {
        bp = pr;
        bw = w;
      }
    }
  }
  best_prio[v] = bp;
  best_w[v] = bw;
  if (bw != kNone) *any_live = 1;
}

__global__ __launch_bounds__(256) void k_match_commit(const unsigned long long* __restrict__ best_prio,
                                                       const uint32_t* __restrict__ best_w,
                                                       int64_t n_vertices,
                                                       const int* __restrict__ live,
                                                       uint32_t* __restrict__ mate) {
  if (*live == 0) return;  // k_match_best of this round did not run or found nothing
  const int64_t v = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (v >= n_vertices) return;
  const uint32_t w = best_w[v];
  if (w == kNone) return;
  if (best_w[w] == uint32_t(v) && best_prio[w] == best_prio[v]) mate[v] = w;
}

// GetSPSSCanonical(fast = false) (spss.h:1208-1322): the reference's one-thread path extension.
// From every node that has no edge yet, a walk leaves through the right side if it has edges
// (else the left) and keeps taking the first edge whose far side is free and that does not
// come back to the walk's start.  Each choice depends on all earlier ones, so this is one
// thread replaying the sweep over the edge table (latency-bound by design, like the reference's).
__global__ __launch_bounds__(64) void k_match_slow(const uint32_t* __restrict__ edges,
                                                    uint32_t* __restrict__ mate, int64_t n_u) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  for (int64_t i = 0; i < n_u; i++) {
    if (mate[2 * i] != kNone || mate[2 * i + 1] != kNone) continue;
    bool any_r = false, any_l = false;
    for (int c = 0; c < 4; c++) {
      any_l |= edges[8 * i + c] != kNone;
      any_r |= edges[8 * i + 4 + c] != kNone;
    }
    if (!any_r && !any_l) continue;
    uint32_t v = uint32_t(2 * i) | (any_r ? 1u : 0u);
    int64_t steps = 0;
    while (mate[v] == kNone && steps++ <= 2 * n_u) {
      uint32_t pick = kNone;
      for (int c = 0; c < 4 && pick == kNone; c++) {
        const uint32_t w = edges[4 * int64_t(v) + c];
        if (w == kNone || (w >> 1) == uint32_t(i) || mate[w] != kNone) continue;
        pick = w;
      }
      if (pick == kNone) break;
      mate[v] = pick;
      mate[pick] = v;
      v = pick ^ 1;  // arrived through side pick & 1, goes on from the other side
    }
  }
}

// ---------------------------------------------------------------------------------- E6
// Loops of the path cover, the reference's way (spss.h:1541-1625): every chosen edge united in a
// ParallelDisjointSet, a component with no node that misses an edge is a loop.  (Round 1 walked
// every open path from its ends with one thread per path; a set with bubbles stitches 10^5..10^6
// unitigs into one path.)
// (the fills of small arrays ride on kernels that have a thread per entry anyway: every fill is a launch)
__global__ __launch_bounds__(256) void k_dsu_init(unsigned long long* __restrict__ a, int64_t n_u,
                                                   uint8_t* __restrict__ has_terminal) {
  const int64_t u = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (u < n_u) {
    a[u] = (unsigned long long)u;
    has_terminal[u] = 0;
  }
}

// One thread per vertex v = 2u + side with an edge; the edge {v, mate[v]} is united once, by its
// smaller vertex (the reference unites it from both ends, :1551-1566; the second call finds both
// in one component already).
__global__ __launch_bounds__(256) void k_dsu_unite_mates(DevDsu dsu, const uint32_t* __restrict__ mate,
                                                          int64_t n_vertices) {
  const int64_t v = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (v >= n_vertices) return;
  const uint32_t w = mate[v];
  if (w == kNone || w < uint32_t(v)) return;
  dsu.unite(uint32_t(v >> 1), w >> 1);
}

// has_terminal[root] = 1 for every component with a node that misses an edge (:1584-1612)
__global__ __launch_bounds__(256) void k_dsu_mark_terminals(DevDsu dsu, const uint32_t* __restrict__ mate, int64_t n_u,
                                                             uint8_t* __restrict__ has_terminal) {
  const int64_t u = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (u >= n_u) return;
  if (mate[2 * u] == kNone || mate[2 * u + 1] == kNone) has_terminal[dsu.find(uint32_t(u))] = 1;
}

// visited[u] = 1: u lies on an open path; 0: on a loop
__global__ __launch_bounds__(256) void k_dsu_open_paths(DevDsu dsu, const uint8_t* __restrict__ has_terminal,
                                                         int64_t n_u, uint8_t* __restrict__ visited) {
  const int64_t u = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (u >= n_u) return;
  visited[u] = has_terminal[dsu.find(uint32_t(u))];
}

// One thread per loop of the path cover (the loop's smallest unitig).  Replays the
// reference's sequential union-by-rank over the loop's nodes in ascending order
// (spss.h:1551-1566, parallel_disjoint_set.h:53-78) to find the root, then drops the
// root's left edge and its mate entry (spss.h:1626-1643).  directed: only the outgoing edges
// are united (spss.h:866-870) and the root's outgoing edge is the one dropped (:922-926).
__global__ __launch_bounds__(64) void k_loop_cut(uint32_t* __restrict__ mate, int64_t n_u, bool directed,
                                                  const uint8_t* __restrict__ visited,
                                                  uint32_t* __restrict__ scratch_nodes,
                                                  uint32_t* __restrict__ scratch_parent,
                                                  uint32_t* __restrict__ scratch_rank,
                                                  unsigned long long* __restrict__ scratch_used) {
  const int64_t u = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (u >= n_u) return;
  if (visited[u]) return;
  // first lap: am I the smallest node, and how long is the loop?
  uint32_t cur = uint32_t(u);
  bool going_right = true;
  int64_t len = 0;
  do {
    const uint32_t w = mate[2 * int64_t(cur) + (going_right ? 1 : 0)];
    if (w == kNone) return;
    cur = w >> 1;
    going_right = (w & 1) == 0;
    if (cur < uint32_t(u)) return;
    len++;
  } while (cur != uint32_t(u) && len <= n_u);
  if (cur != uint32_t(u)) return;
  const uint64_t base = atomicAdd(scratch_used, (unsigned long long)len);
  uint32_t* nodes = scratch_nodes + base;
  uint32_t* parent = scratch_parent + base;
  uint32_t* rank = scratch_rank + base;
  // second lap: collect
  cur = uint32_t(u);
  going_right = true;
  for (int64_t i = 0; i < len; i++) {
    nodes[i] = cur;
    const uint32_t w = mate[2 * int64_t(cur) + (going_right ? 1 : 0)];
    cur = w >> 1;
    going_right = (w & 1) == 0;
  }
  // heap sort ascending
  for (int64_t start = len / 2 - 1; start >= 0; start--) {
    int64_t root = start;
    while (true) {
      int64_t child = 2 * root + 1;
      if (child >= len) break;
      if (child + 1 < len && nodes[child] < nodes[child + 1]) child++;
      if (nodes[root] >= nodes[child]) break;
      const uint32_t tmp = nodes[root];
      nodes[root] = nodes[child];
      nodes[child] = tmp;
      root = child;
    }
  }
  for (int64_t end = len - 1; end > 0; end--) {
    const uint32_t tmp = nodes[0];
    nodes[0] = nodes[end];
    nodes[end] = tmp;
    int64_t root = 0;
    while (true) {
      int64_t child = 2 * root + 1;
      if (child >= end) break;
      if (child + 1 < end && nodes[child] < nodes[child + 1]) child++;
      if (nodes[root] >= nodes[child]) break;
      const uint32_t t2 = nodes[root];
      nodes[root] = nodes[child];
      nodes[child] = t2;
      root = child;
    }
  }
  for (int64_t i = 0; i < len; i++) {
    parent[i] = uint32_t(i);
    rank[i] = 0;
  }
  auto index_of = [&](uint32_t node) {
    int64_t lo = 0, hi = len;
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if (nodes[mid] < node) lo = mid + 1; else hi = mid;
    }
    return uint32_t(lo);
  };
  auto find = [&](uint32_t a) {
    while (parent[a] != a) a = parent[a];
    return a;
  };
  auto unite = [&](uint32_t a, uint32_t b) {
    a = find(a);
    b = find(b);
    if (a == b) return;
    // lower rank, then lower node index, becomes the child; indices are in node order
    if (rank[a] > rank[b] || (rank[a] == rank[b] && a > b)) {
      const uint32_t tmp = a;
      a = b;
      b = tmp;
    }
    parent[a] = b;
    if (rank[a] == rank[b]) rank[b]++;
  };
  for (int64_t i = 0; i < len; i++) {
    const uint32_t a = nodes[i];
    if (!directed) unite(uint32_t(i), index_of(mate[2 * int64_t(a)] >> 1));  // edge_left first
    unite(uint32_t(i), index_of(mate[2 * int64_t(a) + 1] >> 1));
  }
  const uint32_t root_node = nodes[find(0)];
  const int64_t cut = 2 * int64_t(root_node) + (directed ? 1 : 0);
  const uint32_t w = mate[cut];
  mate[cut] = kNone;
  mate[w] = kNone;
}

// ---------------------------------------------------------------------------------- E7
// Walks over the path cover, ranked instead of walked.  A walk is a sequence of states
// S = 2u + going_right: it leaves unitig u through its right side (going_right) or its left one,
// and the state after S is mate[S] ^ 1 (arrive through side w & 1 of unitig w >> 1, go on through
// the other side).  Every state carries one word  done:1 | next:31 | weight:32 :
//   not done: weight = k-mers of the unitigs from S up to, not including, state `next`;
//   done:     weight = k-mers from S to the end of its walk, `next` = the walk's last state.
// A state without a successor starts done (next = itself).  A round replaces (next, weight) of every
// unfinished state by those of two hops (k_walk_jump); whatever snapshot of the successor's word a
// thread reads satisfies the invariant, so the rounds need no double buffering; log2(longest path)
// rounds.  After the loop cut there are no loops left, so every state finishes.
// From the two states of a unitig everything the stitch needs follows without walking
// (spss.h:1649-1829): the path's two end unitigs, the k-mers before it in either direction.
constexpr unsigned long long kWalkDone = 1ull << 63;
__device__ __forceinline__ unsigned long long make_walk(bool done, uint32_t next, uint32_t weight) {
  return (done ? kWalkDone : 0) | ((unsigned long long)(next & 0x7FFFFFFFu) << 32) | weight;
}
__device__ __forceinline__ uint32_t walk_next(unsigned long long w) { return uint32_t(w >> 32) & 0x7FFFFFFFu; }
__device__ __forceinline__ uint32_t walk_weight(unsigned long long w) { return uint32_t(w); }

__global__ __launch_bounds__(256) void k_walk_init(const uint32_t* __restrict__ mate, const uint32_t* __restrict__ u_len,
                                                    int64_t n_states, unsigned long long* __restrict__ walk) {
  const int64_t s = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (s >= n_states) return;
  const uint32_t w = mate[s];
  walk[s] = w == kNone ? make_walk(true, uint32_t(s), u_len[s >> 1]) : make_walk(false, w ^ 1u, u_len[s >> 1]);
}

__global__ __launch_bounds__(256) void k_walk_jump(int64_t n_states, unsigned long long* __restrict__ walk,
                                                    const int* __restrict__ prev, int* __restrict__ changed) {
  if (prev && *prev == 0) return;
  const int64_t s = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (s >= n_states) return;
  unsigned long long mine = walk[s];
  if (mine & kWalkDone) return;
#pragma unroll
  for (int hop = 0; hop < kJumpHops && !(mine & kWalkDone); hop++) {  // (kJumpHops hops per launch, see k_ruler_jump)
    const unsigned long long theirs =
        __hip_atomic_load(&walk[walk_next(mine)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    mine = make_walk((theirs & kWalkDone) != 0, walk_next(theirs), walk_weight(mine) + walk_weight(theirs));
  }
  __hip_atomic_store(&walk[s], mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  *changed = 1;
}

// scls[u]: 0 = kept walk from a left terminal, 1 = kept walk from a right terminal,
//          2 = isolated unitig, 0xFF = not the start of an output string.
// directed: a string starts at every node without an incoming edge and runs forward
// (spss.h:931-1011); isolated unitigs are class 0 like the others.
__global__ __launch_bounds__(256) void k_string_starts(const uint32_t* __restrict__ mate,
                                                        const uint32_t* __restrict__ u_len,
                                                        const unsigned long long* __restrict__ walk,
                                                        int64_t n_u, bool directed,
                                                        uint8_t* __restrict__ scls,
                                                        int64_t* __restrict__ s_nk,
                                                        int64_t* __restrict__ str_start) {
  const int64_t u = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (u >= n_u) return;
  // the strings' base counts land at their ids later (k_string_ids): n_strings <= n_u entries, the rest stay zero
  str_start[u] = 0;
  if (u == 0) str_start[n_u] = 0;
  const bool hl = mate[2 * u] != kNone, hr = mate[2 * u + 1] != kNone;
  uint8_t cls = 0xFF;
  int64_t nk = 0;
  if (directed) {
    if (!hl) {
      nk = walk_weight(walk[2 * u + 1]);
      cls = 0;
    }
  } else if (!hl && !hr) {
    cls = 2;
    nk = u_len[u];
  } else if (!hl || !hr) {
    const unsigned long long w = walk[2 * u + (hl ? 0 : 1)];  // no left edge: the walk goes right
    nk = walk_weight(w);
    const uint32_t far_end = walk_next(w) >> 1;
    if (uint32_t(u) <= far_end) cls = hl ? 1 : 0;  // path.front().first > path.back().first -> skipped
  }
  scls[u] = cls;
  s_nk[u] = nk;
}

// one_sequence (fast = false, spss.h:1328-1350): the strings come in node order whatever their
// class; otherwise class by class (spss.h:1731-1829).
__global__ __launch_bounds__(256) void k_string_counts(const uint8_t* __restrict__ scls, int64_t n_u,
                                                        bool one_sequence,
                                                        int64_t* __restrict__ c01,
                                                        int64_t* __restrict__ c2) {
  const int64_t u = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (u >= n_u) return;
  const uint8_t c = scls[u];
  if (one_sequence) {
    c01[u] = int64_t(c != 0xFF);
    c2[u] = 0;
  } else {
    c01[u] = int64_t(c == 0) | (int64_t(c == 1) << 32);
    c2[u] = int64_t(c == 2);
  }
}

// The string of every start: its id (the reference's push order), its length.
__global__ __launch_bounds__(256) void k_string_ids(
    int64_t n_u, const uint8_t* __restrict__ scls, const int64_t* __restrict__ c01,
    const int64_t* __restrict__ c2, const int64_t* __restrict__ s_nk, const int64_t* __restrict__ class_totals,
    bool one_sequence, int k, uint32_t* __restrict__ sid_at, uint32_t* __restrict__ lens,
    int64_t* __restrict__ str_bases) {
  const int64_t u = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (u >= n_u) return;
  // totals of the two scans (class 0 | class 1 << 32, class 2), still on the device: no host round trip
  const int64_t base1 = class_totals[0] & 0xFFFFFFFF, base2 = base1 + (class_totals[0] >> 32);
  const uint8_t c = scls[u];
  if (c == 0xFF) return;
  int64_t sid;
  if (one_sequence || c == 0) sid = c01[u] & 0xFFFFFFFF;
  else if (c == 1) sid = base1 + (c01[u] >> 32);
  else sid = base2 + c2[u];
  sid_at[u] = uint32_t(sid);
  lens[sid] = uint32_t(s_nk[u] - 1);
  str_bases[sid] = s_nk[u] + k - 1;
}

// Every unitig finds its string, its place in it and its orientation from its own two states: going
// right it ends at e_r, going left at e_l; the string starts at whichever of the two is a start
// (scls), runs towards the other one, and the k-mers before this unitig are those of the walk from
// it BACK to the start, minus its own.
__global__ __launch_bounds__(256) void k_string_assign(
    const uint32_t* __restrict__ u_len, const unsigned long long* __restrict__ walk, int64_t n_u,
    const uint8_t* __restrict__ scls, const uint32_t* __restrict__ sid_at, bool one_sequence,
    uint32_t* __restrict__ u_sid, uint32_t* __restrict__ u_koff, uint8_t* __restrict__ u_flip,
    const int* __restrict__ unfinished) {
  // (the last jump round still changed something: the cover holds a loop that the cut missed, the host reports it
  // after this batch of launches -- until then nothing is derived from walks that have not reached their ends)
  if (unfinished && *unfinished) return;
  const int64_t u = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (u >= n_u) return;
  const unsigned long long wr = walk[2 * u + 1], wl = walk[2 * u];
  const uint32_t e_r = walk_next(wr) >> 1, e_l = walk_next(wl) >> 1;
  if (e_r == uint32_t(u) && e_l == uint32_t(u)) {  // a string of its own
    u_sid[u] = sid_at[u];
    u_koff[u] = 0;
    // fast = false spells an isolated unitig through FindPath(i, false), i.e. reverse-complemented
    u_flip[u] = (scls[u] == 2 && one_sequence) ? 1 : 0;
    return;
  }
  // the start is the end that carries a class; a walk that starts at the left end passes u going right
  const bool from_left = scls[e_l] != 0xFF;
  const uint32_t start = from_left ? e_l : e_r;
  u_sid[u] = sid_at[start];
  u_koff[u] = (from_left ? walk_weight(wl) : walk_weight(wr)) - u_len[u];
  u_flip[u] = from_left ? 0 : 1;
}

// GetUnitigsCanonical output: every unitig is its own string.
__global__ __launch_bounds__(256) void k_unitig_strings(const uint32_t* __restrict__ u_len,
                                                         int64_t n_u, int k,
                                                         uint32_t* __restrict__ u_sid,
                                                         uint32_t* __restrict__ u_koff,
                                                         uint8_t* __restrict__ u_flip,
                                                         uint32_t* __restrict__ lens,
                                                         int64_t* __restrict__ str_bases) {
  const int64_t u = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (u >= n_u) return;
  u_sid[u] = uint32_t(u);
  u_koff[u] = 0;
  u_flip[u] = 0;
  lens[u] = u_len[u] - 1;
  str_bases[u] = int64_t(u_len[u]) + k - 1;
}

// ---------------------------------------------------------------------------------- E8
// Where a unitig's k-mers go in the output base stream.  The records live at the index of the
// unitig's head k-mer (an array of one record per k-mer, touched only at the heads: it reuses the
// two 8-byte scan arrays, dead by then), so that k_emit gets from a k-mer's head straight to its
// place: one dependent random read per k-mer instead of two (unitig id, then place).
struct UnitigPlace {
  int64_t base;    // base position of the unitig's first k-mer slot (in traversal order)
  uint32_t len;    // k-mers in the unitig
  uint32_t flags;  // bit 0: traversed reverse-complemented, bit 1: last unitig of its string
};

__global__ __launch_bounds__(256) void k_unitig_place(const uint32_t* __restrict__ u_len,
                                                       const uint32_t* __restrict__ u_sid,
                                                       const uint32_t* __restrict__ u_koff,
                                                       const uint8_t* __restrict__ u_flip,
                                                       const int64_t* __restrict__ str_start,
                                                       const uint32_t* __restrict__ lens,
                                                       const uint32_t* __restrict__ u_head, int64_t n_u,
                                                       UnitigPlace* __restrict__ place_at_head,
                                                       const int* __restrict__ unfinished) {
  if (unfinished && *unfinished) return;  // (see k_string_assign: string ids from unfinished walks index nothing)
  const int64_t u = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (u >= n_u) return;
  const uint32_t sid = u_sid[u];
  UnitigPlace pl;
  pl.base = str_start[sid] + u_koff[u];
  pl.len = u_len[u];
  const bool is_last = u_koff[u] + u_len[u] == lens[sid] + 1;  // lens = k-mers in the string - 1
  pl.flags = uint32_t(u_flip[u] & 1) | (is_last ? 2u : 0u);
  place_at_head[u_head[u]] = pl;
}

// kEmitBases bases of o (one per byte, first base at the lowest address) when `writer`, all K of
// them when `tail`.
template <int kEmitBases>
__device__ __forceinline__ void store_bases(uint8_t* __restrict__ at, uint64_t o, int k, bool writer, bool tail) {
  if (writer) {
    if (kEmitBases == 1) {
      at[0] = uint8_t((o >> (2 * (k - 1))) & 3);
    } else {
      struct __attribute__((packed, aligned(1))) Run {
        uint32_t w[kEmitBases / 4];
      } run;
#pragma unroll
      for (int j = 0; j < kEmitBases / 4; j++) {
        const uint32_t b = uint32_t(o >> (2 * (k - 4 * (j + 1)))) & 0xFFu;  // bases 4j .. 4j + 3
        run.w[j] = (b >> 6) | (((b >> 4) & 3u) << 8) | (((b >> 2) & 3u) << 16) | ((b & 3u) << 24);
      }
      *reinterpret_cast<Run*>(at) = run;
    }
  }
  if (tail) {
    for (int i = writer ? kEmitBases : 0; i < k; i++) at[i] = uint8_t((o >> (2 * (k - 1 - i))) & 3);
  }
}

// A k-mer at slot q of its string spells the bases q .. q + K - 1 of that string, all of them
// inside the string.  So one k-mer in kEmitBases writes kEmitBases bases in one (unaligned) store
// instead of every k-mer one byte at a scattered address: the writers are the k-mers whose slot
// within the unitig is a multiple of kEmitBases (the slots of a unitig start at 0, so these cover
// it, running over into the next unitig of the string or into the string's tail with the same
// bases those k-mers would write), and the last k-mer of a string adds its whole spelling.  Which
// way round a unitig is traversed is only known at its place record: a k-mer is a candidate if
// either of its two possible slots qualifies (pos, or the far-end distance bits of ori), and only
// candidates look their unitig up -- one k-mer in kEmitBases / 2.
template <typename KeyT, int kEmitBases>
__global__ __launch_bounds__(256) void k_emit(DevSet<KeyT> set, const uint32_t* __restrict__ head,
                                               const uint32_t* __restrict__ pos,
                                               const uint8_t* __restrict__ ori,
                                               const UnitigPlace* __restrict__ place_at_head,
                                               uint8_t* __restrict__ bytes) {
  __shared__ int64_t s_bucket[2];
  const int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  set.block_bucket(s_bucket);
  if (t >= set.n) return;
  constexpr uint32_t kMask = uint32_t(kEmitBases - 1);
  const uint32_t p = pos[t], ob = ori[t];
  if ((p & kMask) != 0 && ((ob >> 1) & kMask) != 0) return;
  const uint32_t h = head[t];
  if (h == kNone) return;
  const UnitigPlace pl = place_at_head[h];
  const uint32_t flip = pl.flags & 1;
  const uint32_t q = flip ? (pl.len - 1 - p) : p;
  const bool tail = (pl.flags & 2) && q == pl.len - 1;  // last k-mer of the string
  const bool writer = (q & kMask) == 0;
  if (!writer && !tail) return;
  const uint64_t x = set.kmer_from_block(t, s_bucket);
  const uint64_t o = ((ob & 1u) ^ flip) ? revcomp(x, set.k) : x;
  store_bases<kEmitBases>(bytes + pl.base + q, o, set.k, writer, tail);
}

// The same bytes without the per-k-mer (head, pos, ori) of k_choose (ranking without stamps,
// k_choose_ends): the chains are walked once more, in string order only.  A sampled k-mer (or
// the unsampled k-mer that starts a chain) works out its unitig and its slot like k_choose does,
// looks up the unitig's place, and the one of its two states that runs with the string walks on
// to the next sampled k-mer or to the chain's end; every k-mer passed knows its slot and
// orientation from the walk.  A walk writes all K bases of its first k-mer and of every K-th one
// after it (each covers the K slots up to the next; the walk that follows starts with a write of
// its own), so the lanes of a wavefront read keys and write in the same iterations, and the last
// k-mer of a string adds the string's tail.
//   kRun = the largest power of two <= K (at most 16): K bases go out as the runs [0, kRun) and
// [K - kRun, K), unaligned stores; kRun = 0: byte by byte (K < 4).
template <int kRun>
__device__ __forceinline__ void store_run(uint8_t* __restrict__ at, uint64_t o, int k, int first) {
  struct __attribute__((packed, aligned(1))) Run {
    uint32_t w[kRun / 4];
  } run;
#pragma unroll
  for (int j = 0; j < kRun / 4; j++) {
    const uint32_t b = uint32_t(o >> (2 * (k - first - 4 * (j + 1)))) & 0xFFu;  // bases first + 4j .. + 3
    run.w[j] = (b >> 6) | (((b >> 4) & 3u) << 8) | (((b >> 2) & 3u) << 16) | ((b & 3u) << 24);
  }
  *reinterpret_cast<Run*>(at + first) = run;
}
template <int kRun>
__device__ __forceinline__ void store_kmer(uint8_t* __restrict__ at, uint64_t o, int k) {
  if (kRun == 0) {
    for (int i = 0; i < k; i++) at[i] = uint8_t((o >> (2 * (k - 1 - i))) & 3);
  } else {
    store_run<(kRun ? kRun : 4)>(at, o, k, 0);
    if (k > kRun) store_run<(kRun ? kRun : 4)>(at, o, k, k - kRun);
  }
}

template <typename KeyT, int kRun>
__device__ __forceinline__ void emit_walk(const DevSet<KeyT>& set, const uint32_t* __restrict__ link,
                                          uint32_t first_state, uint2 own, uint64_t x_first,
                                          const UnitigPlace& pl, uint32_t q, uint8_t* __restrict__ bytes) {
  const int k = set.k;
  uint32_t cur = first_state;
  uint2 pr = own;
  uint64_t x = x_first;
  int since = 0;  // k-mers since the last one written
  while (true) {
    const bool tail = (pl.flags & 2) && q == pl.len - 1;  // last k-mer of the string
    if (since == 0 || tail) {
      if (cur != first_state) x = set.kmer(cur >> 1);
      store_kmer<kRun>(bytes + pl.base + q, (cur & 1) ? revcomp(x, k) : x, k);
    }
    const uint32_t lk = leave_link(pr, cur);
    if (lk == kNone) return;
    cur = step_to(cur, lk);
    q++;
    since = since + 1 == k ? 0 : since + 1;
    if (sampled_ruler(cur) || q >= pl.len) return;  // (q < len always: the bound only guards the stores)
    pr = link_pair(link, cur);
  }
}

// One thread per sampled k-mer.
template <typename KeyT, int kRun>
__global__ __launch_bounds__(256) void k_emit_rulers(DevSet<KeyT> set, const uint32_t* __restrict__ link,
                                                      const unsigned long long* __restrict__ rinfo,
                                                      const unsigned long long* __restrict__ chain_info,
                                                      bool directed,
                                                      const UnitigPlace* __restrict__ place_at_head,
                                                      uint8_t* __restrict__ bytes) {
  __shared__ int64_t s_bucket[2];
  set.block_bucket_at((int64_t(blockIdx.x) * blockDim.x) << kRulerShift, s_bucket);
  const int64_t t = (int64_t(blockIdx.x) * blockDim.x + threadIdx.x) << kRulerShift;
  if (t >= set.n) return;
  const uint2 own = reinterpret_cast<const uint2*>(link)[t];
  Chosen c;
  if (!choose_at(uint32_t(t), own, rinfo, chain_info, directed, &c)) return;
  const UnitigPlace pl = place_at_head[c.head_state >> 1];
  const uint32_t flip = pl.flags & 1;
  emit_walk<KeyT, kRun>(set, link, 2 * uint32_t(t) + (c.d ^ flip), own, set.kmer_from_block(t, s_bucket), pl,
                        flip ? (c.len - 1 - c.p) : c.p, bytes);
}

// One thread per end k-mer: the unsampled ones that start a chain in string order.
template <typename KeyT, int kRun>
__global__ __launch_bounds__(256) void k_emit_heads(DevSet<KeyT> set, const uint32_t* __restrict__ link,
                                                     const unsigned long long* __restrict__ rinfo,
                                                     const unsigned long long* __restrict__ chain_info,
                                                     bool directed, const uint32_t* __restrict__ ends,
                                                     int64_t n_ends,
                                                     const UnitigPlace* __restrict__ place_at_head,
                                                     uint8_t* __restrict__ bytes) {
  const int64_t e = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (e >= n_ends) return;
  const int64_t t = ends[e];
  if ((t & (kRulerEvery - 1)) == 0) return;
  const uint2 own = reinterpret_cast<const uint2*>(link)[t];
  Chosen c;
  if (!choose_at(uint32_t(t), own, rinfo, chain_info, directed, &c)) return;
  const UnitigPlace pl = place_at_head[c.head_state >> 1];
  const uint32_t flip = pl.flags & 1;
  const uint32_t first_state = 2 * uint32_t(t) + (c.d ^ flip);  // the state of t that runs with the string
  if (enter_link(own, first_state) != kNone) return;            // the walk before this k-mer passes it
  emit_walk<KeyT, kRun>(set, link, first_state, own, set.kmer(t), pl, flip ? (c.len - 1 - c.p) : c.p, bytes);
}

// ---- the strings from the logs of the ranking walks (WalkLog): one thread per walker that walked.
// It ranks its first k-mer like k_choose does, looks up the unitig's place, and writes the k-mers
// its log names -- itself, the ones K, 2K, ... steps in, the one it arrived at -- each at the slot
// the step count gives (counting down when the walk ran against the string, and then the k-mer is
// written as its reverse complement).  A stretch longer than the log holds (one in a few hundred)
// is walked once more -- not here, where one such lane would keep its 63 neighbours waiting for
// hundreds of dependent reads, but from a list, by k_emit_log_long.
constexpr uint32_t kLongHead = 0x80000000u;
constexpr int kLongBlocks = 128;  // workgroups of k_emit_log_rulers that walk the listed long stretches

template <typename KeyT, int kRun, bool kLongPass>
__device__ __forceinline__ void emit_logged(const DevSet<KeyT>& set, const uint32_t* __restrict__ link,
                                            const WalkLog& log, int64_t w, uint64_t hdr, uint32_t u,
                                            uint64_t x_u, const Chosen& c, const UnitigPlace& pl,
                                            uint8_t* __restrict__ bytes) {
  const int k = set.k;
  const uint32_t flip = pl.flags & 1;
  const uint32_t q_u = flip ? (c.len - 1 - c.p) : c.p;
  const bool with_string = (u & 1) == (c.d ^ flip);  // the walk ran in string order
  const uint32_t steps = uint32_t(hdr >> 32), arrival = uint32_t(hdr);
  const auto emit = [&](uint32_t state, uint32_t step, uint64_t x) {
    const uint32_t q = with_string ? q_u + step : q_u - step;
    if (q >= pl.len) return;  // (cannot happen: the guard keeps a corrupt log from writing elsewhere)
    const uint32_t as_read = with_string ? state : state ^ 1;
    store_kmer<kRun>(bytes + pl.base + q, (as_read & 1) ? revcomp(x, k) : x, k);
  };
  if (!kLongPass) {
    if (log.is_long(steps)) return;  // on the list (WalkLog::arrived)
    emit(u, 0, x_u);
    if (steps == 0) return;
    const int n_passed = int((steps - 1) / uint32_t(k));
    // (measured and dropped, round 3: the logged k-mers four at a time, level by level -- their states, then
    // keys and coarse bucket entries, then the offsets -- instead of one by one: 941 us per 10^8 against 874;
    // with 7 waves per SIMD in flight the reads of different walkers already overlap)
    for (int j = 0; j < n_passed; j++) {
      const uint32_t st = log.mid[int64_t(j) * log.n_walkers + w];
      emit(st, uint32_t(j + 1) * uint32_t(k), set.kmer(st >> 1));
    }
    // a sampled k-mer the walk arrived at writes itself (emit_ruler_walker: every one of them does, from its own
    // thread, whose k-mer comes with the block's keys): only the end of a chain costs a k-mer's reads here
    if (!sampled_ruler(arrival)) emit(arrival, steps, set.kmer(arrival >> 1));
  } else {
    emit(u, 0, x_u);
    uint32_t cur = u, step = 0;
    int since = 0;
    while (step < steps) {
      const uint32_t lk = leave_link(link_pair(link, cur), cur);
      if (lk == kNone) return;
      cur = step_to(cur, lk);
      step++;
      if (++since == k || step == steps) {
        since = 0;
        emit(cur, step, set.kmer(cur >> 1));
      }
    }
  }
}

template <typename KeyT, int kRun, bool kLongPass>
__device__ __forceinline__ void emit_ruler_walker(const DevSet<KeyT>& set, const uint32_t* __restrict__ link,
                                                  const unsigned long long* __restrict__ rinfo,
                                                  const WalkLog& log, bool directed, int64_t i, uint64_t x_t,
                                                  const UnitigPlace* __restrict__ place_at_head,
                                                  uint8_t* __restrict__ bytes) {
  const uint64_t hdr = log.hdr[i];
  bool self_only = false;
  if (hdr == kRecUnset) {
    // Neither state of this sampled k-mer walked (both stretches beside it were walked from their other ends): the
    // walks that arrived here left the k-mer to itself; the thread of its even state writes it.
    if (kLongPass || (i & 1) || log.hdr[i ^ 1] != kRecUnset) return;
    self_only = true;
  }
  const int64_t t = (i >> 1) << kRulerShift;
  // both states of a sampled k-mer have reached their ends (there is no loop on this path)
  const ulonglong2 ri = reinterpret_cast<const ulonglong2*>(rinfo)[i >> 1];
  const Chosen c = choose_from_ends(uint32_t(ri.x), uint32_t((ri.x >> 32) & 0x7FFFFFFFu), uint32_t(ri.y),
                                    uint32_t((ri.y >> 32) & 0x7FFFFFFFu), directed);
  if (self_only) {
    const UnitigPlace pl = place_at_head[c.head_state >> 1];
    const uint32_t flip = pl.flags & 1;
    const uint32_t q = flip ? (c.len - 1 - c.p) : c.p;
    if (q < pl.len) store_kmer<kRun>(bytes + pl.base + q, ((c.d ^ flip) & 1) ? revcomp(x_t, set.k) : x_t, set.k);
    return;
  }
  emit_logged<KeyT, kRun, kLongPass>(set, link, log, i, hdr, uint32_t(2 * t + (i & 1)), x_t, c,
                                     place_at_head[c.head_state >> 1], bytes);
}

template <typename KeyT, int kRun, bool kLongPass>
__device__ __forceinline__ void emit_head_walker(const DevSet<KeyT>& set, const uint32_t* __restrict__ link,
                                                 const unsigned long long* __restrict__ rinfo,
                                                 const unsigned long long* __restrict__ chain_info,
                                                 const WalkLog& log, bool directed, const uint32_t* __restrict__ ends,
                                                 int64_t e, const UnitigPlace* __restrict__ place_at_head,
                                                 uint8_t* __restrict__ bytes) {
  const uint64_t hdr = log.hdr[e];
  if (hdr == kRecUnset) return;
  const uint32_t t = ends[e];
  const uint2 own = reinterpret_cast<const uint2*>(link)[t];
  Chosen c;
  if (!choose_at(t, own, rinfo, chain_info, directed, &c)) return;
  // the state that walked: the one that starts a chain (state 2t of a k-mer on its own)
  const uint32_t u = (own.y == kNone && own.x != kNone) ? 2 * t + 1 : 2 * t;
  emit_logged<KeyT, kRun, kLongPass>(set, link, log, e, hdr, u, set.kmer(t), c, place_at_head[c.head_state >> 1],
                                     bytes);
}

// The first kLongBlocks workgroups walk the listed long stretches (rulers' and heads'), striding over
// the list -- their few hundred dependent reads each run under the rest of the grid, which takes
// the walkers of the dense ruler array one thread each.
template <typename KeyT, int kRun>
__global__ __launch_bounds__(256) void k_emit_log_rulers(DevSet<KeyT> set, const uint32_t* __restrict__ link,
                                                          const unsigned long long* __restrict__ rinfo,
                                                          const unsigned long long* __restrict__ chain_info,
                                                          WalkLog log, WalkLog log_heads, bool directed,
                                                          const uint32_t* __restrict__ ends,
                                                          const UnitigPlace* __restrict__ place_at_head,
                                                          uint8_t* __restrict__ bytes) {
  __shared__ int64_t s_bucket[2];
  if (blockIdx.x < kLongBlocks) {
    const unsigned int n_long = *log.long_count;
    for (unsigned int at = blockIdx.x * blockDim.x + threadIdx.x; at < n_long; at += kLongBlocks * blockDim.x) {
      const uint32_t w = log.long_walkers[at];
      if (w & kLongHead) {
        emit_head_walker<KeyT, kRun, true>(set, link, rinfo, chain_info, log_heads, directed, ends,
                                           int64_t(w & ~kLongHead), place_at_head, bytes);
      } else {
        emit_ruler_walker<KeyT, kRun, true>(set, link, rinfo, log, directed, int64_t(w),
                                            set.kmer((int64_t(w) >> 1) << kRulerShift), place_at_head, bytes);
      }
    }
    return;
  }
  const int64_t first = int64_t(blockIdx.x - kLongBlocks) * blockDim.x;
  set.block_bucket_at((first >> 1) << kRulerShift, s_bucket);
  const int64_t i = first + threadIdx.x;
  if (i >= log.n_walkers || ((i >> 1) << kRulerShift) >= set.n) return;
  emit_ruler_walker<KeyT, kRun, false>(set, link, rinfo, log, directed, i,
                                       set.kmer_from_block((i >> 1) << kRulerShift, s_bucket), place_at_head, bytes);
}

template <typename KeyT, int kRun>
__global__ __launch_bounds__(256) void k_emit_log_heads(DevSet<KeyT> set, const uint32_t* __restrict__ link,
                                                         const unsigned long long* __restrict__ rinfo,
                                                         const unsigned long long* __restrict__ chain_info,
                                                         WalkLog log, bool directed,
                                                         const uint32_t* __restrict__ ends,
                                                         const UnitigPlace* __restrict__ place_at_head,
                                                         uint8_t* __restrict__ bytes) {
  const int64_t e = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (e >= log.n_walkers) return;
  emit_head_walker<KeyT, kRun, false>(set, link, rinfo, chain_info, log, directed, ends, e, place_at_head, bytes);
}

__global__ __launch_bounds__(256) void k_pack(const uint8_t* __restrict__ bytes, int64_t n_bases,
                                               int64_t n_words, uint64_t* __restrict__ words) {
  const int64_t w = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (w >= n_words) return;
  const int64_t p0 = w * 32;
  uint64_t out = 0;
  if (p0 + 32 <= n_bases) {
    const uint4 lo = *reinterpret_cast<const uint4*>(bytes + p0);
    const uint4 hi = *reinterpret_cast<const uint4*>(bytes + p0 + 16);
    const uint32_t q[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
    for (int i = 0; i < 8; i++) {
#pragma unroll
      for (int b = 0; b < 4; b++) {
        const uint64_t code = (q[i] >> (8 * b)) & 3;
        out |= code << (62 - 2 * (4 * i + b));
      }
    }
  } else {
    for (int j = 0; j < 32 && p0 + j < n_bases; j++) out |= uint64_t(bytes[p0 + j] & 3) << (62 - 2 * j);
  }
  words[w] = out;
}

// ---------------------------------------------------------------------------------- host
// Device control block of one encode plan: every small flag and counter the kernels raise, zeroed by
// ONE fill when the plan starts (they used to be a dozen memsets of 4..16 bytes, each a launch), and
// read back in three copies (after the end list, after the unitig counts, at the end).
constexpr int kJumpRoundsMax = 48;  // pointer-jumping rounds over the rulers: 2 + log2(n) <= 34
constexpr int kMatchBatch = 8;      // matching rounds enqueued between two looks at their flags
constexpr int kMatchFirst = 4;      // ... in the first batch: genomes and their intersections / differences take 1..3
                                    // rounds (a round that finds nothing ends them); graphs with tips and bubbles 4..6
constexpr int kWalkRoundsMax = 40;  // jumping rounds over the path cover: 2 + log2(2 n_u)
struct EncCtl {
  int64_t tot[3 + kLenSums];  // [0..1] unitig counts by class, [2 .. 2 + kLenSums) k-mers the unitigs account
                              // for (partial sums), [2 + kLenSums] (as an int) a ruler on a loop
  int self_rc;                // a canonical set holds a k-mer equal to its own reverse complement
  int rc_marks;               // k_adj_rc ran for a group whose records did not fit k_adj_rc1
  int64_t t2[2];              // strings by class: class 0 | class 1 << 32, class 2
  int64_t n_bases;            // total of the string lengths in bases
  unsigned long long sc_used; // k_loop_cut's scratch cursor
  unsigned int long_count[2]; // WalkLog::long_count
  int rc_batched;             // k_adj_rc: a group's ranges took more than one batch of its LDS window
  int tgt_extra;              // k_tgt_split: parts beyond the first of the windows with long streams
  int jump_live[kJumpRoundsMax + 1];
  int match_live[kMatchBatch + 1];
  int walk_live[kWalkRoundsMax + 1];
};

struct EncPlan {
  int64_t n = 0, n_u = 0, n_strings = 0, n_bases = 0;
  int mode = 0;
  ksh_geom g{};
  ksh_set_view set{};
  // k-mer level (slot kSlotEncode)
  uint32_t *nbr = nullptr, *link = nullptr, *head = nullptr, *pos = nullptr, *hlen = nullptr,
           *hlast = nullptr, *uid = nullptr;
  unsigned long long* info = nullptr;
  uint8_t *ori = nullptr, *hcls = nullptr;
  int64_t *c01 = nullptr, *c23 = nullptr;
  uint32_t* fine = nullptr;
  int fine_bits = 0;
  uint32_t* coarse = nullptr;  // DevSet::coarse (sets of at least four k-mers per bucket)
  // unitig level (own allocation)
  char* ublock = nullptr;
  uint32_t *u_head = nullptr, *u_first = nullptr, *u_last = nullptr, *u_len = nullptr,
           *edges = nullptr, *mate = nullptr, *best_w = nullptr, *u_sid = nullptr,
           *u_koff = nullptr, *lens = nullptr, *sc_nodes = nullptr, *sc_parent = nullptr,
           *sc_rank = nullptr;
  unsigned long long *best_prio = nullptr, *sc_used = nullptr;
  uint8_t *visited = nullptr, *scls = nullptr, *u_flip = nullptr;
  int64_t *s_nk = nullptr, *sc01 = nullptr, *sc2 = nullptr, *str_start = nullptr;
  EncCtl* ctl = nullptr;
  int rounds = 0;
  int64_t routes = 0;  // KSH_ROUTE_* bits: which variants this plan ran (ksh_spss_encode_routes)
  // ranking without stamps (k_choose_ends): the ruler and chain-start records live on in `info`
  // until the strings are written (k_emit_rulers / k_emit_heads)
  bool stamped = true;
  bool directed = false;
  const unsigned long long *rinfo = nullptr, *chain_info = nullptr;
  const uint32_t* ends = nullptr;  // the end k-mers, ascending (aliases pos)
  int64_t n_ends = 0;
  WalkLog log_rulers{}, log_heads{};  // hdr == NULL: the strings are written by walking (k_emit_rulers / k_emit_heads)
};

inline size_t al(size_t x) { return (x + 255) & ~size_t(255); }

// Host-side statement of what a kernel's indexing assumes about the scratch it borrows (carved arrays
// that change hands between stages, LDS byte counts, packed 16-bit fields): checked before the launch.
#define KSH_BOUND(cond)                                                                              \
  do {                                                                                               \
    if (!(cond)) return ::ksh::fail(KSH_INTERNAL, "encode: bound violated: %s (%s:%d)", #cond, __FILE__, __LINE__); \
  } while (0)
static_assert(sizeof(RcRecord<uint16_t>) == 8 && sizeof(RcRecord<uint32_t>) == 8 && sizeof(RcRecord<uint64_t>) == 16,
              "record sizes the scratch layout assumes");

constexpr int64_t kRcRowsMax = 512;          // workgroups (histogram rows) of the rc partition
// LDS window of k_adj_rc (keys + marks + slice index): two workgroups share a CU's 160 KB.  At 10^8
// k-mers and N = 14 the 16 ranges of the second pass hold 6 100 +- 80 keys together and a bucket
// whose k-mers start with C 7 600 (canonical sets are 7 : 5 : 3 : 1 dense by first base): both fit
// 7 980 keys; at 60 KB (6 137 keys) a third of the groups staged the second pass in two batches and
// half of them the first, every record looked up once per batch (2.03 -> 1.76 ms per 10^8)
constexpr int64_t kRcWindowBytes = 78 << 10;

// KSH_ADJACENCY=probe selects the round-1 kernel (every probe a search in global memory) for
// A/B measurements; anything else: the LDS-staged form.
inline bool staged_adjacency() {
  static const bool on = [] {
    const char* e = getenv("KSH_ADJACENCY");
    return !(e && std::string(e) == "probe");
  }();
  return on;
}
// KSH_RANK=stamp: the ranking walks stamp every k-mer and k_choose / k_emit work per k-mer (the
// path every set with a non-branching loop takes anyway).
inline bool rank_with_stamps() {
  static const bool on = [] {
    const char* e = getenv("KSH_RANK");
    return e && std::string(e) == "stamp";
  }();
  return on;
}
// KSH_EMIT=walk: the strings are written by a second walk over the chains instead of from the logs of
// the ranking walks (what a set whose logs do not fit falls back to).
inline bool emit_by_walking() {
  static const bool on = [] {
    const char* e = getenv("KSH_EMIT");
    return e && std::string(e) == "walk";
  }();
  return on;
}
// KSH_L2_MIN: ruler records from which the pointer jumping runs in two levels (tests set it low).
inline int64_t l2_threshold() {
  static const int64_t v = [] {
    const char* e = getenv("KSH_L2_MIN");
    return e ? std::max<int64_t>(4096, atoll(e)) : int64_t(1) << 20;
  }();
  return v;
}
// (Measured and dropped, round 4: with several encodes in flight on different streams (lanes), dynamic LDS added to
// the walk / forward-probe / emit launches to cap how many of their workgroups share a CU, so that another lane's
// kernels find wave slots beside them -- the hope being that the LDS-bound probes and the walks that wait on HBM
// would use the CU together.  64 x 10^8 build at 3 lanes, ms: no cap 779.8; walks at 4 workgroups per CU 782.0;
// walks and emit at 4: 804.2; walks at 8: 777.2; walks at 4 and the forward probe at 3: 797.1; walks at 2: 862.5.
// A kernel trace of the 3-lane build shows three kernels running 62 % of the time and their summed durations
// 2.3 x those of the one-lane build: concurrent kernels share the GPU by time, not by resource; what lanes
// buy is the idle time between one encode's kernels and behind its host round trips.)
inline unsigned nblk(int64_t n) { return unsigned(std::max<int64_t>(1, (n + 255) / 256)); }

template <typename T>
T* carve(char*& at, size_t count) {
  T* p = reinterpret_cast<T*>(at);
  at += al(count * sizeof(T));
  return p;
}

void free_plan(ksh_ctx* ctx) {
  EncPlan* p = static_cast<EncPlan*>(ctx->enc_state);
  if (!p) return;
  if (p->ublock) pool_free(ctx, p->ublock);
  delete p;
  ctx->enc_state = nullptr;
}

// slices of about 1.5 keys (measured on 10^7- and 10^8-key sets: 9 and 12 bits are the
// fastest there; fewer bits mean longer slices, more bits a bigger index to build)
// (at least four values per slice: the four consecutive candidates of a group probe share one)
static int enc_fine_bits(const ksh_geom* g, int64_t n) {
  const int64_t nb = n_buckets(g);
  int fine_bits = 0;
  while (fine_bits < 13 && fine_bits + 2 < key_bits(g) && (int64_t(3) << fine_bits) < 2 * (n / nb + 1)) fine_bits++;
  return fine_bits;
}
static size_t enc_slot_bytes(const ksh_geom* g, int64_t n) {
  const int64_t nb = n_buckets(g);
  const int fine_bits = enc_fine_bits(g, n);
  const size_t fine_entries = fine_bits >= 2 ? (size_t(nb) << fine_bits) + 1 : 0;
  return 2 * al(size_t(2 * n) * 4) + al(size_t(2 * n) * 8) + 5 * al(size_t(n) * 4) + 2 * al(size_t(n)) +
         2 * al(size_t(n) * 8) + al(fine_entries * 4) + al(size_t((n >> kCoarseShift) + 2) * 4) +
         al(sizeof(EncCtl) + size_t(2 * nb) * 4) + 4096;
}
static size_t enc_arena_bytes(const ksh_geom* g, int64_t n) {
  const int64_t nb = n_buckets(g);
  return size_t(n / 256 + 4096) * 8 * 2 + (1u << 16) + size_t(n / kHeadSpan + 64) * 8 +
         (nb <= (1 << 14) ? size_t(kRcRowsMax) * 2 * nb * 4 + size_t(2 * nb + 1) * 16 + size_t(2 * nb) * 280 + 8192 : 0);
}
size_t encode_scratch_bytes(const ksh_geom* g, int64_t n) {
  const size_t slot = enc_slot_bytes(g, n), arena = enc_arena_bytes(g, n);
  return slot + (slot >> 3) + arena + (arena >> 2) + (size_t(2) << 20);  // (with slot_reserve's and arena_reserve's spare)
}
int encode_reserve(ksh_ctx* ctx, const ksh_geom* g, int64_t n) {
  KSH_TRY(slot_reserve(ctx, kSlotEncode, enc_slot_bytes(g, n)));
  return arena_reserve(ctx, enc_arena_bytes(g, n));
}

template <typename KeyT>
int encode_plan_t(ksh_ctx* ctx, const ksh_geom* g, const ksh_set_view* sv, bool directed, int mode,
                  int64_t* n_strings, int64_t* n_bases) {
  free_plan(ctx);
  EncPlan* p = new EncPlan;
  ctx->enc_state = p;
  p->g = *g;
  p->set = *sv;
  p->mode = mode;
  const int64_t n = sv->n_keys;
  p->n = n;
  if (n == 0) {
    *n_strings = 0;
    *n_bases = 0;
    return KSH_OK;
  }
  if (n >= int64_t(0x7FFFFFF0)) return fail(KSH_INVALID_ARGUMENT, "set too large for 32-bit indices");
  const int64_t nb = n_buckets(g);
  const int fine_bits = enc_fine_bits(g, n);
  const bool use_fine = fine_bits >= 2;
  const size_t fine_entries = use_fine ? (size_t(nb) << fine_bits) + 1 : 0;
  KSH_TRY(encode_reserve(ctx, g, n));
  arena_reset(ctx);
  char* at = ctx->slot[kSlotEncode];
  p->nbr = carve<uint32_t>(at, size_t(2 * n));
  p->link = carve<uint32_t>(at, size_t(2 * n));
  p->info = carve<unsigned long long>(at, size_t(2 * n));
  p->head = carve<uint32_t>(at, size_t(n));
  p->pos = carve<uint32_t>(at, size_t(n));
  p->hlen = carve<uint32_t>(at, size_t(n));
  p->hlast = carve<uint32_t>(at, size_t(n));
  p->uid = carve<uint32_t>(at, size_t(n));
  p->ori = carve<uint8_t>(at, size_t(n));
  p->hcls = carve<uint8_t>(at, size_t(n));
  p->c01 = carve<int64_t>(at, size_t(2 * n));  // two arrays of n, one block: later the place records
  p->c23 = p->c01 + n;
  p->fine = use_fine ? carve<uint32_t>(at, fine_entries) : nullptr;
  p->coarse = n >= 4 * nb ? carve<uint32_t>(at, size_t((n >> kCoarseShift) + 2)) : nullptr;
  p->ctl = reinterpret_cast<EncCtl*>(carve<char>(at, sizeof(EncCtl) + size_t(2 * nb) * 4));
  uint32_t* rc_cursor = reinterpret_cast<uint32_t*>(p->ctl + 1);  // k_rc_scatter_l2's per-group cursors: zeroed with the block
  EncCtl* ctl = p->ctl;

  DevSet<KeyT> set{sv->d_offsets, static_cast<const KeyT*>(sv->d_keys), nb, n, g->k, key_bits(g)};
  hipStream_t st = ctx->stream;
  if (use_fine) {
    hipLaunchKernelGGL((k_fine_index<KeyT>), dim3(unsigned(nb)), dim3(256), size_t(4) << fine_bits, st, sv->d_offsets,
                       static_cast<const KeyT*>(sv->d_keys), key_bits(g), fine_bits, p->fine, nb);
    set.fine = p->fine;
    set.fine_bits = fine_bits;
    p->fine_bits = fine_bits;
  }
  if (p->coarse) {
    const int64_t n_entries = (n >> kCoarseShift) + 1;
    hipLaunchKernelGGL((k_coarse_index<KeyT>), dim3(nblk(n_entries)), dim3(256), 0, st, set, n_entries, p->coarse);
    set.coarse = p->coarse;
  }
  KSH_HIP(hipMemsetAsync(ctl, 0, sizeof(EncCtl) + size_t(2 * nb) * 4, st));
  int* self_rc_flag = &ctl->self_rc;
  {
    Timer timer(ctx, 3, n);
    const int nbits = g->n_bucket_bits;
    if (directed) {
      hipLaunchKernelGGL((k_adjacency<KeyT, true>), dim3(nblk(n)), dim3(256), 0, st, set, p->nbr, self_rc_flag);
    } else if (staged_adjacency() && nbits <= 14 && 2 * g->k >= nbits + 4 &&
               key_bits(g) >= nbits + 2 + (nbits & 1) &&
               n / nb <= 4 * ((kRcWindowBytes - 4 * kRcSegs) / int64_t(sizeof(KeyT) + 6))) {
      // (a group four windows long is staged in four batches and every batch looks at all of the group's
      // records: beyond that -- 10^8 k-mers in 2^10 buckets, 5 x 10^8 in 2^14 -- the probing kernel is as fast)
      // E1b: partition by rc-prefix, LDS-staged probe of the reverse-complement half, forward half
      // in place.  The records live in the (still unused) chain-rank records, rc0 / rc1 in the link
      // array; the records are reset by k_link_cut afterwards.
      // (a row per 16K k-mers up to the cap: a 10^7-k-mer set still puts a workgroup on every CU)
      const int64_t rows = std::max<int64_t>(1, std::min<int64_t>(kRcRowsMax, (n + 16383) / 16384));
      const int64_t per_row = ((n + rows - 1) / rows + 1023) / 1024 * 1024;
      KSH_BOUND(rows >= 1 && rows <= kRcRowsMax && rows * per_row >= n);  // k_rc_hist / k_rc_scatter_*: row r owns [r * per_row, ...)
      KSH_BOUND(size_t(n) * sizeof(RcRecord<KeyT>) <= al(size_t(2 * n) * 8));  // the records live in `info`
      static const bool one_level = [] {
        const char* e = getenv("KSH_RC_SCATTER");
        return e && std::string(e) == "direct";
      }();
      static const bool half_groups = [] {
        const char* e = getenv("KSH_RC_GROUPS");
        return e && std::string(e) == "half";
      }();
      // (with KSH_RC_GROUPS=half the two-level scatter starts at 2^12 k-mers, so that small test sets take the route)
      const bool two_level = n >= (int64_t(1) << (half_groups ? 12 : 20)) && nbits > kSgBits && nbits <= 16 && !one_level;
      // Group bits.  KSH_RC_GROUPS=half deals the records into HALF buckets (one more bit of rx): the LDS
      // window of a group halves and four workgroups of 512 threads share a CU instead of two of 1024.
      // Measured on a 10^8-k-mer genome set (profiles/r03_rc_groups_ab.txt): k_adj_rc 1.62 -> 1.54 ms, and the
      // finer histogram and scatter took it back twice over (k_rc_hist + 0.09, k_rc_scatter_l2 + 0.09,
      // columns / bounds + 0.03 ms): not the default.  The kernels take the group bits as a parameter either way.
      const int extra = (half_groups && two_level && nbits == 14 && key_bits(g) >= 2 * ((nbits + 1 + 2 + 1) / 2)) ? 1 : 0;
      const int gbits = nbits + extra;
      const int64_t ng = int64_t(1) << gbits;  // groups
      uint32_t* hist = static_cast<uint32_t*>(arena_alloc(ctx, size_t(rows) * ng * 4));
      int64_t* totals = static_cast<int64_t*>(arena_alloc(ctx, size_t(ng + 1) * 8));
      int64_t* goff = static_cast<int64_t*>(arena_alloc(ctx, size_t(ng + 1) * 8));
      if (!hist || !totals || !goff) return fail(KSH_INTERNAL, "scratch arena too small");
      RcRecord<KeyT>* rec = reinterpret_cast<RcRecord<KeyT>*>(p->info);
      uint32_t* rc0 = p->link;
      uint32_t* rc1 = p->link + n;
      const size_t hist_lds = size_t(ng) * 4;
      {
        // (more than the 64 KB a kernel gets without asking; per context: the attribute is the device's)
        const uint32_t bit = 1u << (sizeof(KeyT) == 2 ? 0 : sizeof(KeyT) == 4 ? 1 : 2);
        if (!(ctx->lds_opt_in & bit)) {
          const int bytes = int(kRcWindowBytes + 2048);
          KSH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_adj_rc<KeyT, 1024>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
          KSH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_adj_rc<KeyT, 512>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
          KSH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_adj_rc<KeyT, 256>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
          KSH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_adj_rc<KeyT, 64>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
          KSH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_adj_rc1<KeyT, 1024>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, kRc1LdsBytes));
          KSH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_adj_rc1<KeyT, 512>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, kRc1LdsBytes));
          KSH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_adj_rc1<KeyT, 256>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, kRc1LdsBytes));
          KSH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_adj_rc1<KeyT, 64>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, kRc1LdsBytes));
          KSH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_rc_hist<KeyT>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, int(size_t(4) << 15)));
          ctx->lds_opt_in |= bit;
        }
      }
      KSH_BOUND(hist_lds <= (size_t(4) << 15) && ng <= 65536);  // k_rc_hist's LDS histogram; group ids travel as u16
      hipLaunchKernelGGL((k_rc_hist<KeyT>), dim3(unsigned(rows)), dim3(1024), hist_lds, st,
                         static_cast<const KeyT*>(sv->d_keys), n, g->k, key_bits(g), gbits, per_row, hist);
      hipLaunchKernelGGL(k_rc_columns, dim3(unsigned((ng + 63) / 64)), dim3(64 * kColTeams), 0, st, hist, rows, int(ng),
                         totals);
      KSH_TRY(scan_exclusive_i64(ctx, totals, goff, ng, goff + ng));
      p->routes |= KSH_ROUTE_PROBE_STAGED | (two_level ? KSH_ROUTE_SCATTER_TWO_LEVEL : 0);
      if (two_level) {
        // two levels of run-wise writes; the intermediate records borrow arrays that are written
        // later: u32 keys: the upper half of the chain-rank records; u64 keys (16-byte records): the
        // neighbour + link arrays (rc0 / rc1 and nbr are only written after level 2); the group ids
        // sit in the head array
        RcRecord<KeyT>* tmp_rec = sizeof(RcRecord<KeyT>) == 8 ? rec + n : reinterpret_cast<RcRecord<KeyT>*>(p->nbr);
        uint16_t* tmp_g = reinterpret_cast<uint16_t*>(p->head);
        // intermediate records: u32 keys, the upper half of `info` (2n records of 8 bytes in 16n bytes);
        // u64 keys, nbr + link (carved back to back: 2 x al(8n) >= 16n); group ids in `head` (2n <= 4n bytes)
        KSH_BOUND(sizeof(RcRecord<KeyT>) == 8 ? size_t(2 * n) * sizeof(RcRecord<KeyT>) <= al(size_t(2 * n) * 8)
                                    : reinterpret_cast<char*>(p->link) == reinterpret_cast<char*>(p->nbr) + al(size_t(2 * n) * 4) &&
                                          size_t(n) * sizeof(RcRecord<KeyT>) <= 2 * al(size_t(2 * n) * 4));
        KSH_BOUND(gbits - kSgBits <= 8);  // k_rc_scatter_l2: two super-groups' groups in its 512 bins
        uint32_t* cursor = rc_cursor;     // (2^gbits zeroed words behind the control block)
        constexpr int kPer = sizeof(RcRecord<KeyT>) == 8 ? 4 : 2;
        hipLaunchKernelGGL((k_rc_scatter_l1<KeyT, kPer>), dim3(unsigned(rows)), dim3(kL1Threads), 0, st, set, gbits,
                           per_row, hist, goff, tmp_rec, tmp_g);
        hipLaunchKernelGGL((k_rc_scatter_l2<KeyT>), dim3(unsigned((n + kL2Tile - 1) / kL2Tile)), dim3(kL2Threads), 0,
                           st, n, gbits, goff, tmp_rec, tmp_g, cursor, rec);
      } else {
        hipLaunchKernelGGL((k_rc_scatter<KeyT>), dim3(unsigned(rows)), dim3(1024), hist_lds, st, set, gbits, 0, 0,
                           per_row, hist, goff, rec);
      }
      // LDS window: a group's range with some to spare (larger ranges are staged in batches).  Half-bucket
      // groups: 15 % to spare, so that four workgroups (window + 1 KB of static LDS each) share a CU's 160 KB.
      const int64_t per_group = n / ng;
      // (a multiple of 8 keys: the 4-byte marks behind the keys are compared-and-swapped in LDS and have to sit on
      // 4-byte boundaries whatever the key width -- an odd window of 2-byte keys faulted the kernel, round 3)
      const int cap = int(std::min<int64_t>((kRcWindowBytes - 4 * kRcSegs) / int64_t(sizeof(KeyT) + 6),
                                            std::max<int64_t>(1024, extra ? per_group * 23 / 20 + 256 : per_group * 5 / 4 + 256))) & ~7;
      const size_t rc_lds = size_t(cap) * (sizeof(KeyT) + 4) + size_t(cap + 2 * kRcSegs) * 2;
      // k_adj_rc keeps window positions, lengths and slice-index offsets in 16 bits (sidx, RcBatch::packed)
      KSH_BOUND(cap >= 8 && cap % 8 == 0 && cap + 2 * kRcSegs < 65536);
      KSH_BOUND(rc_lds + 2048 <= size_t(kRcWindowBytes) + 2048 && rc_lds <= size_t(kRcWindowBytes));
      int64_t* pb = static_cast<int64_t*>(arena_alloc(ctx, size_t(ng) * 2 * kRcSegs * 8));
      int64_t* pb0 = extra ? static_cast<int64_t*>(arena_alloc(ctx, size_t(ng) * 2 * 8)) : nullptr;
      if (!pb || (extra && !pb0)) return fail(KSH_INTERNAL, "scratch arena too small");
      hipLaunchKernelGGL((k_rc_bounds<KeyT>), dim3(nblk(ng * 2 * kRcSegs + (extra ? ng * 2 : 0))), dim3(256), 0, st, set,
                         gbits, pb, pb0);
      p->routes |= per_group > 4096 ? KSH_ROUTE_RC_1024 : per_group > 1024 ? KSH_ROUTE_RC_512
                   : per_group > 256 ? KSH_ROUTE_RC_256 : KSH_ROUTE_RC_64;
      // k_adj_rc1 (the k-mers look for the records) for the groups whose records fit its LDS, k_adj_rc for the others;
      // KSH_RC1=batched: k_adj_rc1 for every group, a group of more records than fit in batches -- measured at the end of
      // round 4 and not the default: the streams run once per batch, and 8 x 5 x 10^8 k = 31 (30 000 16-byte records per
      // group, seven batches) takes 0.757 s against 0.742, 4 x 2 x 10^7 (19, 8) 20.4 ms against 19.9, 4 x 10^8 k = 31
      // (two batches) 62.3 against 59.4; KSH_RC1=marks: k_adj_rc for all (rounds 2-3)
      static const int rc1_mode = [] {
        const char* e = getenv("KSH_RC1");
        return !e ? 1 : (std::string(e) == "marks" ? 2 : (std::string(e) == "batched" ? 0 : 1));
      }();
      const bool rc1_marks = rc1_mode == 2;
      // (LDS for the average group and a third -- the records of a group are a Poisson count around +-19 % by the first
      // base of G -- so that small groups share a CU many at a time; both kernels decide by the same number)
      const int rc1_cap = rc1_marks ? 0 : int(std::min<int64_t>(Rc1Cfg<KeyT>::kCap, (per_group * 4 / 3 + 263) & ~int64_t(7)));
      // (k_adj_rc: groups of at most that many records are k_adj_rc1's)
      const int pass1_cap = rc1_marks ? -1 : rc1_cap;
      const bool with_rc = rc1_mode != 0;  // k_adj_rc is launched at all
      int rc1_sbits = 6;                               // a slice per one or two records; a chain link has sbits + 1 bits
      while ((1 << rc1_sbits) < kRc1SlicesMax && (2 << rc1_sbits) <= rc1_cap + 1) rc1_sbits++;
      const size_t rc1_lds = (size_t(4) << rc1_sbits) + size_t(rc1_cap) * Rc1Cfg<KeyT>::kEntryBytes;
      KSH_BOUND(rc1_lds <= size_t(kRc1LdsBytes) && rc1_cap % 8 == 0 && rc1_cap < (2 << rc1_sbits) - 1);
      if (!rc1_marks) p->routes |= KSH_ROUTE_RC1_STREAMED;
      const int rc1_batched = rc1_mode == 0 ? 1 : 0;
#define KSH_LAUNCH_RC(T)                                                                                              \
  do {                                                                                                                \
    if (with_rc)                                                                                                      \
      hipLaunchKernelGGL((k_adj_rc<KeyT, T>), dim3(unsigned(ng)), dim3(T), rc_lds, st, set, gbits, goff, rec, pb, pb0, \
                         cap, rc0, rc1, &ctl->rc_batched, pass1_cap, &ctl->rc_marks);                                 \
    if (!rc1_marks)                                                                                                   \
      hipLaunchKernelGGL((k_adj_rc1<KeyT, T>), dim3(unsigned(ng)), dim3(T), rc1_lds, st, set, gbits, goff, rec, pb,   \
                         pb0, rc1_cap, rc1_sbits, rc1_batched, rc0, rc1, &ctl->rc_batched);                           \
  } while (0)
      if (per_group > 4096)
        KSH_LAUNCH_RC(1024);
      else if (per_group > 1024)
        KSH_LAUNCH_RC(512);
      else if (per_group > 256)
        KSH_LAUNCH_RC(256);
      else
        KSH_LAUNCH_RC(64);
#undef KSH_LAUNCH_RC
      // KSH_FWD=probe: the forward half in place; =staged: round 3's five staged windows per chunk; default: one
      // window per workgroup, the probes marked at their targets (k_adj_fwd_targets)
      static const int fwd_mode = [] {
        const char* e = getenv("KSH_FWD");
        return !e ? 0 : (std::string(e) == "probe" ? 1 : (std::string(e) == "staged" ? 2 : 0));
      }();
      if (fwd_mode == 1) {
        hipLaunchKernelGGL((k_adj_fwd<KeyT>), dim3(nblk(n)), dim3(256), 0, st, set, rc0, rc1, p->nbr, self_rc_flag);
      } else if (fwd_mode == 0) {
        p->routes |= KSH_ROUTE_FWD_TARGETS;
        // the records are dead by now: the cuts take their place
        constexpr int kTgtChunk = TgtCfg<KeyT>::kChunk, kTgtStreamMax = TgtCfg<KeyT>::kStreamMax;
        const int64_t n_chunks = (n + kTgtChunk - 1) / kTgtChunk;
        const int64_t cap = n / kTgtStreamMax + 1;  // parts beyond a window's first: fewer than its stream / kTgtStreamMax
        // cuts | (lo, hi) cut pairs of the windows, then of the parts | the parts' tasks
        // (a seventh of a byte per k-mer and 300 bytes: in the dead records, or for a set of a few k-mers in the arena)
        const size_t cut_bytes = size_t((n_chunks + 1) * kTgtBounds + (n_chunks + cap) * 2 * kTgtBounds) * 8 + size_t(cap) * sizeof(TgtTask);
        int64_t* cuts = cut_bytes <= al(size_t(2 * n) * 8) ? reinterpret_cast<int64_t*>(p->info)
                                                           : static_cast<int64_t*>(arena_alloc(ctx, cut_bytes));
        if (!cuts) return fail(KSH_INTERNAL, "scratch arena too small");
        int64_t* rec = cuts + (n_chunks + 1) * kTgtBounds;
        TgtTask* task = reinterpret_cast<TgtTask*>(rec + (n_chunks + cap) * 2 * kTgtBounds);
        KSH_BOUND(n_chunks >= 1 && n_chunks * kTgtChunk >= n && n < (int64_t(1) << 31));  // indices travel as idx << 1 in 32 bits
        hipLaunchKernelGGL((k_tgt_bounds<KeyT>), dim3(nblk((n_chunks + 1) * 8)), dim3(256), 0, st, set, n_chunks, cuts);
        hipLaunchKernelGGL(k_tgt_split, dim3(nblk(n_chunks)), dim3(256), 0, st, cuts, n_chunks, cap, kTgtStreamMax, rec, task,
                           &ctl->tgt_extra);
        hipLaunchKernelGGL((k_tgt_subcuts<KeyT>), dim3(nblk(cap * 8)), dim3(256), 0, st, set, cuts, n_chunks, cap, task,
                           &ctl->tgt_extra, rec);
        hipLaunchKernelGGL((k_adj_fwd_targets<KeyT>), dim3(unsigned(n_chunks + cap)), dim3(kTgtThreads), 0, st, set, rec,
                           n_chunks, &ctl->tgt_extra, rc0, rc1, p->nbr, self_rc_flag);
      } else {
        p->routes |= KSH_ROUTE_FWD_STAGED;
        // the records are dead by now: the chunk bounds take their place
        const int64_t n_chunks = (n + kFwdChunk - 1) / kFwdChunk;
        int64_t* bounds = reinterpret_cast<int64_t*>(p->info);  // kFwdBounds * 8 bytes per 512 k-mers
        KSH_BOUND(size_t(n_chunks + 1) * kFwdBounds * 8 <= al(size_t(2 * n) * 8));
        hipLaunchKernelGGL((k_fwd_bounds<KeyT>), dim3(nblk((n_chunks + 1) * 10)), dim3(256), 0, st, set, n_chunks, bounds);
        // (measured and dropped, round 3: a Next window for twice the range, 6144 keys, still four workgroups per
        // CU: 3 % of the k-mers instead of 15 % fall back to probes in global memory, and the kernel takes
        // 1.88 ms per 10^8 against 1.87 -- the fall-backs are not what it waits for.  profiles/r03_probe_stage_ab.txt)
        hipLaunchKernelGGL((k_adj_fwd_staged<KeyT>), dim3(unsigned(n_chunks)), dim3(kFwdChunk), 0, st, set, bounds,
                           rc0, rc1, p->nbr, self_rc_flag);
#ifdef KSH_FWD_DEBUG
        {
          unsigned long long h[8];
          (void)hipStreamSynchronize(st);
          (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_fwd_dbg), sizeof(h));
          fprintf(stderr, "fwd dbg: kmers %llu next!usable %llu next-bucket-miss %llu prev!usable %llu prev-bucket-miss %llu "
                          "next-range>cap %llu two-top-bases %llu\n", h[0], h[1], h[2], h[3], h[4], h[5], h[6]);
          unsigned long long z[8] = {};
          (void)hipMemcpyToSymbol(HIP_SYMBOL(g_fwd_dbg), z, sizeof(z));
        }
#endif
      }
    } else {
      hipLaunchKernelGGL((k_adjacency<KeyT, false>), dim3(nblk(n)), dim3(256), 0, st, set, p->nbr, self_rc_flag);
    }
  }
  if (directed)
    hipLaunchKernelGGL((k_link_cut<KeyT, true>), dim3(nblk(n)), dim3(256), 0, st, set, p->nbr);
  else
    hipLaunchKernelGGL((k_link_cut<KeyT, false>), dim3(nblk(n)), dim3(256), 0, st, set, p->nbr);
  const uint32_t* link = p->nbr;  // the link table from here on
  const int64_t n_hblocks = (n + kHeadSpan - 1) / kHeadSpan;
  int64_t* b01 = p->c23;              // per-workgroup head counts (n / 2048 values each) live in
  int64_t* b23 = p->c23 + n_hblocks;  // the front of c23; c01 still holds the ruler records
  // d_tot[0..1]: unitig counts by class; [2 .. 2 + kLenSums): k-mers the unitigs account for (partial
  // sums), [2 + kLenSums]: a ruler on a loop (an int)
  int64_t* d_tot = ctl->tot;
  // the end k-mers, ascending (k_end_counts / k_end_fill): per-workgroup counts -> offsets, total
  int64_t* end_before = static_cast<int64_t*>(arena_alloc(ctx, size_t(n_hblocks + 1) * 8));
  if (!end_before) return fail(KSH_INTERNAL, "scratch arena too small");
  uint32_t* ends = p->pos;  // (k_choose writes pos after the last kernel that reads the list)
  uint8_t* end_flags = p->ori;  // (the orientations are k_choose_ends' to fill later; a byte per eight k-mers until then)
  KSH_BOUND(size_t(n_hblocks) * 256 <= al(size_t(n)));
  hipLaunchKernelGGL(k_end_counts, dim3(unsigned(n_hblocks)), dim3(256), 0, st, link, n, end_before, end_flags);
  KSH_TRY(scan_exclusive_i64(ctx, end_before, end_before, n_hblocks, end_before + n_hblocks));
  hipLaunchKernelGGL(k_end_fill, dim3(unsigned(n_hblocks)), dim3(256), 0, st, end_flags, n, end_before, ends, p->hcls);
  p->directed = directed;
  p->ends = ends;
  KSH_HIP(hipMemcpyAsync(ctx->h_pinned, end_before + n_hblocks, 8, hipMemcpyDeviceToHost, st));
  KSH_HIP(hipStreamSynchronize(st));
  const int64_t n_ends = ctx->h_pinned[0];
  p->n_ends = n_ends;
  for (bool stamped = rank_with_stamps(), first_pass = true;; stamped = true, first_pass = false) {
    const int64_t ns2 = 2 * n;
    const int64_t n_dense = 2 * ((n + kRulerEvery - 1) / kRulerEvery);
    // with stamps: ruler records in c01 (n_dense * 8 <= 8n), chain starts in c23, both dead after
    // k_choose; without: both in the record array, which nothing stamps, until the strings are written
    unsigned long long* rinfo = stamped ? reinterpret_cast<unsigned long long*>(p->c01) : p->info;
    unsigned long long* chain_info = stamped ? reinterpret_cast<unsigned long long*>(p->c23) : p->info + n_dense;
    {
      Timer timer(ctx, 4, n);
      if (stamped) {
        KSH_HIP(hipMemsetAsync(p->info, 0xFF, size_t(2 * n) * 8, st));  // chain-rank records unset
        hipLaunchKernelGGL(k_ruler_walk, dim3(nblk(n_dense)), dim3(256), 0, st, link, ns2, n_dense, rinfo, p->info);
        hipLaunchKernelGGL(k_ruler_heads, dim3(nblk(n_ends)), dim3(256), 0, st, link, ends, n_ends, p->info, chain_info);
      } else {
        // the walk logs behind the two record arrays, when they fit (16n bytes in all)
        WalkLog lr{}, lh{};
        lr.k = lh.k = g->k;
        const size_t used = (size_t(n_dense) + size_t(n)) * 8;
        const size_t log_bytes = size_t(n_dense) * (8 + 4 * kLogMidRulers + 4) + size_t(n_ends) * (8 + 4 * kLogMidHeads + 4) + 64;  // + the long list
        if (!emit_by_walking() && used + log_bytes <= size_t(2 * n) * 8) {
          unsigned long long* at8 = p->info + n_dense + n;
          lr.hdr = at8;
          lh.hdr = at8 + n_dense;
          uint32_t* at4 = reinterpret_cast<uint32_t*>(at8 + n_dense + n_ends);
          lr.mid = at4;
          lh.mid = at4 + size_t(kLogMidRulers) * n_dense;
          lr.long_walkers = lh.long_walkers = lh.mid + size_t(kLogMidHeads) * n_ends;  // n_dense + n_ends entries at most
          lr.long_count = lh.long_count = ctl->long_count;
          lr.long_tag = 0;
          lh.long_tag = kLongHead;
          lr.n_walkers = n_dense;
          lh.n_walkers = n_ends;
          lr.n_mid = kLogMidRulers;
          lh.n_mid = kLogMidHeads;
          KSH_HIP(hipMemsetAsync(at8, 0xFF, size_t(n_dense + n_ends) * 8, st));  // nobody has walked yet
          p->routes |= KSH_ROUTE_EMIT_LOGS;
        }
        p->log_rulers = lr;
        p->log_heads = lh;
        KSH_HIP(hipMemsetAsync(rinfo, 0xFF, size_t(n_dense) * 8, st));
        hipLaunchKernelGGL(k_rank_unset, dim3(nblk(n_ends)), dim3(256), 0, st, ends, n_ends, chain_info);
        // Enough walkers for the launch to take a while (2^18: half a grid-full): all of them in one launch,
        // the later of two mirror images finds its record filled in -- one walk per stretch and a fifth
        // (by the emit's time: 1.2 logs per stretch) instead of three for two.  The second phase of the split,
        // an eighth of its lanes walking, is bound by how long a wave lives, not by the reads: 10^8 k-mers,
        // k_rank_walk 2.20 + 1.01 ms split, 1.81 + 0.99 split with the look at the start, 2.41 in one
        // launch.  Over the 64 x 10^8 build (ms per build, ranking walks of its timed region halved): from
        // 3 x 2^20 walkers 922 / 356, from 2^20 922 / 342, from 2^19 917 / 329, from 2^18 906 / 329, below that the
        // same.  Fewer walkers start together, mirror images too: the split by hash bit stays
        // (KSH_RANK_PHASES=2 forces it, =1 the single launch).
        static const int64_t race_min = [] {
          const char* e = getenv("KSH_RANK_PHASES");
          if (e && std::string(e) == "1") return int64_t(0);  // one launch at any size (tests)
          if (e && std::string(e) == "2") return int64_t(1) << 60;
          const char* m = getenv("KSH_RANK_RACE_MIN");  // walkers from which one launch takes them all (A/B runs)
          return m ? int64_t(atoll(m)) : int64_t(1) << 18;
        }();
        p->routes |= (n_dense >= race_min ? KSH_ROUTE_RANK_ONE_LAUNCH : 0) | (n_ends >= race_min ? KSH_ROUTE_HEADS_ONE_LAUNCH : 0);
        if (n_dense < race_min)
          hipLaunchKernelGGL(k_rank_walk<1>, dim3(nblk(n_dense)), dim3(256), 0, st, link, ns2, n_dense, rinfo, chain_info, lr,
                             static_cast<unsigned long long*>(nullptr), static_cast<unsigned long long*>(nullptr));
        if (n_ends < race_min)
          hipLaunchKernelGGL(k_rank_heads<1>, dim3(nblk(n_ends)), dim3(256), 0, st, link, ends, n_ends, rinfo, chain_info, lh);
        // (hlen / hlast are k_choose_ends' to fill afterwards: until then they hold the two-level jumping's arrays)
        const bool l2_after = n_dense >= l2_threshold();
        hipLaunchKernelGGL(k_rank_walk<2>, dim3(nblk(n_dense)), dim3(256), 0, st, link, ns2, n_dense, rinfo, chain_info, lr,
                           l2_after ? reinterpret_cast<unsigned long long*>(p->hlen) : nullptr,
                           l2_after ? reinterpret_cast<unsigned long long*>(p->hlast) : nullptr);
        hipLaunchKernelGGL(k_rank_heads<2>, dim3(nblk(n_ends)), dim3(256), 0, st, link, ends, n_ends, rinfo, chain_info, lh);
      }
    }
    // pointer jumping: over all ruler records (with stamps), or in two levels (k_l2_*): the level-2
    // records, the chain heads' and the stamps live in arrays that k_choose_ends fills afterwards
    // (worth its four extra launches once a round over all records costs tens of microseconds)
    const bool two_levels = !stamped && n_dense >= l2_threshold();
    const int64_t n_jump = two_levels ? (((n_dense - 1) >> (kL2Shift + 1)) << 1) + 2 : n_dense;
    unsigned long long* r2 = reinterpret_cast<unsigned long long*>(p->uid);        // n_jump * 8 <= 4n
    unsigned long long* l2_head = reinterpret_cast<unsigned long long*>(p->hlen);  // n_dense * 8 <= 4n
    unsigned long long* l2_stamp = reinterpret_cast<unsigned long long*>(p->hlast);
    if (two_levels) p->routes |= KSH_ROUTE_JUMP_TWO_LEVEL;
    if (stamped) p->routes = (p->routes | KSH_ROUTE_RANK_STAMPED) & ~int64_t(KSH_ROUTE_EMIT_LOGS | KSH_ROUTE_RANK_ONE_LAUNCH | KSH_ROUTE_HEADS_ONE_LAUNCH | KSH_ROUTE_JUMP_TWO_LEVEL);
    if (two_levels) {  // (l2_head / l2_stamp were set to unset by k_rank_walk<2>)
      hipLaunchKernelGGL(k_l2_walk<true>, dim3(nblk(n_jump)), dim3(256), 0, st, rinfo, n_dense, r2, l2_head, l2_stamp);
      hipLaunchKernelGGL(k_l2_walk<false>, dim3(nblk(n_dense)), dim3(256), 0, st, rinfo, n_dense, r2, l2_head, l2_stamp);
    }
    // log5(records) + 3 launches end every chain (a loop of rulers never ends: k_rulers_done sees it); all of
    // them are enqueued at once, a round after the last one that changed anything returns at its first load.
    // (The kernels behind the rounds are safe on records that have not reached an end: every `next` field of a
    // ruler, level-2 or chain-start record is a valid record index at every moment, a record without the end
    // flag resolves to "on a loop" in chain_end_of / k_l2_resolve, and k_rulers_done raises the flag that sends
    // the set through the stamping pass.  The path cover's rounds below are different: string ids derived from
    // an unfinished walk would index the string table, so k_string_assign / k_unitig_place look at the last
    // round's flag first.)
    int max_rounds = 3;  // a record's reach grows five-fold per launch (four hops): log5 of the records, and spare
    for (int64_t x = n_jump; x > 1; x /= (kJumpHops + 1)) max_rounds++;
    max_rounds = std::min(max_rounds, kJumpRoundsMax);
    if (!first_pass) {  // the stamping pass after a loop was found: its flags and sums start over
      KSH_HIP(hipMemsetAsync(ctl->jump_live, 0, sizeof(ctl->jump_live), st));
      KSH_HIP(hipMemsetAsync(d_tot, 0, (3 + kLenSums) * 8, st));
      KSH_HIP(hipMemsetAsync(ctl->long_count, 0, sizeof(ctl->long_count), st));
    }
    for (int round = 0; round < max_rounds; round++) {
      const int* prev = round ? &ctl->jump_live[round - 1] : nullptr;
      if (two_levels)
        hipLaunchKernelGGL(k_l2_jump, dim3(nblk(n_jump)), dim3(256), 0, st, n_jump, r2, prev, &ctl->jump_live[round]);
      else
        hipLaunchKernelGGL(k_ruler_jump, dim3(nblk(n_dense)), dim3(256), 0, st, n_dense, rinfo, prev,
                           &ctl->jump_live[round]);
    }
    if (two_levels)
      hipLaunchKernelGGL(k_l2_resolve, dim3(nblk(n_dense)), dim3(256), 0, st, n_dense, r2, l2_head, l2_stamp, rinfo);
    int* loop_flag = reinterpret_cast<int*>(d_tot + 2 + kLenSums);
    if (stamped) {
      hipLaunchKernelGGL(k_choose, dim3(nblk(n)), dim3(256), 0, st, p->info, rinfo, chain_info, n, directed,
                         p->head, p->pos, p->ori, p->hcls, p->hlen, p->hlast);
      hipLaunchKernelGGL(k_loops, dim3(nblk(n)), dim3(256), 0, st, link, n, p->head, p->pos, p->ori,
                         p->hcls, p->hlen, p->hlast);
    } else {
      hipLaunchKernelGGL(k_choose_ends, dim3(nblk(n_ends)), dim3(256), 0, st, link, ends, n_ends, rinfo, chain_info,
                         directed, p->head, p->ori, p->hcls, p->hlen, p->hlast,
                         reinterpret_cast<unsigned long long*>(d_tot + 2), loop_flag);
      hipLaunchKernelGGL(k_rulers_done, dim3(nblk(n_dense)), dim3(256), 0, st, rinfo, n_dense, loop_flag);
    }
    hipLaunchKernelGGL(k_head_block_counts, dim3(unsigned(n_hblocks)), dim3(256), 0, st, p->hcls, n, b01, b23);
    KSH_TRY(scan_exclusive_i64(ctx, b01, b01, n_hblocks, d_tot));
    KSH_TRY(scan_exclusive_i64(ctx, b23, b23, n_hblocks, d_tot + 1));
    KSH_HIP(hipGetLastError());
    static_assert(offsetof(EncCtl, self_rc) == (3 + kLenSums) * 8 && (4 + kLenSums) <= 64, "one copy: tot + self_rc");
    KSH_HIP(hipMemcpyAsync(ctx->h_pinned, d_tot, (4 + kLenSums) * 8, hipMemcpyDeviceToHost, st));
    KSH_HIP(hipStreamSynchronize(st));
    if (*reinterpret_cast<int*>(ctx->h_pinned + 3 + kLenSums))
      return fail(KSH_INVALID_ARGUMENT, "the canonical set holds a k-mer that is its own reverse "
                                        "complement (even k): not supported");
    static_assert(offsetof(EncCtl, rc_marks) == offsetof(EncCtl, self_rc) + 4, "the word behind self_rc, in the same copy");
    if (reinterpret_cast<int*>(ctx->h_pinned + 3 + kLenSums)[1]) p->routes |= KSH_ROUTE_RC_MARKS_GROUPS;
    p->stamped = stamped;
    p->rinfo = rinfo;
    p->chain_info = chain_info;
    if (stamped) break;
    // a non-branching loop has no end k-mer for k_choose_ends to find: once more, with stamps
    int64_t accounted = 0;
    for (int i = 0; i < kLenSums; i++) accounted += ctx->h_pinned[2 + i];
    if (accounted == n && *reinterpret_cast<int*>(ctx->h_pinned + 2 + kLenSums) == 0) break;
  }
  const int64_t n0 = ctx->h_pinned[0] & 0xFFFFFFFF, n1 = ctx->h_pinned[0] >> 32;
  const int64_t n2 = ctx->h_pinned[1] & 0xFFFFFFFF, n3 = ctx->h_pinned[1] >> 32;
  const int64_t n_u = n0 + n1 + n2 + n3;
  p->n_u = n_u;

  // unitig-level block
  const size_t ub = 10 * al(size_t(n_u) * 4) + al(size_t(8 * n_u) * 4) + 2 * al(size_t(2 * n_u) * 4) +
                    al(size_t(2 * n_u) * 8) + 3 * al(size_t(n_u)) + 4 * al(size_t(n_u + 1) * 8) +
                    3 * al(size_t(n_u) * 4) + 4096;
  KSH_TRY(pool_alloc(ctx, ub, reinterpret_cast<void**>(&p->ublock)));
  at = p->ublock;
  p->u_head = carve<uint32_t>(at, size_t(n_u));
  p->u_first = carve<uint32_t>(at, size_t(n_u));
  p->u_last = carve<uint32_t>(at, size_t(n_u));
  p->u_len = carve<uint32_t>(at, size_t(n_u));
  p->u_sid = carve<uint32_t>(at, size_t(n_u));
  p->u_koff = carve<uint32_t>(at, size_t(n_u));
  p->lens = carve<uint32_t>(at, size_t(n_u));
  p->sc_nodes = carve<uint32_t>(at, size_t(n_u));
  p->sc_parent = carve<uint32_t>(at, size_t(n_u));
  p->sc_rank = carve<uint32_t>(at, size_t(n_u));
  p->edges = carve<uint32_t>(at, size_t(8 * n_u));
  p->mate = carve<uint32_t>(at, size_t(2 * n_u));
  p->best_w = carve<uint32_t>(at, size_t(2 * n_u));
  p->best_prio = carve<unsigned long long>(at, size_t(2 * n_u));
  p->visited = carve<uint8_t>(at, size_t(n_u));
  p->scls = carve<uint8_t>(at, size_t(n_u));
  p->u_flip = carve<uint8_t>(at, size_t(n_u));
  p->s_nk = carve<int64_t>(at, size_t(n_u + 1));
  p->sc01 = carve<int64_t>(at, size_t(n_u + 1));
  p->sc2 = carve<int64_t>(at, size_t(n_u + 1));
  p->str_start = carve<int64_t>(at, size_t(n_u + 1));
  p->sc_used = &ctl->sc_used;

  hipLaunchKernelGGL(k_unitig_fill, dim3(unsigned(n_hblocks)), dim3(256), 0, st, p->hcls, b01, b23, n, n0,
                     n0 + n1, n0 + n1 + n2, p->ori, p->hlen, p->hlast, p->uid, p->u_head, p->u_first,
                     p->u_last, p->u_len);

  int64_t ns = n_u;
  int walk_rounds = 0;
  if (mode == 1) {
    hipLaunchKernelGGL(k_unitig_strings, dim3(nblk(n_u)), dim3(256), 0, st, p->u_len, n_u, g->k,
                       p->u_sid, p->u_koff, p->u_flip, p->lens, p->str_start);
  } else {
    hipLaunchKernelGGL((k_edges<KeyT>), dim3(nblk(2 * n_u)), dim3(256), 0, st, set, 2 * n_u, directed,
                       p->u_first, p->u_last, p->head, p->uid, p->edges, p->mate);
    p->rounds = 0;
    const bool slow = mode == 2;
    if (slow) {
      hipLaunchKernelGGL(k_match_slow, dim3(1), dim3(64), 0, st, p->edges, p->mate, n_u);
    } else {
      // rounds in batches, one look at the flags per batch: four rounds first (every round past the last one that
      // found something is two launches that return at once, some 9 us each), then eight at a time
      for (bool more = true; more;) {
        if (p->rounds) KSH_HIP(hipMemsetAsync(ctl->match_live, 0, sizeof(ctl->match_live), st));
        const int batch = p->rounds ? kMatchBatch : kMatchFirst;
        for (int r = 0; r < batch; r++) {
          hipLaunchKernelGGL(k_match_best, dim3(nblk(2 * n_u)), dim3(256), 0, st, p->edges, p->mate, 2 * n_u, directed,
                             p->best_prio, p->best_w, r ? &ctl->match_live[r - 1] : nullptr, &ctl->match_live[r]);
          hipLaunchKernelGGL(k_match_commit, dim3(nblk(2 * n_u)), dim3(256), 0, st, p->best_prio, p->best_w, 2 * n_u,
                             &ctl->match_live[r], p->mate);
        }
        KSH_HIP(hipMemcpyAsync(ctx->h_pinned, ctl->match_live, batch * sizeof(int), hipMemcpyDeviceToHost, st));
        KSH_HIP(hipStreamSynchronize(st));
        const int* live = reinterpret_cast<const int*>(ctx->h_pinned);
        int ran = 0;
        while (ran < batch && live[ran]) ran++;
        more = ran == batch;
        p->rounds += more ? batch : ran + 1;  // (the round that found nothing counts, as before)
        if (p->rounds > 100000) return fail(KSH_INTERNAL, "matching did not converge");
      }
      // the path extension of fast = false never closes a loop; the greedy matching can:
      // components of the chosen edges (parallel union-find), the ones without a terminal are loops
      DevDsu dsu{reinterpret_cast<unsigned long long*>(p->sc01)};  // sc01 is only filled by k_string_counts
      uint8_t* has_terminal = p->scls;                             // scls only by k_string_starts
      hipLaunchKernelGGL(k_dsu_init, dim3(nblk(n_u)), dim3(256), 0, st, dsu.a, n_u, has_terminal);
      hipLaunchKernelGGL(k_dsu_unite_mates, dim3(nblk(2 * n_u)), dim3(256), 0, st, dsu, p->mate, 2 * n_u);
      hipLaunchKernelGGL(k_dsu_mark_terminals, dim3(nblk(n_u)), dim3(256), 0, st, dsu, p->mate, n_u, has_terminal);
      hipLaunchKernelGGL(k_dsu_open_paths, dim3(nblk(n_u)), dim3(256), 0, st, dsu, has_terminal, n_u, p->visited);
      hipLaunchKernelGGL(k_loop_cut, dim3(unsigned((n_u + 63) / 64)), dim3(64), 0, st, p->mate, n_u,
                         directed, p->visited, p->sc_nodes, p->sc_parent, p->sc_rank, p->sc_used);
    }
    // the walks over the (now loop-free) path cover, by pointer jumping; the words reuse the
    // matching's priorities, the string ids of the starts its candidates
    unsigned long long* walk = p->best_prio;
    uint32_t* sid_at = p->best_w;
    hipLaunchKernelGGL(k_walk_init, dim3(nblk(2 * n_u)), dim3(256), 0, st, p->mate, p->u_len, 2 * n_u, walk);
    // log5(2 n_u) + 3 launches end every walk of a loop-free cover; enqueued at once (a round after the last
    // one that changed anything returns at its first load), the last flag is looked at with the sizes below
    walk_rounds = 3;  // (four hops per launch: log5 of the states, and spare)
    for (int64_t x = 2 * n_u; x > 1; x /= (kJumpHops + 1)) walk_rounds++;
    walk_rounds = std::min(walk_rounds, kWalkRoundsMax);
    for (int round = 0; round < walk_rounds; round++)
      hipLaunchKernelGGL(k_walk_jump, dim3(nblk(2 * n_u)), dim3(256), 0, st, 2 * n_u, walk,
                         round ? &ctl->walk_live[round - 1] : nullptr, &ctl->walk_live[round]);
    hipLaunchKernelGGL(k_string_starts, dim3(nblk(n_u)), dim3(256), 0, st, p->mate, p->u_len, walk, n_u,
                       directed, p->scls, p->s_nk, p->str_start);
    hipLaunchKernelGGL(k_string_counts, dim3(nblk(n_u)), dim3(256), 0, st, p->scls, n_u, slow, p->sc01,
                       p->sc2);
    arena_reset(ctx);
    int64_t* d_t2 = ctl->t2;
    KSH_TRY(scan_exclusive_i64(ctx, p->sc01, p->sc01, n_u, d_t2));
    KSH_TRY(scan_exclusive_i64(ctx, p->sc2, p->sc2, n_u, d_t2 + 1));
    hipLaunchKernelGGL(k_string_ids, dim3(nblk(n_u)), dim3(256), 0, st, n_u, p->scls, p->sc01, p->sc2, p->s_nk, d_t2,
                       slow, g->k, sid_at, p->lens, p->str_start);
    hipLaunchKernelGGL(k_string_assign, dim3(nblk(n_u)), dim3(256), 0, st, p->u_len, walk, n_u, p->scls, sid_at,
                       slow, p->u_sid, p->u_koff, p->u_flip, &ctl->walk_live[walk_rounds - 1]);
  }
  // string starts in bases (a scan over n_u slots: the slots past the last string hold zero)
  arena_reset(ctx);
  KSH_TRY(scan_exclusive_i64(ctx, p->str_start, p->str_start, n_u, &ctl->n_bases));
  hipLaunchKernelGGL(k_unitig_place, dim3(nblk(n_u)), dim3(256), 0, st, p->u_len, p->u_sid, p->u_koff,
                     p->u_flip, p->str_start, p->lens, p->u_head, n_u, reinterpret_cast<UnitigPlace*>(p->c01),
                     walk_rounds > 0 ? &ctl->walk_live[walk_rounds - 1] : nullptr);
  KSH_HIP(hipGetLastError());
  // one look at everything the unitig level left behind: strings by class, bases, the last walk round's flag
  // (one copy: the block from t2 to its end, 436 bytes of the 512 pinned ones)
  constexpr size_t kTail = sizeof(EncCtl) - offsetof(EncCtl, t2);
  static_assert(kTail <= 64 * sizeof(int64_t) && offsetof(EncCtl, n_bases) == offsetof(EncCtl, t2) + 16, "pinned read-back");
  KSH_HIP(hipMemcpyAsync(ctx->h_pinned, ctl->t2, kTail, hipMemcpyDeviceToHost, st));
  KSH_HIP(hipStreamSynchronize(st));
  const int* walk_live_host = reinterpret_cast<const int*>(reinterpret_cast<const char*>(ctx->h_pinned) +
                                                           (offsetof(EncCtl, walk_live) - offsetof(EncCtl, t2)));
  if (walk_rounds > 0 && walk_live_host[walk_rounds - 1])
    return fail(KSH_INTERNAL, "the path cover still holds a loop");
  {
    const char* tail = reinterpret_cast<const char*>(ctx->h_pinned);
    const auto at_tail = [&](size_t member_off) { return tail + (member_off - offsetof(EncCtl, t2)); };
    if (*reinterpret_cast<const int*>(at_tail(offsetof(EncCtl, rc_batched)))) p->routes |= KSH_ROUTE_RC_BATCHED;
    const unsigned int* lc = reinterpret_cast<const unsigned int*>(at_tail(offsetof(EncCtl, long_count)));
    if (!p->stamped && (lc[0] || lc[1])) p->routes |= KSH_ROUTE_LONG_STRETCHES;
    if (p->rounds > kMatchFirst) p->routes |= KSH_ROUTE_MATCH_MORE_ROUNDS;
  }
  if (mode != 1) ns = (ctx->h_pinned[0] & 0xFFFFFFFF) + (ctx->h_pinned[0] >> 32) + ctx->h_pinned[1];
  p->n_strings = ns;
  p->n_bases = ctx->h_pinned[2];
  *n_strings = p->n_strings;
  *n_bases = p->n_bases;
  return KSH_OK;
}

template <typename KeyT>
int encode_write_t(ksh_ctx* ctx, uint64_t* d_words, uint32_t* d_lens) {
  EncPlan* p = static_cast<EncPlan*>(ctx->enc_state);
  if (p->n == 0) return KSH_OK;
  const ksh_geom* g = &p->g;
  const int64_t n = p->n;
  DevSet<KeyT> set{p->set.d_offsets, static_cast<const KeyT*>(p->set.d_keys), n_buckets(g), n, g->k,
                   key_bits(g)};
  set.fine = p->fine;
  set.fine_bits = p->fine_bits;
  set.coarse = p->coarse;
  hipStream_t st = ctx->stream;
  // byte staging aliases the array the staged neighbour probe kept its partial results in (dead
  // since then; 2n * 4 bytes >= n_bases needs checking)
  const size_t need = size_t(p->n_bases) + 64;
  uint8_t* bytes;
  void* tmp = nullptr;
  if (need <= size_t(2 * n) * 4) {
    bytes = reinterpret_cast<uint8_t*>(p->link);
  } else {
    KSH_TRY(pool_alloc(ctx, need, &tmp));
    bytes = static_cast<uint8_t*>(tmp);
  }
  {
    Timer timer(ctx, 5, n);
    const UnitigPlace* place = reinterpret_cast<const UnitigPlace*>(p->c01);
    const int64_t n_sampled = (n + kRulerEvery - 1) / kRulerEvery;
#define KSH_EMIT(W, R)                                                                                         \
  do {                                                                                                         \
    if (p->stamped) {                                                                                          \
      hipLaunchKernelGGL((k_emit<KeyT, W>), dim3(nblk(n)), dim3(256), 0, st, set, p->head, p->pos, p->ori,     \
                         place, bytes);                                                                        \
    } else if (p->log_rulers.hdr) {                                                                            \
      hipLaunchKernelGGL((k_emit_log_rulers<KeyT, R>), dim3(kLongBlocks + nblk(p->log_rulers.n_walkers)),      \
                         dim3(256), 0, st, set, p->nbr, p->rinfo, p->chain_info, p->log_rulers, p->log_heads,  \
                         p->directed, p->ends, place, bytes);                                                  \
      hipLaunchKernelGGL((k_emit_log_heads<KeyT, R>), dim3(nblk(p->log_heads.n_walkers)), dim3(256), 0, st,    \
                         set, p->nbr, p->rinfo, p->chain_info, p->log_heads, p->directed, p->ends, place,      \
                         bytes);                                                                               \
    } else {                                                                                                   \
      hipLaunchKernelGGL((k_emit_rulers<KeyT, R>), dim3(nblk(n_sampled)), dim3(256), 0, st, set, p->nbr,       \
                         p->rinfo, p->chain_info, p->directed, place, bytes);                                  \
      hipLaunchKernelGGL((k_emit_heads<KeyT, R>), dim3(nblk(p->n_ends)), dim3(256), 0, st, set, p->nbr,        \
                         p->rinfo, p->chain_info, p->directed, p->ends, p->n_ends, place, bytes);              \
    }                                                                                                          \
  } while (0)
    if (g->k >= 16)
      KSH_EMIT(16, 16);
    else if (g->k >= 8)
      KSH_EMIT(4, 8);
    else if (g->k >= 4)
      KSH_EMIT(4, 4);
    else
      KSH_EMIT(1, 0);
#undef KSH_EMIT
  }
  const int64_t n_words = (p->n_bases + 31) / 32;
  hipLaunchKernelGGL(k_pack, dim3(nblk(n_words)), dim3(256), 0, st, bytes, p->n_bases, n_words,
                     d_words);
  KSH_HIP(hipMemcpyAsync(d_lens, p->lens, size_t(p->n_strings) * 4, hipMemcpyDeviceToDevice, st));
  KSH_HIP(hipGetLastError());
  if (tmp) pool_free(ctx, tmp);
  return KSH_OK;
}

}  // namespace ksh

using namespace ksh;

extern "C" {

int ksh_spss_encode_plan(ksh_ctx* ctx, const ksh_geom* g, const ksh_set_view* set, int canonical_flag,
                         int mode, int64_t* n_strings, int64_t* n_bases) {
  if (!ctx || !set || !n_strings || !n_bases) return fail(KSH_INVALID_ARGUMENT, "NULL argument");
  KSH_TRY(check_geom(g));
  if (mode < 0 || mode > 2)
    return fail(KSH_INVALID_ARGUMENT, "mode must be 0 (SPSS), 1 (unitigs) or 2 (SPSS, fast = false)");
  const bool directed = !canonical_flag;
  if (directed && mode == 2) mode = 0;  // FromKmerSet ignores `fast` for non-canonical sets
  if (set->n_keys < 0 || !set->d_offsets || (set->n_keys > 0 && !set->d_keys))
    return fail(KSH_INVALID_ARGUMENT, "bad set view");
  KSH_HIP(hipSetDevice(ctx->device));
  return KSH_BY_KEY(g->key_bytes, encode_plan_t, ctx, g, set, directed, mode, n_strings, n_bases);
}

int ksh_spss_encode_write(ksh_ctx* ctx, uint64_t* d_words, uint32_t* d_lens) {
  if (!ctx) return fail(KSH_INVALID_ARGUMENT, "ctx is NULL");
  EncPlan* p = static_cast<EncPlan*>(ctx->enc_state);
  if (!p) return fail(KSH_FAILED_PRECONDITION, "ksh_spss_encode_write without ksh_spss_encode_plan");
  if (p->n > 0 && (!d_words || !d_lens)) return fail(KSH_INVALID_ARGUMENT, "NULL output");
  KSH_HIP(hipSetDevice(ctx->device));
  return KSH_BY_KEY(p->g.key_bytes, encode_write_t, ctx, d_words, d_lens);
}

int ksh_spss_encode_routes(ksh_ctx* ctx, int64_t* routes) {
  if (!ctx || !routes) return fail(KSH_INVALID_ARGUMENT, "NULL argument");
  EncPlan* p = static_cast<EncPlan*>(ctx->enc_state);
  if (!p) return fail(KSH_FAILED_PRECONDITION, "no encode plan");
  *routes = p->routes;
  return KSH_OK;
}

int ksh_spss_encode_stats(ksh_ctx* ctx, int64_t stats[4]) {
  if (!ctx || !stats) return fail(KSH_INVALID_ARGUMENT, "NULL argument");
  EncPlan* p = static_cast<EncPlan*>(ctx->enc_state);
  if (!p) return fail(KSH_FAILED_PRECONDITION, "no encode plan");
  stats[0] = p->n_u;
  stats[1] = p->rounds;
  stats[2] = p->n_strings;
  stats[3] = p->n_bases;
  return KSH_OK;
}

#ifdef KSH_TRACE
int ksh_debug_set_probe_trace(void* d_buf, long long rows) {
  unsigned long long* p = static_cast<unsigned long long*>(d_buf);
  if (hipMemcpyToSymbol(HIP_SYMBOL(ksh::g_probe_trace), &p, sizeof(p)) != hipSuccess) return 13;
  return hipMemcpyToSymbol(HIP_SYMBOL(ksh::g_probe_trace_rows), &rows, sizeof(rows)) == hipSuccess ? 0 : 13;
}
#endif

int ksh_spss_encode_release(ksh_ctx* ctx) {
  if (ctx) free_plan(ctx);
  return KSH_OK;
}

}  // extern "C"
