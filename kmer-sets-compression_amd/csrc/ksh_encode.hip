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
//   k_adjacency   8 membership probes per k-mer (4 Next, 4 Prev, forward or reverse
//                 complement) -> per side: none / the single neighbour / many
//   k_links       mutual singles
//   k_ruler_*     chains of states ranked through a sparse ruler set: (end, distance to end)
//   k_choose      per k-mer: the chain that starts at the larger end (spss.h:511,555)
//   k_loops       non-branching loops, spelled from their smallest k-mer (spss.h:585-610)
//   k_head_counts / scans / k_unitig_fill   unitig ids in the reference's push order
//   k_edges       <= 4 edges per unitig side, in the reference's enumeration order
//   k_match_*     lexicographically-first maximal matching by rounds of mutual minima
//                 == the sequential greedy sweep of spss.h:1445-1499
//   k_cover_mark / k_loop_cut    loops of the path cover, cut where the reference's
//                 union-by-rank root says (spss.h:1541-1647)
//   k_string_*    stitch order and orientation (spss.h:1649-1829)
//   k_emit / k_pack   bases -> 2-bit words, len - K per string
//
// All of it is integer gather/scatter work bounded by HBM random-access rate; no MFMA.
#include "ksh_internal.h"
#include "ksh_kmer.h"

#include <algorithm>

namespace ksh {

// Sampled rulers: both states of every kRulerEvery-th k-mer (E2).
#ifndef KSH_RULER_SHIFT
#define KSH_RULER_SHIFT 4
#endif
constexpr int kRulerShift = KSH_RULER_SHIFT;
constexpr uint32_t kRulerEvery = 1u << kRulerShift;

constexpr uint32_t kNone = 0xFFFFFFFFu;
constexpr uint32_t kMulti = 0xFFFFFFFEu;
constexpr uint64_t kUnset = ~uint64_t(0);

// ---------------------------------------------------------------------------------- E1
// Fine index of a set: one workgroup per bucket walks its sorted keys once and records
// where the top fine_bits key bits change.
template <typename KeyT>
__global__ __launch_bounds__(256) void k_fine_index(const int64_t* __restrict__ off,
                                                     const KeyT* __restrict__ keys, int key_bits,
                                                     int fine_bits, uint32_t* __restrict__ fine,
                                                     int64_t n_buckets) {
  const int64_t b = blockIdx.x;
  const int64_t lo = off[b], hi = off[b + 1];
  const int kSlices = 1 << fine_bits;
  const int sh = key_bits - fine_bits;
  uint32_t* f = fine + (b << fine_bits);
  if (lo == hi) {
    for (int sub = threadIdx.x; sub < kSlices; sub += 256) f[sub] = uint32_t(lo);
  } else {
    for (int64_t i = lo + threadIdx.x; i < hi; i += 256) {
      const int cur = int(uint64_t(keys[i]) >> sh);
      const int prev = i > lo ? int(uint64_t(keys[i - 1]) >> sh) : -1;
      for (int sub = prev + 1; sub <= cur; sub++) f[sub] = uint32_t(i);
    }
    const int last = int(uint64_t(keys[hi - 1]) >> sh);
    for (int sub = last + 1 + int(threadIdx.x); sub < kSlices; sub += 256) f[sub] = uint32_t(hi);
  }
  if (b == n_buckets - 1 && threadIdx.x == 0) fine[n_buckets << fine_bits] = uint32_t(hi);
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
  const int k = set.k;
  int cnt[2] = {0, 0};
  uint32_t single[2] = {kNone, kNone};
  if (kDirected) {
    // Non-canonical sets (GetUnitigs, spss.h:76-96): k-mers as they are, side 1 = outgoing edges
    // Next(x, .) (four consecutive values, one bounded search), side 0 = incoming edges
    // Prev(x, .) (four buckets, probed one by one); no edge flips an orientation.
    set.for_group4(kmer_next(x, k, 0), [&](int64_t idx) {
      if (idx == t) return;  // next != kmer (spss.h:81)
      cnt[1]++;
      single[1] = uint32_t(idx) << 1;
    });
#pragma unroll
    for (int c = 0; c < 4; c++) {
      const uint64_t z = kmer_prev(x, k, c);
      if (z == x) continue;  // prev != kmer (spss.h:91)
      const int64_t idx = set.find(z);
      if (idx < 0) continue;
      cnt[0]++;
      single[0] = uint32_t(idx) << 1;
    }
  } else {
  // The 8 candidates of the reference (Next / Prev of x, each as is or reverse-complemented,
  // spss.h:238-273) in two kinds.  The four Next(x, c) are consecutive values, and so are the
  // four Next(rc(x), c) = rc(Prev(x, 3 - c)): one bounded search each finds all of their
  // members that are in the set (only a canonical k-mer can be).  The other candidates,
  // Prev(x, c) and Prev(rc(x), c) = rc(Next(x, 3 - c)), sit in four different buckets and are
  // probed one by one, and only when they are the canonical form.
  const uint64_t rx = revcomp(x, k);
  if (rx == x) *self_rc = 1;
#pragma unroll
  for (int side = 0; side < 2; side++) {
    // group: side 1 -> Next(x, .) (neighbour as is);  side 0 -> Next(rc(x), .) (neighbour
    // reverse-complemented, i.e. a same-side edge)
    set.for_group4(kmer_next(side ? x : rx, k, 0), [&](int64_t idx) {
      if (idx == t) return;  // kmer != next (spss.h:242,248)
      cnt[side]++;
      single[side] = (uint32_t(idx) << 1) | (side ? 0u : 1u);
    });
    // singles: side 1 -> Prev(rc(x), c) = rc(Next(x, 3 - c));  side 0 -> Prev(x, c) as is
    const uint64_t base = side ? rx : x;
#pragma unroll
    for (int c = 0; c < 4; c++) {
      const uint64_t z = kmer_prev(base, k, c);
      if (revcomp(z, k) < z) continue;  // not canonical: cannot be in the set
      if (z == x) continue;
      const int64_t idx = set.find(z);
      if (idx < 0) continue;
      cnt[side]++;
      single[side] = (uint32_t(idx) << 1) | (side ? 1u : 0u);
    }
  }
  }
  nbr[2 * t] = cnt[0] == 0 ? kNone : (cnt[0] == 1 ? single[0] : kMulti);
  nbr[2 * t + 1] = cnt[1] == 0 ? kNone : (cnt[1] == 1 ? single[1] : kMulti);
}


// One thread per k-mer: both links; the chain-rank records of its two states start unset.
__global__ __launch_bounds__(256) void k_links(const uint32_t* __restrict__ nbr, int64_t n,
                                                uint32_t* __restrict__ link,
                                                unsigned long long* __restrict__ info,
                                                uint8_t* __restrict__ start_flags) {
  const int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const uint2 nb2 = reinterpret_cast<const uint2*>(nbr)[t];
  const uint32_t v[2] = {nb2.x, nb2.y};
  uint32_t out[2];
#pragma unroll
  for (uint32_t side = 0; side < 2; side++) {
    out[side] = kNone;
    if (v[side] < kMulti) {
      const uint32_t y = v[side] >> 1, same = v[side] & 1;
      const uint32_t facing = same ? side : side ^ 1;
      if (nbr[2 * int64_t(y) + facing] < kMulti) out[side] = v[side];
    }
  }
  reinterpret_cast<uint2*>(link)[t] = make_uint2(out[0], out[1]);
  reinterpret_cast<ulonglong2*>(info)[t] = make_ulonglong2(kUnset, kUnset);  // chain-rank records
  // which of its two states start a chain without being a sampled ruler (k_ruler_heads):
  // state 2t enters through side 0, state 2t + 1 through side 1
  const bool sampled = (t & (kRulerEvery - 1)) == 0;
  start_flags[t] = sampled ? 0 : uint8_t((out[0] == kNone ? 1 : 0) | (out[1] == kNone ? 2 : 0));
}

// ---------------------------------------------------------------------------------- E2
// Chains of states are ranked with a sparse ruler set instead of one serial walk per
// chain (a 10^7-k-mer unitig would otherwise be a 10^7-step dependent walk):
//   rulers = both states of every 16th k-mer (index order is unrelated to chain order);
//   k_ruler_walk  every sampled ruler walks to the next one (about 16 steps), stamping the
//                 states it passes with (ruler, offset);  k_ruler_heads does the same for the
//                 head segment of a chain that starts between two sampled k-mers;
//   k_ruler_jump  pointer jumping over the dense ruler array only (1/16 of the states), until
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

// Sampled rulers are both states of every 16th k-mer (index order is unrelated to chain
// order, so this is as good as a hash), which makes them enumerable without compaction:
// dense thread i <-> state 32 * (i >> 1) + (i & 1).  The other rulers are chain starts and
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
// The dense ruler array is 1/16 of the states (L2/MALL resident at 10^8 k-mers), so pointer
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

// One thread per sampled ruler (dense index i <-> state 32 * (i >> 1) + (i & 1)).
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

// One thread per k-mer; acts on those of its two states that start a chain without being a
// sampled ruler (flags from k_links).  One walk from the start S to the first sampled ruler ahead,
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
                                                      const uint8_t* __restrict__ start_flags,
                                                      int64_t n,
                                                      unsigned long long* __restrict__ rec,
                                                      unsigned long long* __restrict__ chain_info) {
  const int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const uint32_t flags = start_flags[t];
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

// Pointer jumping over the dense ruler array until every ruler on a path points at its end.
__global__ __launch_bounds__(256) void k_ruler_jump(int64_t n_dense,
                                                     unsigned long long* __restrict__ rinfo,
                                                     int* __restrict__ changed) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n_dense) return;
  const uint64_t mine = rinfo[i];
  if (mine & kEndFlag) return;
  const uint64_t theirs = rinfo[dense_index(uint32_t(mine))];
  const uint32_t dist = uint32_t((mine >> 32) & 0x7FFFFFFFu) + uint32_t((theirs >> 32) & 0x7FFFFFFFu);
  rinfo[i] = (theirs & kEndFlag) | (uint64_t(dist & 0x7FFFFFFFu) << 32) | uint32_t(theirs);
  *changed = 1;
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
  hcls[t] = 0xFF;
  const uint32_t fwd_start = e1 >> 1, fwd_end = e0 >> 1;
  const uint32_t d = (directed || fwd_start >= fwd_end) ? 0u : 1u;
  const uint32_t start_state = (d ? e0 : e1) ^ 1;
  const uint32_t p = d ? d0 : d1;
  head[t] = start_state >> 1;
  pos[t] = p;
  ori[t] = uint8_t(d);
  if (p == 0) {
    hcls[t] = (directed || fwd_start == fwd_end) ? 0 : ((start_state & 1) == 0 ? 1 : 2);
    hlen[t] = d0 + d1 + 1;
    hlast[t] = d ? e1 : e0;
  }
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
  uint32_t p = 0, last = start;
  s = start;
  do {
    const uint32_t y = s >> 1;
    head[y] = uint32_t(t);
    pos[y] = p;
    ori[y] = uint8_t(s & 1);
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
    u_first[u] = uint32_t(2 * t) | ori[t];
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
                                                uint32_t* __restrict__ edges) {
  const int64_t v = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (v >= n_vertices) return;
  const uint32_t u = uint32_t(v >> 1), side = uint32_t(v & 1);
  const uint32_t st = side ? u_last[u] : u_first[u];
  const int k = set.k;
  const uint64_t x = set.kmer(st >> 1);
  const uint64_t o = (st & 1) ? revcomp(x, k) : x;
#pragma unroll
  for (int c = 0; c < 4; c++) {
    const uint64_t y = side ? kmer_next(o, k, c) : kmer_prev(o, k, c);
    uint32_t out = kNone;
    if (directed) {
      const int64_t i = set.find(y);
      if (i >= 0) {
        const uint32_t u2 = uid[head[i]];
        if (u2 != u) out = 2 * u2 + (side ? 0u : 1u);
      }
    } else {
      const uint64_t r = revcomp(y, k);
      const uint64_t z = y < r ? y : r;
      const int64_t i = set.find(z);
      if (i >= 0) {
        const uint32_t u2 = uid[head[i]];
        if (u2 != u) {
          // side of k-mer z this edge touches
          const uint32_t f = side ? (y == z ? 0u : 1u) : (y == z ? 1u : 0u);
          const uint32_t fs = u_first[u2];
          const uint32_t side2 = ((fs >> 1) == uint32_t(i) && (fs & 1) == f) ? 0u : 1u;
          out = 2 * u2 + side2;
        }
      }
    }
    edges[4 * v + c] = out;
  }
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
                                                     int* __restrict__ any_live) {
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
                                                       uint32_t* __restrict__ mate) {
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
__global__ __launch_bounds__(256) void k_cover_mark(const uint32_t* __restrict__ mate, int64_t n_u,
                                                     uint8_t* __restrict__ visited) {
  const int64_t u = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (u >= n_u) return;
  const bool hl = mate[2 * u] != kNone, hr = mate[2 * u + 1] != kNone;
  if (hl && hr) return;
  uint32_t cur = uint32_t(u);
  bool going_right = !hl;
  int64_t steps = 0;
  while (true) {
    visited[cur] = 1;
    const uint32_t w = mate[2 * int64_t(cur) + (going_right ? 1 : 0)];
    if (w == kNone || steps++ > n_u) break;
    cur = w >> 1;
    going_right = (w & 1) == 0;
  }
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
// scls[u]: 0 = kept walk from a left terminal, 1 = kept walk from a right terminal,
//          2 = isolated unitig, 0xFF = not the start of an output string.
// directed: a string starts at every node without an incoming edge and runs forward
// (spss.h:931-1011); isolated unitigs are class 0 like the others.
__global__ __launch_bounds__(256) void k_string_starts(const uint32_t* __restrict__ mate,
                                                        const uint32_t* __restrict__ u_len,
                                                        int64_t n_u, bool directed,
                                                        uint8_t* __restrict__ scls,
                                                        int64_t* __restrict__ s_nk) {
  const int64_t u = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (u >= n_u) return;
  const bool hl = mate[2 * u] != kNone, hr = mate[2 * u + 1] != kNone;
  uint8_t cls = 0xFF;
  int64_t nk = 0;
  if (directed) {
    if (!hl) {
      uint32_t cur = uint32_t(u);
      int64_t steps = 0;
      while (true) {
        nk += u_len[cur];
        const uint32_t w = mate[2 * int64_t(cur) + 1];
        if (w == kNone || steps++ > n_u) break;
        cur = w >> 1;
      }
      cls = 0;
    }
  } else if (!hl && !hr) {
    cls = 2;
    nk = u_len[u];
  } else if (!hl || !hr) {
    uint32_t cur = uint32_t(u);
    bool going_right = !hl;
    int64_t steps = 0;
    while (true) {
      nk += u_len[cur];
      const uint32_t w = mate[2 * int64_t(cur) + (going_right ? 1 : 0)];
      if (w == kNone || steps++ > n_u) break;
      cur = w >> 1;
      going_right = (w & 1) == 0;
    }
    if (uint32_t(u) <= cur) cls = hl ? 1 : 0;  // path.front().first > path.back().first -> skipped
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

__global__ __launch_bounds__(256) void k_string_assign(
    const uint32_t* __restrict__ mate, const uint32_t* __restrict__ u_len, int64_t n_u,
    const uint8_t* __restrict__ scls, const int64_t* __restrict__ c01,
    const int64_t* __restrict__ c2, const int64_t* __restrict__ s_nk, int64_t base1, int64_t base2,
    bool one_sequence, int k, uint32_t* __restrict__ u_sid, uint32_t* __restrict__ u_koff,
    uint8_t* __restrict__ u_flip, uint32_t* __restrict__ lens, int64_t* __restrict__ str_bases) {
  const int64_t u = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (u >= n_u) return;
  const uint8_t c = scls[u];
  if (c == 0xFF) return;
  int64_t sid;
  if (one_sequence || c == 0) sid = c01[u] & 0xFFFFFFFF;
  else if (c == 1) sid = base1 + (c01[u] >> 32);
  else sid = base2 + c2[u];
  lens[sid] = uint32_t(s_nk[u] - 1);
  str_bases[sid] = s_nk[u] + k - 1;
  uint32_t cur = uint32_t(u);
  // fast = false spells an isolated unitig through FindPath(i, false), i.e. reverse-complemented
  bool going_right = c == 2 ? !one_sequence : c != 1;
  uint32_t koff = 0;
  int64_t steps = 0;
  while (true) {
    u_sid[cur] = uint32_t(sid);
    u_koff[cur] = koff;
    u_flip[cur] = going_right ? 0 : 1;
    koff += u_len[cur];
    if (c == 2) break;
    const uint32_t w = mate[2 * int64_t(cur) + (going_right ? 1 : 0)];
    if (w == kNone || steps++ > n_u) break;
    cur = w >> 1;
    going_right = (w & 1) == 0;
  }
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
                                                       UnitigPlace* __restrict__ place_at_head) {
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

template <typename KeyT>
__global__ __launch_bounds__(256) void k_emit(DevSet<KeyT> set, const uint32_t* __restrict__ head,
                                               const uint32_t* __restrict__ pos,
                                               const uint8_t* __restrict__ ori,
                                               const UnitigPlace* __restrict__ place_at_head,
                                               uint8_t* __restrict__ bytes) {
  __shared__ int64_t s_bucket[2];
  const int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  const uint64_t x = set.kmer_in_block(t, s_bucket);
  if (t >= set.n) return;
  const uint32_t h = head[t];
  if (h == kNone) return;
  const UnitigPlace pl = place_at_head[h];
  const uint32_t flip = pl.flags & 1;
  const uint32_t q = flip ? (pl.len - 1 - pos[t]) : pos[t];
  const int k = set.k;
  const uint64_t o = (uint32_t(ori[t]) ^ flip) ? revcomp(x, k) : x;
  const int64_t at = pl.base + q;
  bytes[at] = uint8_t((o >> (2 * (k - 1))) & 3);
  if ((pl.flags & 2) && q == pl.len - 1) {  // last k-mer of the string: its remaining K - 1 bases
    for (int i = 1; i < k; i++) bytes[at + i] = uint8_t((o >> (2 * (k - 1 - i))) & 3);
  }
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
  // unitig level (own allocation)
  char* ublock = nullptr;
  uint32_t *u_head = nullptr, *u_first = nullptr, *u_last = nullptr, *u_len = nullptr,
           *edges = nullptr, *mate = nullptr, *best_w = nullptr, *u_sid = nullptr,
           *u_koff = nullptr, *lens = nullptr, *sc_nodes = nullptr, *sc_parent = nullptr,
           *sc_rank = nullptr;
  unsigned long long *best_prio = nullptr, *sc_used = nullptr;
  uint8_t *visited = nullptr, *scls = nullptr, *u_flip = nullptr;
  int64_t *s_nk = nullptr, *sc01 = nullptr, *sc2 = nullptr, *str_start = nullptr;
  int* any_live = nullptr;
  int rounds = 0;
};

inline size_t al(size_t x) { return (x + 255) & ~size_t(255); }
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
  // slices of about 1.5 keys (measured on 10^7- and 10^8-key sets: 9 and 12 bits are the
  // fastest there; fewer bits mean longer slices, more bits a bigger index to build)
  int fine_bits = 0;
  // (at least four values per slice: the four consecutive candidates of a group probe share one)
  while (fine_bits < 13 && fine_bits + 2 < key_bits(g) && (int64_t(3) << fine_bits) < 2 * (n / nb + 1)) fine_bits++;
  const bool use_fine = fine_bits >= 2;
  const size_t fine_entries = use_fine ? (size_t(nb) << fine_bits) + 1 : 0;
  const size_t bytes = 2 * al(size_t(2 * n) * 4) + al(size_t(2 * n) * 8) + 5 * al(size_t(n) * 4) +
                       2 * al(size_t(n)) + 2 * al(size_t(n) * 8) + al(fine_entries * 4) + 4096;
  KSH_TRY(slot_reserve(ctx, kSlotEncode, bytes));
  KSH_TRY(arena_reserve(ctx, size_t(n / 256 + 4096) * 8 * 2 + (1u << 16)));
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

  DevSet<KeyT> set{sv->d_offsets, static_cast<const KeyT*>(sv->d_keys), nb, n, g->k, key_bits(g)};
  hipStream_t st = ctx->stream;
  if (use_fine) {
    hipLaunchKernelGGL((k_fine_index<KeyT>), dim3(unsigned(nb)), dim3(256), 0, st, sv->d_offsets,
                       static_cast<const KeyT*>(sv->d_keys), key_bits(g), fine_bits, p->fine, nb);
    set.fine = p->fine;
    set.fine_bits = fine_bits;
    p->fine_bits = fine_bits;
  }
  int* flags = static_cast<int*>(arena_alloc(ctx, 16));  // [0] pointer-jumping progress, [1] self_rc
  if (!flags) return fail(KSH_INTERNAL, "scratch arena too small");
  KSH_HIP(hipMemsetAsync(flags, 0, 16, st));
  {
    Timer timer(ctx, 3, n);
    if (directed)
      hipLaunchKernelGGL((k_adjacency<KeyT, true>), dim3(nblk(n)), dim3(256), 0, st, set, p->nbr, flags + 1);
    else
      hipLaunchKernelGGL((k_adjacency<KeyT, false>), dim3(nblk(n)), dim3(256), 0, st, set, p->nbr, flags + 1);
  }
  hipLaunchKernelGGL(k_links, dim3(nblk(n)), dim3(256), 0, st, p->nbr, n, p->link, p->info,
                     p->hcls);  // hcls doubles as the start-flag bytes until k_choose
  {
    int* changed = flags;
    const int64_t ns2 = 2 * n;
    const int64_t n_dense = 2 * ((n + kRulerEvery - 1) / kRulerEvery);
    unsigned long long* rinfo = reinterpret_cast<unsigned long long*>(p->c01);  // n_dense * 8 <= 8n
    {
      Timer timer(ctx, 4, n);
      hipLaunchKernelGGL(k_ruler_walk, dim3(nblk(n_dense)), dim3(256), 0, st, p->link, ns2, n_dense, rinfo,
                         p->info);
    }
    hipLaunchKernelGGL(k_ruler_heads, dim3(nblk(n)), dim3(256), 0, st, p->link, p->hcls, n, p->info,
                       reinterpret_cast<unsigned long long*>(p->c23));  // c23: chain_info until k_choose is done
    int max_rounds = 2;
    for (int64_t x = n_dense; x > 1; x >>= 1) max_rounds++;
    for (int round = 0; round < max_rounds;) {
      KSH_HIP(hipMemsetAsync(changed, 0, sizeof(int), st));
      for (int b = 0; b < 4 && round < max_rounds; b++, round++)
        hipLaunchKernelGGL(k_ruler_jump, dim3(nblk(n_dense)), dim3(256), 0, st, n_dense, rinfo, changed);
      KSH_HIP(hipMemcpyAsync(ctx->h_pinned, flags, 2 * sizeof(int), hipMemcpyDeviceToHost, st));
      KSH_HIP(hipStreamSynchronize(st));
      if (reinterpret_cast<int*>(ctx->h_pinned)[1])
        return fail(KSH_INVALID_ARGUMENT, "the canonical set holds a k-mer that is its own reverse "
                                          "complement (even k): not supported");
      if (reinterpret_cast<int*>(ctx->h_pinned)[0] == 0) break;
    }
  }
  hipLaunchKernelGGL(k_choose, dim3(nblk(n)), dim3(256), 0, st, p->info,
                     reinterpret_cast<const unsigned long long*>(p->c01),
                     reinterpret_cast<const unsigned long long*>(p->c23), n, directed, p->head, p->pos, p->ori,
                     p->hcls, p->hlen, p->hlast);
  hipLaunchKernelGGL(k_loops, dim3(nblk(n)), dim3(256), 0, st, p->link, n, p->head, p->pos, p->ori,
                     p->hcls, p->hlen, p->hlast);
  const int64_t n_hblocks = (n + kHeadSpan - 1) / kHeadSpan;
  int64_t* b01 = p->c23;              // per-workgroup head counts (n / 2048 values each) live in
  int64_t* b23 = p->c23 + n_hblocks;  // the front of c23; c01 still holds the ruler records
  hipLaunchKernelGGL(k_head_block_counts, dim3(unsigned(n_hblocks)), dim3(256), 0, st, p->hcls, n, b01, b23);
  int64_t* d_tot = static_cast<int64_t*>(arena_alloc(ctx, 16));
  if (!d_tot) return fail(KSH_INTERNAL, "scratch arena too small");
  KSH_TRY(scan_exclusive_i64(ctx, b01, b01, n_hblocks, d_tot));
  KSH_TRY(scan_exclusive_i64(ctx, b23, b23, n_hblocks, d_tot + 1));
  KSH_HIP(hipGetLastError());
  KSH_HIP(hipMemcpyAsync(ctx->h_pinned, d_tot, 16, hipMemcpyDeviceToHost, st));
  KSH_HIP(hipStreamSynchronize(st));
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
  p->sc_used = carve<unsigned long long>(at, 1);
  p->any_live = carve<int>(at, 1);

  hipLaunchKernelGGL(k_unitig_fill, dim3(unsigned(n_hblocks)), dim3(256), 0, st, p->hcls, b01, b23, n, n0,
                     n0 + n1, n0 + n1 + n2, p->ori, p->hlen, p->hlast, p->uid, p->u_head, p->u_first,
                     p->u_last, p->u_len);

  int64_t ns = n_u;
  if (mode == 1) {
    hipLaunchKernelGGL(k_unitig_strings, dim3(nblk(n_u)), dim3(256), 0, st, p->u_len, n_u, g->k,
                       p->u_sid, p->u_koff, p->u_flip, p->lens, p->str_start);
  } else {
    hipLaunchKernelGGL((k_edges<KeyT>), dim3(nblk(2 * n_u)), dim3(256), 0, st, set, 2 * n_u, directed,
                       p->u_first, p->u_last, p->head, p->uid, p->edges);
    KSH_HIP(hipMemsetAsync(p->mate, 0xFF, size_t(2 * n_u) * 4, st));
    p->rounds = 0;
    const bool slow = mode == 2;
    if (slow) {
      hipLaunchKernelGGL(k_match_slow, dim3(1), dim3(64), 0, st, p->edges, p->mate, n_u);
    } else {
      while (true) {
        KSH_HIP(hipMemsetAsync(p->any_live, 0, sizeof(int), st));
        hipLaunchKernelGGL(k_match_best, dim3(nblk(2 * n_u)), dim3(256), 0, st, p->edges, p->mate,
                           2 * n_u, directed, p->best_prio, p->best_w, p->any_live);
        hipLaunchKernelGGL(k_match_commit, dim3(nblk(2 * n_u)), dim3(256), 0, st, p->best_prio,
                           p->best_w, 2 * n_u, p->mate);
        KSH_HIP(hipMemcpyAsync(ctx->h_pinned, p->any_live, sizeof(int), hipMemcpyDeviceToHost, st));
        KSH_HIP(hipStreamSynchronize(st));
        p->rounds++;
        if (*reinterpret_cast<int*>(ctx->h_pinned) == 0) break;
        if (p->rounds > 100000) return fail(KSH_INTERNAL, "matching did not converge");
      }
      // the path extension of fast = false never closes a loop; the greedy matching can
      KSH_HIP(hipMemsetAsync(p->visited, 0, size_t(n_u), st));
      KSH_HIP(hipMemsetAsync(p->sc_used, 0, 8, st));
      hipLaunchKernelGGL(k_cover_mark, dim3(nblk(n_u)), dim3(256), 0, st, p->mate, n_u, p->visited);
      hipLaunchKernelGGL(k_loop_cut, dim3(unsigned((n_u + 63) / 64)), dim3(64), 0, st, p->mate, n_u,
                         directed, p->visited, p->sc_nodes, p->sc_parent, p->sc_rank, p->sc_used);
    }
    hipLaunchKernelGGL(k_string_starts, dim3(nblk(n_u)), dim3(256), 0, st, p->mate, p->u_len, n_u,
                       directed, p->scls, p->s_nk);
    hipLaunchKernelGGL(k_string_counts, dim3(nblk(n_u)), dim3(256), 0, st, p->scls, n_u, slow, p->sc01,
                       p->sc2);
    arena_reset(ctx);
    int64_t* d_t2 = static_cast<int64_t*>(arena_alloc(ctx, 16));
    KSH_TRY(scan_exclusive_i64(ctx, p->sc01, p->sc01, n_u, d_t2));
    KSH_TRY(scan_exclusive_i64(ctx, p->sc2, p->sc2, n_u, d_t2 + 1));
    KSH_HIP(hipMemcpyAsync(ctx->h_pinned, d_t2, 16, hipMemcpyDeviceToHost, st));
    KSH_HIP(hipStreamSynchronize(st));
    const int64_t s0 = ctx->h_pinned[0] & 0xFFFFFFFF, s1 = ctx->h_pinned[0] >> 32,
                  s2 = ctx->h_pinned[1];
    ns = s0 + s1 + s2;
    hipLaunchKernelGGL(k_string_assign, dim3(nblk(n_u)), dim3(256), 0, st, p->mate, p->u_len, n_u,
                       p->scls, p->sc01, p->sc2, p->s_nk, s0, s0 + s1, slow, g->k, p->u_sid, p->u_koff,
                       p->u_flip, p->lens, p->str_start);
  }
  // string starts in bases
  arena_reset(ctx);
  KSH_TRY(scan_exclusive_i64(ctx, p->str_start, p->str_start, ns, p->str_start + ns));
  hipLaunchKernelGGL(k_unitig_place, dim3(nblk(n_u)), dim3(256), 0, st, p->u_len, p->u_sid, p->u_koff,
                     p->u_flip, p->str_start, p->lens, p->u_head, n_u, reinterpret_cast<UnitigPlace*>(p->c01));
  KSH_HIP(hipGetLastError());
  KSH_HIP(hipMemcpyAsync(ctx->h_pinned, p->str_start + ns, 8, hipMemcpyDeviceToHost, st));
  KSH_HIP(hipStreamSynchronize(st));
  p->n_strings = ns;
  p->n_bases = ctx->h_pinned[0];
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
  hipStream_t st = ctx->stream;
  // byte staging aliases the neighbour array (2n * 4 bytes >= n_bases needs checking)
  const size_t need = size_t(p->n_bases) + 64;
  uint8_t* bytes;
  void* tmp = nullptr;
  if (need <= size_t(2 * n) * 4) {
    bytes = reinterpret_cast<uint8_t*>(p->nbr);
  } else {
    KSH_TRY(pool_alloc(ctx, need, &tmp));
    bytes = static_cast<uint8_t*>(tmp);
  }
  {
    Timer timer(ctx, 5, n);
    hipLaunchKernelGGL((k_emit<KeyT>), dim3(nblk(n)), dim3(256), 0, st, set, p->head, p->pos, p->ori,
                       reinterpret_cast<const UnitigPlace*>(p->c01), bytes);
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
  return g->key_bytes == 4 ? encode_plan_t<uint32_t>(ctx, g, set, directed, mode, n_strings, n_bases)
                           : encode_plan_t<uint64_t>(ctx, g, set, directed, mode, n_strings, n_bases);
}

int ksh_spss_encode_write(ksh_ctx* ctx, uint64_t* d_words, uint32_t* d_lens) {
  if (!ctx) return fail(KSH_INVALID_ARGUMENT, "ctx is NULL");
  EncPlan* p = static_cast<EncPlan*>(ctx->enc_state);
  if (!p) return fail(KSH_FAILED_PRECONDITION, "ksh_spss_encode_write without ksh_spss_encode_plan");
  if (p->n > 0 && (!d_words || !d_lens)) return fail(KSH_INVALID_ARGUMENT, "NULL output");
  KSH_HIP(hipSetDevice(ctx->device));
  return p->g.key_bytes == 4 ? encode_write_t<uint32_t>(ctx, d_words, d_lens)
                             : encode_write_t<uint64_t>(ctx, d_words, d_lens);
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

int ksh_spss_encode_release(ksh_ctx* ctx) {
  if (ctx) free_plan(ctx);
  return KSH_OK;
}

}  // extern "C"
