// SPSS decode: packed 2-bit strings -> bucketed sorted key set.
//
// Replaces KmerSetCompact::ToKmerSet = ToStrings + GetKmerSetFromSPSS
// (lib/core/kmer_set_compact.h:52-55,290-336; lib/core/spss.h:1861-1941), which
// rebuilds a std::string per SPSS string, parses every k-mer from a substr and
// inserts it into 2^N hash sets behind try_lock spins.  Here:
//
//   k_str_bases      lens (len - K) -> bases per string; scan -> string starts
//   k_mark_ends      one bit per base position: "last base of a string"
//   k_decode<false>  every thread owns one 64-bit word (32 positions), builds the
//                    forward and reverse-complement k-mers by rolling, keeps the
//                    canonical one, counts buckets in an LDS histogram;
//                    per-workgroup histograms go to a [groups][2^N] matrix
//   k_hist_columns   column sums -> bucket offsets, and per-(group, bucket) bases
//   k_decode<true>   same walk again, scatters the key into its bucket slot
//   k_bucket_sort    one workgroup per bucket: distribution + insertion sort in LDS (bitonic
//                    network for skewed or oversize buckets), duplicates dropped
//                    (GetKmerSetFromSPSS inserts into sets, so repeated k-mers of a
//                    hand-written input collapse; a real SPSS has none)
//
// Compulsory HBM traffic per k-mer: w/4 bytes of bases read twice + the key written,
// read and written once more = about 3 * key_bytes + w/2.
#include "ksh_internal.h"
#include "ksh_kmer.h"

#include <algorithm>
#include <cstdlib>
#include <string>
#include <vector>

namespace ksh {

constexpr int kDecThreads = 1024;       // (two workgroups per CU by their 64 KB histograms: 2048 threads)
constexpr int kMaxLdsBuckets = 16384;   // 64 KiB of u32 counters
// LDS window of the bucket sort; with 2 x 8 KiB of counters two workgroups share a CU's 160 KB.  It holds
// the densest buckets of a canonical 10^8-k-mer set at N = 14 with 4-byte keys (first base A: 7/4 of
// the average, 10 700 keys) and all but those with 8-byte keys (7 936 keys: at 44 KB = 5 632 keys every
// bucket of a k = 31 set took the partition-first path for oversize buckets)
constexpr int kSortLdsBytes = 63488;
constexpr int kMaxSubBits = 11;
constexpr int kSortThreads = 1024;      // one workgroup per bucket (two fit a CU by LDS: 2048 threads)

__global__ __launch_bounds__(256) void k_str_bases(const uint32_t* __restrict__ lens, int64_t n,
                                                    int k, int64_t* __restrict__ bases) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) bases[i] = int64_t(lens[i]) + k;
}

// end_bits: bit (p & 63) of word p >> 6 is set iff base p is the last base of a string.
__global__ __launch_bounds__(256) void k_mark_ends(const int64_t* __restrict__ str_start,
                                                    int64_t n_strings,
                                                    unsigned long long* __restrict__ end_bits) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n_strings) return;
  const int64_t p = str_start[i + 1] - 1;
  atomicOr(&end_bits[p >> 6], 1ull << (p & 63));
}

// Sum of (len - K + 1) = number of k-mers = KmerSetCompact::Size (kmer_set_compact.h:90-112).
__global__ __launch_bounds__(256) void k_sum_lens(const uint32_t* __restrict__ lens, int64_t n,
                                                   unsigned long long* __restrict__ out) {
  unsigned long long acc = 0;
  for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n;
       i += int64_t(gridDim.x) * blockDim.x)
    acc += lens[i];
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d, 64);
  if ((threadIdx.x & 63) == 0 && acc) atomicAdd(out, acc);
}

// kScatter == false: hist_matrix[group][bucket] = k-mers of this group's words in that bucket.
// kScatter == true : hist_matrix holds the exclusive per-bucket base of each group; keys go to
//                    keys[offsets[bucket] + base + rank].
template <typename KeyT, bool kScatter>
__global__ __launch_bounds__(kDecThreads) void k_decode(
    const uint64_t* __restrict__ words, int64_t n_words, int64_t n_bases,
    const unsigned long long* __restrict__ end_bits, int64_t n_end_words, int k, int key_bits,
    int n_buckets, int canonical_flag, int64_t words_per_group, uint32_t* __restrict__ hist_matrix,
    const int64_t* __restrict__ offsets, KeyT* __restrict__ keys) {
  extern __shared__ uint32_t lds_hist[];
  uint32_t* my_row = hist_matrix + int64_t(blockIdx.x) * n_buckets;
  for (int b = threadIdx.x; b < n_buckets; b += kDecThreads) lds_hist[b] = kScatter ? my_row[b] : 0u;
  __syncthreads();

  const int64_t w_begin = int64_t(blockIdx.x) * words_per_group;
  const int64_t w_end = min(w_begin + words_per_group, n_words);
  const uint64_t mask = kmer_mask(k);
  const uint64_t key_mask = (uint64_t(1) << key_bits) - 1;
  const uint64_t no_end_mask = (uint64_t(1) << (k - 1)) - 1;  // bits p .. p+K-2 must be clear

  for (int64_t w = w_begin + threadIdx.x; w < w_end; w += kDecThreads) {
    const uint64_t w0 = words[w];
    const uint64_t w1 = (w + 1 < n_words) ? words[w + 1] : 0;
    // end bits for positions 32w .. 32w + 63
    const int64_t p0 = w << 5;
    const int64_t ew = p0 >> 6;
    const int esh = int(p0 & 63);
    uint64_t eb = end_bits[ew] >> esh;
    if (esh && ew + 1 < n_end_words) eb |= end_bits[ew + 1] << (64 - esh);

    // first k-mer of the word: bases p0 .. p0+K-1 = top 2K bits of w0:w1
    uint64_t fw = (k <= 32) ? (w0 >> (64 - 2 * k)) : 0;  // k <= 31 always
    uint64_t rc = revcomp(fw, k);
#pragma unroll 4
    for (int j = 0; j < 32; j++) {
      const int64_t p = p0 + j;
      const bool valid = (p + k <= n_bases) && (((eb >> j) & no_end_mask) == 0);
      if (valid) {
        const uint64_t cn = (canonical_flag && rc < fw) ? rc : fw;
        const uint32_t bucket = uint32_t(cn >> key_bits);
        if (kScatter) {
          const uint32_t rel = atomicAdd(&lds_hist[bucket], 1u);
          keys[offsets[bucket] + rel] = KeyT(cn & key_mask);
        } else {
          atomicAdd(&lds_hist[bucket], 1u);
        }
      }
      // roll: the base entering is base p + K
      const int q = j + k;  // position inside the 64-base window w0:w1
      const uint64_t nb = q < 32 ? (w0 >> (62 - 2 * q)) & 3 : (w1 >> (62 - 2 * (q - 32))) & 3;
      fw = ((fw << 2) & mask) | nb;
      rc = (rc >> 2) | ((3 - nb) << (2 * (k - 1)));
    }
  }
  if (!kScatter) {
    __syncthreads();
    for (int b = threadIdx.x; b < n_buckets; b += kDecThreads) my_row[b] = lds_hist[b];
  }
}

// ---- the scatter in two levels, for large inputs ---------------------------------------------------
// k_decode<true> writes every key to a random one of 2^N buckets: 10^8 scattered 4-byte stores, the slowest
// thing the memory system does (1.29 ms per 10^8).  Here every store instruction writes runs, as the SPSS
// encode's record scatter does (ksh_encode.hip, k_rc_scatter_l1 / l2).  Level 1 (k_decode_l1, the same
// groups of words and histogram rows as the counting pass): a group takes its words tile by tile
// (256 words, 8192 positions: a thread rolls a quarter of a word), sorts the tile's keys by SUPER-BUCKET
// (the top kDecSgBits bucket bits) in LDS and appends each super-bucket's run to the group's share of
// it -- known exactly from the histogram rows: no atomics in global memory.  Level 2 (k_decode_l2): a
// tile of the intermediate array lies in one super-bucket (two at a seam), is sorted by bucket in LDS,
// and a bucket's run goes where one atomicAdd per (tile, bucket) says.  k_bucket_sort orders the buckets.
constexpr int kDecSgBits = 7;
constexpr int kDecL1Threads = 1024;
// positions per thread and tile: 8 (a tile = 256 words; 49 KB of LDS with 4-byte keys), 4 with 8-byte keys
// (128 words; 41 KB instead of 81 KB: one workgroup per CU otherwise, 14.4 against 7.1 ms per 4 x 10^8 k-mers)
template <typename KeyT>
struct DecL1 {
  static constexpr int kPer = sizeof(KeyT) == 8 ? 4 : 8;
  static constexpr int kTile = kDecL1Threads * kPer;
};
constexpr int kDecL2Threads = 256;
constexpr int kDecL2Per = 8;
constexpr int kDecL2Tile = kDecL2Threads * kDecL2Per;

template <typename KeyT>
__global__ __launch_bounds__(kDecL1Threads) void k_decode_l1(
    const uint64_t* __restrict__ words, int64_t n_words, int64_t n_bases,
    const unsigned long long* __restrict__ end_bits, int64_t n_end_words, int k, int key_bits, int n_bucket_bits,
    int canonical_flag, int64_t words_per_group, const uint32_t* __restrict__ hist_matrix,
    const int64_t* __restrict__ offsets, KeyT* __restrict__ tmp_keys, uint16_t* __restrict__ tmp_b) {
  constexpr int kSg = 1 << kDecSgBits;
  constexpr int kDecL1Per = DecL1<KeyT>::kPer, kDecL1Tile = DecL1<KeyT>::kTile;
  constexpr int kParts = 32 / kDecL1Per;  // threads per word
  __shared__ KeyT s_key[kDecL1Tile];
  __shared__ uint16_t s_bkt[kDecL1Tile];
  __shared__ uint32_t s_cur[kSg], s_cnt[kSg], s_lbase[kSg + 1];
  const int tid = threadIdx.x;
  const int nb = 1 << n_bucket_bits, lo_bits = n_bucket_bits - kDecSgBits;
  const uint32_t* my_row = hist_matrix + int64_t(blockIdx.x) * nb;
  if (tid < kSg) {
    // where this group's share of super-bucket tid starts: the super-bucket's start, plus what the groups
    // before this one put into each of its buckets (the intermediate layout is [super-bucket][group][tile order])
    uint32_t at = uint32_t(offsets[int64_t(tid) << lo_bits]);
    for (int bi = 0; bi < (1 << lo_bits); bi++) at += my_row[(tid << lo_bits) + bi];
    s_cur[tid] = at;
    s_cnt[tid] = 0;
  }
  const int64_t w_begin = int64_t(blockIdx.x) * words_per_group;
  const int64_t w_end = min(w_begin + words_per_group, n_words);
  const uint64_t mask = kmer_mask(k);
  const uint64_t key_mask = (uint64_t(1) << key_bits) - 1;
  const uint64_t no_end_mask = (uint64_t(1) << (k - 1)) - 1;  // bits p .. p+K-2 must be clear
  __syncthreads();
  for (int64_t w0i = w_begin; w0i < w_end; w0i += kDecL1Tile / 32) {
    const int64_t w = w0i + tid / kParts;
    const int j0 = (tid % kParts) * kDecL1Per;
    KeyT key[kDecL1Per];
    uint32_t bkt[kDecL1Per], rank[kDecL1Per];
#pragma unroll
    for (int j = 0; j < kDecL1Per; j++) bkt[j] = 0xFFFFFFFFu;
    if (w < w_end) {
      const uint64_t w0 = words[w];
      const uint64_t w1 = (w + 1 < n_words) ? words[w + 1] : 0;
      const int64_t p0 = w << 5;
      const int64_t ew = p0 >> 6;
      const int esh = int(p0 & 63);
      uint64_t eb = end_bits[ew] >> esh;
      if (esh && ew + 1 < n_end_words) eb |= end_bits[ew + 1] << (64 - esh);
      // the k-mer at position j0 of the word: bases j0 .. j0 + K - 1 of the window w0:w1 (j0 + K <= 55)
      const int top = 2 * (j0 + k);  // bits of the window up to and including the k-mer's last base
      uint64_t fw = (top <= 64 ? (w0 >> (64 - top)) : ((w0 << (top - 64)) | (w1 >> (128 - top)))) & mask;
      uint64_t rc = revcomp(fw, k);
#pragma unroll
      for (int j = 0; j < kDecL1Per; j++) {
        const int jj = j0 + j;
        const int64_t p = p0 + jj;
        const bool valid = (p + k <= n_bases) && (((eb >> jj) & no_end_mask) == 0);
        if (valid) {
          const uint64_t cn = (canonical_flag && rc < fw) ? rc : fw;
          bkt[j] = uint32_t(cn >> key_bits);
          key[j] = KeyT(cn & key_mask);
          rank[j] = atomicAdd(&s_cnt[bkt[j] >> lo_bits], 1u);
        }
        const int q = jj + k;  // position inside the 64-base window of the base that enters
        const uint64_t nbase = q < 32 ? (w0 >> (62 - 2 * q)) & 3 : (w1 >> (62 - 2 * (q - 32))) & 3;
        fw = ((fw << 2) & mask) | nbase;
        rc = (rc >> 2) | ((3 - nbase) << (2 * (k - 1)));
      }
    }
    __syncthreads();
    if (tid < 64) {
      // exclusive scan of the kSg counts by the first wave, two counts per lane (see k_rc_scatter_l1)
      static_assert(kSg == 128, "two super-buckets per lane of one wave");
      const uint32_t c0 = s_cnt[2 * tid], c1 = s_cnt[2 * tid + 1];
      uint32_t inc = c0 + c1;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(inc, d, 64);
        if (tid >= d) inc += o;
      }
      s_lbase[2 * tid] = inc - c0 - c1;
      s_lbase[2 * tid + 1] = inc - c1;
      if (tid == 63) s_lbase[kSg] = inc;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kDecL1Per; j++) {
      if (bkt[j] == 0xFFFFFFFFu) continue;
      const uint32_t pos = s_lbase[bkt[j] >> lo_bits] + rank[j];
      s_key[pos] = key[j];
      s_bkt[pos] = uint16_t(bkt[j]);
    }
    __syncthreads();
    const int tile_n = int(s_lbase[kSg]);
    for (int i = tid; i < tile_n; i += kDecL1Threads) {
      const uint32_t sg = uint32_t(s_bkt[i]) >> lo_bits;
      const uint32_t dst = s_cur[sg] + (uint32_t(i) - s_lbase[sg]);
      tmp_keys[dst] = s_key[i];
      tmp_b[dst] = s_bkt[i];
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
__global__ __launch_bounds__(kDecL2Threads) void k_decode_l2(int64_t n, int n_bucket_bits,
                                                             const int64_t* __restrict__ offsets,
                                                             const KeyT* __restrict__ tmp_keys,
                                                             const uint16_t* __restrict__ tmp_b,
                                                             uint32_t* __restrict__ cursor, KeyT* __restrict__ keys) {
  constexpr int kBins = 2 * kDecL2Threads;  // two super-buckets' worth of buckets (a super-bucket has at most 256)
  __shared__ KeyT s_key[kDecL2Tile];
  __shared__ uint16_t s_bin[kDecL2Tile];
  __shared__ uint32_t s_cnt[kBins], s_lbase[kBins], s_gbase[kBins];
  __shared__ uint32_t s_wave[kDecL2Threads / 64];
  __shared__ uint32_t s_first;
  const int tid = threadIdx.x;
  const int lo_bits = n_bucket_bits - kDecSgBits;
  const int64_t p0 = int64_t(blockIdx.x) * kDecL2Tile;
  const int tile_n = int(min<int64_t>(kDecL2Tile, n - p0));
  s_cnt[2 * tid] = 0;
  s_cnt[2 * tid + 1] = 0;
  if (tid == 0) s_first = (uint32_t(tmp_b[p0]) >> lo_bits) << lo_bits;  // first bucket of the tile's first super-bucket
  __syncthreads();
  const uint32_t base_b = s_first;
  const uint32_t n_bins = min<uint32_t>(uint32_t(kBins), (2u << lo_bits));
  KeyT mine[kDecL2Per];
  uint32_t bin[kDecL2Per], rank[kDecL2Per];
#pragma unroll
  for (int j = 0; j < kDecL2Per; j++) {
    const int i = tid + j * kDecL2Threads;
    bin[j] = 0xFFFFFFFFu;
    if (i >= tile_n) continue;
    mine[j] = tmp_keys[p0 + i];
    const uint32_t b = tmp_b[p0 + i];
    if (b - base_b < n_bins) {
      bin[j] = b - base_b;
      rank[j] = atomicAdd(&s_cnt[bin[j]], 1u);
    } else {
      // a tile over three or more super-buckets (tiny or very skewed inputs): one key at a time
      keys[uint32_t(offsets[b]) + atomicAdd(&cursor[b], 1u)] = mine[j];
    }
  }
  __syncthreads();
  {
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
    s_gbase[2 * tid] = c0 ? uint32_t(offsets[base_b + 2 * tid]) + atomicAdd(&cursor[base_b + 2 * tid], c0) : 0u;
    s_gbase[2 * tid + 1] = c1 ? uint32_t(offsets[base_b + 2 * tid + 1]) + atomicAdd(&cursor[base_b + 2 * tid + 1], c1) : 0u;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < kDecL2Per; j++) {
    if (bin[j] == 0xFFFFFFFFu) continue;
    const uint32_t pos = s_lbase[bin[j]] + rank[j];
    s_key[pos] = mine[j];
    s_bin[pos] = uint16_t(bin[j]);
  }
  __syncthreads();
  const uint32_t staged = s_lbase[kBins - 1] + s_cnt[kBins - 1];  // (uniform; the far keys were written directly)
  for (uint32_t i = tid; i < staged; i += kDecL2Threads) {
    const uint32_t b = s_bin[i];
    keys[s_gbase[b] + (i - s_lbase[b])] = s_key[i];
  }
}

// Exclusive running sum down the groups, per bucket; totals[b] = column sum.  A workgroup takes 64
// buckets, its 16 teams of 64 threads a sixteenth of the groups each, stitched through LDS (the chain
// of dependent reads down a column is 1/16 as long).
constexpr int kColTeams = 16;
__global__ __launch_bounds__(64 * kColTeams) void k_hist_columns(uint32_t* __restrict__ hist_matrix,
                                                                  int64_t n_groups, int n_buckets,
                                                                  int64_t* __restrict__ totals,
                                                                  unsigned long long* __restrict__ max_total) {
  __shared__ uint32_t team_total[kColTeams][64];
  const int lane = threadIdx.x & 63, team = threadIdx.x >> 6;
  const int b = blockIdx.x * 64 + lane;
  const int64_t per_team = (n_groups + kColTeams - 1) / kColTeams;
  const int64_t g0 = min(int64_t(team) * per_team, n_groups), g1 = min(g0 + per_team, n_groups);
  uint32_t run = 0;
  if (b < n_buckets)
    for (int64_t g = g0; g < g1; g++) run += hist_matrix[g * n_buckets + b];
  team_total[team][lane] = run;
  __syncthreads();
  if (b >= n_buckets) return;
  uint32_t before = 0, all = 0;
#pragma unroll
  for (int t2 = 0; t2 < kColTeams; t2++) {
    const uint32_t v = team_total[t2][lane];
    if (t2 < team) before += v;
    all += v;
  }
  run = before;
  for (int64_t g = g0; g < g1; g++) {
    uint32_t* cell = hist_matrix + g * n_buckets + b;
    const uint32_t c = *cell;
    *cell = run;
    run += c;
  }
  if (team == 0) {
    totals[b] = all;
    // the largest bucket (the sort needs a scratch copy when one exceeds its LDS window): one atomic per wave
    unsigned long long m = all;
    if (int(blockIdx.x + 1) * 64 <= n_buckets) {  // a full wave (the others returned above: every lane for itself)
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) {
        const unsigned long long o = __shfl_xor(m, d, 64);
        m = o > m ? o : m;
      }
      if (lane == 0 && max_total) atomicMax(max_total, m);
    } else if (max_total) {
      atomicMax(max_total, m);
    }
  }
}

// ---- per-bucket sort --------------------------------------------------------------------------
struct SortLds {
  uint32_t sub_cnt[(1 << kMaxSubBits) + 1];
  int wave_cnt[kSortThreads / 64];
  int wave_below[kSortThreads / 64];
  int overflow;
  int kept;
};

// Sorts src[0, cnt) (cnt <= capacity of `lds`) into `lds`, ascending.  Keys that differ only in
// their low `eff_bits` bits: one distribution pass on the top bits of those leaves sub-bins of
// a few keys each (keys of a bucket are close to uniform), finished by insertion sort, one
// thread per sub-bin; a skewed range (a sub-bin over 64 keys) is sorted by a bitonic network
// instead (all-ascending form, so the slots past cnt stay virtual).
template <typename KeyT>
__device__ void block_sort_into_lds(const KeyT* __restrict__ src, int cnt, int eff_bits,
                                    KeyT* __restrict__ lds, SortLds* __restrict__ sh) {
  int bits = 0;
  while ((4 << bits) < cnt && bits < kMaxSubBits) bits++;
  if (bits > eff_bits) bits = eff_bits;
  const int n_sub = 1 << bits;
  const int shift = eff_bits - bits;
  for (int i = threadIdx.x; i <= n_sub; i += kSortThreads) sh->sub_cnt[i] = 0;
  if (threadIdx.x == 0) sh->overflow = 0;
  // a range of up to kSortRegs keys per thread is read from memory once and kept in registers between the
  // counting and the placing pass (all loads in flight together; it used to be read twice, one key per trip)
  constexpr int kSortRegs = 8;
  const bool in_regs = cnt <= kSortRegs * kSortThreads;
  KeyT held[kSortRegs];
  if (in_regs) {
#pragma unroll
    for (int u = 0; u < kSortRegs; u++) {
      const int i = int(threadIdx.x) + u * kSortThreads;
      if (i < cnt) held[u] = src[i];
    }
  }
  __syncthreads();
  if (in_regs) {
#pragma unroll
    for (int u = 0; u < kSortRegs; u++)
      if (int(threadIdx.x) + u * kSortThreads < cnt)
        atomicAdd(&sh->sub_cnt[uint32_t(uint64_t(held[u]) >> shift) & uint32_t(n_sub - 1)], 1u);
  } else {
    for (int i = threadIdx.x; i < cnt; i += kSortThreads)
      atomicAdd(&sh->sub_cnt[uint32_t(uint64_t(src[i]) >> shift) & uint32_t(n_sub - 1)], 1u);
  }
  __syncthreads();
  {  // exclusive scan of the sub-bin counts
    const int per = (n_sub + kSortThreads - 1) / kSortThreads;
    const int c0 = min(int(threadIdx.x) * per, n_sub), c1 = min(c0 + per, n_sub);
    int mine = 0;
    for (int i = c0; i < c1; i++) mine += int(sh->sub_cnt[i]);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int o = __shfl_up(inc, d, 64);
      if (lane >= d) inc += o;
    }
    if (lane == 63) sh->wave_cnt[wave] = inc;
    __syncthreads();
    int run = inc - mine;
    for (int w = 0; w < wave; w++) run += sh->wave_cnt[w];
    for (int i = c0; i < c1; i++) {
      const int c = int(sh->sub_cnt[i]);
      sh->sub_cnt[i] = uint32_t(run);
      run += c;
    }
  }
  __syncthreads();
  if (in_regs) {
#pragma unroll
    for (int u = 0; u < kSortRegs; u++)
      if (int(threadIdx.x) + u * kSortThreads < cnt) {
        const uint32_t pos = atomicAdd(&sh->sub_cnt[uint32_t(uint64_t(held[u]) >> shift) & uint32_t(n_sub - 1)], 1u);
        lds[pos] = held[u];
      }
  } else {
    for (int i = threadIdx.x; i < cnt; i += kSortThreads) {
      const KeyT key = src[i];
      const uint32_t pos = atomicAdd(&sh->sub_cnt[uint32_t(uint64_t(key) >> shift) & uint32_t(n_sub - 1)], 1u);
      lds[pos] = key;
    }
  }
  __syncthreads();
  // sub_cnt[b] is now the end of sub-bin b (= start of b + 1)
  for (int b2 = threadIdx.x; b2 < n_sub; b2 += kSortThreads) {
    const int s0 = b2 ? int(sh->sub_cnt[b2 - 1]) : 0, s1 = int(sh->sub_cnt[b2]);
    if (s1 - s0 > 64) {
      sh->overflow = 1;
    } else {
      for (int i = s0 + 1; i < s1; i++) {
        const KeyT x = lds[i];
        int j = i - 1;
        while (j >= s0 && lds[j] > x) {
          lds[j + 1] = lds[j];
          j--;
        }
        lds[j + 1] = x;
      }
    }
  }
  __syncthreads();
  if (sh->overflow == 0) return;
  int padded = 1;
  while (padded < cnt) padded <<= 1;
  const int half = padded >> 1;
  for (int size = 2; size <= padded; size <<= 1) {
    const int hs = size >> 1;
    for (int p = threadIdx.x; p < half; p += kSortThreads) {
      const int block = p / hs, o = p - block * hs;
      const int i = block * size + o, j = block * size + size - 1 - o;
      if (j < cnt) {
        const KeyT x = lds[i], y = lds[j];
        if (x > y) {
          lds[i] = y;
          lds[j] = x;
        }
      }
    }
    __syncthreads();
    for (int stride = size >> 2; stride > 0; stride >>= 1) {
      for (int p = threadIdx.x; p < half; p += kSortThreads) {
        const int i = 2 * stride * (p / stride) + (p % stride), j = i + stride;
        if (j < cnt) {
          const KeyT x = lds[i], y = lds[j];
          if (x > y) {
            lds[i] = y;
            lds[j] = x;
          }
        }
      }
      __syncthreads();
    }
  }
}

// Writes the distinct keys of the sorted lds[0, cnt) that occur at least `cutoff` times to dst
// and returns how many (block-uniform); *below (thread 0 adds to it) counts the distinct keys
// that occur less often.  cutoff == 1 is plain duplicate removal (KmerSet semantics); larger
// cutoffs are KmerCounter::ToKmerSet (lib/core/kmer_counter.h:209-243; its saturating uint8
// counts compare like the exact ones for every cutoff a uint8 can hold).
template <typename KeyT>
__device__ int block_unique_write(const KeyT* __restrict__ lds, int cnt, KeyT* __restrict__ dst,
                                  SortLds* __restrict__ sh, int cutoff, int64_t* below) {
  const int per = (cnt + kSortThreads - 1) / kSortThreads;
  const int c0 = min(int(threadIdx.x) * per, cnt), c1 = min(c0 + per, cnt);
  int mine = 0, mine_below = 0;
  for (int i = c0; i < c1; i++) {
    if (i == 0 || lds[i] != lds[i - 1]) {
      bool keep = true;
      if (cutoff > 1) {
        int run = 1;
        while (run < cutoff && i + run < cnt && lds[i + run] == lds[i]) run++;
        keep = run >= cutoff;
      }
      mine += keep;
      mine_below += !keep;
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int inc = mine, inc_below = mine_below;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int o = __shfl_up(inc, d, 64);
    if (lane >= d) inc += o;
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) inc_below += __shfl_xor(inc_below, d, 64);
  __syncthreads();  // wave_cnt may still be read by the scan of a previous call
  if (lane == 63) sh->wave_cnt[wave] = inc;
  if (lane == 0) sh->wave_below[wave] = inc_below;
  __syncthreads();
  int at = inc - mine;
  for (int w = 0; w < wave; w++) at += sh->wave_cnt[w];
  int total = 0;
#pragma unroll
  for (int w = 0; w < kSortThreads / 64; w++) total += sh->wave_cnt[w];
  if (threadIdx.x == 0 && below)
    for (int w = 0; w < kSortThreads / 64; w++) *below += sh->wave_below[w];
  if (total == cnt) {
    // nothing dropped (every real SPSS: no k-mer twice): the sorted range goes out as it lies, in whole lines --
    // from the loop below a wave's stores are `per` keys apart
    for (int i = threadIdx.x; i < cnt; i += kSortThreads) dst[i] = lds[i];
    __syncthreads();
    return total;
  }
  for (int i = c0; i < c1; i++) {
    if (i == 0 || lds[i] != lds[i - 1]) {
      bool keep = true;
      if (cutoff > 1) {
        int run = 1;
        while (run < cutoff && i + run < cnt && lds[i + run] == lds[i]) run++;
        keep = run >= cutoff;
      }
      if (keep) dst[at++] = lds[i];
    }
  }
  __syncthreads();
  return total;
}

// One workgroup per bucket: sorts keys[lo, hi) ascending and drops duplicates; uniq[b] = number
// kept.  A bucket that fits the LDS is sorted there in one go.  A larger one (5 x 10^8 keys per
// set put 30 k keys in a bucket) is first partitioned by its top key bits into `scratch`
// (same index range as `keys`), then every part is sorted in LDS and written back in order.
template <typename KeyT>
__global__ __launch_bounds__(kSortThreads, 8) void k_bucket_sort(const int64_t* __restrict__ offsets,
                                                      KeyT* __restrict__ keys,
                                                      KeyT* __restrict__ scratch,
                                                      int64_t* __restrict__ uniq, int key_bits,
                                                      int cutoff, int64_t* __restrict__ below) {
  extern __shared__ unsigned char lds_raw[];
  __shared__ SortLds sh;
  __shared__ uint32_t part_end[(1 << kMaxSubBits) + 1];
  KeyT* lds = reinterpret_cast<KeyT*>(lds_raw);
  constexpr int kCap = kSortLdsBytes / int(sizeof(KeyT));
  const int64_t b = blockIdx.x;
  const int64_t lo = offsets[b], hi = offsets[b + 1];
  const int64_t cnt64 = hi - lo;
  if (cnt64 == 0) {
    if (threadIdx.x == 0) {
      uniq[b] = 0;
      if (below) below[b] = 0;
    }
    return;
  }
  KeyT* g = keys + lo;
  int64_t n_below = 0;  // thread 0's copy is the bucket's count
  if (cnt64 <= kCap) {
    block_sort_into_lds(g, int(cnt64), key_bits, lds, &sh);
    const int kept = block_unique_write(lds, int(cnt64), g, &sh, cutoff, &n_below);
    if (threadIdx.x == 0) {
      uniq[b] = kept;
      if (below) below[b] = n_below;
    }
    return;
  }
  // ---- oversize bucket: partition by the top `bits` key bits so that a part holds about
  // kCap / 4 keys, then sort part by part
  KeyT* tmp = scratch + lo;
  int bits = 1;
  while ((cnt64 >> bits) > kCap / 4 && bits < kMaxSubBits) bits++;
  if (bits > key_bits) bits = key_bits;
  const int n_part = 1 << bits;
  const int shift = key_bits - bits;
  for (int i = threadIdx.x; i <= n_part; i += kSortThreads) part_end[i] = 0;
  __syncthreads();
  for (int64_t i = threadIdx.x; i < cnt64; i += kSortThreads)
    atomicAdd(&part_end[uint32_t(uint64_t(g[i]) >> shift) & uint32_t(n_part - 1)], 1u);
  __syncthreads();
  if (threadIdx.x == 0) {  // serial exclusive scan of <= 2048 counters
    uint32_t run = 0;
    for (int i = 0; i < n_part; i++) {
      const uint32_t c = part_end[i];
      part_end[i] = run;
      run += c;
    }
  }
  __syncthreads();
  for (int64_t i = threadIdx.x; i < cnt64; i += kSortThreads) {
    const KeyT key = g[i];
    const uint32_t pos = atomicAdd(&part_end[uint32_t(uint64_t(key) >> shift) & uint32_t(n_part - 1)], 1u);
    tmp[pos] = key;
  }
  __syncthreads();
  // part_end[p] is now the end of part p
  int64_t out = 0;
  for (int p = 0; p < n_part; p++) {
    const int64_t s0 = p ? int64_t(part_end[p - 1]) : 0, s1 = int64_t(part_end[p]);
    const int64_t pc = s1 - s0;
    if (pc == 0) continue;
    if (pc <= kCap) {
      block_sort_into_lds(tmp + s0, int(pc), shift, lds, &sh);
      out += block_unique_write(lds, int(pc), g + out, &sh, cutoff, &n_below);
    } else {
      // a part that still does not fit (heavily skewed keys): bitonic network in global memory
      KeyT* buf = tmp + s0;
      int64_t padded = 1;
      while (padded < pc) padded <<= 1;
      const int64_t half = padded >> 1;
      for (int64_t size = 2; size <= padded; size <<= 1) {
        const int64_t hs = size >> 1;
        for (int64_t q = threadIdx.x; q < half; q += kSortThreads) {
          const int64_t block = q / hs, o = q - block * hs;
          const int64_t i = block * size + o, j = block * size + size - 1 - o;
          if (j < pc) {
            const KeyT x = buf[i], y = buf[j];
            if (x > y) {
              buf[i] = y;
              buf[j] = x;
            }
          }
        }
        __syncthreads();
        for (int64_t stride = size >> 2; stride > 0; stride >>= 1) {
          for (int64_t q = threadIdx.x; q < half; q += kSortThreads) {
            const int64_t i = 2 * stride * (q / stride) + (q % stride), j = i + stride;
            if (j < pc) {
              const KeyT x = buf[i], y = buf[j];
              if (x > y) {
                buf[i] = y;
                buf[j] = x;
              }
            }
          }
          __syncthreads();
        }
      }
      if (threadIdx.x == 0) {
        int64_t at = out;
        for (int64_t i = 0; i < pc;) {
          int64_t run = 1;
          while (i + run < pc && buf[i + run] == buf[i]) run++;
          if (run >= cutoff) g[at++] = buf[i];
          else n_below++;
          i += run;
        }
        sh.kept = int(at - out);
      }
      __syncthreads();
      out += sh.kept;
      __syncthreads();
    }
  }
  if (threadIdx.x == 0) {
    uniq[b] = out;
    if (below) below[b] = n_below;
  }
}

template <typename KeyT>
__global__ __launch_bounds__(256) void k_compact_buckets(const int64_t* __restrict__ old_off,
                                                          const int64_t* __restrict__ new_off,
                                                          const KeyT* __restrict__ src,
                                                          KeyT* __restrict__ dst) {
  const int64_t b = blockIdx.x;
  const int64_t cnt = new_off[b + 1] - new_off[b];
  const KeyT* s = src + old_off[b];
  KeyT* d = dst + new_off[b];
  for (int64_t i = threadIdx.x; i < cnt; i += 256) d[i] = s[i];
}

struct DecodeState {
  uint32_t* hist;       // [groups][n_buckets]
  unsigned long long* end_bits;
  int64_t* str_start;   // [n_strings + 1]
  int64_t* totals;      // [n_buckets + 1]
};

inline size_t a256(size_t x) { return (x + 255) & ~size_t(255); }

int check_spss(const ksh_spss_view* s) {
  if (!s) return fail(KSH_INVALID_ARGUMENT, "spss is NULL");
  if (s->n_strings < 0 || s->n_bases < 0) return fail(KSH_INVALID_ARGUMENT, "negative spss size");
  if (s->n_strings > 0 && (!s->d_words || !s->d_lens))
    return fail(KSH_INVALID_ARGUMENT, "spss pointers are NULL");
  return KSH_OK;
}

void decode_geometry(const ksh_spss_view* s, int64_t* n_words, int64_t* n_end_words, int64_t* groups,
                     int64_t* words_per_group) {
  *n_words = (s->n_bases + 31) / 32;
  *n_end_words = (s->n_bases + 63) / 64 + 1;
  int64_t g = std::min<int64_t>(512, (*n_words + 255) / 256);
  if (g < 1) g = 1;
  *groups = g;
  *words_per_group = (*n_words + g - 1) / g;
}

void decode_carve(ksh_ctx* ctx, int64_t groups, int64_t nb, int64_t n_end_words, int64_t n_strings,
                  DecodeState* st) {
  char* at = ctx->slot[kSlotDecode];
  st->hist = reinterpret_cast<uint32_t*>(at);
  at += a256(size_t(groups) * nb * 4);
  st->end_bits = reinterpret_cast<unsigned long long*>(at);
  at += a256(size_t(n_end_words) * 8);
  st->str_start = reinterpret_cast<int64_t*>(at);
  at += a256(size_t(n_strings + 1) * 8);
  st->totals = reinterpret_cast<int64_t*>(at);
}

template <typename KeyT>
int decode_plan_t(ksh_ctx* ctx, const ksh_geom* g, const ksh_spss_view* s, int canonical_flag,
                  int64_t* d_offsets, int64_t* n_keys) {
  const int64_t nb = n_buckets(g);
  if (s->n_strings == 0 || s->n_bases == 0) {
    KSH_HIP(hipMemsetAsync(d_offsets, 0, size_t(nb + 1) * 8, ctx->stream));
    *n_keys = 0;
    ctx->dec_kmers = 0;
    ctx->dec_src = s->d_words;
    return KSH_OK;
  }
  int64_t n_words, n_end_words, groups, wpg;
  decode_geometry(s, &n_words, &n_end_words, &groups, &wpg);
  const size_t bytes = a256(size_t(groups) * nb * 4) + a256(size_t(n_end_words) * 8) +
                       a256(size_t(s->n_strings + 1) * 8) + a256(size_t(nb + 2) * 8);
  KSH_TRY(slot_reserve(ctx, kSlotDecode, bytes));
  KSH_TRY(arena_reserve(ctx, size_t(s->n_strings / 256 + 4096) * 8 + (1u << 16)));
  arena_reset(ctx);
  DecodeState st;
  decode_carve(ctx, groups, nb, n_end_words, s->n_strings, &st);

  const unsigned sb = unsigned((s->n_strings + 255) / 256);
  hipLaunchKernelGGL(k_str_bases, dim3(sb), dim3(256), 0, ctx->stream, s->d_lens, s->n_strings, g->k,
                     st.str_start);
  KSH_TRY(scan_exclusive_i64(ctx, st.str_start, st.str_start, s->n_strings,
                             st.str_start + s->n_strings));
  KSH_HIP(hipMemsetAsync(st.end_bits, 0, size_t(n_end_words) * 8, ctx->stream));
  hipLaunchKernelGGL(k_mark_ends, dim3(sb), dim3(256), 0, ctx->stream, st.str_start, s->n_strings,
                     st.end_bits);
  hipLaunchKernelGGL((k_decode<KeyT, false>), dim3(unsigned(groups)), dim3(kDecThreads),
                     size_t(nb) * 4, ctx->stream, s->d_words, n_words, s->n_bases, st.end_bits,
                     n_end_words, g->k, key_bits(g), int(nb), canonical_flag, wpg, st.hist, nullptr,
                     static_cast<KeyT*>(nullptr));
  unsigned long long* d_max = reinterpret_cast<unsigned long long*>(st.totals + nb + 1);
  KSH_HIP(hipMemsetAsync(d_max, 0, 8, ctx->stream));
  hipLaunchKernelGGL(k_hist_columns, dim3(unsigned((nb + 63) / 64)), dim3(64 * kColTeams), 0, ctx->stream,
                     st.hist, groups, int(nb), st.totals, d_max);
  KSH_TRY(scan_exclusive_i64(ctx, st.totals, d_offsets, nb, d_offsets + nb));
  KSH_HIP(hipGetLastError());
  // consistency: the strings must tile the base stream exactly
  KSH_HIP(hipMemcpyAsync(ctx->h_pinned, d_offsets + nb, 8, hipMemcpyDeviceToHost, ctx->stream));
  KSH_HIP(hipMemcpyAsync(ctx->h_pinned + 1, st.str_start + s->n_strings, 8, hipMemcpyDeviceToHost,
                         ctx->stream));
  KSH_HIP(hipMemcpyAsync(ctx->h_pinned + 2, d_max, 8, hipMemcpyDeviceToHost, ctx->stream));
  KSH_HIP(hipStreamSynchronize(ctx->stream));
  ctx->dec_max_bucket = ctx->h_pinned[2];
  if (ctx->h_pinned[1] != s->n_bases)
    return fail(KSH_INVALID_ARGUMENT, "spss: sum of string lengths (%lld) != n_bases (%lld)",
                (long long)ctx->h_pinned[1], (long long)s->n_bases);
  *n_keys = ctx->h_pinned[0];
  ctx->dec_kmers = *n_keys;
  ctx->dec_words = n_words;
  ctx->dec_groups = groups;
  ctx->dec_src = s->d_words;
  return KSH_OK;
}

template <typename KeyT>
int decode_write_t(ksh_ctx* ctx, const ksh_geom* g, const ksh_spss_view* s, int canonical_flag,
                   int64_t* d_offsets, void* d_keys, int64_t* n_keys, int cutoff, int64_t* n_below) {
  const int64_t nb = n_buckets(g);
  if (ctx->dec_src != s->d_words)
    return fail(KSH_FAILED_PRECONDITION, "ksh_spss_decode_write without a matching decode_plan");
  if (n_below) *n_below = 0;
  if (ctx->dec_kmers == 0) {
    *n_keys = 0;
    return KSH_OK;
  }
  int64_t n_words, n_end_words, groups, wpg;
  decode_geometry(s, &n_words, &n_end_words, &groups, &wpg);
  DecodeState st;
  decode_carve(ctx, groups, nb, n_end_words, s->n_strings, &st);
  KeyT* keys = static_cast<KeyT*>(d_keys);
  // KSH_DECODE_SCATTER=direct: one scattered store per k-mer (k_decode<true>), for A/B runs and small inputs
  static const bool direct = [] {
    const char* e = getenv("KSH_DECODE_SCATTER");
    return e && std::string(e) == "direct";
  }();
  static const int64_t two_level_min = [] {
    const char* e = getenv("KSH_DECODE_L2_MIN");  // (tests set it low)
    return e ? std::max<int64_t>(1024, atoll(e)) : int64_t(1) << 20;
  }();
  if (!direct && ctx->dec_kmers >= two_level_min && g->n_bucket_bits > kDecSgBits && g->n_bucket_bits <= 15 &&
      ctx->dec_kmers < int64_t(0xFFFFFFF0)) {
    void* tmp = nullptr;
    const size_t tk = (size_t(ctx->dec_kmers) * sizeof(KeyT) + 255) & ~size_t(255);
    KSH_TRY(pool_alloc(ctx, tk + size_t(ctx->dec_kmers) * 2 + size_t(nb) * 4 + 256, &tmp));
    KeyT* tmp_keys = static_cast<KeyT*>(tmp);
    uint16_t* tmp_b = reinterpret_cast<uint16_t*>(static_cast<char*>(tmp) + tk);
    uint32_t* cursor = reinterpret_cast<uint32_t*>(static_cast<char*>(tmp) + tk + ((size_t(ctx->dec_kmers) * 2 + 255) & ~size_t(255)));
    KSH_HIP(hipMemsetAsync(cursor, 0, size_t(nb) * 4, ctx->stream));
    hipLaunchKernelGGL((k_decode_l1<KeyT>), dim3(unsigned(groups)), dim3(kDecL1Threads), 0, ctx->stream, s->d_words,
                       n_words, s->n_bases, st.end_bits, n_end_words, g->k, key_bits(g), g->n_bucket_bits,
                       canonical_flag, wpg, st.hist, d_offsets, tmp_keys, tmp_b);
    hipLaunchKernelGGL((k_decode_l2<KeyT>), dim3(unsigned((ctx->dec_kmers + kDecL2Tile - 1) / kDecL2Tile)),
                       dim3(kDecL2Threads), 0, ctx->stream, ctx->dec_kmers, g->n_bucket_bits, d_offsets, tmp_keys, tmp_b,
                       cursor, keys);
    pool_free(ctx, tmp);  // (single stream: the block is reused only behind these launches)
  } else {
    hipLaunchKernelGGL((k_decode<KeyT, true>), dim3(unsigned(groups)), dim3(kDecThreads),
                       size_t(nb) * 4, ctx->stream, s->d_words, n_words, s->n_bases, st.end_bits,
                       n_end_words, g->k, key_bits(g), int(nb), canonical_flag, wpg, st.hist, d_offsets,
                       keys);
  }
  // per-bucket sort + duplicate removal; uniq counts reuse st.totals.  Buckets larger than the
  // LDS capacity are partitioned through a scratch copy first; whether any exists is read off
  // the bucket offsets (128 KiB to the host).
  KeyT* scratch = nullptr;
  {
    // (the largest bucket came back with the plan's sizes: no copy of the offsets, no round trip here)
    const int64_t max_bucket = ctx->dec_max_bucket;
    if (max_bucket > int64_t(kSortLdsBytes / sizeof(KeyT))) {
      void* ptr = nullptr;
      KSH_TRY(pool_alloc(ctx, size_t(ctx->dec_kmers) * sizeof(KeyT), &ptr));
      scratch = static_cast<KeyT*>(ptr);
    }
  }
  arena_reset(ctx);
  KSH_TRY(arena_reserve(ctx, 2 * a256(size_t(nb + 1) * 8) + (1u << 16)));
  int64_t* new_off = static_cast<int64_t*>(arena_alloc(ctx, size_t(nb + 1) * 8));
  int64_t* below = static_cast<int64_t*>(arena_alloc(ctx, size_t(nb + 1) * 8));
  if (!new_off || !below) return fail(KSH_INTERNAL, "scratch arena too small");
  {
    // (more than the 64 KB a kernel gets without asking)
    const uint32_t bit = 8u << (sizeof(KeyT) == 2 ? 0 : sizeof(KeyT) == 4 ? 1 : 2);
    if (!(ctx->lds_opt_in & bit)) {
      KSH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_bucket_sort<KeyT>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, kSortLdsBytes));
      ctx->lds_opt_in |= bit;
    }
  }
  hipLaunchKernelGGL((k_bucket_sort<KeyT>), dim3(unsigned(nb)), dim3(kSortThreads), kSortLdsBytes, ctx->stream,
                     d_offsets, keys, scratch, st.totals, key_bits(g), cutoff, n_below ? below : nullptr);
  if (scratch) pool_free(ctx, scratch);
  KSH_HIP(hipGetLastError());
  KSH_TRY(scan_exclusive_i64(ctx, st.totals, new_off, nb, new_off + nb));
  if (n_below) KSH_TRY(scan_exclusive_i64(ctx, below, below, nb, below + nb));
  KSH_HIP(hipMemcpyAsync(ctx->h_pinned, new_off + nb, 8, hipMemcpyDeviceToHost, ctx->stream));
  if (n_below)
    KSH_HIP(hipMemcpyAsync(ctx->h_pinned + 1, below + nb, 8, hipMemcpyDeviceToHost, ctx->stream));
  KSH_HIP(hipStreamSynchronize(ctx->stream));
  const int64_t kept = ctx->h_pinned[0];
  if (n_below) *n_below = ctx->h_pinned[1];
  if (kept != ctx->dec_kmers) {
    // repeated k-mers in the input: close the gaps (rare; through a temporary copy)
    void* tmp = nullptr;
    KSH_TRY(device_alloc(ctx, size_t(ctx->dec_kmers) * sizeof(KeyT), &tmp));
    KSH_HIP(hipMemcpyAsync(tmp, keys, size_t(ctx->dec_kmers) * sizeof(KeyT), hipMemcpyDeviceToDevice,
                           ctx->stream));
    hipLaunchKernelGGL((k_compact_buckets<KeyT>), dim3(unsigned(nb)), dim3(256), 0, ctx->stream,
                       d_offsets, new_off, static_cast<const KeyT*>(tmp), keys);
    KSH_HIP(hipMemcpyAsync(d_offsets, new_off, size_t(nb + 1) * 8, hipMemcpyDeviceToDevice,
                           ctx->stream));
    KSH_HIP(hipStreamSynchronize(ctx->stream));
    KSH_HIP(hipFree(tmp));
  }
  *n_keys = kept;
  return KSH_OK;
}

}  // namespace ksh

using namespace ksh;

extern "C" {

int ksh_spss_size(ksh_ctx* ctx, const ksh_geom* g, const ksh_spss_view* s, int64_t* n_kmers) {
  if (!ctx || !n_kmers) return fail(KSH_INVALID_ARGUMENT, "NULL argument");
  KSH_TRY(check_geom(g));
  KSH_TRY(check_spss(s));
  KSH_HIP(hipSetDevice(ctx->device));
  if (s->n_strings == 0) {
    *n_kmers = 0;
    return KSH_OK;
  }
  arena_reset(ctx);
  unsigned long long* d_acc = static_cast<unsigned long long*>(arena_alloc(ctx, 8));
  KSH_HIP(hipMemsetAsync(d_acc, 0, 8, ctx->stream));
  const unsigned blocks = unsigned(std::min<int64_t>((s->n_strings + 255) / 256, 1024));
  hipLaunchKernelGGL(k_sum_lens, dim3(blocks), dim3(256), 0, ctx->stream, s->d_lens, s->n_strings,
                     d_acc);
  KSH_HIP(hipGetLastError());
  KSH_HIP(hipMemcpyAsync(ctx->h_pinned, d_acc, 8, hipMemcpyDeviceToHost, ctx->stream));
  KSH_HIP(hipStreamSynchronize(ctx->stream));
  *n_kmers = ctx->h_pinned[0] + s->n_strings;  // sum(len - K) + n = sum(len - K + 1)
  return KSH_OK;
}

int ksh_spss_decode_plan(ksh_ctx* ctx, const ksh_geom* g, const ksh_spss_view* s, int canonical_flag,
                         int64_t* d_offsets, int64_t* n_keys) {
  if (!ctx || !d_offsets || !n_keys) return fail(KSH_INVALID_ARGUMENT, "NULL argument");
  KSH_TRY(check_geom(g));
  KSH_TRY(check_spss(s));
  if (n_buckets(g) > kMaxLdsBuckets)
    return fail(KSH_INVALID_ARGUMENT, "decode supports n_bucket_bits <= 14 (got %d)", g->n_bucket_bits);
  KSH_HIP(hipSetDevice(ctx->device));
  return KSH_BY_KEY(g->key_bytes, decode_plan_t, ctx, g, s, canonical_flag, d_offsets, n_keys);
}

int ksh_spss_decode_write(ksh_ctx* ctx, const ksh_geom* g, const ksh_spss_view* s, int canonical_flag,
                          int64_t* d_offsets, void* d_keys, int64_t* n_keys) {
  if (!ctx || !d_offsets || !n_keys) return fail(KSH_INVALID_ARGUMENT, "NULL argument");
  KSH_TRY(check_geom(g));
  KSH_TRY(check_spss(s));
  if (ctx->dec_kmers > 0 && !d_keys) return fail(KSH_INVALID_ARGUMENT, "d_keys is NULL");
  KSH_HIP(hipSetDevice(ctx->device));
  return KSH_BY_KEY(g->key_bytes, decode_write_t, ctx, g, s, canonical_flag, d_offsets, d_keys, n_keys, 1, nullptr);
}

/* KmerCounter::FromReads + ToKmerSet (lib/core/kmer_counter.h:64-133,209-243): the k-mers of
 * the strings of `reads` counted with multiplicity, those seen at least `cutoff` times kept.
 * Same two calls as the decode (ksh_spss_decode_plan sizes the key buffer: one slot per k-mer
 * occurrence). */
int ksh_kmer_count_write(ksh_ctx* ctx, const ksh_geom* g, const ksh_spss_view* reads, int canonical_flag,
                         int32_t cutoff, int64_t* d_offsets, void* d_keys, int64_t* n_keys,
                         int64_t* n_cut) {
  if (!ctx || !d_offsets || !n_keys || !n_cut) return fail(KSH_INVALID_ARGUMENT, "NULL argument");
  KSH_TRY(check_geom(g));
  KSH_TRY(check_spss(reads));
  if (cutoff < 0 || cutoff > 255) return fail(KSH_INVALID_ARGUMENT, "cutoff must be in 0..255 (uint8 counts)");
  if (cutoff < 1) cutoff = 1;  // "count < 0" never holds: a cutoff of 0 keeps every k-mer, like 1
  if (ctx->dec_kmers > 0 && !d_keys) return fail(KSH_INVALID_ARGUMENT, "d_keys is NULL");
  KSH_HIP(hipSetDevice(ctx->device));
  return KSH_BY_KEY(g->key_bytes, decode_write_t, ctx, g, reads, canonical_flag, d_offsets, d_keys, n_keys, cutoff, n_cut);
}

}  // extern "C"
