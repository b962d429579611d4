// Device-side k-mer primitives (lib/core/kmer.h:103-186, lib/core/kmer_set.h:22-43)
// and membership in a bucketed sorted set (KmerSet::Contains, kmer_set.h:99-105).
// gfx950 only.
#ifndef KSH_KMER_H_
#define KSH_KMER_H_

#include <hip/hip_runtime.h>

#include <cstdint>

namespace ksh {

// Reverse complement of a 2-bit packed k-mer, bit-parallel (the reference loops K
// times, kmer.h:103-129; same function).
__device__ __forceinline__ uint64_t revcomp(uint64_t x, int k) {
  x = ~x;
  x = ((x >> 2) & 0x3333333333333333ull) | ((x & 0x3333333333333333ull) << 2);
  x = ((x >> 4) & 0x0F0F0F0F0F0F0F0Full) | ((x & 0x0F0F0F0F0F0F0F0Full) << 4);
  x = __builtin_bswap64(x);
  return x >> (64 - 2 * k);
}

__device__ __forceinline__ uint64_t canonical(uint64_t x, int k) {
  const uint64_t r = revcomp(x, k);
  return x < r ? x : r;
}

__device__ __forceinline__ uint64_t kmer_mask(int k) { return ~uint64_t(0) >> (64 - 2 * k); }

// Kmer::Next / Kmer::Prev (kmer.h:136-186), c in 0..3.
__device__ __forceinline__ uint64_t kmer_next(uint64_t x, int k, int c) {
  return ((x << 2) & kmer_mask(k)) | uint64_t(c);
}
__device__ __forceinline__ uint64_t kmer_prev(uint64_t x, int k, int c) {
  return (x >> 2) | (uint64_t(c) << (2 * (k - 1)));
}

constexpr int kCoarseShift = 8;

// A resident set seen from a kernel.
template <typename KeyT>
struct DevSet {
  const int64_t* off;  // [n_buckets + 1]
  const KeyT* keys;
  int64_t n_buckets;
  int64_t n;
  int k;
  int key_bits;
  // Optional fine index: fine[(b << fine_bits) + sub] = first key index of bucket b whose top
  // fine_bits key bits are >= sub (nb * 2^fine_bits + 1 entries).  Narrows a membership probe
  // from the whole bucket (~10-13 dependent loads) to a slice of one or two keys; the index is
  // read at the same rate whether it is 16 MB or 256 MB (only an L2-sized one would be faster),
  // so it is sized for the shortest slices (fine_bits_for).  Slices are at least four values
  // wide (fine_bits <= key_bits - 2).
  const uint32_t* fine = nullptr;
  int fine_bits = 0;
  // Optional coarse inverse of `off`: coarse[t >> kCoarseShift] = bucket of index (t >> kCoarseShift)
  // << kCoarseShift, so that the bucket of an arbitrary index is one table read and a step or two
  // along `off` instead of a binary search over it (kmer(t) for indices that come out of walks).
  const uint32_t* coarse = nullptr;

  __device__ __forceinline__ uint64_t key_mask() const {
    return key_bits == 64 ? ~uint64_t(0) : ((uint64_t(1) << key_bits) - 1);
  }

  // Two adjacent array elements in one request (they need not be aligned as a pair).  The encode
  // kernels are bound by the number of load requests they issue, not by the bytes or the
  // distinct lines behind them, so the probes below ask for as much as they can per request.
  template <typename T>
  __device__ static __forceinline__ void load_pair(const T* p, T* a, T* b) {
    struct __attribute__((packed, aligned(alignof(T)))) Pair {
      T x, y;
    };
    const Pair v = *reinterpret_cast<const Pair*>(p);
    *a = v.x;
    *b = v.y;
  }

  // [lo, hi) of the keys that can equal `key` (its slice of the fine index, or its bucket).
  __device__ __forceinline__ void probe_range(int64_t b, KeyT key, int64_t* lo, int64_t* hi) const {
    if (fine) {
      uint32_t x, y;
      load_pair(fine + (b << fine_bits) + int64_t(uint64_t(key) >> (key_bits - fine_bits)), &x, &y);
      *lo = x;
      *hi = y;
    } else {
      load_pair(off + b, lo, hi);
    }
  }

  // Index of k-mer z in the set's ascending order, or -1.
  __device__ int64_t find(uint64_t z) const {
    const int64_t b = int64_t(z >> key_bits);
    const KeyT key = KeyT(z & key_mask());
    int64_t lo, hi;
    probe_range(b, key, &lo, &hi);
    if (lo >= hi) return -1;
    if (hi - lo == 1) return keys[lo] == key ? lo : int64_t(-1);
    KeyT k0, k1;
    load_pair(keys + lo, &k0, &k1);
    if (key <= k1) return key == k0 ? lo : (key == k1 ? lo + 1 : int64_t(-1));
    lo += 2;
    const int64_t end = hi;
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if (keys[mid] < key) lo = mid + 1; else hi = mid;
    }
    return (lo < end && keys[lo] == key) ? lo : int64_t(-1);
  }

  // The members among the four consecutive k-mers g0 .. g0 + 3 (g0 ends in base A: they share
  // a bucket and, slices being at least four values wide, a slice): f(index) for each, ascending.
  template <typename F>
  __device__ __forceinline__ void for_group4(uint64_t g0, F f) const {
    const int64_t b = int64_t(g0 >> key_bits);
    const KeyT gkey = KeyT(g0 & key_mask());
    int64_t lo, hi;
    probe_range(b, gkey, &lo, &hi);
    if (lo >= hi) return;
    if (hi - lo == 1) {
      if (uint64_t(keys[lo]) - uint64_t(gkey) < 4) f(lo);
      return;
    }
    KeyT k0, k1;
    load_pair(keys + lo, &k0, &k1);
    if (hi - lo == 2 || k1 >= gkey + 3) {  // nothing after these two can be a member
      if (uint64_t(k0) - uint64_t(gkey) < 4) f(lo);
      if (uint64_t(k1) - uint64_t(gkey) < 4) f(lo + 1);
      return;
    }
    const int64_t end = hi;
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if (keys[mid] < gkey) lo = mid + 1; else hi = mid;
    }
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const int64_t idx = lo + e;
      if (idx >= end) break;
      if (uint64_t(keys[idx]) - uint64_t(gkey) >= 4) break;
      f(idx);
    }
  }

  // Bucket holding index t (largest b with off[b] <= t).
  __device__ int64_t bucket_of(int64_t t) const {
    if (coarse) {
      int64_t b = coarse[t >> kCoarseShift];
      while (off[b + 1] <= t) b++;
      return b;
    }
    return bucket_search(t);
  }
  __device__ int64_t bucket_search(int64_t t) const {
    int64_t lo = 0, hi = n_buckets;
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if (off[mid + 1] <= t) lo = mid + 1; else hi = mid;
    }
    return lo;
  }

  __device__ __forceinline__ uint64_t kmer(int64_t t) const {
    return (uint64_t(bucket_of(t)) << key_bits) | uint64_t(keys[t]);
  }

  // For kernels whose workgroup owns 256 consecutive indices: one binary search per
  // workgroup (its first index); lds2[0] = that bucket, lds2[1] = where it ends.  A thread
  // inside the first bucket asks memory for nothing but its key; the others step forward.
  __device__ __forceinline__ uint64_t kmer_in_block(int64_t t, int64_t* lds2) const {
    if (threadIdx.x == 0) {
      const int64_t t0 = int64_t(blockIdx.x) * blockDim.x;
      const int64_t b0 = bucket_of(t0 < n ? t0 : n - 1);
      lds2[0] = b0;
      lds2[1] = off[b0 + 1];
    }
    __syncthreads();
    if (t >= n) return 0;
    int64_t b = lds2[0];
    if (t >= lds2[1]) {
      b++;
      while (off[b + 1] <= t) b++;
    }
    return (uint64_t(b) << key_bits) | uint64_t(keys[t]);
  }

  // The same in two halves, for kernels in which only a few threads need their k-mer: every
  // thread of the workgroup calls block_bucket, those few kmer_from_block.
  __device__ __forceinline__ void block_bucket(int64_t* lds2) const {
    block_bucket_at(int64_t(blockIdx.x) * blockDim.x, lds2);
  }
  // (t0 = the smallest index any thread of the workgroup will ask for)
  __device__ __forceinline__ void block_bucket_at(int64_t t0, int64_t* lds2) const {
    if (threadIdx.x == 0) {
      const int64_t b0 = bucket_of(t0 < n ? t0 : n - 1);
      lds2[0] = b0;
      lds2[1] = off[b0 + 1];
    }
    __syncthreads();
  }
  __device__ __forceinline__ uint64_t kmer_from_block(int64_t t, const int64_t* lds2) const {
    int64_t b = lds2[0];
    if (t >= lds2[1]) {
      b++;
      while (off[b + 1] <= t) b++;
    }
    return (uint64_t(b) << key_bits) | uint64_t(keys[t]);
  }
};

}  // namespace ksh

#endif
