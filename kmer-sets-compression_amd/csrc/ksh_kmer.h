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
  // so it is sized for the shortest slices (fine_bits_for).
  const uint32_t* fine = nullptr;
  int fine_bits = 0;

  __device__ __forceinline__ uint64_t key_mask() const {
    return key_bits == 64 ? ~uint64_t(0) : ((uint64_t(1) << key_bits) - 1);
  }

  // Index of k-mer z in the set's ascending order, or -1.
  __device__ int64_t find(uint64_t z) const {
    const int64_t b = int64_t(z >> key_bits);
    const KeyT key = KeyT(z & key_mask());
    int64_t lo, hi;
    if (fine) {
      const int64_t f = (b << fine_bits) + int64_t(uint64_t(key) >> (key_bits - fine_bits));
      lo = fine[f];
      hi = fine[f + 1];
    } else {
      lo = off[b];
      hi = off[b + 1];
    }
    const int64_t end = hi;
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if (keys[mid] < key) lo = mid + 1; else hi = mid;
    }
    return (lo < end && keys[lo] == key) ? lo : int64_t(-1);
  }

  // First index whose k-mer is >= z inside z's bucket; *end = end of that bucket.
  __device__ int64_t lower_bound(uint64_t z, int64_t* end) const {
    const int64_t b = int64_t(z >> key_bits);
    const KeyT key = KeyT(z & key_mask());
    int64_t lo, hi;
    if (fine) {
      const int64_t f = (b << fine_bits) + int64_t(uint64_t(key) >> (key_bits - fine_bits));
      lo = fine[f];
      hi = fine[f + 1];
    } else {
      lo = off[b];
      hi = off[b + 1];
    }
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if (keys[mid] < key) lo = mid + 1; else hi = mid;
    }
    *end = off[b + 1];
    return lo;
  }

  // Bucket holding index t (largest b with off[b] <= t).
  __device__ int64_t bucket_of(int64_t t) const {
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
  // workgroup (its first index), then every thread steps forward from that bucket.
  __device__ __forceinline__ uint64_t kmer_in_block(int64_t t, int64_t* lds_first_bucket) const {
    if (threadIdx.x == 0) {
      const int64_t t0 = int64_t(blockIdx.x) * blockDim.x;
      *lds_first_bucket = bucket_of(t0 < n ? t0 : n - 1);
    }
    __syncthreads();
    if (t >= n) return 0;
    int64_t b = *lds_first_bucket;
    while (off[b + 1] <= t) b++;
    return (uint64_t(b) << key_bits) | uint64_t(keys[t]);
  }
};

}  // namespace ksh

#endif
