// ParallelDisjointSet on device (lib/core/parallel_disjoint_set.h:15-111): the wait-free union-find of
// Anderson and Woll.  One 64-bit word per node = rank << 32 | parent; Find walks to the root and
// compresses the path with compare-and-swap guarded by the (rank, index) order (:24-40); Unite links
// the root of lower (rank, index) under the other one and bumps the survivor's rank on a tie (:53-78),
// retrying when a concurrent Unite changed the root in between (:88-95).  Every access is an
// agent-scope atomic: the XCDs' L2s are not coherent with each other, and a stale parent word would
// only cost retries but a stale rank could link a root under a node of lower rank.
// No thread ever waits for another one (the loops retry on THEIR OWN failed swap), so divergent
// lanes of a wave cannot block each other.  gfx950 only.
#ifndef KSH_DSU_H_
#define KSH_DSU_H_

#include <hip/hip_runtime.h>

#include <cstdint>

namespace ksh {

struct DevDsu {
  unsigned long long* a;  // [n]

  __device__ __forceinline__ unsigned long long load(uint32_t i) const {
    return __hip_atomic_load(&a[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __device__ __forceinline__ static uint32_t rank_of(unsigned long long w) { return uint32_t(w >> 32); }
  __device__ __forceinline__ static uint32_t next_of(unsigned long long w) { return uint32_t(w); }
  __device__ __forceinline__ bool cas(uint32_t i, unsigned long long expected, unsigned long long desired) const {
    return atomicCAS(&a[i], expected, desired) == expected;
  }
  // (rank, index) order of :97-105
  __device__ __forceinline__ bool less_than(uint32_t x, uint32_t y) const {
    const uint32_t rx = rank_of(load(x)), ry = rank_of(load(y));
    return rx < ry || (rx == ry && x < y);
  }

  __device__ uint32_t find(uint32_t x) const {
    uint32_t y = x;
    while (true) {
      const uint32_t nx = next_of(load(x));
      if (nx == x) break;
      x = nx;
    }
    while (less_than(y, x)) {
      const unsigned long long expected = load(y);
      const unsigned long long desired = (expected >> 32 << 32) | x;
      (void)cas(y, expected, desired);  // compare_exchange_weak: a failure is fine
      y = next_of(expected);
    }
    return x;
  }

  __device__ bool update_root(uint32_t x, uint32_t old_rank, uint32_t y, uint32_t new_rank) const {
    const unsigned long long old = load(x);
    if (next_of(old) != x || rank_of(old) != old_rank) return false;
    return cas(x, old, ((unsigned long long)new_rank << 32) | y);
  }

  __device__ void unite(uint32_t x, uint32_t y) const {
    while (true) {
      x = find(x);
      y = find(y);
      if (x == y) return;
      uint32_t rx = rank_of(load(x)), ry = rank_of(load(y));
      if (rx > ry || (rx == ry && x > y)) {
        const uint32_t t = x;
        x = y;
        y = t;
        const uint32_t r = rx;
        rx = ry;
        ry = r;
      }
      if (!update_root(x, rx, y, rx)) continue;
      if (rx == ry) (void)update_root(y, ry, y, ry + 1);
      return;
    }
  }
};

}  // namespace ksh

#endif
