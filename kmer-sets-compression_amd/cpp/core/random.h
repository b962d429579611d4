// Seeded replacement for GetRandomInts((1 << N) / 50, unique, sorted, 0, (1 << N) - 1)
// (lib/core/random.h:12-41 as called at lib/core/kmer_set_set.h:123-124).  The reference
// draws from an unseeded absl::InsecureBitGen, which makes its merge sequence
// irreproducible; here the sample is the first m entries of a Fisher-Yates shuffle driven
// by a splitmix64 counter (the same function as kmersets/synth.py sample_bucket_ids), so
// runs and the oracle agree.  Seed: KSC_BUCKET_SEED (default 1).
#ifndef KSC_CORE_RANDOM_H_
#define KSC_CORE_RANDOM_H_

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <numeric>
#include <vector>

namespace ksc {

inline std::uint64_t Mix64(std::uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

inline std::vector<int> SampleBucketIds(int n_bits, std::uint64_t seed, int divisor = 50) {
  const int n = 1 << n_bits;
  const int m = n / divisor;
  std::vector<int> perm(static_cast<std::size_t>(n));
  std::iota(perm.begin(), perm.end(), 0);
  for (int i = 0; i < m; i++) {
    const std::uint64_t r = Mix64((seed << 32) + static_cast<std::uint64_t>(i));
    const int j = i + static_cast<int>(r % static_cast<std::uint64_t>(n - i));
    std::swap(perm[i], perm[j]);
  }
  std::vector<int> out(perm.begin(), perm.begin() + m);
  std::sort(out.begin(), out.end());
  return out;
}

inline std::uint64_t BucketSeedFromEnv() {
  const char* e = std::getenv("KSC_BUCKET_SEED");
  return e ? std::strtoull(e, nullptr, 10) : 1;
}

}  // namespace ksc

#endif
