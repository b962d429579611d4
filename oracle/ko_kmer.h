// ORACLE (test infrastructure, not product code).
//
// CPU restatement of the reference's k-mer primitives and (bucket, key) split.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use
// anything under oracle/.  The product path (HIP kernels behind the C ABI in
// include/kmersets_hip.h) never calls into this directory.
//
// Follows:
//   lib/core/kmer.h:22-46     Kmer(string)          -> kmer_from_string
//   lib/core/kmer.h:53-79     String()              -> kmer_to_string
//   lib/core/kmer.h:103-129   Complement()          -> complement  (K-step loop,
//                                                      kept as the reference has it)
//   lib/core/kmer.h:133       Canonical()           -> canonical
//   lib/core/kmer.h:136-161   Next(c)               -> next
//   lib/core/kmer.h:164-186   Prev(c)               -> prev
//   lib/core/kmer_set.h:22-43 GetBucketAndKeyFromKmer / GetKmerFromBucketAndKey
//   lib/core/spss.h:45-68     internal::Complement(string)
//
// K and N are run-time values here (the reference has them as template
// parameters); bases are 0..3 for A, C, G, T.
#ifndef ORACLE_KO_KMER_H_
#define ORACLE_KO_KMER_H_

#include <algorithm>
#include <cassert>
#include <cstdint>
#include <string>

namespace ko {

struct Geom {
  int k = 0;          // k-mer length
  int n = 0;          // bucket bits (the reference's template parameter N)
  int key_bytes = 4;  // sizeof(KeyType)
  int key_bits() const { return 2 * k - n; }
  std::uint64_t kmer_mask() const { return ~std::uint64_t(0) >> (64 - 2 * k); }
  std::int64_t n_buckets() const { return std::int64_t(1) << n; }
};

inline int base_code(char c) {
  switch (c) {
    case 'A': return 0;
    case 'C': return 1;
    case 'G': return 2;
    case 'T': return 3;
  }
  return -1;
}

inline char base_char(int code) { return "ACGT"[code & 3]; }

inline std::uint64_t kmer_from_string(const char* s, int k) {
  std::uint64_t bits = 0;
  for (int i = 0; i < k; i++) {
    bits <<= 2;
    int c = base_code(s[i]);
    assert(c >= 0);
    bits += static_cast<std::uint64_t>(c);
  }
  return bits;
}

inline std::string kmer_to_string(std::uint64_t bits, int k) {
  std::string s(k, 'A');
  for (int i = 0; i < k; i++) {
    s[k - 1 - i] = base_char(static_cast<int>(bits % 4));
    bits >>= 2;
  }
  return s;
}

inline std::uint64_t complement(std::uint64_t bits, int k) {
  std::uint64_t out = 0;
  for (int i = 0; i < k; i++) {
    out <<= 2;
    out += 3 - (bits % 4);
    bits >>= 2;
  }
  return out;
}

inline std::uint64_t canonical(std::uint64_t bits, int k) {
  return std::min(bits, complement(bits, k));
}

inline std::uint64_t next(std::uint64_t bits, int k, int c) {
  bits <<= 2;
  bits &= ~std::uint64_t(0) >> (64 - k * 2);
  return bits + static_cast<std::uint64_t>(c);
}

inline std::uint64_t prev(std::uint64_t bits, int k, int c) {
  bits >>= 2;
  return bits + (static_cast<std::uint64_t>(c) << ((k - 1) * 2));
}

inline char kmer_last(std::uint64_t bits) { return base_char(static_cast<int>(bits % 4)); }

inline void bucket_and_key(const Geom& g, std::uint64_t bits, std::int64_t* bucket,
                           std::uint64_t* key) {
  const int n_key_bits = g.key_bits();
  *bucket = static_cast<std::int64_t>(bits >> n_key_bits);
  *key = bits % (std::uint64_t(1) << n_key_bits);
}

inline std::uint64_t kmer_from_bucket_and_key(const Geom& g, std::int64_t bucket,
                                              std::uint64_t key) {
  return (static_cast<std::uint64_t>(bucket) << g.key_bits()) + key;
}

inline std::string complement_string(std::string s) {
  std::reverse(s.begin(), s.end());
  for (char& c : s) c = base_char(3 - base_code(c));
  return s;
}

}  // namespace ko

#endif
