// Kmer<K>: a k-mer in 2 * K bits, A C G T = 00 01 10 11 -- the value type of the
// reference (lib/core/kmer.h:17-240), host side.  Same constructors and methods
// (String, Last, Complement, Canonical, Next, Prev, Nexts, Prevs, Bits, Hash) and
// the same comparison operators; Complement is bit-parallel here (the reference
// loops K times, kmer.h:103-129) -- the device kernels use the same trick.
#ifndef KSC_CORE_KMER_H_
#define KSC_CORE_KMER_H_

#include <array>
#include <cassert>
#include <cstddef>
#include <cstdint>
#include <string>

template <int K>
class Kmer {
  static_assert(K >= 2 && K <= 31, "k-mers are held in 62 bits");

 public:
  Kmer() = default;

  explicit Kmer(const std::string& s) {
    std::uint64_t bits = 0;
    for (int i = 0; i < K; i++) bits = (bits << 2) | Code(s[i]);
    bits_ = bits;
  }

  explicit Kmer(std::uint64_t bits) : bits_(bits) {}

  std::string String() const {
    std::string s(K, 'A');
    for (int i = 0; i < K; i++) s[i] = "ACGT"[(bits_ >> (2 * (K - 1 - i))) & 3];
    return s;
  }

  char Last() const { return "ACGT"[bits_ & 3]; }

  // Reverse complement ("AACCG" -> "CGGTT").
  Kmer<K> Complement() const {
    std::uint64_t x = ~bits_;
    x = ((x >> 2) & 0x3333333333333333ull) | ((x & 0x3333333333333333ull) << 2);
    x = ((x >> 4) & 0x0F0F0F0F0F0F0F0Full) | ((x & 0x0F0F0F0F0F0F0F0Full) << 4);
    x = __builtin_bswap64(x);
    return Kmer<K>(x >> (64 - 2 * K));
  }

  Kmer<K> Canonical() const {
    const Kmer<K> c = Complement();
    return bits_ < c.bits_ ? *this : c;
  }

  // (K-1)-suffix + c.
  Kmer<K> Next(char c) const { return Kmer<K>(((bits_ << 2) & kMask) | Code(c)); }
  // c + (K-1)-prefix.
  Kmer<K> Prev(char c) const { return Kmer<K>((bits_ >> 2) | (Code(c) << (2 * (K - 1)))); }

  std::array<Kmer<K>, 4> Nexts() const { return {Next('A'), Next('C'), Next('G'), Next('T')}; }
  std::array<Kmer<K>, 4> Prevs() const { return {Prev('A'), Prev('C'), Prev('G'), Prev('T')}; }

  std::uint64_t Bits() const { return bits_; }
  std::size_t Hash() const { return bits_; }

 private:
  static constexpr std::uint64_t kMask = ~std::uint64_t(0) >> (64 - 2 * K);

  static std::uint64_t Code(char c) {
    switch (c) {
      case 'A': return 0;
      case 'C': return 1;
      case 'G': return 2;
      case 'T': return 3;
    }
    assert(false);
    return 0;
  }

  std::uint64_t bits_ = 0;
};

template <int K>
bool operator==(const Kmer<K>& l, const Kmer<K>& r) { return l.Bits() == r.Bits(); }
template <int K>
bool operator!=(const Kmer<K>& l, const Kmer<K>& r) { return !(l == r); }
template <int K>
bool operator<(const Kmer<K>& l, const Kmer<K>& r) { return l.Bits() < r.Bits(); }
template <int K>
bool operator>(const Kmer<K>& l, const Kmer<K>& r) { return l.Bits() > r.Bits(); }

#endif
