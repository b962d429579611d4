// Free functions of lib/core/spss.h on the device path: GetUnitigs (:73-227), GetUnitigsCanonical
// (:230-615), GetSPSS (:1018-1036), GetSPSSCanonical (:1835-1858, fast or not) and
// GetKmerSetFromSPSS (:1861-1941).  Strings come back
// as std::vector<std::string>, as in the reference; their order is the oracle's
// (n_workers == 1 control flow, ascending iteration).
#ifndef KSC_CORE_SPSS_H_
#define KSC_CORE_SPSS_H_

#include <string>
#include <vector>

#include "core/kmer_set.h"
#include "core/kmer_set_compact.h"

namespace internal {
// "ACGTT" -> "AACGT" (lib/core/spss.h:45-68).
inline std::string Complement(std::string s) {
  std::string out(s.rbegin(), s.rend());
  for (char& c : out) c = c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : 'A';
  return out;
}
}  // namespace internal

namespace ksc_detail {
template <int K, int N, typename KeyType>
std::vector<std::string> Encode(const KmerSet<K, N, KeyType>& kmer_set, bool canonical, int mode) {
  const ksh_geom g = KmerSet<K, N, KeyType>::Geom();
  const ksh_set_view v = kmer_set.View();
  std::int64_t n = 0, n_bases = 0;
  ksc::Check(ksh_spss_encode_plan(ksc::Ctx(), &g, &v, canonical ? 1 : 0, mode, &n, &n_bases));
  ksc::DeviceBuffer words(std::size_t((n_bases + 31) / 32) * 8), lens(std::size_t(n) * 4);
  ksc::Check(ksh_spss_encode_write(ksc::Ctx(), static_cast<std::uint64_t*>(words.get()),
                                   static_cast<std::uint32_t*>(lens.get())));
  ksc::Check(ksh_ctx_sync(ksc::Ctx()));
  ksc::Check(ksh_spss_encode_release(ksc::Ctx()));
  return KmerSetCompact<K, N, KeyType>::FromDevice(std::move(words), std::move(lens), n, n_bases)
      .ToStrings(1);
}
}  // namespace ksc_detail

template <int K, int N, typename KeyType>
std::vector<std::string> GetUnitigsCanonical(const KmerSet<K, N, KeyType>& kmer_set, int /*n_workers*/) {
  return ksc_detail::Encode(kmer_set, true, 1);
}

// The non-canonical variant (k-mers as they are, edges only forward).
template <int K, int N, typename KeyType>
std::vector<std::string> GetUnitigs(const KmerSet<K, N, KeyType>& kmer_set, int /*n_workers*/) {
  return ksc_detail::Encode(kmer_set, false, 1);
}

template <int K, int N, typename KeyType>
std::vector<std::string> GetSPSS(const KmerSet<K, N, KeyType>& kmer_set, int /*n_workers*/,
                                 int /*n_buckets*/ = 64) {
  return ksc_detail::Encode(kmer_set, false, 0);
}

template <int K, int N, typename KeyType>
std::vector<std::string> GetSPSSCanonical(const KmerSet<K, N, KeyType>& kmer_set, bool fast,
                                          int /*n_workers*/, int /*n_buckets*/ = 512) {
  return ksc_detail::Encode(kmer_set, true, fast ? 0 : 2);
}

template <int K, int N, typename KeyType>
KmerSet<K, N, KeyType> GetKmerSetFromSPSS(const std::vector<std::string>& spss, bool canonical,
                                          int n_workers) {
  return KmerSetCompact<K, N, KeyType>::FromStrings(spss).ToKmerSet(canonical, n_workers);
}

#endif
