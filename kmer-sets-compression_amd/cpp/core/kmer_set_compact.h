// KmerSetCompact<K, N, KeyType>: the reference's SPSS container
// (lib/core/kmer_set_compact.h:29-348) with its bases resident in HBM: 2 bits per base
// in 64-bit words + (len - K) per string (layout in include/kmersets_hip.h).
//   FromKmerSet  -> ksh_spss_encode_plan/write  (GetSPSSCanonical fast / slow, or GetSPSS for
//                   canonical == false)
//   ToKmerSet    -> ksh_spss_decode_plan/write  (ToStrings + GetKmerSetFromSPSS)
//   Size, Weight -> ksh_spss_size, n_bases
//   GetSampledKmerSet -> decode, then the listed buckets (already sorted)
//   Dump / Load  -> the reference's text format (one string per line)
#ifndef KSC_CORE_KMER_SET_COMPACT_H_
#define KSC_CORE_KMER_SET_COMPACT_H_

#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "core/device.h"
#include "core/io.h"
#include "core/kmer.h"
#include "core/kmer_set.h"
#include "core/status.h"

template <int K, int N, typename KeyType>
class KmerSetCompact {
 public:
  using Set = KmerSet<K, N, KeyType>;

  KmerSetCompact() = default;

  // lib/core/kmer_set_compact.h:36-47: canonical ? GetSPSSCanonical(set, fast) : GetSPSS(set).
  static KmerSetCompact FromKmerSet(const Set& kmer_set, bool canonical, bool fast, int /*n_workers*/) {
    const ksh_geom g = Set::Geom();
    const ksh_set_view v = kmer_set.View();
    KmerSetCompact c;
    ksc::Check(ksh_spss_encode_plan(ksc::Ctx(), &g, &v, canonical ? 1 : 0, fast ? 0 : 2, &c.n_,
                                    &c.n_bases_));
    c.words_ = ksc::DeviceBuffer(std::size_t((c.n_bases_ + 31) / 32) * 8);
    c.lens_ = ksc::DeviceBuffer(std::size_t(c.n_) * 4);
    ksc::Check(ksh_spss_encode_write(ksc::Ctx(), static_cast<std::uint64_t*>(c.words_.get()),
                                     static_cast<std::uint32_t*>(c.lens_.get())));
    ksc::Check(ksh_ctx_sync(ksc::Ctx()));
    ksc::Check(ksh_spss_encode_release(ksc::Ctx()));
    return c;
  }

  // From SPSS strings over ACGT (the reference's private constructor, :206-266).
  static KmerSetCompact FromStrings(const std::vector<std::string>& spss) {
    KmerSetCompact c;
    c.n_ = static_cast<std::int64_t>(spss.size());
    std::vector<std::uint32_t> lens(spss.size());
    std::int64_t total = 0;
    for (std::size_t i = 0; i < spss.size(); i++) {
      lens[i] = static_cast<std::uint32_t>(spss[i].length()) - static_cast<std::uint32_t>(K);
      total += static_cast<std::int64_t>(spss[i].length());
    }
    c.n_bases_ = total;
    std::vector<std::uint64_t> words(static_cast<std::size_t>((total + 31) / 32), 0);
    std::int64_t at = 0;
    for (const std::string& s : spss) {
      for (char ch : s) {
        std::uint64_t code = 0;
        switch (ch) {
          case 'C': code = 1; break;
          case 'G': code = 2; break;
          case 'T': code = 3; break;
          default: break;
        }
        words[at / 32] |= code << (62 - 2 * (at % 32));
        at++;
      }
    }
    c.words_ = ksc::DeviceBuffer::FromHost(words);
    c.lens_ = ksc::DeviceBuffer::FromHost(lens);
    return c;
  }

  // Takes ownership of device buffers in the C-ABI layout.
  static KmerSetCompact FromDevice(ksc::DeviceBuffer words, ksc::DeviceBuffer lens, std::int64_t n_strings,
                                   std::int64_t n_bases) {
    KmerSetCompact c;
    c.words_ = std::move(words);
    c.lens_ = std::move(lens);
    c.n_ = n_strings;
    c.n_bases_ = n_bases;
    return c;
  }

  Set ToKmerSet(bool canonical, int /*n_workers*/) const {
    const ksh_geom g = Set::Geom();
    const ksh_spss_view v = View();
    ksc::DeviceBuffer off(std::size_t(Set::kBucketsNum + 1) * 8);
    std::int64_t n = 0;
    ksc::Check(ksh_spss_decode_plan(ksc::Ctx(), &g, &v, canonical ? 1 : 0,
                                    static_cast<std::int64_t*>(off.get()), &n));
    ksc::DeviceBuffer keys(std::size_t(n) * Set::kDeviceKeyBytes);
    ksc::Check(ksh_spss_decode_write(ksc::Ctx(), &g, &v, canonical ? 1 : 0,
                                     static_cast<std::int64_t*>(off.get()), keys.get(), &n));
    return Set::FromDevice(std::move(off), std::move(keys), n);
  }

  // The lines are spelled on the device (ksh_spss_to_text); the host writes one buffer.
  ksc::Status Dump(const std::string& file_name, const std::string& compressor, int /*n_workers*/) const {
    const std::string text = ToText();
    return ksc::WriteBytes(file_name, compressor, text.data(), text.size());
  }

  // The file's bytes go to the device as they are and are parsed there
  // (ksh_spss_from_text_plan / _write).
  static ksc::StatusOr<KmerSetCompact> Load(const std::string& file_name, const std::string& decompressor) {
    ksc::StatusOr<std::string> bytes = ksc::ReadBytes(file_name, decompressor);
    if (!bytes.ok()) return bytes.status();
    return FromText(bytes.value());
  }

  // "line\nline\n...": the text Dump writes.
  std::string ToText() const {
    std::string text(static_cast<std::size_t>(n_bases_ + n_), '\0');
    if (text.empty()) return text;
    const ksh_geom g = Set::Geom();
    const ksh_spss_view v = View();
    ksc::DeviceBuffer d_text(text.size());
    ksc::Check(ksh_spss_to_text(ksc::Ctx(), &g, &v, static_cast<char*>(d_text.get())));
    ksc::Check(ksh_ctx_sync(ksc::Ctx()));
    ksc::Check(ksh_ctx_memcpy_d2h(ksc::Ctx(), &text[0], d_text.get(), text.size()));
    return text;
  }

  static KmerSetCompact FromText(const std::string& text) {
    KmerSetCompact c;
    if (text.empty()) return c;
    const ksh_geom g = Set::Geom();
    ksc::DeviceBuffer d_text(text.size());
    ksc::Check(ksh_ctx_memcpy_h2d(ksc::Ctx(), d_text.get(), text.data(), text.size()));
    ksc::Check(ksh_spss_from_text_plan(ksc::Ctx(), &g, static_cast<const char*>(d_text.get()),
                                       static_cast<std::int64_t>(text.size()), &c.n_, &c.n_bases_));
    c.words_ = ksc::DeviceBuffer(std::size_t((c.n_bases_ + 31) / 32) * 8);
    c.lens_ = ksc::DeviceBuffer(std::size_t(c.n_) * 4);
    ksc::Check(ksh_spss_from_text_write(ksc::Ctx(), static_cast<std::uint64_t*>(c.words_.get()),
                                        static_cast<std::uint32_t*>(c.lens_.get())));
    return c;
  }

  std::int64_t Size(int /*n_workers*/) const {
    const ksh_geom g = Set::Geom();
    const ksh_spss_view v = View();
    std::int64_t n = 0;
    ksc::Check(ksh_spss_size(ksc::Ctx(), &g, &v, &n));
    return n;
  }

  std::int64_t Weight() const { return n_bases_; }

  // bucket_ids[i] = j  <->  result[i] holds the sorted keys of bucket j.
  std::vector<std::vector<KeyType>> GetSampledKmerSet(const std::vector<int>& bucket_ids, bool canonical,
                                                      int n_workers) const {
    const Set s = ToKmerSet(canonical, n_workers);
    const std::vector<std::uint64_t>& bits = s.HostBits();
    std::vector<std::vector<KeyType>> out(bucket_ids.size());
    for (std::size_t i = 0; i < bucket_ids.size(); i++) {
      const std::uint64_t lo = std::uint64_t(bucket_ids[i]) << Set::kKeyBits;
      const std::uint64_t hi = std::uint64_t(bucket_ids[i] + 1) << Set::kKeyBits;
      auto b = std::lower_bound(bits.begin(), bits.end(), lo);
      auto e = std::lower_bound(bits.begin(), bits.end(), hi);
      for (auto it = b; it != e; ++it) out[i].push_back(static_cast<KeyType>(*it - lo));
    }
    return out;
  }

  // The stored strings (downloaded).
  std::vector<std::string> ToStrings(int /*n_workers*/) const {
    const std::vector<std::uint64_t> words = words_.ToHost<std::uint64_t>(std::size_t((n_bases_ + 31) / 32));
    const std::vector<std::uint32_t> lens = lens_.ToHost<std::uint32_t>(std::size_t(n_));
    std::vector<std::string> strings(static_cast<std::size_t>(n_));
    std::int64_t at = 0;
    for (std::int64_t i = 0; i < n_; i++) {
      const std::uint32_t len = lens[i] + K;
      strings[i].resize(len);
      for (std::uint32_t j = 0; j < len; j++, at++)
        strings[i][j] = "ACGT"[(words[at / 32] >> (62 - 2 * (at % 32))) & 3];
    }
    return strings;
  }

  ksh_spss_view View() const {
    return ksh_spss_view{static_cast<const std::uint64_t*>(words_.get()),
                         static_cast<const std::uint32_t*>(lens_.get()), n_, n_bases_};
  }

  std::int64_t StringCount() const { return n_; }

 private:
  std::int64_t n_ = 0;        // number of stored strings (value-initialised, unlike :339)
  std::int64_t n_bases_ = 0;  // = Weight()
  ksc::DeviceBuffer words_, lens_;
};

#endif
