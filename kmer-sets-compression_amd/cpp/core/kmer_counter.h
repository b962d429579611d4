// KmerCounter<K, N, KeyType, ValueType>: the reference's k-mer counter
// (lib/core/kmer_counter.h:49-300), the step that produces the loop's inputs
// (src/kmerset-build.cc).  The reads stay as FASTA text; counting happens on the device:
//   FromFASTA / FromReads  -> the text is kept (FromFASTA validates it on the device right
//                             away, ksh_fasta_plan)
//   ToKmerSet(cutoff)      -> ksh_fasta_plan / ksh_fasta_write (reads -> ACGT fragments in HBM),
//                             ksh_spss_decode_plan + ksh_kmer_count_write (sort per bucket,
//                             run lengths against the cutoff)
//   Size()                 -> the distinct k-mers = ToKmerSet(0).first.Size()
//   Add(kmer, v) / Get     -> host-side accessors for callers that build or inspect a counter
//                             k-mer by k-mer (tests): Add appends the k-mer v times as reads, Get
//                             bisects the cutoff (counts are uint8: 8 device runs at most).
// ValueType is the reference's saturating count type; the device path implements uint8.
// n_workers stays in the signatures and is ignored.
#ifndef KSC_CORE_KMER_COUNTER_H_
#define KSC_CORE_KMER_COUNTER_H_

#include <cstdint>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include "core/device.h"
#include "core/io.h"
#include "core/kmer.h"
#include "core/kmer_set.h"
#include "core/status.h"

template <int K, int N, typename KeyType, typename ValueType = std::uint8_t>
class KmerCounter {
  static_assert(std::is_same<ValueType, std::uint8_t>::value, "the device path counts in uint8");

 public:
  using Set = KmerSet<K, N, KeyType>;

  KmerCounter() = default;

  std::int64_t Size() const { return ToKmerSet(0, 1).first.Size(); }

  static KmerCounter FromReads(std::vector<std::string> reads, bool canonical, int /*n_workers*/) {
    KmerCounter c;
    c.canonical_ = canonical;
    for (const std::string& read : reads) c.AppendRead(read);
    return c;
  }

  static ksc::StatusOr<KmerCounter> FromFASTA(const std::string& file_name, const std::string& decompressor,
                                              bool canonical, int /*n_workers*/) {
    ksc::StatusOr<std::string> bytes = ksc::ReadBytes(file_name, decompressor);
    if (!bytes.ok()) return bytes.status();
    return FromFASTAText(std::move(bytes).value(), canonical);
  }

  static ksc::StatusOr<KmerCounter> FromFASTA(std::vector<std::string> lines, bool canonical, int /*n_workers*/) {
    std::string text;
    for (const std::string& line : lines) text += line + "\n";
    return FromFASTAText(std::move(text), canonical);
  }

  // Counts the k-mers of a FASTA file given as its bytes.
  static ksc::StatusOr<KmerCounter> FromFASTAText(std::string text, bool canonical) {
    KmerCounter c;
    c.canonical_ = canonical;
    c.fasta_ = std::move(text);
    // validate now, as the reference does at construction
    const ksh_geom g = Set::Geom();
    std::int64_t n_frag = 0, n_bases = 0;
    ksc::DeviceBuffer d_text(c.fasta_.size());
    if (!c.fasta_.empty())
      ksc::Check(ksh_ctx_memcpy_h2d(ksc::Ctx(), d_text.get(), c.fasta_.data(), c.fasta_.size()));
    const int rc = ksh_fasta_plan(ksc::Ctx(), &g, static_cast<const char*>(d_text.get()),
                                  static_cast<std::int64_t>(c.fasta_.size()), &n_frag, &n_bases);
    if (rc == KSH_FAILED_PRECONDITION) return ksc::FailedPreconditionError(ksh_last_error());
    ksc::Check(rc);
    return c;
  }

  // The k-mers seen at least `cutoff` times, and how many distinct k-mers were seen less often.
  std::pair<Set, std::int64_t> ToKmerSet(ValueType cutoff, int /*n_workers*/) const {
    const ksh_geom g = Set::Geom();
    ksc::DeviceBuffer off(std::size_t(Set::kBucketsNum + 1) * 8);
    std::int64_t n_frag = 0, n_bases = 0;
    ksc::DeviceBuffer d_text(fasta_.size());
    if (!fasta_.empty()) ksc::Check(ksh_ctx_memcpy_h2d(ksc::Ctx(), d_text.get(), fasta_.data(), fasta_.size()));
    ksc::Check(ksh_fasta_plan(ksc::Ctx(), &g, static_cast<const char*>(d_text.get()),
                              static_cast<std::int64_t>(fasta_.size()), &n_frag, &n_bases));
    ksc::DeviceBuffer words(std::size_t((n_bases + 31) / 32) * 8), lens(std::size_t(n_frag) * 4);
    ksc::Check(ksh_fasta_write(ksc::Ctx(), static_cast<std::uint64_t*>(words.get()),
                               static_cast<std::uint32_t*>(lens.get())));
    const ksh_spss_view v{static_cast<const std::uint64_t*>(words.get()),
                          static_cast<const std::uint32_t*>(lens.get()), n_frag, n_bases};
    std::int64_t n = 0, n_cut = 0;
    ksc::Check(ksh_spss_decode_plan(ksc::Ctx(), &g, &v, canonical_ ? 1 : 0, static_cast<std::int64_t*>(off.get()), &n));
    ksc::DeviceBuffer keys(std::size_t(n) * Set::kDeviceKeyBytes);
    ksc::Check(ksh_kmer_count_write(ksc::Ctx(), &g, &v, canonical_ ? 1 : 0, static_cast<std::int32_t>(cutoff),
                                    static_cast<std::int64_t*>(off.get()), keys.get(), &n, &n_cut));
    return std::make_pair(Set::FromDevice(std::move(off), std::move(keys), n), n_cut);
  }

  // Count of one k-mer (saturating at 255, like the reference's uint8).
  ValueType Get(const Kmer<K>& kmer) const {
    int lo = 0, hi = 255;  // the count is the largest c in [0, 255] with kmer in ToKmerSet(c)
    while (lo < hi) {
      const int mid = (lo + hi + 1) / 2;
      if (ToKmerSet(static_cast<ValueType>(mid), 1).first.Contains(kmer)) lo = mid; else hi = mid - 1;
    }
    return static_cast<ValueType>(lo);
  }

  // Increments the count of a k-mer by v (as v more reads holding just that k-mer).  The
  // reference's Add stores the k-mer as given (kmer_counter.h:262-271); a counter that was not
  // built by FromReads / FromFASTA(canonical = true) does the same here, a canonical one folds
  // it onto its canonical form when counting.
  KmerCounter& Add(const Kmer<K>& kmer, ValueType v) {
    for (int i = 0; i < int(v); i++) AppendRead(kmer.String());
    return *this;
  }

 private:
  void AppendRead(const std::string& read) {
    fasta_ += ">\n";
    fasta_ += read;
    fasta_ += '\n';
  }

  std::string fasta_;  // the reads, as FASTA text
  bool canonical_ = false;
};

#endif
