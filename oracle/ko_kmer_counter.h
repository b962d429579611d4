// ORACLE (test infrastructure, not product code).
//
// CPU restatement of the reference's k-mer counter (the step before the loop).
//
// Follows lib/core/kmer_counter.h:
//   :27-38    AddWithMax (saturating add for integral ValueType)   -> add_with_max
//   :49-58    class KmerCounter, 2^N buckets of key -> uint8 count  -> KmerCounter
//   :53-61    Size(): distinct k-mers
//   :64-133   FromReads: every read split at 'N' (absl::StrSplit), every K-long window of a
//             fragment counted, canonical or not                     -> from_reads
//   :136-206  FromFASTA: an even number of lines, even lines non-empty and starting with '>',
//             odd lines over ACGTN; otherwise FailedPrecondition       -> from_fasta_lines
//   :209-243  ToKmerSet(cutoff): keys with count >= cutoff; second result = number of
//             distinct k-mers with count < cutoff                    -> to_kmer_set
//   :246-271  Get / Add (single k-mer)                               -> get / add
// Lines come from ReadLines (lib/core/io.h:20-76): the std::getline split is restated in
// split_lines (a final line without '\n' counts, an empty input has no lines).
//
// The reference keeps each bucket in an absl::flat_hash_map (not vendored); std::map is used
// here: only lookups and whole-bucket iteration are needed, and nothing depends on the order.
#ifndef ORACLE_KO_KMER_COUNTER_H_
#define ORACLE_KO_KMER_COUNTER_H_

#include <algorithm>
#include <cstdint>
#include <map>
#include <string>
#include <utility>
#include <vector>

#include "ko_kmer.h"
#include "ko_kmer_set.h"

namespace ko {

inline std::uint8_t add_with_max(std::uint8_t x, std::uint8_t y) {
  return static_cast<std::uint8_t>(std::min<std::int64_t>(255, std::int64_t(x) + std::int64_t(y)));
}

inline std::vector<std::string> split_lines(const std::string& text) {
  std::vector<std::string> lines;
  std::size_t at = 0;
  while (at < text.size()) {
    const std::size_t nl = text.find('\n', at);
    if (nl == std::string::npos) {
      lines.push_back(text.substr(at));
      break;
    }
    lines.push_back(text.substr(at, nl - at));
    at = nl + 1;
  }
  return lines;
}

class KmerCounter {
 public:
  explicit KmerCounter(const Geom& g) : g_(g), buckets_(static_cast<std::size_t>(g.n_buckets())) {}

  std::int64_t size() const {
    std::int64_t sum = 0;
    for (const auto& b : buckets_) sum += static_cast<std::int64_t>(b.size());
    return sum;
  }

  void add(std::uint64_t kmer, std::uint8_t v) {
    std::int64_t bucket;
    std::uint64_t key;
    bucket_and_key(g_, kmer, &bucket, &key);
    std::uint8_t& slot = buckets_[static_cast<std::size_t>(bucket)][key];
    slot = add_with_max(slot, v);
  }

  std::uint8_t get(std::uint64_t kmer) const {
    std::int64_t bucket;
    std::uint64_t key;
    bucket_and_key(g_, kmer, &bucket, &key);
    const auto& b = buckets_[static_cast<std::size_t>(bucket)];
    const auto it = b.find(key);
    return it == b.end() ? 0 : it->second;
  }

  // kmer_counter.h:64-133 with n_workers == 1
  void from_reads(const std::vector<std::string>& reads, bool canonical_flag) {
    for (const std::string& read : reads) {
      std::size_t at = 0;
      while (true) {  // absl::StrSplit(read, 'N')
        const std::size_t n_at = read.find('N', at);
        const std::string fragment = read.substr(at, n_at == std::string::npos ? std::string::npos : n_at - at);
        for (std::size_t j = 0; j + static_cast<std::size_t>(g_.k) <= fragment.length(); j++) {
          const std::uint64_t kmer = kmer_from_string(fragment.c_str() + j, g_.k);
          add(canonical_flag ? canonical(kmer, g_.k) : kmer, 1);
        }
        if (n_at == std::string::npos) break;
        at = n_at + 1;
      }
    }
  }

  // kmer_counter.h:157-206.  0 = ok, 1 = odd number of lines, 2 = invalid FASTA file.
  int from_fasta_lines(const std::vector<std::string>& lines, bool canonical_flag) {
    if (lines.size() % 2 != 0) return 1;
    std::vector<std::string> reads(lines.size() / 2);
    for (std::size_t i = 0; i < lines.size(); i++) {
      const std::string& line = lines[i];
      if (i % 2 == 0) {
        if (line.empty() || line[0] != '>') return 2;
      } else {
        for (char c : line)
          if (c != 'A' && c != 'C' && c != 'G' && c != 'T' && c != 'N') return 2;
        reads[i / 2] = line;
      }
    }
    from_reads(reads, canonical_flag);
    return 0;
  }

  // kmer_counter.h:209-243
  template <typename KeyT>
  std::pair<KmerSet<KeyT>, std::int64_t> to_kmer_set(int cutoff) const {
    KmerSet<KeyT> set(g_);
    std::int64_t cutoff_count = 0;
    for (std::size_t b = 0; b < buckets_.size(); b++) {
      for (const auto& p : buckets_[b]) {
        if (int(p.second) < cutoff) {
          cutoff_count += 1;
          continue;
        }
        set.add(kmer_from_bucket_and_key(g_, static_cast<std::int64_t>(b), p.first));
      }
    }
    return std::make_pair(std::move(set), cutoff_count);
  }

 private:
  Geom g_;
  std::vector<std::map<std::uint64_t, std::uint8_t>> buckets_;
};

}  // namespace ko

#endif
