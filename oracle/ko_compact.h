// ORACLE (test infrastructure, not product code).
//
// CPU restatement of the reference's SPSS container.
//
// Follows lib/core/kmer_set_compact.h:
//   :36-47    FromKmerSet                         -> Compact::from_kmer_set
//   :52-55    ToKmerSet                           -> to_kmer_set
//   :62-87    Dump / Load (text, one string per line; lib/core/io.h:20-126)
//   :90-112   Size  = sum(len - K + 1)
//   :115      Weight = data bits / 2
//   :120-203  GetSampledKmerSet (re-decode every string, keep keys whose bucket
//             is listed, sort each bucket)
//   :206-266  ctor: 2 bits per base, first bit = high bit of the code
//             (A=00 C=01 G=10 T=11), lengths stored as len - K, StreamVByte 0124
//   :269-336  GetLengths / ToStrings
//
// Bit container.  The reference keeps the bits in a std::vector<bool> whose word
// layout is libstdc++'s business and never leaves the process.  The oracle (and
// the device format of this repo) fixes one: 64-bit words, base j of the
// concatenated stream in word j/32 at bits [63-2(j%32), 62-2(j%32)], i.e. the
// reference's bit index 2j is the more significant of the pair.
//
// StreamVByte.  lemire/streamvbyte v0.4.1 (extern/install.sh:70-80) is not under
// /root/reference.  The "0124" variant is restated here from its published
// format: ceil(n/4) control bytes, then the data bytes; value i has a 2-bit code
// in control byte i/4 at bits [2(i%4), 2(i%4)+1]; code 0/1/2/3 = 0/1/2/4 data
// bytes, little-endian, smallest width that holds the value.  The compressed
// bytes never reach a file in the reference (kmer_set_compact.h:339-347), so
// byte-level parity with upstream is "parity unpinned"; round trips are pinned
// by test/kmer_set_compact.cc:53-69.
#ifndef ORACLE_KO_COMPACT_H_
#define ORACLE_KO_COMPACT_H_

#include <cstdint>
#include <cstdio>
#include <fstream>
#include <string>
#include <vector>

#include "ko_kmer.h"
#include "ko_kmer_set.h"
#include "ko_spss.h"

namespace ko {

inline std::size_t svb_max_compressed_bytes(std::uint32_t n) {
  return (static_cast<std::size_t>(n) + 3) / 4 + static_cast<std::size_t>(n) * 4;
}

inline std::size_t svb_encode_0124(const std::uint32_t* in, std::uint32_t n, std::uint8_t* out) {
  std::uint8_t* ctrl = out;
  std::uint8_t* data = out + (static_cast<std::size_t>(n) + 3) / 4;
  for (std::size_t i = 0; i < (static_cast<std::size_t>(n) + 3) / 4; i++) ctrl[i] = 0;
  for (std::uint32_t i = 0; i < n; i++) {
    const std::uint32_t v = in[i];
    int code, bytes;
    if (v == 0) {
      code = 0, bytes = 0;
    } else if (v < (1u << 8)) {
      code = 1, bytes = 1;
    } else if (v < (1u << 16)) {
      code = 2, bytes = 2;
    } else {
      code = 3, bytes = 4;
    }
    ctrl[i / 4] |= static_cast<std::uint8_t>(code << (2 * (i % 4)));
    for (int b = 0; b < bytes; b++) *data++ = static_cast<std::uint8_t>(v >> (8 * b));
  }
  return static_cast<std::size_t>(data - out);
}

inline std::size_t svb_decode_0124(const std::uint8_t* in, std::uint32_t* out, std::uint32_t n) {
  const std::uint8_t* ctrl = in;
  const std::uint8_t* data = in + (static_cast<std::size_t>(n) + 3) / 4;
  for (std::uint32_t i = 0; i < n; i++) {
    const int code = (ctrl[i / 4] >> (2 * (i % 4))) & 3;
    const int bytes = code == 3 ? 4 : code;
    std::uint32_t v = 0;
    for (int b = 0; b < bytes; b++) v |= static_cast<std::uint32_t>(*data++) << (8 * b);
    out[i] = v;
  }
  return static_cast<std::size_t>(data - in);
}

// lib/core/io.h:20-126 without the popen branch (a (de)compressor pipe is
// outside the hot path; the text format is what matters for parity).
inline bool read_lines(const std::string& file_name, std::vector<std::string>* lines) {
  std::ifstream file(file_name);
  if (file.fail()) return false;
  std::string s;
  while (std::getline(file, s)) lines->push_back(s);
  return true;
}

inline bool write_lines(const std::string& file_name, const std::vector<std::string>& lines) {
  std::ofstream file(file_name);
  if (file.fail()) return false;
  for (const std::string& line : lines) file << line << '\n';
  return true;
}

class Compact {
 public:
  Compact() = default;

  Compact(const Geom& g, const std::vector<std::string>& spss) : g_(g) {
    n_ = static_cast<std::int64_t>(spss.size());
    std::vector<std::uint32_t> lengths(static_cast<std::size_t>(n_));
    std::int64_t size = 0;
    for (std::int64_t i = 0; i < n_; i++) {
      const std::uint32_t length = static_cast<std::uint32_t>(spss[i].length());
      lengths[i] = length - static_cast<std::uint32_t>(g.k);
      size += static_cast<std::int64_t>(length) * 2;
    }
    n_bits_ = size;
    words_.assign(static_cast<std::size_t>((size + 63) / 64), 0);
    std::int64_t base = 0;
    for (std::int64_t i = 0; i < n_; i++) {
      for (char ch : spss[i]) {
        const std::uint64_t code = static_cast<std::uint64_t>(base_code(ch));
        words_[base / 32] |= code << (62 - 2 * (base % 32));
        base++;
      }
    }
    lengths_compressed_.resize(svb_max_compressed_bytes(static_cast<std::uint32_t>(n_)));
    const std::size_t sz = svb_encode_0124(lengths.data(), static_cast<std::uint32_t>(n_),
                                           lengths_compressed_.data());
    lengths_compressed_.resize(sz);
    lengths_compressed_.shrink_to_fit();
  }

  // FromKmerSet(kmer_set, canonical, fast, n_workers) (kmer_set_compact.h:36-47).
  template <typename KeyT>
  static Compact from_kmer_set(const KmerSet<KeyT>& kmer_set, bool canon = true, bool fast = true) {
    return Compact(kmer_set.geom(), canon ? spss_canonical(kmer_set, fast) : spss_directed(kmer_set));
  }

  const Geom& geom() const { return g_; }
  std::int64_t n_strings() const { return n_; }
  std::int64_t weight() const { return n_bits_ / 2; }
  const std::vector<std::uint64_t>& words() const { return words_; }
  const std::vector<std::uint8_t>& lengths_compressed() const { return lengths_compressed_; }

  std::vector<std::uint32_t> lengths() const {
    std::vector<std::uint32_t> lengths(static_cast<std::size_t>(n_));
    svb_decode_0124(lengths_compressed_.data(), lengths.data(), static_cast<std::uint32_t>(n_));
    for (auto& l : lengths) l += static_cast<std::uint32_t>(g_.k);
    return lengths;
  }

  std::int64_t size() const {
    std::int64_t s = 0;
    for (std::uint32_t l : lengths()) s += static_cast<std::int64_t>(l) - g_.k + 1;
    return s;
  }

  std::vector<std::string> to_strings() const {
    const std::vector<std::uint32_t> lens = lengths();
    std::vector<std::string> strings(static_cast<std::size_t>(n_));
    std::int64_t base = 0;
    for (std::int64_t i = 0; i < n_; i++) {
      strings[i].resize(lens[i]);
      for (std::uint32_t j = 0; j < lens[i]; j++, base++)
        strings[i][j] = base_char(static_cast<int>((words_[base / 32] >> (62 - 2 * (base % 32))) & 3));
    }
    return strings;
  }

  template <typename KeyT>
  KmerSet<KeyT> to_kmer_set(bool canon) const {
    return kmer_set_from_spss<KeyT>(g_, to_strings(), canon);
  }

  // bucket_ids[i] = j  <->  out[i] holds the sorted keys of bucket j.
  template <typename KeyT>
  std::vector<std::vector<KeyT>> sampled(const std::vector<int>& bucket_ids, bool canon) const {
    const std::vector<std::string> spss = to_strings();
    std::unordered_map<int, int> map;
    for (std::size_t i = 0; i < bucket_ids.size(); i++) map[bucket_ids[i]] = static_cast<int>(i);
    std::vector<std::vector<KeyT>> buckets(bucket_ids.size());
    for (const std::string& s : spss) {
      for (int j = 0; j < static_cast<int>(s.length()) - g_.k + 1; j++) {
        std::uint64_t kmer = kmer_from_string(s.data() + j, g_.k);
        if (canon) kmer = canonical(kmer, g_.k);
        std::int64_t bucket;
        std::uint64_t key;
        bucket_and_key(g_, kmer, &bucket, &key);
        auto it = map.find(static_cast<int>(bucket));
        if (it == map.end()) continue;
        buckets[it->second].push_back(static_cast<KeyT>(key));
      }
    }
    for (auto& b : buckets) std::sort(b.begin(), b.end());
    return buckets;
  }

  bool dump(const std::string& file_name) const { return write_lines(file_name, to_strings()); }

  static bool load(const Geom& g, const std::string& file_name, Compact* out) {
    std::vector<std::string> lines;
    if (!read_lines(file_name, &lines)) return false;
    *out = Compact(g, lines);
    return true;
  }

 private:
  Geom g_;
  std::int64_t n_ = 0;
  std::int64_t n_bits_ = 0;
  std::vector<std::uint8_t> lengths_compressed_;
  std::vector<std::uint64_t> words_;
};

}  // namespace ko

#endif
