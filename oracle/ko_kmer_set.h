// ORACLE (test infrastructure, not product code).
//
// CPU restatement of the reference's hash-bucketed k-mer set.
//
// Follows lib/core/kmer_set.h:
//   :57-62    class KmerSet, 2^N buckets               -> KmerSet<KeyT>
//   :65-71    Size()
//   :81-105   Add / Remove / Contains (single k-mer)
//   :116-161  Find(pred) / Find()                      -> find() (see ordering note)
//   :164-187  Add(other) / Sub(other)
//   :191-219  Diff / Equals
//   :224-244  Hash (XOR of Kmer::Bits())
//   :286-305  free Add / Sub / Intersection (Intersection = lhs.Sub(Sub(lhs, rhs)))
//
// The reference's bucket is absl::flat_hash_set<KeyType> (absl 20200923.2, not
// vendored); FlatSet below is an open-addressing stand-in with the same
// operations (insert / erase / find / size / iterate).  Hash iteration order is
// the one thing the reference leaves unspecified; the oracle pins it:
// find() returns k-mers in ascending order.  With n_workers == 1 that is a
// legal execution of the reference (every place that depends on the order
// either sorts afterwards or accepts any order).
#ifndef ORACLE_KO_KMER_SET_H_
#define ORACLE_KO_KMER_SET_H_

#include <algorithm>
#include <cstdint>
#include <utility>
#include <vector>

#include "ko_kmer.h"

namespace ko {

template <typename KeyT>
class FlatSet {
 public:
  FlatSet() = default;

  std::size_t size() const { return size_; }

  void reserve(std::size_t n) {
    std::size_t want = 8;
    while (want * 3 < n * 4 + 4) want <<= 1;  // load factor <= 0.75
    if (want > keys_.size()) rehash(want);
  }

  bool contains(KeyT key) const {
    if (keys_.empty()) return false;
    std::size_t m = keys_.size() - 1;
    for (std::size_t i = slot(key) & m;; i = (i + 1) & m) {
      if (ctrl_[i] == kEmpty) return false;
      if (ctrl_[i] == kFull && keys_[i] == key) return true;
    }
  }

  bool insert(KeyT key) {
    if (keys_.empty() || (used_ + 1) * 4 > keys_.size() * 3) {
      std::size_t cap = std::max<std::size_t>(8, keys_.size());
      if ((size_ + 1) * 2 > cap) cap *= 2;  // else mostly tombstones: same capacity
      rehash(cap);
    }
    std::size_t m = keys_.size() - 1;
    std::size_t first_deleted = static_cast<std::size_t>(-1);
    for (std::size_t i = slot(key) & m;; i = (i + 1) & m) {
      if (ctrl_[i] == kEmpty) {
        std::size_t at = first_deleted != static_cast<std::size_t>(-1) ? first_deleted : i;
        if (ctrl_[at] == kEmpty) used_++;
        ctrl_[at] = kFull;
        keys_[at] = key;
        size_++;
        return true;
      }
      if (ctrl_[i] == kDeleted) {
        if (first_deleted == static_cast<std::size_t>(-1)) first_deleted = i;
      } else if (keys_[i] == key) {
        return false;
      }
    }
  }

  bool erase(KeyT key) {
    if (keys_.empty()) return false;
    std::size_t m = keys_.size() - 1;
    for (std::size_t i = slot(key) & m;; i = (i + 1) & m) {
      if (ctrl_[i] == kEmpty) return false;
      if (ctrl_[i] == kFull && keys_[i] == key) {
        ctrl_[i] = kDeleted;
        size_--;
        return true;
      }
    }
  }

  template <typename F>
  void for_each(F f) const {
    for (std::size_t i = 0; i < keys_.size(); i++) {
      if (ctrl_[i] == kFull) f(keys_[i]);
    }
  }

  void clear() {
    std::vector<KeyT>().swap(keys_);
    std::vector<std::uint8_t>().swap(ctrl_);
    size_ = used_ = 0;
  }

 private:
  static constexpr std::uint8_t kEmpty = 0, kFull = 1, kDeleted = 2;

  static std::size_t slot(KeyT key) {
    std::uint64_t h = static_cast<std::uint64_t>(key) * 0x9E3779B97F4A7C15ull;
    return static_cast<std::size_t>(h >> 20);
  }

  void rehash(std::size_t cap) {
    std::vector<KeyT> old_keys;
    std::vector<std::uint8_t> old_ctrl;
    old_keys.swap(keys_);
    old_ctrl.swap(ctrl_);
    keys_.assign(cap, KeyT());
    ctrl_.assign(cap, kEmpty);
    size_ = used_ = 0;
    std::size_t m = cap - 1;
    for (std::size_t j = 0; j < old_keys.size(); j++) {
      if (old_ctrl[j] != kFull) continue;
      std::size_t i = slot(old_keys[j]) & m;
      while (ctrl_[i] != kEmpty) i = (i + 1) & m;
      ctrl_[i] = kFull;
      keys_[i] = old_keys[j];
      size_++;
      used_++;
    }
  }

  std::vector<KeyT> keys_;
  std::vector<std::uint8_t> ctrl_;
  std::size_t size_ = 0;  // full slots
  std::size_t used_ = 0;  // full + deleted slots
};

template <typename KeyT>
class KmerSet {
 public:
  explicit KmerSet(const Geom& g) : g_(g), buckets_(static_cast<std::size_t>(g.n_buckets())) {}

  const Geom& geom() const { return g_; }

  std::int64_t size() const {
    std::int64_t sum = 0;
    for (const auto& b : buckets_) sum += static_cast<std::int64_t>(b.size());
    return sum;
  }

  void clear() {
    for (auto& b : buckets_) b.clear();
  }

  void add(std::uint64_t kmer) {
    std::int64_t bucket;
    std::uint64_t key;
    bucket_and_key(g_, kmer, &bucket, &key);
    buckets_[bucket].insert(static_cast<KeyT>(key));
  }

  void remove(std::uint64_t kmer) {
    std::int64_t bucket;
    std::uint64_t key;
    bucket_and_key(g_, kmer, &bucket, &key);
    buckets_[bucket].erase(static_cast<KeyT>(key));
  }

  bool contains(std::uint64_t kmer) const {
    std::int64_t bucket;
    std::uint64_t key;
    bucket_and_key(g_, kmer, &bucket, &key);
    return buckets_[bucket].contains(static_cast<KeyT>(key));
  }

  void reserve(std::int64_t n) {
    for (auto& b : buckets_) b.reserve(static_cast<std::size_t>(n / g_.n_buckets()));
  }

  // Find(pred): k-mers matching pred, ascending (ordering rule of the oracle).
  template <typename Pred>
  std::vector<std::uint64_t> find(Pred pred) const {
    std::vector<std::uint64_t> out;
    for (std::int64_t b = 0; b < g_.n_buckets(); b++) {
      std::size_t begin = out.size();
      buckets_[b].for_each([&](KeyT key) {
        std::uint64_t kmer = kmer_from_bucket_and_key(g_, b, key);
        if (pred(kmer)) out.push_back(kmer);
      });
      std::sort(out.begin() + static_cast<std::ptrdiff_t>(begin), out.end());
    }
    return out;
  }

  std::vector<std::uint64_t> find_all() const {
    return find([](std::uint64_t) { return true; });
  }

  KmerSet& add_set(const KmerSet& other) {
    for (std::int64_t b = 0; b < g_.n_buckets(); b++) {
      other.buckets_[b].for_each([&](KeyT key) { buckets_[b].insert(key); });
    }
    return *this;
  }

  KmerSet& sub_set(const KmerSet& other) {
    for (std::int64_t b = 0; b < g_.n_buckets(); b++) {
      other.buckets_[b].for_each([&](KeyT key) { buckets_[b].erase(key); });
    }
    return *this;
  }

  std::int64_t diff(const KmerSet& other) const {
    std::int64_t count = 0;
    for (std::int64_t b = 0; b < g_.n_buckets(); b++) {
      other.buckets_[b].for_each([&](KeyT key) {
        if (!buckets_[b].contains(key)) count += 1;
      });
    }
    for (std::int64_t b = 0; b < g_.n_buckets(); b++) {
      buckets_[b].for_each([&](KeyT key) {
        if (!other.buckets_[b].contains(key)) count += 1;
      });
    }
    return count;
  }

  bool equals(const KmerSet& other) const { return diff(other) == 0; }

  std::uint64_t hash() const {
    std::uint64_t h = 0;
    for (std::int64_t b = 0; b < g_.n_buckets(); b++) {
      buckets_[b].for_each([&](KeyT key) { h ^= kmer_from_bucket_and_key(g_, b, key); });
    }
    return h;
  }

  // Sorted keys of one bucket (used by the sampled-set restatement and by tests
  // that compare against the device layout: offsets[2^N+1] + sorted keys).
  std::vector<KeyT> sorted_bucket(std::int64_t b) const {
    std::vector<KeyT> v;
    v.reserve(buckets_[b].size());
    buckets_[b].for_each([&](KeyT key) { v.push_back(key); });
    std::sort(v.begin(), v.end());
    return v;
  }

  std::size_t bucket_size(std::int64_t b) const { return buckets_[b].size(); }

  // Bucket access for the multi-worker branches (ko_mt.h: ForEachBucket, kmer_set.h:260-282).
  const FlatSet<KeyT>& bucket(std::int64_t b) const { return buckets_[b]; }
  FlatSet<KeyT>& bucket(std::int64_t b) { return buckets_[b]; }

 private:
  Geom g_;
  std::vector<FlatSet<KeyT>> buckets_;
};

// Free functions, by value as the reference has them (kmer_set.h:286-305).
template <typename KeyT>
KmerSet<KeyT> set_add(KmerSet<KeyT> lhs, const KmerSet<KeyT>& rhs) {
  lhs.add_set(rhs);
  return lhs;
}

template <typename KeyT>
KmerSet<KeyT> set_sub(KmerSet<KeyT> lhs, const KmerSet<KeyT>& rhs) {
  lhs.sub_set(rhs);
  return lhs;
}

template <typename KeyT>
KmerSet<KeyT> set_intersection(KmerSet<KeyT> lhs, const KmerSet<KeyT>& rhs) {
  KmerSet<KeyT> d = set_sub(lhs, rhs);
  lhs.sub_set(d);
  return lhs;
}

}  // namespace ko

#endif
