// ORACLE (test infrastructure, not product code).
//
// The reference's n_workers > 1 branches of the hot path, restated for the CPU baseline of
// bench.py ("cpu_baseline": the same loop on the host's cores) and checked against the
// n_workers == 1 oracle at the level the reference itself guarantees for them: set membership
// of every node, the merge sequence and N_proc -- NOT the SPSS strings, which depend on thread
// interleaving in the reference as soon as n_workers > 1 (lock order of the greedy matching,
// spss.h:1448-1496; append order of the per-thread buffers).
//
// Structure follows the reference: every parallel step builds a NEW pool of n_workers threads,
// posts n_workers^2 range chunks (Range::Split, range.h:52-77) and joins
// (SURVEY.md 8b "Threading").
//   lib/core/kmer_set.h:116-187,260-282   ForEachBucket, Find, Add, Sub    -> for_each_bucket, find, ...
//   lib/core/kmer_set_compact.h:120-203   GetSampledKmerSet                 -> sampled
//   lib/core/kmer_set_compact.h:290-336   ToStrings                         -> to_strings
//   lib/core/spss.h:230-615               GetUnitigsCanonical               -> unitigs_canonical
//   lib/core/spss.h:619-695               prefix / suffix maps              -> end_map
//   lib/core/spss.h:1358-1832             greedy matching under bucket locks, ParallelDisjointSet,
//                                         loop cut, terminals, stitch       -> spss_from_unitigs
//   lib/core/spss.h:1861-1941             GetKmerSetFromSPSS                -> kmer_set_from_spss
//   lib/core/kmer_set_set.h:109-427       the constructor                   -> KmerSetSetMT
#ifndef ORACLE_KO_MT_H_
#define ORACLE_KO_MT_H_

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <deque>
#include <functional>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <unordered_set>
#include <utility>
#include <vector>

#include "ko_compact.h"
#include "ko_dsu.h"
#include "ko_kmer.h"
#include "ko_kmer_set.h"
#include "ko_kmer_set_set.h"
#include "ko_spss.h"

namespace ko {
namespace mt {

// boost::asio::thread_pool as the reference uses it: construct, post, join, destroy.
class Pool {
 public:
  explicit Pool(int n) {
    for (int i = 0; i < n; i++) threads_.emplace_back([this] { run(); });
  }
  ~Pool() { join(); }
  void post(std::function<void()> f) {
    {
      std::lock_guard<std::mutex> lck(mu_);
      tasks_.push_back(std::move(f));
    }
    cv_.notify_one();
  }
  void join() {
    {
      std::lock_guard<std::mutex> lck(mu_);
      closed_ = true;
    }
    cv_.notify_all();
    for (std::thread& t : threads_)
      if (t.joinable()) t.join();
  }

 private:
  void run() {
    while (true) {
      std::function<void()> f;
      {
        std::unique_lock<std::mutex> lck(mu_);
        cv_.wait(lck, [this] { return closed_ || !tasks_.empty(); });
        if (tasks_.empty()) return;
        f = std::move(tasks_.front());
        tasks_.pop_front();
      }
      f();
    }
  }
  std::mutex mu_;
  std::condition_variable cv_;
  std::deque<std::function<void()>> tasks_;
  std::vector<std::thread> threads_;
  bool closed_ = false;
};

// Range::Split (range.h:52-77): n chunks, the first n - r of size s / n, then r of size s / n + 1.
inline std::vector<std::pair<std::int64_t, std::int64_t>> split(std::int64_t begin, std::int64_t end,
                                                                 std::int64_t n) {
  const std::int64_t size = end - begin;
  const std::int64_t small = size / n, large_n = size - small * n, small_n = n - large_n;
  std::vector<std::pair<std::int64_t, std::int64_t>> out;
  std::int64_t at = begin;
  for (std::int64_t i = 0; i < small_n; i++, at += small) out.emplace_back(at, at + small);
  for (std::int64_t i = 0; i < large_n; i++, at += small + 1) out.emplace_back(at, at + small + 1);
  return out;
}

// for (range : Range(b, e).Split(w * w)) post(pool, [&, range] { f(range) }); pool.join();
template <typename F>
void for_chunks(std::int64_t begin, std::int64_t end, int w, F f) {
  Pool pool(w);
  for (const auto& r : split(begin, end, std::int64_t(w) * w)) pool.post([&f, r] { f(r.first, r.second); });
  pool.join();
}

// kmer_set.h:260-282
template <typename KeyT, typename F>
void for_each_bucket(const KmerSet<KeyT>& s, int w, F f) {
  for_chunks(0, s.geom().n_buckets(), w, [&](std::int64_t b0, std::int64_t b1) {
    for (std::int64_t b = b0; b < b1; b++) f(s.bucket(b), b);
  });
}

// kmer_set.h:116-161 (unsorted: per-bucket buffers appended under one mutex)
template <typename KeyT, typename Pred>
std::vector<std::uint64_t> find(const KmerSet<KeyT>& s, int w, Pred pred) {
  std::vector<std::uint64_t> kmers;
  std::mutex mu;
  for_each_bucket(s, w, [&](const FlatSet<KeyT>& bucket, std::int64_t b) {
    std::vector<std::uint64_t> buf;
    buf.reserve(bucket.size());
    bucket.for_each([&](KeyT key) {
      const std::uint64_t kmer = kmer_from_bucket_and_key(s.geom(), b, key);
      if (pred(kmer)) buf.push_back(kmer);
    });
    std::lock_guard<std::mutex> lck(mu);
    kmers.insert(kmers.end(), buf.begin(), buf.end());
  });
  return kmers;
}

template <typename KeyT>
void sub_set(KmerSet<KeyT>* lhs, const KmerSet<KeyT>& other, int w) {  // kmer_set.h:177-187
  for_each_bucket(other, w, [&](const FlatSet<KeyT>& ob, std::int64_t b) {
    FlatSet<KeyT>& mine = lhs->bucket(b);
    ob.for_each([&](KeyT key) { mine.erase(key); });
  });
}

template <typename KeyT>
void add_set(KmerSet<KeyT>* lhs, const KmerSet<KeyT>& other, int w) {  // kmer_set.h:164-174
  for_each_bucket(other, w, [&](const FlatSet<KeyT>& ob, std::int64_t b) {
    FlatSet<KeyT>& mine = lhs->bucket(b);
    ob.for_each([&](KeyT key) { mine.insert(key); });
  });
}

// Intersection(lhs, rhs) = lhs.Sub(Sub(lhs, rhs)) with both by-value copies (kmer_set.h:294-305)
template <typename KeyT>
KmerSet<KeyT> intersection(KmerSet<KeyT> lhs, const KmerSet<KeyT>& rhs, int w) {
  KmerSet<KeyT> d = lhs;
  sub_set(&d, rhs, w);
  sub_set(&lhs, d, w);
  return lhs;
}

// boost::sort::block_indirect_sort stand-in: chunk sorts on the pool, then a merge tree.
inline void parallel_sort(std::vector<std::uint64_t>* v, int w) {
  const std::int64_t n = std::int64_t(v->size());
  if (n < 4096 || w == 1) {
    std::sort(v->begin(), v->end());
    return;
  }
  auto chunks = split(0, n, w);
  {
    Pool pool(w);
    for (const auto& c : chunks)
      pool.post([v, c] { std::sort(v->begin() + c.first, v->begin() + c.second); });
    pool.join();
  }
  while (chunks.size() > 1) {
    std::vector<std::pair<std::int64_t, std::int64_t>> next;
    Pool pool(w);
    for (std::size_t i = 0; i + 1 < chunks.size(); i += 2) {
      const auto a = chunks[i], b = chunks[i + 1];
      pool.post([v, a, b] { std::inplace_merge(v->begin() + a.first, v->begin() + a.second, v->begin() + b.second); });
      next.emplace_back(a.first, b.second);
    }
    if (chunks.size() % 2) next.push_back(chunks.back());
    pool.join();
    chunks.swap(next);
  }
}

// ---- Compact: ToStrings / GetSampledKmerSet / GetKmerSetFromSPSS ----------------------------------
inline std::vector<std::string> to_strings(const Compact& c, int w) {  // kmer_set_compact.h:290-336
  const std::vector<std::uint32_t> lens = c.lengths();
  const std::int64_t n = c.n_strings();
  std::vector<std::int64_t> positions(static_cast<std::size_t>(n) + 1, 0);
  for (std::int64_t i = 0; i < n; i++) positions[i + 1] = positions[i] + lens[i];
  std::vector<std::string> strings(static_cast<std::size_t>(n));
  const std::vector<std::uint64_t>& words = c.words();
  for_chunks(0, n, w, [&](std::int64_t i0, std::int64_t i1) {
    for (std::int64_t i = i0; i < i1; i++) {
      strings[i].resize(lens[i]);
      std::int64_t base = positions[i];
      for (std::uint32_t j = 0; j < lens[i]; j++, base++)
        strings[i][j] = base_char(static_cast<int>((words[base / 32] >> (62 - 2 * (base % 32))) & 3));
    }
  });
  return strings;
}

template <typename KeyT>
KmerSet<KeyT> kmer_set_from_spss(const Geom& g, const std::vector<std::string>& spss, bool canon, int w) {
  KmerSet<KeyT> kmer_set(g);
  std::atomic<std::int64_t> size{0};
  for_chunks(0, std::int64_t(spss.size()), w, [&](std::int64_t i0, std::int64_t i1) {
    std::int64_t buf = 0;
    for (std::int64_t i = i0; i < i1; i++) buf += std::int64_t(spss[i].length()) - g.k + 1;
    size += buf;
  });
  kmer_set.reserve(size);
  const std::int64_t nb = g.n_buckets();
  std::vector<std::mutex> mus(static_cast<std::size_t>(nb));
  for_chunks(0, std::int64_t(spss.size()), w, [&](std::int64_t i0, std::int64_t i1) {
    std::vector<std::vector<KeyT>> buf(static_cast<std::size_t>(nb));
    for (std::int64_t i = i0; i < i1; i++) {
      const std::string& s = spss[i];
      for (int j = 0; j < int(s.length()) - g.k + 1; j++) {
        std::uint64_t kmer = kmer_from_string(s.data() + j, g.k);
        if (canon) kmer = canonical(kmer, g.k);
        std::int64_t bucket;
        std::uint64_t key;
        bucket_and_key(g, kmer, &bucket, &key);
        buf[bucket].push_back(static_cast<KeyT>(key));
      }
    }
    std::vector<bool> done(static_cast<std::size_t>(nb));
    std::int64_t done_count = 0;
    while (done_count < nb) {
      for (std::int64_t b = 0; b < nb; b++) {
        if (!done[b] && mus[b].try_lock()) {
          FlatSet<KeyT>& dst = kmer_set.bucket(b);
          for (KeyT key : buf[b]) dst.insert(key);
          mus[b].unlock();
          done[b] = true;
          done_count++;
        }
      }
    }
  });
  return kmer_set;
}

template <typename KeyT>
std::vector<std::vector<KeyT>> sampled(const Compact& c, const std::vector<int>& bucket_ids, bool canon, int w) {
  const Geom& g = c.geom();
  const std::vector<std::string> spss = to_strings(c, w);
  const int n_buckets = int(bucket_ids.size());
  std::unordered_map<int, int> map;
  for (int i = 0; i < n_buckets; i++) map[bucket_ids[i]] = i;
  std::vector<std::vector<KeyT>> buckets(static_cast<std::size_t>(n_buckets));
  std::vector<std::mutex> mus(static_cast<std::size_t>(n_buckets));
  for_chunks(0, std::int64_t(spss.size()), w, [&](std::int64_t i0, std::int64_t i1) {
    std::vector<std::vector<KeyT>> buf(static_cast<std::size_t>(n_buckets));
    for (std::int64_t i = i0; i < i1; i++) {
      const std::string& s = spss[i];
      for (int j = 0; j < int(s.length()) - g.k + 1; j++) {
        std::uint64_t kmer = kmer_from_string(s.data() + j, g.k);
        if (canon) kmer = canonical(kmer, g.k);
        std::int64_t bucket;
        std::uint64_t key;
        bucket_and_key(g, kmer, &bucket, &key);
        auto it = map.find(int(bucket));
        if (it == map.end()) continue;
        buf[it->second].push_back(static_cast<KeyT>(key));
      }
    }
    int done_count = 0;
    std::vector<bool> done(static_cast<std::size_t>(n_buckets));
    while (done_count < n_buckets) {
      for (int b = 0; b < n_buckets; b++) {
        if (!done[b] && mus[b].try_lock()) {
          buckets[b].insert(buckets[b].end(), buf[b].begin(), buf[b].end());
          mus[b].unlock();
          done[b] = true;
          done_count++;
        }
      }
    }
  });
  {
    Pool pool(w);
    for (int b = 0; b < n_buckets; b++) pool.post([&buckets, b] { std::sort(buckets[b].begin(), buckets[b].end()); });
    pool.join();
  }
  return buckets;
}

// ---- GetUnitigsCanonical, n_workers > 1 (spss.h:230-615) ---------------------------------------
template <typename KeyT>
std::vector<std::string> unitigs_canonical(const KmerSet<KeyT>& kmer_set, int w) {
  const int k = kmer_set.geom().k;
  using Neighbor = std::pair<std::uint64_t, bool>;
  const auto neighbors_right = [&](std::uint64_t kmer, Neighbor* out) {
    int cnt = 0;
    for (int c = 0; c < 4; c++) {
      const std::uint64_t nx = next(kmer, k, c);
      if (kmer != nx && kmer_set.contains(nx)) out[cnt++] = {nx, false};
      const std::uint64_t nxc = complement(nx, k);
      if (kmer != nxc && kmer_set.contains(nxc)) out[cnt++] = {nxc, true};
    }
    return cnt;
  };
  const auto neighbors_left = [&](std::uint64_t kmer, Neighbor* out) {
    int cnt = 0;
    for (int c = 0; c < 4; c++) {
      const std::uint64_t pv = prev(kmer, k, c);
      if (kmer != pv && kmer_set.contains(pv)) out[cnt++] = {pv, false};
      const std::uint64_t pvc = complement(pv, k);
      if (kmer != pvc && kmer_set.contains(pvc)) out[cnt++] = {pvc, true};
    }
    return cnt;
  };
  const auto is_terminal_left = [&](std::uint64_t kmer) {
    Neighbor nb[8], tmp[8];
    if (neighbors_left(kmer, nb) != 1) return true;
    return (nb[0].second ? neighbors_left(nb[0].first, tmp) : neighbors_right(nb[0].first, tmp)) != 1;
  };
  const auto is_terminal_right = [&](std::uint64_t kmer) {
    Neighbor nb[8], tmp[8];
    if (neighbors_right(kmer, nb) != 1) return true;
    return (nb[0].second ? neighbors_right(nb[0].first, tmp) : neighbors_left(nb[0].first, tmp)) != 1;
  };

  std::vector<std::uint64_t> terminals_left = find(kmer_set, w, is_terminal_left);
  parallel_sort(&terminals_left, w);
  std::vector<std::uint64_t> terminals_right = find(kmer_set, w, is_terminal_right);
  parallel_sort(&terminals_right, w);
  std::vector<std::uint64_t> terminals_both;
  std::set_intersection(terminals_left.begin(), terminals_left.end(), terminals_right.begin(),
                        terminals_right.end(), std::back_inserter(terminals_both));
  {
    std::vector<std::uint64_t> buf;
    std::set_difference(terminals_left.begin(), terminals_left.end(), terminals_both.begin(),
                        terminals_both.end(), std::back_inserter(buf));
    buf.swap(terminals_left);
  }
  {
    std::vector<std::uint64_t> buf;
    std::set_difference(terminals_right.begin(), terminals_right.end(), terminals_both.begin(),
                        terminals_both.end(), std::back_inserter(buf));
    buf.swap(terminals_right);
  }

  const auto find_path = [&](std::uint64_t start, bool is_right_side) {
    std::uint64_t current = start;
    std::vector<std::uint64_t> path;
    while (true) {
      path.push_back(is_right_side ? current : complement(current, k));
      if (is_right_side ? is_terminal_right(current) : is_terminal_left(current)) break;
      Neighbor nb[8];
      if (is_right_side) neighbors_right(current, nb); else neighbors_left(current, nb);
      current = nb[0].first;
      if (nb[0].second) is_right_side = !is_right_side;
    }
    return path;
  };

  std::vector<std::string> unitigs;
  FlatSet<std::uint64_t> visited;
  unitigs.reserve(terminals_both.size() + (terminals_left.size() + terminals_right.size()) / 2);
  visited.reserve(static_cast<std::size_t>(kmer_set.size()));

  const auto move_from_buffer = [&](std::mutex& mu_unitigs, std::mutex& mu_visited,
                                    std::vector<std::string>& buf_unitigs,
                                    std::vector<std::uint64_t>& buf_visited) {
    bool done_unitigs = false, done_visited = false;
    while (!done_unitigs || !done_visited) {
      if (!done_unitigs && mu_unitigs.try_lock()) {
        for (std::string& u : buf_unitigs) unitigs.push_back(std::move(u));
        mu_unitigs.unlock();
        done_unitigs = true;
      }
      if (!done_visited && mu_visited.try_lock()) {
        for (std::uint64_t x : buf_visited) visited.insert(x);
        mu_visited.unlock();
        done_visited = true;
      }
    }
  };

  {
    std::mutex mu_unitigs, mu_visited;
    for_chunks(0, std::int64_t(terminals_both.size()), w, [&](std::int64_t i0, std::int64_t i1) {
      std::vector<std::string> buf_unitigs;
      std::vector<std::uint64_t> buf_visited;
      for (std::int64_t i = i0; i < i1; i++) {
        buf_unitigs.push_back(kmer_to_string(terminals_both[i], k));
        buf_visited.push_back(terminals_both[i]);
      }
      move_from_buffer(mu_unitigs, mu_visited, buf_unitigs, buf_visited);
    });
  }
  for (int side = 0; side < 2; side++) {
    const std::vector<std::uint64_t>& terminals = side == 0 ? terminals_left : terminals_right;
    std::mutex mu_unitigs, mu_visited;
    for_chunks(0, std::int64_t(terminals.size()), w, [&](std::int64_t i0, std::int64_t i1) {
      std::vector<std::string> buf_unitigs;
      std::vector<std::uint64_t> buf_visited;
      for (std::int64_t i = i0; i < i1; i++) {
        std::vector<std::uint64_t> path = find_path(terminals[i], side == 0);
        if (canonical(path.front(), k) < canonical(path.back(), k)) continue;
        for (std::uint64_t x : path) buf_visited.push_back(canonical(x, k));
        buf_unitigs.push_back(concatenate_kmers(path, k));
      }
      move_from_buffer(mu_unitigs, mu_visited, buf_unitigs, buf_visited);
    });
  }

  std::vector<std::uint64_t> not_visited = find(kmer_set, w, [&](std::uint64_t x) { return !visited.contains(x); });
  std::sort(not_visited.begin(), not_visited.end());  // the oracle's ordering rule for the serial loop pass
  for (std::uint64_t start : not_visited) {
    if (visited.contains(start)) continue;
    bool is_right_side = true;
    std::uint64_t current = start;
    std::vector<std::uint64_t> path;
    while (!visited.contains(current)) {
      visited.insert(current);
      path.push_back(is_right_side ? current : complement(current, k));
      Neighbor nb[8];
      if (is_right_side) neighbors_right(current, nb); else neighbors_left(current, nb);
      current = nb[0].first;
      if (nb[0].second) is_right_side = !is_right_side;
    }
    unitigs.push_back(concatenate_kmers(path, k));
  }
  return unitigs;
}

using EndHash = std::unordered_map<std::uint64_t, std::vector<std::int64_t>>;

// spss.h:619-695: per-chunk maps merged under one mutex with map.insert
inline EndHash end_map(const std::vector<std::string>& unitigs, int k, bool suffix, int w) {
  EndHash out;
  std::mutex mu;
  for_chunks(0, std::int64_t(unitigs.size()), w, [&](std::int64_t i0, std::int64_t i1) {
    EndHash buf;
    for (std::int64_t i = i0; i < i1; i++) {
      const std::string& u = unitigs[i];
      buf[kmer_from_string(suffix ? u.data() + u.length() - k : u.data(), k)].push_back(i);
    }
    std::lock_guard<std::mutex> lck(mu);
    out.insert(buf.begin(), buf.end());
  });
  return out;
}

// spss.h:1039-1206 helpers + :1358-1832 with n_workers > 1
inline std::vector<std::string> spss_from_unitigs(const std::vector<std::string>& unitigs, const EndHash& prefixes,
                                                  const EndHash& suffixes, int k, int w) {
  const std::int64_t n = std::int64_t(unitigs.size());
  const int n_buckets = 512;  // spss.h:1044
  using Edge = std::pair<std::int64_t, bool>;
  const auto edges_right = [&](std::int64_t i) {
    std::vector<Edge> edges;
    const std::string& u = unitigs[i];
    const std::uint64_t suffix = kmer_from_string(u.data() + u.length() - k, k);
    for (int c = 0; c < 4; c++) {
      const std::uint64_t sn = next(suffix, k, c);
      auto it = prefixes.find(sn);
      if (it != prefixes.end())
        for (std::int64_t j : it->second)
          if (i != j) edges.emplace_back(j, false);
      auto it2 = suffixes.find(complement(sn, k));
      if (it2 != suffixes.end())
        for (std::int64_t j : it2->second)
          if (i != j) edges.emplace_back(j, true);
    }
    return edges;
  };
  const auto edges_left = [&](std::int64_t i) {
    std::vector<Edge> edges;
    const std::uint64_t prefix = kmer_from_string(unitigs[i].data(), k);
    for (int c = 0; c < 4; c++) {
      const std::uint64_t pp = prev(prefix, k, c);
      auto it = suffixes.find(pp);
      if (it != suffixes.end())
        for (std::int64_t j : it->second)
          if (i != j) edges.emplace_back(j, false);
      auto it2 = prefixes.find(complement(pp, k));
      if (it2 != prefixes.end())
        for (std::int64_t j : it2->second)
          if (i != j) edges.emplace_back(j, true);
    }
    return edges;
  };

  std::unordered_map<std::int64_t, Edge> edge_left, edge_right;
  {
    std::vector<std::mutex> mus(n_buckets);
    std::vector<std::unordered_map<std::int64_t, Edge>> buf_left(n_buckets), buf_right(n_buckets);
    const auto acquire = [&](std::int64_t i, std::int64_t j) {
      const int bi = int(i % n_buckets), bj = int(j % n_buckets);
      if (bi == bj) {
        mus[bi].lock();
        return;
      }
      mus[std::min(bi, bj)].lock();
      mus[std::max(bi, bj)].lock();
    };
    const auto release = [&](std::int64_t i, std::int64_t j) {
      const int bi = int(i % n_buckets), bj = int(j % n_buckets);
      if (bi == bj) {
        mus[bi].unlock();
        return;
      }
      mus[std::max(bi, bj)].unlock();
      mus[std::min(bi, bj)].unlock();
    };
    const auto has_left = [&](std::int64_t i) { return buf_left[i % n_buckets].count(i) != 0; };
    const auto has_right = [&](std::int64_t i) { return buf_right[i % n_buckets].count(i) != 0; };
    const auto add_left = [&](std::int64_t i, std::int64_t j, bool s) { buf_left[i % n_buckets][i] = {j, s}; };
    const auto add_right = [&](std::int64_t i, std::int64_t j, bool s) { buf_right[i % n_buckets][i] = {j, s}; };
    for_chunks(0, n, w, [&](std::int64_t i0, std::int64_t i1) {
      for (std::int64_t i = i0; i < i1; i++) {
        for (const Edge& e : edges_right(i)) {
          const std::int64_t j = e.first;
          acquire(i, j);
          if (e.second) {
            if (!has_right(i) && !has_right(j)) {
              add_right(i, j, true);
              add_right(j, i, true);
            }
          } else if (!has_right(i) && !has_left(j)) {
            add_right(i, j, false);
            add_left(j, i, false);
          }
          release(i, j);
        }
        for (const Edge& e : edges_left(i)) {
          const std::int64_t j = e.first;
          acquire(i, j);
          if (e.second) {
            if (!has_left(i) && !has_left(j)) {
              add_left(i, j, true);
              add_left(j, i, true);
            }
          } else if (!has_left(i) && !has_right(j)) {
            add_left(i, j, false);
            add_right(j, i, false);
          }
          release(i, j);
        }
      }
    });
    {
      Pool pool(w);
      pool.post([&] {
        for (int b = 0; b < n_buckets; b++) edge_left.insert(buf_left[b].begin(), buf_left[b].end());
      });
      pool.post([&] {
        for (int b = 0; b < n_buckets; b++) edge_right.insert(buf_right[b].begin(), buf_right[b].end());
      });
      pool.join();
    }
  }

  {
    DisjointSet ds{int(n)};  // ParallelDisjointSet, united from all workers (parallel_disjoint_set.h:53-78)
    for_chunks(0, n, w, [&](std::int64_t i0, std::int64_t i1) {
      for (std::int64_t i = i0; i < i1; i++) {
        auto it = edge_left.find(i);
        if (it != edge_left.end()) ds.unite(int(i), int(it->second.first));
        auto it2 = edge_right.find(i);
        if (it2 != edge_right.end()) ds.unite(int(i), int(it2->second.first));
      }
    });
    std::unordered_set<int> groups, groups_with_terminals;
    std::mutex mu_groups, mu_gwt;
    for_chunks(0, n, w, [&](std::int64_t i0, std::int64_t i1) {
      std::unordered_set<int> bg, bt;
      for (std::int64_t i = i0; i < i1; i++) {
        const int group = ds.find(int(i));
        bg.insert(group);
        if (edge_left.find(i) == edge_left.end() || edge_right.find(i) == edge_right.end()) bt.insert(group);
      }
      {
        std::lock_guard<std::mutex> lck(mu_groups);
        groups.insert(bg.begin(), bg.end());
      }
      {
        std::lock_guard<std::mutex> lck(mu_gwt);
        groups_with_terminals.insert(bt.begin(), bt.end());
      }
    });
    for (int i : groups) {
      if (groups_with_terminals.count(i)) continue;
      auto it = edge_left.find(i);
      const std::int64_t j = it->second.first;
      const bool same = it->second.second;
      edge_left.erase(i);
      if (same) edge_left.erase(j); else edge_right.erase(j);
    }
  }

  std::vector<std::int64_t> terminals_left, terminals_right, terminals_both;
  {
    std::mutex mu_l, mu_r, mu_b;
    for_chunks(0, n, w, [&](std::int64_t i0, std::int64_t i1) {
      std::vector<std::int64_t> bl, br, bb;
      for (std::int64_t i = i0; i < i1; i++) {
        const bool hl = edge_left.find(i) != edge_left.end(), hr = edge_right.find(i) != edge_right.end();
        if (!hl && !hr) bb.push_back(i); else if (!hl) bl.push_back(i); else if (!hr) br.push_back(i);
      }
      bool dl = false, dr = false, db = false;
      while (!dl || !dr || !db) {
        if (!dl && mu_l.try_lock()) {
          terminals_left.insert(terminals_left.end(), bl.begin(), bl.end());
          mu_l.unlock();
          dl = true;
        }
        if (!dr && mu_r.try_lock()) {
          terminals_right.insert(terminals_right.end(), br.begin(), br.end());
          mu_r.unlock();
          dr = true;
        }
        if (!db && mu_b.try_lock()) {
          terminals_both.insert(terminals_both.end(), bb.begin(), bb.end());
          mu_b.unlock();
          db = true;
        }
      }
    });
  }

  using Path = std::vector<std::pair<std::int64_t, bool>>;
  const auto find_path = [&](std::int64_t start, bool is_right_side) {
    Path path;
    std::int64_t current = start;
    while (true) {
      bool same;
      if (is_right_side) {
        path.emplace_back(current, false);
        auto it = edge_right.find(current);
        if (it == edge_right.end()) break;
        current = it->second.first;
        same = it->second.second;
      } else {
        path.emplace_back(current, true);
        auto it = edge_left.find(current);
        if (it == edge_left.end()) break;
        current = it->second.first;
        same = it->second.second;
      }
      if (same) is_right_side = !is_right_side;
    }
    return path;
  };
  const auto string_from_path = [&](const Path& path) {
    std::string s;
    bool first = true;
    for (const auto& p : path) {
      const std::string& u = unitigs[p.first];
      if (first) {
        s += p.second ? complement_string(u) : u;
        first = false;
      } else {
        s += p.second ? complement_string(u).substr(k - 1, u.length() - (k - 1)) : u.substr(k - 1, u.length() - (k - 1));
      }
    }
    return s;
  };

  std::vector<std::string> spss;
  for (int side = 0; side < 2; side++) {
    const std::vector<std::int64_t>& terminals = side == 0 ? terminals_left : terminals_right;
    std::mutex mu;
    for_chunks(0, std::int64_t(terminals.size()), w, [&](std::int64_t i0, std::int64_t i1) {
      std::vector<std::string> buf;
      for (std::int64_t i = i0; i < i1; i++) {
        Path path = find_path(terminals[i], side == 0);
        if (path.front().first > path.back().first) continue;
        buf.push_back(string_from_path(path));
      }
      std::lock_guard<std::mutex> lck(mu);
      spss.insert(spss.end(), buf.begin(), buf.end());
    });
  }
  {
    std::mutex mu;
    for_chunks(0, std::int64_t(terminals_both.size()), w, [&](std::int64_t i0, std::int64_t i1) {
      std::vector<std::string> buf;
      for (std::int64_t i = i0; i < i1; i++) buf.push_back(unitigs[terminals_both[i]]);
      std::lock_guard<std::mutex> lck(mu);
      spss.insert(spss.end(), buf.begin(), buf.end());
    });
  }
  return spss;
}

// FromKmerSet(set, canonical = true, fast = true, n_workers) (kmer_set_compact.h:36-47)
template <typename KeyT>
Compact from_kmer_set(const KmerSet<KeyT>& s, int w) {
  const int k = s.geom().k;
  const std::vector<std::string> unitigs = unitigs_canonical(s, w);
  const EndHash prefixes = end_map(unitigs, k, false, w), suffixes = end_map(unitigs, k, true, w);
  return Compact(s.geom(), spss_from_unitigs(unitigs, prefixes, suffixes, k, w));
}

// ---- the KmerSetSet constructor with n_workers > 1 (kmer_set_set.h:109-427), canonical sets -----
template <typename KeyT>
struct KmerSetSetMT {
  std::vector<Compact> compacts;
  AdjacencyList children;
  std::vector<IterationTrace> iterations;
  std::vector<CheckpointTrace> checkpoints;
  std::int64_t n_processed = 0, initial_total_size = 0, final_total_size = 0;

  KmerSetSetMT(std::vector<Compact> in, const std::vector<int>& bucket_ids, int max_iterations, int w)
      : compacts(std::move(in)) {
    using Sampled = std::vector<std::vector<KeyT>>;
    const int n_buckets = int(bucket_ids.size());
    std::vector<Sampled> samples(compacts.size());
    {
      Pool pool(w);  // :138-153: one task per set, each with n_workers = 1
      for (std::size_t i = 0; i < compacts.size(); i++)
        pool.post([&, i] { samples[i] = compacts[i].template sampled<KeyT>(bucket_ids, true); });
      pool.join();
    }
    const auto edge_weight = [&](int i, int j) {
      std::int64_t count = 0;
      for (int b = 0; b < n_buckets; b++) {
        const std::vector<KeyT>&bi = samples[i][b], &bj = samples[j][b];
        auto it_i = bi.begin();
        auto it_j = bj.begin();
        while (it_i != bi.end() && it_j != bj.end()) {
          if (*it_i < *it_j) ++it_i;
          else if (*it_i > *it_j) ++it_j;
          else {
            count++;
            ++it_i;
            ++it_j;
          }
        }
      }
      return count;
    };
    std::map<std::pair<int, int>, std::int64_t> weights;
    const auto weigh = [&](const std::vector<std::pair<int, int>>& pairs) {  // :191-219, :409-420
      std::mutex mu;
      for_chunks(0, std::int64_t(pairs.size()), w, [&](std::int64_t i0, std::int64_t i1) {
        for (std::int64_t i = i0; i < i1; i++) {
          const std::int64_t wt = edge_weight(pairs[i].first, pairs[i].second);
          std::lock_guard<std::mutex> lck(mu);
          weights[pairs[i]] = wt;
        }
      });
    };
    {
      std::vector<std::pair<int, int>> pairs;
      const int n = int(compacts.size());
      for (int i = 0; i < n; i++)
        for (int j = i + 1; j < n; j++) pairs.emplace_back(i, j);
      weigh(pairs);
    }
    std::atomic<std::int64_t> total_size{0};
    {
      Pool pool(w);
      for (std::size_t i = 0; i < compacts.size(); i++) pool.post([&, i] { total_size += compacts[i].size(); });
      pool.join();
    }
    initial_total_size = total_size;
    n_processed = total_size;
    const auto total_spss_weight_now = [&] {
      std::int64_t total = 0;
      for (const Compact& c : compacts) total += c.weight();
      return total;
    };
    std::int64_t total_spss_weight = total_spss_weight_now();
    const int interval = int(compacts.size() / 8 + 1);
    const float improvement_threshold = 0.1 * interval / compacts.size();
    for (int i = 0;; i++) {
      if (max_iterations >= 0 && i >= max_iterations) break;
      if (i > 0 && i % interval == 0) {
        const std::int64_t updated = total_spss_weight_now();
        const float improvement = static_cast<float>(total_spss_weight - updated) / total_spss_weight;
        const bool stop = improvement <= improvement_threshold;
        checkpoints.push_back({i, total_spss_weight, updated, improvement, stop});
        if (stop) break;
        total_spss_weight = updated;
      }
      const int n = int(compacts.size());
      std::int64_t weight = 0;
      int j = -1, kk = -1;
      for (const auto& p : weights)
        if (p.second > weight) {
          j = p.first.first;
          kk = p.first.second;
          weight = p.second;
        }
      if (weight == 0) break;
      const std::int64_t original_size = compacts[j].size() + compacts[kk].size();
      {
        const Geom& g = compacts[j].geom();
        KmerSet<KeyT> set_j = kmer_set_from_spss<KeyT>(g, to_strings(compacts[j], w), true, w);
        KmerSet<KeyT> set_k = kmer_set_from_spss<KeyT>(g, to_strings(compacts[kk], w), true, w);
        {
          const KmerSet<KeyT> set_n = intersection(set_j, set_k, w);
          sub_set(&set_j, set_n, w);
          sub_set(&set_k, set_n, w);
          compacts.push_back(from_kmer_set(set_n, w));
          samples.push_back(sampled<KeyT>(compacts[n], bucket_ids, true, w));
        }
        compacts[j] = from_kmer_set(set_j, w);
        samples[j] = sampled<KeyT>(compacts[j], bucket_ids, true, w);
        compacts[kk] = from_kmer_set(set_k, w);
        samples[kk] = sampled<KeyT>(compacts[kk], bucket_ids, true, w);
        children[j].push_back(n);
        children[kk].push_back(n);
      }
      const std::int64_t size_diff = compacts[n].size() + compacts[j].size() + compacts[kk].size() - original_size;
      total_size += size_diff;
      n_processed += original_size;
      iterations.push_back({j, kk, weight, original_size, size_diff});
      {
        std::vector<std::pair<int, int>> pairs;
        for (int l = 0; l < n; l++)
          if (j != l) pairs.emplace_back(std::min(j, l), std::max(j, l));
        for (int l = 0; l < n; l++)
          if (kk != l) pairs.emplace_back(std::min(kk, l), std::max(kk, l));
        for (int l = 0; l < n; l++) pairs.emplace_back(l, n);
        weigh(pairs);
      }
    }
    final_total_size = total_size;
  }
};

}  // namespace mt
}  // namespace ko

#endif
