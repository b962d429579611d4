// ORACLE (test infrastructure, not product code).
//
// CPU restatement of the reference's set-of-sets compressor (the hot loop that
// kmerset-multiple-compress runs), n_workers == 1 control flow.
//
// Follows lib/core/kmer_set_set.h:
//   :45-85    Serialize/DeserializeAdjacencyList ("meta" line 0)
//   :109-427  KmerSetSet constructor                     -> KmerSetSet::KmerSetSet
//       :123-124  bucket_ids (2 % of buckets)   -- explicit input here, see below
//       :138-153  initial sampled sets
//       :158-184  GetEdgeWeight (sorted-merge intersection count per bucket)
//       :191-219  initial weights for all i < j
//       :225-264  total_size, total_spss_weight
//       :267-273  interval = n0/8 + 1, threshold = 0.1 * interval / n0 (float)
//       :278-426  loop: convergence check, argmax, decode j,k, Intersection,
//                 Sub, Sub, three re-encodes, three re-samples, children_,
//                 size_diff, re-weighting
//   :433-454  Get(i): BFS over children_, union
//   :459-530  Dump (meta.<ext>, <i>.<ext>)   :533-547 DumpGraph   :550-615 Load
//
// Two things the reference leaves to chance are pinned, both as a legal
// execution of the reference:
//  * bucket_ids come from an unseeded generator (lib/core/random.h:17); here the
//    caller passes them.
//  * the argmax over `weights` walks an absl::flat_hash_map (:310-316, strict >);
//    here the map is ordered, so the first maximal (j, k) in lexicographic order
//    wins.
#ifndef ORACLE_KO_KMER_SET_SET_H_
#define ORACLE_KO_KMER_SET_SET_H_

#include <cstdint>
#include <map>
#include <queue>
#include <sstream>
#include <string>
#include <utility>
#include <vector>

#include "ko_compact.h"
#include "ko_kmer_set.h"

namespace ko {

using AdjacencyList = std::map<int, std::vector<int>>;

inline std::string serialize_adjacency_list(const AdjacencyList& a) {
  std::stringstream ss;
  ss << a.size();
  for (const auto& p : a) {
    ss << ' ' << p.first;
    ss << ' ' << p.second.size();
    for (int i : p.second) ss << ' ' << i;
  }
  return ss.str();
}

inline AdjacencyList deserialize_adjacency_list(const std::string& s) {
  std::stringstream ss(s);
  AdjacencyList a;
  std::size_t size;
  ss >> size;
  for (std::size_t i = 0; i < size; i++) {
    int key;
    ss >> key;
    std::size_t value_size;
    ss >> value_size;
    std::vector<int> value(value_size);
    for (std::size_t j = 0; j < value_size; j++) ss >> value[j];
    a[key] = std::move(value);
  }
  return a;
}

struct IterationTrace {
  int j, k;
  std::int64_t weight;
  std::int64_t original_size;  // |S_j| + |S_k| before the merge (kmer_set_set.h:326-327)
  std::int64_t size_diff;
};

struct CheckpointTrace {
  int iteration;
  std::int64_t previous, updated;
  float improvement;
  bool stopped;
};

template <typename KeyT>
class KmerSetSet {
 public:
  using Sampled = std::vector<std::vector<KeyT>>;

  KmerSetSet() = default;

  // max_iterations < 0: run to the reference's own stopping rule.
  KmerSetSet(std::vector<Compact> compacts, const std::vector<int>& bucket_ids, bool canon,
             int max_iterations = -1)
      : compacts_(std::move(compacts)) {
    const int n_buckets = static_cast<int>(bucket_ids.size());
    std::vector<Sampled> sampled(compacts_.size());
    for (std::size_t i = 0; i < compacts_.size(); i++)
      sampled[i] = compacts_[i].template sampled<KeyT>(bucket_ids, canon);

    const auto edge_weight = [&](int i, int j) {
      std::int64_t count = 0;
      for (int b = 0; b < n_buckets; b++) {
        const std::vector<KeyT>& bi = sampled[i][b];
        const std::vector<KeyT>& bj = sampled[j][b];
        auto it_i = bi.begin();
        auto it_j = bj.begin();
        while (it_i != bi.end() && it_j != bj.end()) {
          if (*it_i < *it_j) {
            ++it_i;
          } else if (*it_i > *it_j) {
            ++it_j;
          } else {
            count += 1;
            ++it_i;
            ++it_j;
          }
        }
      }
      return count;
    };

    std::map<std::pair<int, int>, std::int64_t> weights;
    {
      const int n = static_cast<int>(compacts_.size());
      for (int i = 0; i < n; i++)
        for (int j = i + 1; j < n; j++) weights[{i, j}] = edge_weight(i, j);
    }
    initial_weights_ = weights;

    std::int64_t total_size = 0;
    for (const Compact& c : compacts_) total_size += c.size();
    initial_total_size_ = total_size;
    n_processed_ = total_size;

    const auto total_spss_weight_now = [&] {
      std::int64_t total = 0;
      for (const Compact& c : compacts_) total += c.weight();
      return total;
    };

    std::int64_t total_spss_weight = total_spss_weight_now();
    initial_total_spss_weight_ = total_spss_weight;

    const int interval = static_cast<int>(compacts_.size() / 8 + 1);
    const float improvement_threshold = 0.1 * interval / compacts_.size();

    for (int i = 0;; i++) {
      if (max_iterations >= 0 && i >= max_iterations) break;

      if (i > 0 && i % interval == 0) {
        const std::int64_t updated = total_spss_weight_now();
        const float improvement =
            static_cast<float>(total_spss_weight - updated) / total_spss_weight;
        const bool stop = improvement <= improvement_threshold;
        checkpoints_.push_back({i, total_spss_weight, updated, improvement, stop});
        if (stop) break;
        total_spss_weight = updated;
      }

      const int n = static_cast<int>(compacts_.size());

      std::int64_t weight = 0;
      int j = -1, k = -1;
      for (const auto& p : weights) {
        if (p.second > weight) {
          j = p.first.first;
          k = p.first.second;
          weight = p.second;
        }
      }
      if (weight == 0) break;

      const std::int64_t original_size = compacts_[j].size() + compacts_[k].size();

      {
        KmerSet<KeyT> set_j = compacts_[j].template to_kmer_set<KeyT>(canon);
        KmerSet<KeyT> set_k = compacts_[k].template to_kmer_set<KeyT>(canon);
        {
          const KmerSet<KeyT> set_n = set_intersection(set_j, set_k);
          set_j.sub_set(set_n);
          set_k.sub_set(set_n);
          compacts_.push_back(Compact::from_kmer_set(set_n, canon, true));
          sampled.push_back(compacts_[n].template sampled<KeyT>(bucket_ids, canon));
        }
        compacts_[j] = Compact::from_kmer_set(set_j, canon, true);
        sampled[j] = compacts_[j].template sampled<KeyT>(bucket_ids, canon);
        compacts_[k] = Compact::from_kmer_set(set_k, canon, true);
        sampled[k] = compacts_[k].template sampled<KeyT>(bucket_ids, canon);
        children_[j].push_back(n);
        children_[k].push_back(n);
      }

      const std::int64_t size_diff =
          compacts_[n].size() + compacts_[j].size() + compacts_[k].size() - original_size;
      total_size += size_diff;
      n_processed_ += original_size;
      iterations_.push_back({j, k, weight, original_size, size_diff});

      {
        std::vector<std::pair<int, int>> pairs;
        for (int l = 0; l < n; l++) {
          if (j == l) continue;
          pairs.emplace_back(std::min(j, l), std::max(j, l));
        }
        for (int l = 0; l < n; l++) {
          if (k == l) continue;
          pairs.emplace_back(std::min(k, l), std::max(k, l));
        }
        for (int l = 0; l < n; l++) pairs.emplace_back(l, n);
        for (const auto& p : pairs) weights[p] = edge_weight(p.first, p.second);
      }
    }
    final_total_size_ = total_size;
  }

  KmerSetSet(AdjacencyList children, std::vector<Compact> compacts)
      : children_(std::move(children)), compacts_(std::move(compacts)) {}

  // Result of the multi-worker constructor (ko_mt.h) behind the same accessors.
  void adopt_trace(std::vector<IterationTrace> iterations, std::vector<CheckpointTrace> checkpoints,
                   std::int64_t n_processed, std::int64_t initial_total_size, std::int64_t final_total_size) {
    iterations_ = std::move(iterations);
    checkpoints_ = std::move(checkpoints);
    n_processed_ = n_processed;
    initial_total_size_ = initial_total_size;
    final_total_size_ = final_total_size;
  }

  int size() const { return static_cast<int>(compacts_.size()); }

  KmerSet<KeyT> get(int i, bool canon) const {
    KmerSet<KeyT> kmer_set(compacts_[i].geom());
    std::queue<int> queue;
    queue.push(i);
    while (!queue.empty()) {
      int current = queue.front();
      queue.pop();
      kmer_set.add_set(compacts_[current].template to_kmer_set<KeyT>(canon));
      auto it = children_.find(current);
      if (it != children_.end())
        for (int child : it->second) queue.push(child);
    }
    return kmer_set;
  }

  bool dump(const std::string& dir, const std::string& ext) const {
    std::vector<std::string> v;
    v.push_back(serialize_adjacency_list(children_));
    v.push_back(std::to_string(compacts_.size()));
    if (!write_lines(dir + "/meta." + ext, v)) return false;
    for (std::size_t i = 0; i < compacts_.size(); i++)
      if (!compacts_[i].dump(dir + "/" + std::to_string(i) + "." + ext)) return false;
    return true;
  }

  static bool load(const Geom& g, const std::string& dir, const std::string& ext,
                   KmerSetSet* out) {
    std::vector<std::string> lines;
    if (!read_lines(dir + "/meta." + ext, &lines) || lines.size() < 2) return false;
    AdjacencyList children = deserialize_adjacency_list(lines[0]);
    int n = 0;
    {
      std::stringstream ss(lines[1]);
      ss >> n;
    }
    std::vector<Compact> compacts(static_cast<std::size_t>(n));
    for (int i = 0; i < n; i++)
      if (!Compact::load(g, dir + "/" + std::to_string(i) + "." + ext, &compacts[i])) return false;
    *out = KmerSetSet(std::move(children), std::move(compacts));
    return true;
  }

  std::vector<std::string> graph_lines() const {
    std::vector<std::string> lines;
    lines.emplace_back("digraph G {");
    for (const auto& p : children_)
      for (int i : p.second)
        lines.push_back("v" + std::to_string(p.first) + " -> v" + std::to_string(i));
    lines.emplace_back("}");
    return lines;
  }

  const AdjacencyList& children() const { return children_; }
  const std::vector<Compact>& compacts() const { return compacts_; }
  const std::vector<IterationTrace>& iterations() const { return iterations_; }
  const std::vector<CheckpointTrace>& checkpoints() const { return checkpoints_; }
  const std::map<std::pair<int, int>, std::int64_t>& initial_weights() const {
    return initial_weights_;
  }
  std::int64_t initial_total_size() const { return initial_total_size_; }
  std::int64_t final_total_size() const { return final_total_size_; }
  std::int64_t initial_total_spss_weight() const { return initial_total_spss_weight_; }
  // N_proc of SURVEY.md 8(d): sum |S_i| + sum over iterations (|S_j| + |S_k|).
  std::int64_t n_processed() const { return n_processed_; }

 private:
  AdjacencyList children_;
  std::vector<Compact> compacts_;
  std::vector<IterationTrace> iterations_;
  std::vector<CheckpointTrace> checkpoints_;
  std::map<std::pair<int, int>, std::int64_t> initial_weights_;
  std::int64_t initial_total_size_ = 0, final_total_size_ = 0;
  std::int64_t initial_total_spss_weight_ = 0;
  std::int64_t n_processed_ = 0;
};

}  // namespace ko

#endif
