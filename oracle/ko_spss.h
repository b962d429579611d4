// ORACLE (test infrastructure, not product code).
//
// CPU restatement of the reference's SPSS algorithms (canonical fast and slow, and the
// non-canonical variant), with the reference's n_workers == 1 control flow and the oracle's
// ordering rule (KmerSet::find ascending).  Strings are ASCII over ACGT as in the reference.
//
// Follows lib/core/spss.h:
//   :27-41      internal::ConcatenateKmers            -> concatenate_kmers
//   :73-227     GetUnitigs (non-canonical)            -> unitigs_directed
//   :697-1014   GetSPSS(unitigs, prefixes)            -> spss_directed_from_unitigs
//   :1018-1036  GetSPSS(kmer_set)                     -> spss_directed
//   :1208-1356  GetSPSSCanonical, fast == false       -> spss_canonical_from_unitigs(fast = false)
//   :230-615    GetUnitigsCanonical                   -> unitigs_canonical
//       :238-273  GetNeighborsRight / GetNeighborsLeft
//       :276-313  IsTerminalLeft / IsTerminalRight
//       :318-392  terminals_left / _right / _both (sorted, both removed)
//       :396-423  FindPath
//       :461-579  emit: both, then kept walks from left, then from right
//                 (kept iff !(canon(front) < canon(back)), :511,555)
//       :585-610  non-branching loops, serial, from each unvisited k-mer
//   :619-695    GetPrefixesFromUnitigs / GetSuffixesFromUnitigs -> end maps
//   :1039-1206  GetSPSSCanonical(unitigs, ...) helpers: GetEdgesRight/Left,
//               FindPath, GetStringFromPath
//   :1358-1539  fast == true greedy edge selection (n_workers == 1 branch)
//   :1541-1647  disjoint set + loop removal
//   :1649-1729  terminals of the path cover
//   :1731-1829  stitch (left walks, right walks, isolated), kept iff
//               !(front.index > back.index)
//   :1835-1858  GetSPSSCanonical(kmer_set, ...)
//   :1861-1941  GetKmerSetFromSPSS
//
// Ordering rule where the reference leaves it to hash iteration order: the
// loop pass (:585-610) visits unvisited k-mers in ascending order, so a loop is
// spelled from its smallest canonical k-mer, walking out of its right side.
#ifndef ORACLE_KO_SPSS_H_
#define ORACLE_KO_SPSS_H_

#include <cstdint>
#include <iterator>
#include <map>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <utility>
#include <vector>

#include "ko_dsu.h"
#include "ko_kmer.h"
#include "ko_kmer_set.h"

namespace ko {

inline std::string concatenate_kmers(const std::vector<std::uint64_t>& kmers, int k) {
  std::string s;
  s.reserve(static_cast<std::size_t>(k) + kmers.size() - 1);
  s += kmer_to_string(kmers[0], k);
  for (std::size_t i = 1; i < kmers.size(); i++) s += kmer_last(kmers[i]);
  return s;
}

template <typename KeyT>
std::vector<std::string> unitigs_canonical(const KmerSet<KeyT>& kmer_set) {
  const int k = kmer_set.geom().k;
  using Neighbor = std::pair<std::uint64_t, bool>;  // (k-mer, same side)

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
    if (nb[0].second) {
      if (neighbors_left(nb[0].first, tmp) != 1) return true;
    } else {
      if (neighbors_right(nb[0].first, tmp) != 1) return true;
    }
    return false;
  };

  const auto is_terminal_right = [&](std::uint64_t kmer) {
    Neighbor nb[8], tmp[8];
    if (neighbors_right(kmer, nb) != 1) return true;
    if (nb[0].second) {
      if (neighbors_right(nb[0].first, tmp) != 1) return true;
    } else {
      if (neighbors_left(nb[0].first, tmp) != 1) return true;
    }
    return false;
  };

  // find() is ascending, so the reference's std::sort is already satisfied.
  std::vector<std::uint64_t> terminals_left = kmer_set.find(is_terminal_left);
  std::vector<std::uint64_t> terminals_right = kmer_set.find(is_terminal_right);

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
      if (is_right_side) {
        if (is_terminal_right(current)) break;
      } else {
        if (is_terminal_left(current)) break;
      }
      Neighbor nb[8];
      int cnt = is_right_side ? neighbors_right(current, nb) : neighbors_left(current, nb);
      (void)cnt;
      assert(cnt == 1);
      current = nb[0].first;
      if (nb[0].second) is_right_side = !is_right_side;
    }
    return path;
  };

  std::vector<std::string> unitigs;
  std::unordered_set<std::uint64_t> visited;
  visited.reserve(static_cast<std::size_t>(kmer_set.size()));

  for (std::uint64_t kmer : terminals_both) {
    unitigs.push_back(kmer_to_string(kmer, k));
    visited.insert(kmer);
  }

  for (std::uint64_t start : terminals_left) {
    std::vector<std::uint64_t> path = find_path(start, true);
    if (canonical(path.front(), k) < canonical(path.back(), k)) continue;
    for (std::uint64_t kmer : path) visited.insert(canonical(kmer, k));
    unitigs.push_back(concatenate_kmers(path, k));
  }

  for (std::uint64_t start : terminals_right) {
    std::vector<std::uint64_t> path = find_path(start, false);
    if (canonical(path.front(), k) < canonical(path.back(), k)) continue;
    for (std::uint64_t kmer : path) visited.insert(canonical(kmer, k));
    unitigs.push_back(concatenate_kmers(path, k));
  }

  std::vector<std::uint64_t> not_visited =
      kmer_set.find([&](std::uint64_t kmer) { return visited.find(kmer) == visited.end(); });

  for (std::uint64_t start : not_visited) {
    if (visited.find(start) != visited.end()) continue;
    bool is_right_side = true;
    std::uint64_t current = start;
    std::vector<std::uint64_t> path;
    while (visited.find(current) == visited.end()) {
      visited.insert(current);
      path.push_back(is_right_side ? current : complement(current, k));
      Neighbor nb[8];
      int cnt = is_right_side ? neighbors_right(current, nb) : neighbors_left(current, nb);
      (void)cnt;
      assert(cnt == 1);
      current = nb[0].first;
      if (nb[0].second) is_right_side = !is_right_side;
    }
    unitigs.push_back(concatenate_kmers(path, k));
  }

  return unitigs;
}

using EndMap = std::map<std::uint64_t, std::vector<std::int64_t>>;

inline EndMap prefixes_from_unitigs(const std::vector<std::string>& unitigs, int k) {
  EndMap m;
  for (std::size_t i = 0; i < unitigs.size(); i++)
    m[kmer_from_string(unitigs[i].data(), k)].push_back(static_cast<std::int64_t>(i));
  return m;
}

inline EndMap suffixes_from_unitigs(const std::vector<std::string>& unitigs, int k) {
  EndMap m;
  for (std::size_t i = 0; i < unitigs.size(); i++)
    m[kmer_from_string(unitigs[i].data() + unitigs[i].length() - k, k)].push_back(
        static_cast<std::int64_t>(i));
  return m;
}

// n_workers == 1; fast == false is the reference's one-thread path extension (:1208-1356).
inline std::vector<std::string> spss_canonical_from_unitigs(const std::vector<std::string>& unitigs,
                                                            const EndMap& prefixes,
                                                            const EndMap& suffixes, int k,
                                                            bool fast = true) {
  const std::int64_t n = static_cast<std::int64_t>(unitigs.size());
  using Edge = std::pair<std::int64_t, bool>;  // (node, same side)

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

  using Path = std::vector<std::pair<std::int64_t, bool>>;

  const auto find_path = [&](std::int64_t start, bool is_right_side) {
    Path path;
    std::int64_t current = start;
    while (true) {
      bool is_same_side;
      if (is_right_side) {
        path.emplace_back(current, false);
        auto it = edge_right.find(current);
        if (it == edge_right.end()) break;
        current = it->second.first;
        is_same_side = it->second.second;
      } else {
        path.emplace_back(current, true);
        auto it = edge_left.find(current);
        if (it == edge_left.end()) break;
        current = it->second.first;
        is_same_side = it->second.second;
      }
      if (is_same_side) is_right_side = !is_right_side;
    }
    return path;
  };

  const auto string_from_path = [&](const Path& path) {
    std::string s;
    bool is_first = true;
    for (const auto& p : path) {
      const std::string& u = unitigs[p.first];
      if (is_first) {
        s += p.second ? complement_string(u) : u;
        is_first = false;
      } else {
        s += p.second ? complement_string(u).substr(k - 1, u.length() - (k - 1))
                      : u.substr(k - 1, u.length() - (k - 1));
      }
    }
    return s;
  };

  const auto has_left = [&](std::int64_t i) { return edge_left.find(i) != edge_left.end(); };
  const auto has_right = [&](std::int64_t i) { return edge_right.find(i) != edge_right.end(); };

  if (!fast) {
    // :1208-1322: from every node without an edge yet, extend one path as far as it goes.
    for (std::int64_t i = 0; i < n; i++) {
      if (has_left(i) || has_right(i)) continue;
      std::int64_t current = i;
      bool is_right_side;
      {
        const std::vector<Edge> er = edges_right(current);
        const std::vector<Edge> el = edges_left(current);
        if (er.empty() && el.empty()) continue;
        is_right_side = !er.empty();
      }
      while (true) {
        auto& mine = is_right_side ? edge_right : edge_left;
        if (mine.find(current) != mine.end()) break;
        const std::vector<Edge> edges = is_right_side ? edges_right(current) : edges_left(current);
        if (edges.empty()) break;
        std::int64_t nx = -1;
        bool is_same_side = false, found = false;
        for (const Edge& e : edges) {
          nx = e.first;
          is_same_side = e.second;
          if (nx == i) continue;  // would close a loop
          // the side of nx this edge lands on: the same side as ours, or the opposite one
          const bool lands_right = is_same_side ? is_right_side : !is_right_side;
          if (lands_right ? has_right(nx) : has_left(nx)) continue;
          found = true;
          break;
        }
        if (!found) break;
        mine[current] = {nx, is_same_side};
        if (is_same_side) {
          mine[nx] = {current, is_same_side};
          is_right_side = !is_right_side;
        } else {
          (is_right_side ? edge_left : edge_right)[nx] = {current, is_same_side};
        }
        current = nx;
      }
    }
    // :1328-1350
    std::vector<std::string> spss;
    for (std::int64_t i = 0; i < n; i++) {
      const bool hl = has_left(i), hr = has_right(i);
      if (hl && hr) continue;
      const Path path = hr ? find_path(i, true) : find_path(i, false);
      if (path.front().first <= path.back().first) spss.push_back(string_from_path(path));
    }
    return spss;
  }

  for (std::int64_t i = 0; i < n; i++) {
    for (const Edge& e : edges_right(i)) {
      const std::int64_t j = e.first;
      if (e.second) {
        if (!has_right(i) && !has_right(j)) {
          edge_right[i] = {j, true};
          edge_right[j] = {i, true};
        }
      } else {
        if (!has_right(i) && !has_left(j)) {
          edge_right[i] = {j, false};
          edge_left[j] = {i, false};
        }
      }
    }
    for (const Edge& e : edges_left(i)) {
      const std::int64_t j = e.first;
      if (e.second) {
        if (!has_left(i) && !has_left(j)) {
          edge_left[i] = {j, true};
          edge_left[j] = {i, true};
        }
      } else {
        if (!has_left(i) && !has_right(j)) {
          edge_left[i] = {j, false};
          edge_right[j] = {i, false};
        }
      }
    }
  }

  {
    DisjointSet ds(static_cast<int>(n));
    for (std::int64_t i = 0; i < n; i++) {
      auto it = edge_left.find(i);
      if (it != edge_left.end()) ds.unite(static_cast<int>(i), static_cast<int>(it->second.first));
      auto it2 = edge_right.find(i);
      if (it2 != edge_right.end())
        ds.unite(static_cast<int>(i), static_cast<int>(it2->second.first));
    }

    std::unordered_set<int> groups, groups_with_terminals;
    for (std::int64_t i = 0; i < n; i++) {
      int group = ds.find(static_cast<int>(i));
      groups.insert(group);
      if (!has_left(i) || !has_right(i)) groups_with_terminals.insert(group);
    }

    for (int i : groups) {
      if (groups_with_terminals.find(i) != groups_with_terminals.end()) continue;
      auto it = edge_left.find(i);
      assert(it != edge_left.end());
      const std::int64_t j = it->second.first;
      const bool is_same_side = it->second.second;
      edge_left.erase(i);
      if (is_same_side) {
        edge_left.erase(j);
      } else {
        edge_right.erase(j);
      }
    }
  }

  std::vector<std::int64_t> terminals_left, terminals_right, terminals_both;
  for (std::int64_t i = 0; i < n; i++) {
    const bool hl = has_left(i), hr = has_right(i);
    if (!hl && !hr) {
      terminals_both.push_back(i);
    } else if (!hl) {
      terminals_left.push_back(i);
    } else if (!hr) {
      terminals_right.push_back(i);
    }
  }

  std::vector<std::string> spss;
  for (std::int64_t t : terminals_left) {
    Path path = find_path(t, true);
    if (path.front().first > path.back().first) continue;
    spss.push_back(string_from_path(path));
  }
  for (std::int64_t t : terminals_right) {
    Path path = find_path(t, false);
    if (path.front().first > path.back().first) continue;
    spss.push_back(string_from_path(path));
  }
  for (std::int64_t t : terminals_both) spss.push_back(unitigs[t]);

  return spss;
}

template <typename KeyT>
std::vector<std::string> spss_canonical(const KmerSet<KeyT>& kmer_set, bool fast = true) {
  const int k = kmer_set.geom().k;
  const std::vector<std::string> unitigs = unitigs_canonical(kmer_set);
  const EndMap prefixes = prefixes_from_unitigs(unitigs, k);
  const EndMap suffixes = suffixes_from_unitigs(unitigs, k);
  return spss_canonical_from_unitigs(unitigs, prefixes, suffixes, k, fast);
}

// ---- non-canonical variant: k-mers as they are, edges only forward -------------------------
// :73-227.  Ordering rule as above: start k-mers and the loop pass ascend.
template <typename KeyT>
std::vector<std::string> unitigs_directed(const KmerSet<KeyT>& kmer_set) {
  const int k = kmer_set.geom().k;

  const auto nexts = [&](std::uint64_t kmer, std::uint64_t* out) {
    int cnt = 0;
    for (int c = 0; c < 4; c++) {
      const std::uint64_t nx = next(kmer, k, c);
      if (nx != kmer && kmer_set.contains(nx)) out[cnt++] = nx;
    }
    return cnt;
  };
  const auto prevs = [&](std::uint64_t kmer, std::uint64_t* out) {
    int cnt = 0;
    for (int c = 0; c < 4; c++) {
      const std::uint64_t pv = prev(kmer, k, c);
      if (pv != kmer && kmer_set.contains(pv)) out[cnt++] = pv;
    }
    return cnt;
  };

  // :96-116: no incoming edge, several, or the one predecessor branches.
  const auto is_start = [&](std::uint64_t kmer) {
    std::uint64_t p[4], q[4];
    const int np = prevs(kmer, p);
    if (np != 1) return true;
    return nexts(p[0], q) >= 2;
  };
  // :119-146
  const auto is_end = [&](std::uint64_t kmer) {
    std::uint64_t p[4], q[4];
    const int nn = nexts(kmer, p);
    if (nn != 1) return true;
    return prevs(p[0], q) >= 2;
  };

  const std::vector<std::uint64_t> start_kmers = kmer_set.find(is_start);

  std::vector<std::string> unitigs;
  std::unordered_set<std::uint64_t> visited;
  visited.reserve(static_cast<std::size_t>(kmer_set.size()));

  for (std::uint64_t start : start_kmers) {  // :159-199
    std::vector<std::uint64_t> path;
    std::uint64_t current = start;
    while (true) {
      visited.insert(current);
      path.push_back(current);
      if (is_end(current)) break;
      std::uint64_t q[4];
      nexts(current, q);
      current = q[0];
    }
    unitigs.push_back(concatenate_kmers(path, k));
  }

  // :203-224: loops in which every k-mer has one incoming and one outgoing edge.
  const std::vector<std::uint64_t> not_visited =
      kmer_set.find([&](std::uint64_t kmer) { return visited.find(kmer) == visited.end(); });
  for (std::uint64_t kmer : not_visited) {
    if (visited.find(kmer) != visited.end()) continue;
    std::uint64_t current = kmer;
    std::vector<std::uint64_t> path;
    while (visited.find(current) == visited.end()) {
      path.push_back(current);
      visited.insert(current);
      std::uint64_t q[4];
      nexts(current, q);
      current = q[0];
    }
    unitigs.push_back(concatenate_kmers(path, k));
  }
  return unitigs;
}

// :697-1014, n_workers == 1.
inline std::vector<std::string> spss_directed_from_unitigs(const std::vector<std::string>& unitigs,
                                                           const EndMap& prefixes, int k) {
  const std::int64_t n = static_cast<std::int64_t>(unitigs.size());

  const auto edges_out = [&](std::int64_t i) {  // :706-726
    std::vector<std::int64_t> edges;
    const std::string& u = unitigs[i];
    const std::uint64_t suffix = kmer_from_string(u.data() + u.length() - k, k);
    for (int c = 0; c < 4; c++) {
      auto it = prefixes.find(next(suffix, k, c));
      if (it == prefixes.end()) continue;
      for (std::int64_t j : it->second)
        if (i != j) edges.push_back(j);
    }
    return edges;
  };

  std::unordered_map<std::int64_t, std::int64_t> edge_in, edge_out;

  for (std::int64_t i = 0; i < n; i++) {  // :797-815
    for (std::int64_t j : edges_out(i)) {
      if (edge_out.find(i) == edge_out.end() && edge_in.find(j) == edge_in.end()) {
        edge_out[i] = j;
        edge_in[j] = i;
      }
    }
  }

  {  // :853-929
    DisjointSet ds(static_cast<int>(n));
    for (std::int64_t i = 0; i < n; i++) {
      auto it = edge_out.find(i);
      if (it != edge_out.end()) ds.unite(static_cast<int>(i), static_cast<int>(it->second));
    }
    std::unordered_set<int> groups, groups_with_terminals;
    for (std::int64_t i = 0; i < n; i++) {
      const int group = ds.find(static_cast<int>(i));
      groups.insert(group);
      if (edge_out.find(i) == edge_out.end()) groups_with_terminals.insert(group);
    }
    for (int i : groups) {
      if (groups_with_terminals.find(i) != groups_with_terminals.end()) continue;
      const std::int64_t j = edge_out[i];
      edge_out.erase(i);
      edge_in.erase(j);
    }
  }

  std::vector<std::string> spss;  // :931-1011
  for (std::int64_t start = 0; start < n; start++) {
    if (edge_in.find(start) != edge_in.end()) continue;
    std::string s = unitigs[start];
    std::int64_t current = start;
    while (true) {
      auto it = edge_out.find(current);
      if (it == edge_out.end()) break;
      current = it->second;
      s += unitigs[current].substr(k - 1, unitigs[current].length() - (k - 1));
    }
    spss.push_back(std::move(s));
  }
  return spss;
}

template <typename KeyT>
std::vector<std::string> spss_directed(const KmerSet<KeyT>& kmer_set) {
  const int k = kmer_set.geom().k;
  const std::vector<std::string> unitigs = unitigs_directed(kmer_set);
  return spss_directed_from_unitigs(unitigs, prefixes_from_unitigs(unitigs, k), k);
}

template <typename KeyT>
KmerSet<KeyT> kmer_set_from_spss(const Geom& g, const std::vector<std::string>& spss,
                                 bool canon) {
  KmerSet<KeyT> kmer_set(g);
  std::int64_t size = 0;
  for (const std::string& s : spss) size += static_cast<std::int64_t>(s.length()) - g.k + 1;
  kmer_set.reserve(size);
  for (const std::string& s : spss) {
    for (int j = 0; j < static_cast<int>(s.length()) - g.k + 1; j++) {
      // The reference builds every k-mer with Kmer(s.substr(j, K)) (spss.h:1906).
      std::uint64_t kmer = kmer_from_string(s.data() + j, g.k);
      if (canon) kmer = canonical(kmer, g.k);
      kmer_set.add(kmer);
    }
  }
  return kmer_set;
}

}  // namespace ko

#endif
