// ORACLE (test infrastructure, not product code).
//
// Restatement of lib/core/parallel_disjoint_set.h:15-111 (Anderson-Woll
// union-find; one 64-bit word per node = rank << 32 | parent).  std::atomic is
// kept so the multi-threaded property test of the reference
// (test/parallel_disjoint_set.cc) can be replayed against it.
//   :24-40   Find   (walk to root, then CAS path compression guarded by LessThan)
//   :43-50   IsSame
//   :53-78   Unite  (lower rank, then lower index, becomes the child)
//   :88-95   UpdateRoot
//   :97-105  LessThan
#ifndef ORACLE_KO_DSU_H_
#define ORACLE_KO_DSU_H_

#include <atomic>
#include <cstdint>
#include <utility>
#include <vector>

namespace ko {

class DisjointSet {
 public:
  explicit DisjointSet(int size) : a_(static_cast<std::size_t>(size)) {
    for (int i = 0; i < size; i++) a_[i] = static_cast<std::uint64_t>(i);
  }

  int find(int x) {
    int y = x;
    while (x != get_next(x)) x = get_next(x);
    while (less_than(y, x)) {
      std::uint64_t expected = (static_cast<std::uint64_t>(get_rank(y)) << 32) +
                               static_cast<std::uint64_t>(get_next(y));
      std::uint64_t desired = ((expected >> 32) << 32) + static_cast<std::uint64_t>(x);
      a_[y].compare_exchange_weak(expected, desired);
      y = get_next(y);
    }
    return x;
  }

  bool is_same(int x, int y) {
    while (true) {
      x = find(x);
      y = find(y);
      if (x == y) return true;
      if (get_next(x) == x) return false;
    }
  }

  void unite(int x, int y) {
    while (true) {
      x = find(x);
      y = find(y);
      if (x == y) return;
      int rank_x = get_rank(x);
      int rank_y = get_rank(y);
      if (rank_x > rank_y || (rank_x == rank_y && x > y)) {
        std::swap(x, y);
        std::swap(rank_x, rank_y);
      }
      if (!update_root(x, rank_x, y, rank_x)) continue;
      if (rank_x == rank_y) update_root(y, rank_y, y, rank_y + 1);
      break;
    }
  }

 private:
  int get_rank(int i) const { return static_cast<int>(a_[i] >> 32); }
  int get_next(int i) const { return static_cast<int>((a_[i] << 32) >> 32); }

  bool update_root(int x, int old_rank, int y, int new_rank) {
    std::uint64_t old = a_[x];
    if ((old << 32) >> 32 != static_cast<std::uint64_t>(x) ||
        old >> 32 != static_cast<std::uint64_t>(old_rank))
      return false;
    std::uint64_t updated =
        (static_cast<std::uint64_t>(new_rank) << 32) + static_cast<std::uint64_t>(y);
    return a_[x].compare_exchange_strong(old, updated);
  }

  bool less_than(int x, int y) const {
    int rank_x = get_rank(x);
    int rank_y = get_rank(y);
    if (rank_x < rank_y) return true;
    if (rank_x > rank_y) return false;
    return x < y;
  }

  std::vector<std::atomic<std::uint64_t>> a_;
};

}  // namespace ko

#endif
