// KmerSet<K, N, KeyType>: the reference's public surface (lib/core/kmer_set.h:57-305)
// over a set resident in HBM.
//
// The reference keeps 2^N absl::flat_hash_set<KeyType>; here the set is
// offsets[2^N + 1] + keys sorted inside each bucket, on the GPU, and every bulk
// operation is one or two kernel launches through the C ABI:
//   Size / Hash / Diff / Equals          ksh_set_view.n_keys / ksh_set_hash / ksh_set_diff
//   Add(other) / Sub(other)              ksh_set_union_* / ksh_pair_plan + ksh_pair_write
//   free Add / Sub / Intersection        the same (Intersection is NOT lhs.Sub(Sub(lhs, rhs)):
//                                        one merge emits it directly)
// Single-k-mer Add / Remove go through a pending list (flushed by the next bulk call);
// Contains(kmer) looks at the pending edits, then at the host copy of the resident part (the one Find
// keeps) or asks the device (ksh_set_contains; the batched overload takes many k-mers in one
// launch); Find needs the k-mers on the host for its host predicate: they are expanded on the
// device (ksh_set_kmers) and downloaded once.  n_workers stays in the signatures and is ignored.
// Const methods may be called from several host threads at once, as the reference's are from its pool
// threads (lib/core/kmer_set.h:51-52, spss.h:80-90): what they fill in lazily -- the flush of pending
// edits, the host copy -- is filled in under a lock, each thread talks to the device through its own
// context (ksc::Ctx()).  Non-const methods need the set to themselves, as in the reference.
// KeyType keeps its meaning for the host-side accessors; on the device keys are 2 bytes when
// 2K - N <= 16 (the reference's (15, 14, uint16_t)), 4 when <= 32, else 8.
#ifndef KSC_CORE_KMER_SET_H_
#define KSC_CORE_KMER_SET_H_

#include <algorithm>
#include <atomic>
#include <cstdint>
#include <mutex>
#include <tuple>
#include <unordered_set>
#include <utility>
#include <vector>

#include "core/device.h"
#include "core/kmer.h"

// The first N bits select the bucket, the last 2K - N bits are the key
// (lib/core/kmer_set.h:22-43).
template <int K, int N, typename KeyType>
std::pair<int, KeyType> GetBucketAndKeyFromKmer(const Kmer<K>& kmer) {
  const int n_key_bits = K * 2 - N;
  static_assert(n_key_bits <= static_cast<int>(sizeof(KeyType) * 8));
  const std::uint64_t bits = kmer.Bits();
  return std::make_pair(static_cast<int>(bits >> n_key_bits),
                        static_cast<KeyType>(bits % (std::uint64_t(1) << n_key_bits)));
}

template <int K, int N, typename KeyType>
Kmer<K> GetKmerFromBucketAndKey(int bucket_id, KeyType key) {
  const int n_key_bits = K * 2 - N;
  return Kmer<K>((static_cast<std::uint64_t>(bucket_id) << n_key_bits) +
                 static_cast<std::uint64_t>(key));
}

template <int K, int N, typename KeyType>
class KmerSetCompact;

template <int K, int N, typename KeyType>
class KmerSet {
  static_assert(2 * K - N <= static_cast<int>(sizeof(KeyType) * 8));

 public:
  static constexpr int kBucketsNum = 1 << N;
  static constexpr int kKeyBits = 2 * K - N;
  static constexpr int kDeviceKeyBytes = kKeyBits <= 16 ? 2 : (kKeyBits <= 32 ? 4 : 8);
  static ksh_geom Geom() { return ksh_geom{K, N, kDeviceKeyBytes, 0}; }

  KmerSet() = default;

  // From an ascending list of distinct k-mer bit patterns.
  static KmerSet FromSortedBits(const std::vector<std::uint64_t>& bits) {
    KmerSet s;
    s.Upload(bits);
    return s;
  }

  // Takes ownership of device buffers in the C-ABI layout.
  static KmerSet FromDevice(ksc::DeviceBuffer offsets, ksc::DeviceBuffer keys, std::int64_t n) {
    KmerSet s;
    s.offsets_ = std::move(offsets);
    s.keys_ = std::move(keys);
    s.n_ = n;
    s.resident_ = true;
    return s;
  }

  std::int64_t Size() const {
    Flush();
    return n_;
  }

  void Clear() { *this = KmerSet(); }

  // (the host copy mirrors the RESIDENT part and stays valid while edits are only pending: the reference's
  // `while (!visited.Contains(cur)) visited.Add(cur)` (spss.h:206-216) then never touches the device)
  void Add(const Kmer<K>& kmer) {
    if (!pending_remove_.empty()) Flush();  // keeps Remove-then-Add order
    pending_add_.push_back(kmer.Bits());
    pending_add_index_.insert(kmer.Bits());
  }

  void Remove(const Kmer<K>& kmer) {
    Flush();  // keeps Add-then-Remove order
    pending_remove_.push_back(kmer.Bits());
  }

  // One k-mer: the pending edits first (removes are always younger than adds: Remove flushes), then a binary
  // search in the host copy of the resident part -- fetched here, under the lock, for sets of up to 2^24 k-mers
  // (128 MB: a loop of single queries then costs one download instead of an allocation, a launch and two copies
  // per k-mer) -- and a larger set without a host copy answers on the device.  Nothing else is modified, so
  // concurrent readers are safe (the reference calls this from its pool threads, spss.h:80-90).
  bool Contains(const Kmer<K>& kmer) const {
    const std::uint64_t b = kmer.Bits();
    if (!pending_remove_.empty() && std::find(pending_remove_.begin(), pending_remove_.end(), b) != pending_remove_.end())
      return false;
    if (pending_add_index_.count(b)) return true;
    if (!resident_.load(std::memory_order_acquire)) return false;
    if (!host_valid_.load(std::memory_order_acquire) && n_ > (std::int64_t(1) << 24)) return ResidentContains({b})[0];
    const std::vector<std::uint64_t>& bits = ResidentHostBits();
    return std::binary_search(bits.begin(), bits.end(), b);
  }

  // Batched membership: one launch for all queries, nothing of the set leaves the device.
  std::vector<bool> Contains(const std::vector<Kmer<K>>& kmers) const {
    std::vector<std::uint64_t> bits;
    bits.reserve(kmers.size());
    for (const Kmer<K>& kmer : kmers) bits.push_back(kmer.Bits());
    Flush();
    return ResidentContains(bits);
  }

  void Reserve(std::int64_t) {}

  // K-mers matching the predicate, ascending.
  template <typename PredType>
  std::vector<Kmer<K>> Find(PredType pred, int /*n_workers*/, std::int64_t estimated_size = 0) const {
    std::vector<Kmer<K>> kmers;
    if (estimated_size > 0) kmers.reserve(static_cast<std::size_t>(estimated_size));
    for (std::uint64_t b : HostBits()) {
      const Kmer<K> kmer(b);
      if (pred(kmer)) kmers.push_back(kmer);
    }
    return kmers;
  }

  std::vector<Kmer<K>> Find(int n_workers) const {
    return Find([](const Kmer<K>&) { return true; }, n_workers, Size());
  }

  KmerSet& Add(const KmerSet& other, int /*n_workers*/) {
    Flush();
    other.Flush();
    const ksh_geom g = Geom();
    const ksh_set_view va = View(), vb = other.View();
    ksc::DeviceBuffer off(std::size_t(kBucketsNum + 1) * 8);
    std::int64_t total = 0;
    ksc::Check(ksh_set_union_plan(ksc::Ctx(), &g, &va, &vb, static_cast<std::int64_t*>(off.get()), &total));
    ksc::DeviceBuffer keys(std::size_t(total) * kDeviceKeyBytes);
    ksc::Check(ksh_set_union_write(ksc::Ctx(), &g, &va, &vb, keys.get()));
    ksc::Check(ksh_ctx_sync(ksc::Ctx()));
    Adopt(std::move(off), std::move(keys), total);
    return *this;
  }

  KmerSet& Sub(const KmerSet& other, int /*n_workers*/) {
    KmerSet amb;
    Algebra(*this, other, nullptr, &amb, nullptr);
    *this = std::move(amb);
    return *this;
  }

  std::int64_t Diff(const KmerSet& other, int /*n_workers*/) const {
    Flush();
    other.Flush();
    const ksh_geom g = Geom();
    const ksh_set_view va = View(), vb = other.View();
    std::int64_t d = 0;
    ksc::Check(ksh_set_diff(ksc::Ctx(), &g, &va, &vb, &d));
    return d;
  }

  bool Equals(const KmerSet& other, int n_workers) const { return Diff(other, n_workers) == 0; }

  std::size_t Hash(int /*n_workers*/) const {
    Flush();
    const ksh_geom g = Geom();
    const ksh_set_view v = View();
    std::uint64_t h = 0;
    ksc::Check(ksh_set_hash(ksc::Ctx(), &g, &v, &h));
    return h;
  }

  // A&B, A\B, B\A of one pair in one merge (what kmer_set_set.h:339-343 asks for).
  static void Algebra(const KmerSet& a, const KmerSet& b, KmerSet* inter, KmerSet* a_minus_b,
                      KmerSet* b_minus_a) {
    a.Flush();
    b.Flush();
    const ksh_geom g = Geom();
    const ksh_set_view va = a.View(), vb = b.View();
    ksc::DeviceBuffer oi(std::size_t(kBucketsNum + 1) * 8), oa(std::size_t(kBucketsNum + 1) * 8),
        ob(std::size_t(kBucketsNum + 1) * 8);
    std::int64_t totals[3];
    ksc::Check(ksh_pair_plan(ksc::Ctx(), &g, &va, &vb, static_cast<std::int64_t*>(oi.get()),
                             static_cast<std::int64_t*>(oa.get()), static_cast<std::int64_t*>(ob.get()),
                             totals));
    ksc::DeviceBuffer ki, ka, kb;
    if (inter) ki = ksc::DeviceBuffer(std::size_t(totals[0]) * kDeviceKeyBytes);
    if (a_minus_b) ka = ksc::DeviceBuffer(std::size_t(totals[1]) * kDeviceKeyBytes);
    if (b_minus_a) kb = ksc::DeviceBuffer(std::size_t(totals[2]) * kDeviceKeyBytes);
    ksc::Check(ksh_pair_write(ksc::Ctx(), &g, &va, &vb, inter ? ki.get() : nullptr,
                              a_minus_b ? ka.get() : nullptr, b_minus_a ? kb.get() : nullptr));
    ksc::Check(ksh_ctx_sync(ksc::Ctx()));
    if (inter) *inter = FromDevice(std::move(oi), std::move(ki), totals[0]);
    if (a_minus_b) *a_minus_b = FromDevice(std::move(oa), std::move(ka), totals[1]);
    if (b_minus_a) *b_minus_a = FromDevice(std::move(ob), std::move(kb), totals[2]);
  }

  // Device view (flushes pending single-k-mer edits first).
  ksh_set_view View() const {
    Flush();
    return ksh_set_view{static_cast<const std::int64_t*>(offsets_.get()), keys_.get(), n_};
  }

  // Ascending bit patterns of all k-mers: expanded on the device, downloaded once, cached.
  const std::vector<std::uint64_t>& HostBits() const {
    Flush();
    return ResidentHostBits();
  }

 private:
  // A mutex / flag that a copy or a move of the set starts afresh (the set is a value type, lib/core/kmer_set.h:57).
  struct Lock {
    std::recursive_mutex m;
    Lock() = default;
    Lock(const Lock&) {}
    Lock& operator=(const Lock&) { return *this; }
  };
  struct Flag {
    std::atomic<bool> v{false};
    Flag() = default;
    Flag(const Flag& o) : v(o.v.load()) {}
    Flag& operator=(const Flag& o) {
      v.store(o.v.load());
      return *this;
    }
    Flag& operator=(bool b) {
      v.store(b, std::memory_order_release);
      return *this;
    }
    bool load(std::memory_order mo = std::memory_order_acquire) const { return v.load(mo); }
    operator bool() const { return load(); }
  };

  // The host copy of the resident part (pending edits are not in it), filled in once under the lock.
  const std::vector<std::uint64_t>& ResidentHostBits() const {
    if (!host_valid_.load()) {
      std::lock_guard<std::recursive_mutex> lock(mu_.m);
      if (!host_valid_.load()) {
        ksc::DeviceBuffer d_bits(static_cast<std::size_t>(n_) * 8);
        const ksh_geom g = Geom();
        const ksh_set_view v{static_cast<const std::int64_t*>(offsets_.get()), keys_.get(), n_};
        ksc::Check(ksh_set_kmers(ksc::Ctx(), &g, &v, static_cast<std::uint64_t*>(d_bits.get())));
        ksc::Check(ksh_ctx_sync(ksc::Ctx()));
        host_ = d_bits.ToHost<std::uint64_t>(static_cast<std::size_t>(n_));
        host_valid_ = true;
      }
    }
    return host_;
  }

  // Membership in the resident part, on the device (the calling thread's context; the set is only read).
  std::vector<bool> ResidentContains(const std::vector<std::uint64_t>& bits) const {
    const ksc::DeviceBuffer d_q = ksc::DeviceBuffer::FromHost(bits);
    ksc::DeviceBuffer d_f(bits.size());
    const ksh_geom g = Geom();
    const ksh_set_view v{static_cast<const std::int64_t*>(offsets_.get()), keys_.get(), n_};
    ksc::Check(ksh_set_contains(ksc::Ctx(), &g, &v, static_cast<const std::uint64_t*>(d_q.get()),
                                static_cast<std::int64_t>(bits.size()), static_cast<std::uint8_t*>(d_f.get())));
    ksc::Check(ksh_ctx_sync(ksc::Ctx()));
    const std::vector<std::uint8_t> f = d_f.ToHost<std::uint8_t>(bits.size());
    return std::vector<bool>(f.begin(), f.end());
  }

  void Adopt(ksc::DeviceBuffer off, ksc::DeviceBuffer keys, std::int64_t n) const {
    offsets_ = std::move(off);
    keys_ = std::move(keys);
    n_ = n;
    host_valid_ = false;
    resident_ = true;
  }

  void Upload(std::vector<std::uint64_t> bits) const {
    std::vector<std::int64_t> off(kBucketsNum + 1, 0);
    for (std::uint64_t b : bits) off[(b >> kKeyBits) + 1]++;
    for (int b = 0; b < kBucketsNum; b++) off[b + 1] += off[b];
    ksc::DeviceBuffer keys;
    if (kDeviceKeyBytes == 2) {
      std::vector<std::uint16_t> k16(bits.size());
      for (std::size_t i = 0; i < bits.size(); i++)
        k16[i] = static_cast<std::uint16_t>(bits[i] & ((std::uint64_t(1) << kKeyBits) - 1));
      keys = ksc::DeviceBuffer::FromHost(k16);
    } else if (kDeviceKeyBytes == 4) {
      std::vector<std::uint32_t> k32(bits.size());
      for (std::size_t i = 0; i < bits.size(); i++)
        k32[i] = static_cast<std::uint32_t>(bits[i] & ((std::uint64_t(1) << kKeyBits) - 1));
      keys = ksc::DeviceBuffer::FromHost(k32);
    } else {
      std::vector<std::uint64_t> k64(bits.size());
      for (std::size_t i = 0; i < bits.size(); i++) k64[i] = bits[i] & ((std::uint64_t(1) << kKeyBits) - 1);
      keys = ksc::DeviceBuffer::FromHost(k64);
    }
    Adopt(ksc::DeviceBuffer::FromHost(off), std::move(keys), static_cast<std::int64_t>(bits.size()));
    host_ = std::move(bits);
    host_valid_ = true;
  }

  // Applies pending single-k-mer edits: set = (set | adds) \ removes.  Under the lock: two const calls that both
  // find edits pending apply them once.
  void Flush() const {
    if (resident_.load() && pending_add_.empty() && pending_remove_.empty()) return;
    std::lock_guard<std::recursive_mutex> lock(mu_.m);
    if (!resident_.load()) Upload({});
    if (pending_add_.empty() && pending_remove_.empty()) return;
    std::vector<std::uint64_t> adds, removes;
    adds.swap(pending_add_);
    removes.swap(pending_remove_);
    pending_add_index_.clear();
    if (!adds.empty()) {
      std::sort(adds.begin(), adds.end());
      adds.erase(std::unique(adds.begin(), adds.end()), adds.end());
      KmerSet tmp = FromSortedBits(adds);
      const_cast<KmerSet*>(this)->Add(tmp, 1);
    }
    if (!removes.empty()) {
      std::sort(removes.begin(), removes.end());
      removes.erase(std::unique(removes.begin(), removes.end()), removes.end());
      KmerSet tmp = FromSortedBits(removes);
      const_cast<KmerSet*>(this)->Sub(tmp, 1);
    }
  }

  mutable ksc::DeviceBuffer offsets_, keys_;
  mutable std::int64_t n_ = 0;
  mutable Flag resident_;
  mutable std::vector<std::uint64_t> pending_add_, pending_remove_;
  mutable std::unordered_set<std::uint64_t> pending_add_index_;
  mutable std::vector<std::uint64_t> host_;
  mutable Flag host_valid_;  // host_ mirrors the resident part (not the pending edits)
  mutable Lock mu_;

  friend class KmerSetCompact<K, N, KeyType>;
};

// Union / difference / intersection by value, as the reference has them
// (lib/core/kmer_set.h:286-305).
template <int K, int N, typename KeyType>
KmerSet<K, N, KeyType> Add(KmerSet<K, N, KeyType> lhs, const KmerSet<K, N, KeyType>& rhs, int n_workers) {
  return lhs.Add(rhs, n_workers);
}

template <int K, int N, typename KeyType>
KmerSet<K, N, KeyType> Sub(KmerSet<K, N, KeyType> lhs, const KmerSet<K, N, KeyType>& rhs, int n_workers) {
  return lhs.Sub(rhs, n_workers);
}

template <int K, int N, typename KeyType>
KmerSet<K, N, KeyType> Intersection(KmerSet<K, N, KeyType> lhs, const KmerSet<K, N, KeyType>& rhs,
                                    int /*n_workers*/) {
  KmerSet<K, N, KeyType> inter;
  KmerSet<K, N, KeyType>::Algebra(lhs, rhs, &inter, nullptr, nullptr);
  return inter;
}

#endif
