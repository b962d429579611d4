// KmerSetSet<K, N, KeyType> and KmerSetSetReader: the reference's set-of-sets compressor
// (lib/core/kmer_set_set.h:102-775) over the device loop ksh_kss_build.
//   ctor(vector<KmerSetCompact>, canonical, n_workers)  -> ksh_kss_build (the hot path)
//   Size / Get / Dump / DumpGraph / Load                 -> same meaning and file format:
//       meta.<ext> line 0 = "<#keys> {<key> <#children> <child>...}", line 1 = node count;
//       <i>.<ext> = one SPSS string per line; DOT graph "digraph G {" / "v%d -> v%d" / "}".
// bucket_ids come from ksc::SampleBucketIds (seeded) or from the caller.
#ifndef KSC_CORE_KMER_SET_SET_H_
#define KSC_CORE_KMER_SET_SET_H_

#include <algorithm>
#include <atomic>
#include <cstdint>
#include <filesystem>
#include <map>
#include <queue>
#include <sstream>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "core/device.h"
#include "core/io.h"
#include "core/kmer_set.h"
#include "core/kmer_set_compact.h"
#include "core/random.h"
#include "core/status.h"

namespace internal {

// The reference's map is an absl::flat_hash_map (iteration order unspecified); ordered here.
using AdjacencyList = std::map<int, std::vector<int>>;

inline std::string SerializeAdjacencyList(const AdjacencyList& adjacency_list) {
  std::stringstream ss;
  ss << adjacency_list.size();
  for (const auto& p : adjacency_list) {
    ss << ' ' << p.first << ' ' << p.second.size();
    for (int i : p.second) ss << ' ' << i;
  }
  return ss.str();
}

inline AdjacencyList DeserializeAdjacencyList(const std::string& s) {
  std::stringstream ss(s);
  AdjacencyList adjacency_list;
  std::size_t size;
  ss >> size;
  for (std::size_t i = 0; i < size; i++) {
    int key;
    std::size_t value_size;
    ss >> key >> value_size;
    std::vector<int> value(value_size);
    for (std::size_t j = 0; j < value_size; j++) ss >> value[j];
    adjacency_list[key] = std::move(value);
  }
  return adjacency_list;
}

}  // namespace internal

template <int K, int N, typename KeyType>
class KmerSetSet {
 public:
  using Compact = KmerSetCompact<K, N, KeyType>;
  using Set = KmerSet<K, N, KeyType>;

  struct Iteration {
    std::int64_t j, k, weight, original_size, size_diff;
  };

  KmerSetSet() = default;

  KmerSetSet(std::vector<Compact> kmer_sets_compact, bool canonical, int n_workers)
      : KmerSetSet(std::move(kmer_sets_compact), canonical, n_workers,
                   ksc::SampleBucketIds(N, ksc::BucketSeedFromEnv())) {}

  // One process per GPU: every rank constructs the same KmerSetSet with its own rank, and the
  // SPSS encodes are dealt out (ksh_kss_build_sharded).  `gather` is the all-gather the build
  // needs a few times (ksh_allgather_i64: RCCL ncclAllGather on a staging buffer, MPI, ...).
  // Nodes held by another rank stay empty here (Holder(i) says which); Dump writes the files
  // of the nodes this rank holds, so the ranks together write the directory.
  struct Shard {
    int rank = 0, world = 1;
    ksh_allgather_i64 gather = nullptr;
    void* user = nullptr;
    // The owner-sharded build (ksh_kss_build_owned) when `comm` is set: a node's set and SPSS live on
    // one rank; `owners[i]` holds input i, whose container only that rank needs to pass (the others'
    // entries are ignored); comm comes from ksh_comm_create_rccl (or _custom) with this rank / world.
    ksh_comm* comm = nullptr;
    std::vector<std::int32_t> owners;
  };

  KmerSetSet(std::vector<Compact> kmer_sets_compact, bool canonical, int /*n_workers*/,
             const std::vector<int>& bucket_ids, int max_iterations = -1, Shard shard = Shard()) {
    const ksh_geom g = Set::Geom();
    std::vector<ksh_spss_view> views;
    for (const Compact& c : kmer_sets_compact) views.push_back(c.View());
    std::vector<std::int32_t> ids(bucket_ids.begin(), bucket_ids.end());
    ksh_kss* kss = nullptr;
    if (shard.comm) {
      std::vector<std::int32_t> owners = shard.owners;
      if (owners.empty())  // contiguous blocks
        for (std::size_t i = 0; i < views.size(); i++)
          owners.push_back(static_cast<std::int32_t>(i * std::size_t(shard.world) / views.size()));
      ksc::Check(ksh_kss_build_owned(ksc::Ctx(), shard.comm, &g, views.data(), static_cast<std::int32_t>(views.size()),
                                     owners.data(), ids.data(), static_cast<std::int32_t>(ids.size()),
                                     canonical ? 1 : 0, max_iterations, &kss));
    } else if (shard.world > 1)
      ksc::Check(ksh_kss_build_sharded(ksc::Ctx(), &g, views.data(), static_cast<std::int32_t>(views.size()),
                                       ids.data(), static_cast<std::int32_t>(ids.size()), canonical ? 1 : 0,
                                       max_iterations, shard.rank, shard.world, shard.gather, shard.user, &kss));
    else
      ksc::Check(ksh_kss_build(ksc::Ctx(), &g, views.data(), static_cast<std::int32_t>(views.size()),
                               ids.data(), static_cast<std::int32_t>(ids.size()), canonical ? 1 : 0,
                               max_iterations, &kss));
    std::int32_t n_nodes = 0;
    ksc::Check(ksh_kss_size(kss, &n_nodes));
    for (std::int32_t i = 0; i < n_nodes; i++) {
      ksh_spss_view v{nullptr, nullptr, 0, 0};
      std::int32_t holder = -1;
      ksc::Check(ksh_kss_node_holder(kss, i, &holder));
      holders_.push_back(holder);
      my_rank_ = shard.rank;
      if (holder < 0 || holder == shard.rank) ksc::Check(ksh_kss_node(kss, i, &v, nullptr, nullptr));
      ksc::DeviceBuffer words(std::size_t((v.n_bases + 31) / 32) * 8), lens(std::size_t(v.n_strings) * 4);
      if (v.n_bases) ksc::Check(ksh_ctx_memcpy_d2d(ksc::Ctx(), words.get(), v.d_words, std::size_t((v.n_bases + 31) / 32) * 8));
      if (v.n_strings) ksc::Check(ksh_ctx_memcpy_d2d(ksc::Ctx(), lens.get(), v.d_lens, std::size_t(v.n_strings) * 4));
      kmer_sets_compact_.push_back(Compact::FromDevice(std::move(words), std::move(lens), v.n_strings, v.n_bases));
      const std::int32_t* ch = nullptr;
      std::int32_t n_ch = 0;
      ksc::Check(ksh_kss_children(kss, i, &ch, &n_ch));
      if (n_ch) children_[i] = std::vector<int>(ch, ch + n_ch);
    }
    std::int64_t n_it = 0;
    const std::int64_t* rows = nullptr;
    ksc::Check(ksh_kss_trace(kss, &n_it, &rows, nullptr, nullptr, nullptr));
    for (std::int64_t i = 0; i < n_it; i++)
      iterations_.push_back({rows[5 * i], rows[5 * i + 1], rows[5 * i + 2], rows[5 * i + 3], rows[5 * i + 4]});
    ksc::Check(ksh_kss_stats(kss, stats_));
    ksc::Check(ksh_kss_destroy(kss));
  }

  int Size() const { return static_cast<int>(kmer_sets_compact_.size()); }

  // Rank holding node i's SPSS after a sharded construction; -1: every rank (and always after
  // a single-process construction or Load).
  int Holder(int i) const { return i < static_cast<int>(holders_.size()) ? holders_[i] : -1; }

  // Reconstructs the ith k-mer set: union over the nodes reachable from i.
  Set Get(int i, bool canonical, int n_workers) const {
    Set kmer_set;
    std::queue<int> queue;
    queue.push(i);
    while (!queue.empty()) {
      const int current = queue.front();
      queue.pop();
      kmer_set.Add(kmer_sets_compact_[current].ToKmerSet(canonical, n_workers), n_workers);
      auto it = children_.find(current);
      if (it != children_.end())
        for (int child : it->second) queue.push(child);
    }
    return kmer_set;
  }

  ksc::Status Dump(const std::string& directory_name, const std::string& compressor,
                   const std::string& extension, int n_workers) {
    try {
      std::filesystem::create_directories(directory_name);
    } catch (...) {
      return ksc::InternalError("failed to create a directory");
    }
    const std::filesystem::path dir(directory_name);
    if (my_rank_ == 0) {
      std::vector<std::string> v;
      v.push_back(internal::SerializeAdjacencyList(children_));
      v.push_back(std::to_string(kmer_sets_compact_.size()));
      ksc::Status status = ksc::WriteLines((dir / ("meta." + extension)).string(), compressor, v);
      if (!status.ok()) return status;
    }
    // One writer per worker, a node's file per task (kmer_set_set.h:497-519): each writer thread has
    // its own device context (ksc::Ctx() is per thread), spells its node's lines on the GPU and feeds
    // its own file or compressor pipe, so the (de)compressor processes of different nodes overlap.
    // After a sharded construction a node's file is written by the rank that holds it (the inputs,
    // held everywhere, by rank 0).
    std::vector<std::size_t> mine;
    for (std::size_t i = 0; i < kmer_sets_compact_.size(); i++) {
      const int holder = Holder(static_cast<int>(i));
      if (holder >= 0 ? holder == my_rank_ : my_rank_ == 0) mine.push_back(i);
    }
    std::atomic<std::size_t> next{0};
    std::atomic<int> fail_count{0};
    const auto writer = [&] {
      for (std::size_t q = next++; q < mine.size(); q = next++) {
        const std::size_t i = mine[q];
        ksc::Status status = ksc::InternalError("exception");
        try {
          status = kmer_sets_compact_[i].Dump((dir / (std::to_string(i) + "." + extension)).string(), compressor, 1);
        } catch (...) {
        }
        if (!status.ok()) fail_count += 1;
      }
    };
    const int n_threads = std::max(1, std::min<int>(n_workers, static_cast<int>(mine.size())));
    if (n_threads == 1) {
      writer();
    } else {
      std::vector<std::thread> pool;
      for (int w = 0; w < n_threads; w++) pool.emplace_back(writer);
      for (std::thread& t : pool) t.join();
    }
    if (fail_count > 0) return ksc::InternalError("failed to write " + std::to_string(fail_count) + " files");
    return ksc::OkStatus();
  }

  ksc::Status DumpGraph(const std::string& file_name) const {
    std::vector<std::string> lines;
    lines.emplace_back("digraph G {");
    for (const auto& p : children_)
      for (int i : p.second) lines.push_back("v" + std::to_string(p.first) + " -> v" + std::to_string(i));
    lines.emplace_back("}");
    return ksc::WriteLines(file_name, "", lines);
  }

  static ksc::StatusOr<KmerSetSet> Load(const std::string& directory_name, const std::string& decompressor,
                                        const std::string& extension, int /*n_workers*/) {
    const std::filesystem::path dir(directory_name);
    ksc::StatusOr<std::vector<std::string>> meta =
        ksc::ReadLines((dir / ("meta." + extension)).string(), decompressor);
    if (!meta.ok()) return meta.status();
    if (meta.value().size() < 2) return ksc::InternalError("malformed meta file");
    KmerSetSet out;
    out.children_ = internal::DeserializeAdjacencyList(meta.value()[0]);
    const int n = std::stoi(meta.value()[1]);
    int n_fail = 0;
    for (int i = 0; i < n; i++) {
      ksc::StatusOr<Compact> c = Compact::Load((dir / (std::to_string(i) + "." + extension)).string(), decompressor);
      if (!c.ok()) {
        n_fail += 1;
        continue;
      }
      out.kmer_sets_compact_.push_back(std::move(c).value());
    }
    if (n_fail > 0) return ksc::InternalError("failed to dump " + std::to_string(n_fail) + " files");
    return out;
  }

  // What the reference logs per iteration (kmer_set_set.h:324,380) and the totals.
  const std::vector<Iteration>& Iterations() const { return iterations_; }
  const std::map<int, std::vector<int>>& Children() const { return children_; }
  const Compact& Node(int i) const { return kmer_sets_compact_[i]; }
  std::int64_t ProcessedKmers() const { return stats_[3]; }        // N_proc of SURVEY.md 8(d)
  std::int64_t TotalSpssWeight() const { return stats_[4]; }

 private:
  internal::AdjacencyList children_;
  std::vector<Compact> kmer_sets_compact_;
  std::vector<int> holders_;  // sharded construction: the rank holding each node's SPSS (-1: all)
  int my_rank_ = 0;
  std::vector<Iteration> iterations_;
  std::int64_t stats_[8] = {};
};

// Reconstructs sets from a dumped directory without loading every node
// (lib/core/kmer_set_set.h:629-775).
template <int K, int N, typename KeyType>
class KmerSetSetReader {
 public:
  using Compact = KmerSetCompact<K, N, KeyType>;
  using Set = KmerSet<K, N, KeyType>;

  KmerSetSetReader() = default;

  static ksc::StatusOr<KmerSetSetReader> FromDirectory(std::string directory_name, std::string extension,
                                                       std::string decompressor, bool canonical) {
    const std::filesystem::path dir(directory_name);
    ksc::StatusOr<std::vector<std::string>> meta =
        ksc::ReadLines((dir / ("meta." + extension)).string(), decompressor);
    if (!meta.ok()) return meta.status();
    if (meta.value().size() < 2) return ksc::InternalError("malformed meta file");
    KmerSetSetReader r;
    r.directory_name_ = std::move(directory_name);
    r.extension_ = std::move(extension);
    r.decompressor_ = std::move(decompressor);
    r.canonical_ = canonical;
    r.children_ = internal::DeserializeAdjacencyList(meta.value()[0]);
    r.size_ = std::stoi(meta.value()[1]);
    return r;
  }

  int Size() const { return size_; }

  ksc::StatusOr<Set> Get(int i, int n_workers) const {
    std::vector<int> ids;
    std::queue<int> queue;
    queue.push(i);
    while (!queue.empty()) {
      const int current = queue.front();
      queue.pop();
      ids.push_back(current);
      auto it = children_.find(current);
      if (it == children_.end()) continue;
      for (int child : it->second) queue.push(child);
    }
    Set to_return;
    int n_fail = 0;
    const std::filesystem::path dir(directory_name_);
    for (int id : ids) {
      ksc::StatusOr<Compact> c = Compact::Load((dir / (std::to_string(id) + "." + extension_)).string(), decompressor_);
      if (!c.ok()) {
        n_fail += 1;
        continue;
      }
      to_return.Add(c.value().ToKmerSet(canonical_, 1), n_workers);
    }
    if (n_fail > 0) return ksc::InternalError("failed to load data from " + std::to_string(n_fail) + " files");
    return to_return;
  }

 private:
  std::string directory_name_, extension_, decompressor_;
  bool canonical_ = true;
  internal::AdjacencyList children_;
  int size_ = 0;
};

#endif
