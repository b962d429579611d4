// Host-side tests of the C++17 mirror of the reference API, written after the
// reference's own tests (test/kmer.cc, test/kmer_set.cc, test/spss.cc,
// test/kmer_set_compact.cc, test/kmer_set_set.cc) with seeded inputs.  Needs a GPU:
// every set operation below runs through libkmersets_hip.so.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <filesystem>
#include <functional>
#include <set>
#include <string>
#include <thread>
#include <vector>

#include "core/kmer.h"
#include "core/kmer_counter.h"
#include "core/kmer_set.h"
#include "core/kmer_set_compact.h"
#include "core/kmer_set_set.h"
#include "core/random.h"
#include "core/spss.h"

static int g_failed = 0, g_checks = 0;
#define EXPECT_TRUE(x)                                                        \
  do {                                                                        \
    g_checks++;                                                               \
    if (!(x)) {                                                               \
      g_failed++;                                                             \
      std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #x);     \
    }                                                                         \
  } while (0)
#define EXPECT_EQ(a, b) EXPECT_TRUE((a) == (b))

// ---- seeded inputs (shape of lib/random.h:37-110) ------------------------------------
static std::uint64_t g_ctr = 0;
static std::uint64_t Rand() { return ksc::Mix64(0xC0FFEE00 + g_ctr++); }

template <int K>
std::string RandomRead() {
  std::string s;
  const int n = 1 + int(Rand() % 100);
  for (int j = 0; j < n; j++) s += Kmer<K>(Rand() & (~std::uint64_t(0) >> (64 - 2 * K))).String();
  if (Rand() % 2 == 0) s += s;  // creates a loop
  return s;
}

template <int K, int N, typename KeyType>
KmerSet<K, N, KeyType> RandomKmerSet(int n, bool canonical) {
  std::set<std::uint64_t> kmers;
  while (static_cast<int>(kmers.size()) < n) {
    const std::string s = RandomRead<K>();
    for (int j = 0; j + K <= static_cast<int>(s.size()) && static_cast<int>(kmers.size()) < n; j++) {
      Kmer<K> kmer(s.substr(j, K));
      if (canonical) kmer = kmer.Canonical();
      kmers.insert(kmer.Bits());
    }
  }
  return KmerSet<K, N, KeyType>::FromSortedBits(std::vector<std::uint64_t>(kmers.begin(), kmers.end()));
}

// A correlated family: one random genome, point substitutions per member.
template <int K, int N, typename KeyType>
std::vector<KmerSet<K, N, KeyType>> Family(int n_sets, int length) {
  std::string genome;
  for (int i = 0; i < length; i++) genome += "ACGT"[Rand() % 4];
  std::vector<KmerSet<K, N, KeyType>> out;
  for (int s = 0; s < n_sets; s++) {
    std::string g = genome;
    for (int i = 0; i < length; i++)
      if (Rand() % 300 == 0) g[i] = "ACGT"[Rand() % 4];
    std::set<std::uint64_t> kmers;
    for (int j = 0; j + K <= length; j++) kmers.insert(Kmer<K>(g.substr(j, K)).Canonical().Bits());
    out.push_back(KmerSet<K, N, KeyType>::FromSortedBits(std::vector<std::uint64_t>(kmers.begin(), kmers.end())));
  }
  return out;
}

// ---- test/kmer.cc ----------------------------------------------------------------------
static void TestKmer() {
  EXPECT_EQ(Kmer<5>("AGCTG").String(), "AGCTG");
  EXPECT_EQ(Kmer<5>("AAAAT").Canonical().String(), "AAAAT");
  EXPECT_EQ(Kmer<5>("TTTTA").Canonical().String(), "TAAAA");
  EXPECT_EQ(Kmer<5>("CCCCG").Canonical().String(), "CCCCG");
  EXPECT_EQ(Kmer<5>("GGGGC").Canonical().String(), "GCCCC");
  EXPECT_EQ(Kmer<5>("AGCTA").Complement().String(), "TAGCT");
  EXPECT_EQ(Kmer<5>("AGCTG").Next('C').String(), "GCTGC");
  EXPECT_EQ(Kmer<5>("AGCTG").Prev('C').String(), "CAGCT");
  EXPECT_EQ(internal::Complement("ACGTT"), "AACGT");  // test/spss.cc:13
}

// ---- test/kmer_set.cc ---------------------------------------------------------------------
static void TestKmerSet() {
  const int K = 5, N = 3;
  using KeyType = std::uint8_t;
  {
    Kmer<K> kmer("AGCTG");
    int bucket;
    KeyType key;
    std::tie(bucket, key) = GetBucketAndKeyFromKmer<K, N, KeyType>(kmer);
    EXPECT_EQ(kmer.String(), (GetKmerFromBucketAndKey<K, N, KeyType>(bucket, key)).String());
  }
  {
    Kmer<K> kmer("AAAAA");
    KmerSet<K, N, KeyType> s;
    EXPECT_EQ(s.Size(), 0);
    EXPECT_TRUE(!s.Contains(kmer));
    s.Add(kmer);
    EXPECT_EQ(s.Size(), 1);
    EXPECT_TRUE(s.Contains(kmer));
    s.Remove(kmer);
    EXPECT_EQ(s.Size(), 0);
    EXPECT_TRUE(!s.Contains(kmer));
  }
  {
    KmerSet<K, N, KeyType> s;
    s.Add(Kmer<K>("AAAAA"));
    s.Add(Kmer<K>("CCCCC"));
    auto a = s.Find([](const Kmer<K>& k) { return k.String()[0] == 'A'; }, 1);
    EXPECT_EQ(a.size(), 1u);
    EXPECT_EQ(a[0].String(), "AAAAA");
    auto c = s.Find([](const Kmer<K>& k) { return k.String()[1] == 'C'; }, 1);
    EXPECT_EQ(c.size(), 1u);
    EXPECT_EQ(c[0].String(), "CCCCC");
  }
  {
    KmerSet<K, N, KeyType> s1, s2;
    for (const char* x : {"AAAAA", "TTTTT", "CCCCC"}) s1.Add(Kmer<K>(x));
    for (const char* x : {"AAAAA", "TTTTT", "GGGGG"}) s2.Add(Kmer<K>(x));
    EXPECT_EQ(Add(s1, s2, 1).Size(), 4);
    EXPECT_EQ(Sub(s1, s2, 1).Size(), 1);
    EXPECT_EQ(Sub(s2, s1, 1).Size(), 1);
    EXPECT_EQ(Intersection(s2, s1, 1).Size(), 2);
    EXPECT_EQ(Intersection(s1, s2, 1).Size(), 2);
  }
  {
    KmerSet<K, N, KeyType> s1, s2, s3;
    for (const char* x : {"AAAAA", "TTTTT"}) {
      s1.Add(Kmer<K>(x));
      s2.Add(Kmer<K>(x));
    }
    for (const char* x : {"AAAAA", "CCCCC", "GGGGG"}) s3.Add(Kmer<K>(x));
    EXPECT_TRUE(s1.Equals(s1, 1) && s2.Equals(s2, 1) && s3.Equals(s3, 1));
    EXPECT_TRUE(s1.Equals(s2, 1) && s2.Equals(s1, 1));
    EXPECT_TRUE(!s1.Equals(s3, 1) && !s3.Equals(s1, 1));
  }
}

// Single-k-mer edits interleaved with queries (the reference's `while (!visited.Contains(cur)) visited.Add(cur)`,
// lib/core/spss.h:206-216), Remove-then-Add order, and const Contains from several threads at once on a set
// without a host copy (spss.h:80-90 calls it from pool threads): every thread fills in the same cache.
static void TestKmerSetEditsAndThreads() {
  const int K = 9, N = 10;
  using KeyType = std::uint8_t;
  using Set = KmerSet<K, N, KeyType>;
  {
    Set visited;
    std::set<std::uint64_t> model;
    std::uint64_t cur = 12345;
    int steps = 0;
    while (!visited.Contains(Kmer<K>(cur))) {
      EXPECT_TRUE(model.count(cur) == 0);
      visited.Add(Kmer<K>(cur));
      model.insert(cur);
      cur = ksc::Mix64(cur) % 4000;  // a rho-shaped walk: comes back to a k-mer it has seen
      steps++;
    }
    EXPECT_TRUE(steps > 10 && model.count(cur) == 1);
    EXPECT_EQ(visited.Size(), static_cast<std::int64_t>(model.size()));
    for (std::uint64_t b = 0; b < 4000; b += 7) EXPECT_EQ(visited.Contains(Kmer<K>(b)), model.count(b) == 1);
    // Remove, then Add of the same k-mer: it is in; Add, then Remove: it is out
    const Kmer<K> x(*model.begin());
    visited.Remove(x);
    EXPECT_TRUE(!visited.Contains(x));
    visited.Add(x);
    EXPECT_TRUE(visited.Contains(x));
    EXPECT_EQ(visited.Size(), static_cast<std::int64_t>(model.size()));
    visited.Add(Kmer<K>(5000));
    visited.Remove(Kmer<K>(5000));
    EXPECT_TRUE(!visited.Contains(Kmer<K>(5000)));
    EXPECT_EQ(visited.Size(), static_cast<std::int64_t>(model.size()));
  }
  {
    const Set s = RandomKmerSet<K, N, KeyType>(20000, true);
    Set empty;
    EXPECT_TRUE(!empty.Contains(Kmer<K>(1)));  // not resident yet
    EXPECT_EQ(empty.Size(), 0);
    EXPECT_TRUE(!empty.Contains(Kmer<K>(1)));  // an empty resident set
    // a copy whose host cache is empty: rebuilt from the device by whichever thread gets there first
    Set inter = Intersection(s, s, 1);
    const std::vector<Kmer<K>> all = s.Find(1);
    std::vector<int> bad(8, 0);
    std::vector<std::thread> threads;
    for (int t = 0; t < 8; t++)
      threads.emplace_back([&, t]() {
        const Set& r = inter;
        for (std::size_t i = t; i < all.size(); i += 8)
          if (!r.Contains(all[i])) bad[t]++;
        for (std::uint64_t b = t; b < 3000; b += 8) {
          const Kmer<K> q(b * 77 + 1);
          const bool want = std::binary_search(all.begin(), all.end(), q, [](const Kmer<K>& l, const Kmer<K>& rr) { return l.Bits() < rr.Bits(); });
          if (r.Contains(q) != want) bad[t]++;
        }
      });
    for (std::thread& t : threads) t.join();
    for (int t = 0; t < 8; t++) EXPECT_EQ(bad[t], 0);
  }
}

// ---- test/spss.cc ----------------------------------------------------------------------------
template <int K, int N, typename KeyType>
static void CheckStrings(const std::vector<std::string>& strings, const KmerSet<K, N, KeyType>& want,
                         bool canonical = true) {
  std::set<std::uint64_t> seen;
  bool ok = true;
  for (const std::string& s : strings) {
    if (static_cast<int>(s.length()) < K) ok = false;
    for (int i = 0; i + K <= static_cast<int>(s.length()); i++) {
      const Kmer<K> kmer(s.substr(i, K));
      if (!seen.insert((canonical ? kmer.Canonical() : kmer).Bits()).second) ok = false;  // no k-mer twice
    }
  }
  EXPECT_TRUE(ok);
  auto got = KmerSet<K, N, KeyType>::FromSortedBits(std::vector<std::uint64_t>(seen.begin(), seen.end()));
  EXPECT_TRUE(want.Equals(got, 1));
}

static void TestSpss() {
  const int K = 9, N = 10;
  using KeyType = std::uint8_t;
  for (int size : {1, 300, 20000, 65536}) {
    auto s = RandomKmerSet<K, N, KeyType>(size, true);
    CheckStrings(GetUnitigsCanonical(s, 4), s);
    const auto spss = GetSPSSCanonical(s, true, 4);
    CheckStrings(spss, s);
    EXPECT_TRUE(s.Equals(GetKmerSetFromSPSS<K, N, KeyType>(spss, true, 4), 4));
    // test/spss.cc:99-124: fast == false
    const auto slow = GetSPSSCanonical(s, false, 4);
    CheckStrings(slow, s);
    EXPECT_TRUE(s.Equals(KmerSetCompact<K, N, KeyType>::FromKmerSet(s, true, false, 4).ToKmerSet(true, 4), 4));
    // test/spss.cc:15-40,71-96,155-170: the non-canonical variant
    auto f = RandomKmerSet<K, N, KeyType>(size, false);
    CheckStrings(GetUnitigs(f, 4), f, false);
    const auto fw = GetSPSS(f, 4);
    CheckStrings(fw, f, false);
    EXPECT_TRUE(f.Equals(GetKmerSetFromSPSS<K, N, KeyType>(fw, false, 4), 4));
    EXPECT_TRUE(f.Equals(KmerSetCompact<K, N, KeyType>::FromKmerSet(f, false, true, 4).ToKmerSet(false, 4), 4));
  }
  {  // outputs of the reference's own headers
    auto of = [](const std::string& seq) {
      std::set<std::uint64_t> k;
      for (int i = 0; i + 5 <= static_cast<int>(seq.size()); i++) k.insert(Kmer<5>(seq.substr(i, 5)).Canonical().Bits());
      return KmerSet<5, 3, std::uint8_t>::FromSortedBits(std::vector<std::uint64_t>(k.begin(), k.end()));
    };
    auto a = of("AACCGTTAGCAT");
    EXPECT_EQ(a.Size(), 8);
    EXPECT_EQ(a.Hash(1), 493u);
    EXPECT_TRUE((GetSPSSCanonical(a, true, 1) == std::vector<std::string>{"ATGCTAACGGTT"}));
    auto b = of("ACGTACGTACG");
    EXPECT_EQ(b.Hash(1), 477u);
    EXPECT_TRUE((GetSPSSCanonical(b, true, 1) == std::vector<std::string>{"GTACGT"}));
    EXPECT_EQ(of("AAAAAAAAA").Hash(1), 0u);
  }
}

// ---- test/kmer_counter.cc ----------------------------------------------------------------------
static void TestCounter() {
  const int K = 5, N = 3;
  using KeyType = std::uint8_t;
  using Counter = KmerCounter<K, N, KeyType, std::uint8_t>;
  {  // AddAndGet
    Counter c;
    c.Add(Kmer<K>("AAAAA"), 1);
    c.Add(Kmer<K>("CCCCC"), 2);
    c.Add(Kmer<K>("TTTTT"), 3);
    c.Add(Kmer<K>("AAAAA"), 1);
    EXPECT_EQ(int(c.Get(Kmer<K>("AAAAA"))), 2);
    EXPECT_EQ(int(c.Get(Kmer<K>("CCCCC"))), 2);
    EXPECT_EQ(int(c.Get(Kmer<K>("TTTTT"))), 3);
    EXPECT_EQ(int(c.Get(Kmer<K>("GGGGG"))), 0);
    EXPECT_EQ(c.Size(), 3);
  }
  {  // ToKmerSet
    Counter c;
    c.Add(Kmer<K>("AAAAA"), 3);
    c.Add(Kmer<K>("CCCCC"), 1);
    c.Add(Kmer<K>("GGGGG"), 2);
    c.Add(Kmer<K>("TTTTT"), 4);
    auto r = c.ToKmerSet(3, 1);
    EXPECT_EQ(r.second, 2);
    EXPECT_EQ(r.first.Size(), 2);
    EXPECT_TRUE(r.first.Contains(Kmer<K>("AAAAA")));
    EXPECT_TRUE(r.first.Contains(Kmer<K>("TTTTT")));
  }
  {  // FromReads
    const Counter c = Counter::FromReads({"AACCGTT", "AACCGTA"}, false, 1);
    EXPECT_EQ(int(c.Get(Kmer<K>("AACCG"))), 2);
    EXPECT_EQ(int(c.Get(Kmer<K>("ACCGT"))), 2);
    EXPECT_EQ(int(c.Get(Kmer<K>("CCGTT"))), 1);
    EXPECT_EQ(int(c.Get(Kmer<K>("CCGTA"))), 1);
  }
  {  // FromFASTA: a file, the same through a pipe, and the reference's two failure modes
    const std::string file = (std::filesystem::temp_directory_path() / "ksc_test_reads.fa").string();
    const std::string text = ">r1\nAACCGTTNNAACCGTA\n>r2\nACGTNACGTACGT\n";
    EXPECT_TRUE(ksc::WriteBytes(file, "", text.data(), text.size()).ok());
    auto c = Counter::FromFASTA(file, "", false, 4);
    EXPECT_TRUE(c.ok());
    EXPECT_EQ(int(c.value().Get(Kmer<K>("AACCG"))), 2);
    EXPECT_EQ(int(c.value().Get(Kmer<K>("GTTAA"))), 0);
    auto piped = Counter::FromFASTA(file, "cat", true, 4);
    EXPECT_TRUE(piped.ok());
    EXPECT_TRUE(piped.value().Size() <= c.value().Size());
    auto odd = Counter::FromFASTA(std::vector<std::string>{">r1", "ACGTA", ">r2"}, true, 1);
    EXPECT_TRUE(!odd.ok());
    EXPECT_TRUE(odd.status().message() == "FASTA files should have an even number of lines");
    auto bad = Counter::FromFASTA(std::vector<std::string>{"r1", "ACGTA"}, true, 1);
    EXPECT_TRUE(!bad.ok());
    EXPECT_TRUE(bad.status().message() == "invalid FASTA file");
  }
}

// ---- test/kmer_set_compact.cc -----------------------------------------------------------------------
static void TestCompact() {
  const int K = 9, N = 10;
  using KeyType = std::uint8_t;
  const auto s = RandomKmerSet<K, N, KeyType>(100000, true);
  const auto c = KmerSetCompact<K, N, KeyType>::FromKmerSet(s, true, true, 4);
  const std::string file = (std::filesystem::temp_directory_path() / "ksc_test_compact.txt").string();
  EXPECT_TRUE(c.Dump(file, "", 4).ok());
  {
    auto loaded = KmerSetCompact<K, N, KeyType>::Load(file, "");
    EXPECT_TRUE(loaded.ok());
    EXPECT_TRUE(s.Equals(loaded.value().ToKmerSet(true, 4), 4));
  }
  {
    auto loaded = KmerSetCompact<K, N, KeyType>::Load(file, "cat");  // through a (de)compressor pipe
    EXPECT_TRUE(loaded.ok());
    EXPECT_TRUE(s.Equals(loaded.value().ToKmerSet(true, 4), 4));
  }
  EXPECT_TRUE((!KmerSetCompact<K, N, KeyType>::Load(file + ".missing", "").ok()));
  {
    // the dumped bytes are the strings, one per line (device-spelled == host-spelled), and a
    // compressor pipe sees the same bytes
    std::string want;
    for (const std::string& line : c.ToStrings(4)) want += line + "\n";
    EXPECT_TRUE(c.ToText() == want);
    auto bytes = ksc::ReadBytes(file, "");
    EXPECT_TRUE(bytes.ok());
    EXPECT_TRUE(bytes.value() == want);
    EXPECT_TRUE(c.Dump(file + ".piped", "cat", 4).ok());
    auto piped = ksc::ReadBytes(file + ".piped", "cat");
    EXPECT_TRUE(piped.ok());
    EXPECT_TRUE(piped.value() == want);
    const auto again = (KmerSetCompact<K, N, KeyType>::FromText(want));
    EXPECT_TRUE(again.ToStrings(4) == c.ToStrings(4));
    EXPECT_EQ((KmerSetCompact<K, N, KeyType>::FromText("")).StringCount(), 0);
  }
  EXPECT_EQ(s.Size(), c.Size(4));
  EXPECT_TRUE(s.Equals(c.ToKmerSet(true, 4), 4));
  std::vector<int> ids;
  for (int i = (1 << N) - 1; i >= 0; i--) ids.push_back(i);
  const auto sampled = c.GetSampledKmerSet(ids, true, 4);
  KmerSet<K, N, KeyType> rebuilt;
  bool sorted = true;
  for (std::size_t i = 0; i < ids.size(); i++) {
    for (std::size_t j = 1; j < sampled[i].size(); j++) sorted = sorted && sampled[i][j - 1] < sampled[i][j];
    for (KeyType key : sampled[i]) rebuilt.Add(GetKmerFromBucketAndKey<K, N, KeyType>(ids[i], key));
  }
  EXPECT_TRUE(sorted);
  EXPECT_TRUE(s.Equals(rebuilt, 4));
  std::filesystem::remove(file);
}

// ---- test/kmer_set_set.cc -------------------------------------------------------------------------------
template <int K, int N, typename KeyType>
static void TestSetSet(int n_sets, int length) {
  auto sets = Family<K, N, KeyType>(n_sets, length);
  std::vector<KmerSetCompact<K, N, KeyType>> compacts;
  for (const auto& s : sets) compacts.push_back(KmerSetCompact<K, N, KeyType>::FromKmerSet(s, true, true, 4));
  KmerSetSet<K, N, KeyType> kss(compacts, true, 4);
  EXPECT_TRUE(kss.Size() >= n_sets);
  for (int i = 0; i < n_sets; i++) EXPECT_TRUE(kss.Get(i, true, 4).Equals(sets[i], 4));
  const std::string dir = (std::filesystem::temp_directory_path() / "ksc_test_kss").string();
  std::filesystem::remove_all(dir);
  EXPECT_TRUE(kss.Dump(dir, "", "txt", 4).ok());
  EXPECT_TRUE(kss.DumpGraph(dir + "/graph.dot").ok());
  auto loaded = KmerSetSet<K, N, KeyType>::Load(dir, "", "txt", 4);
  EXPECT_TRUE(loaded.ok());
  EXPECT_EQ(loaded.value().Size(), kss.Size());
  for (int i = 0; i < kss.Size(); i++)
    EXPECT_TRUE(kss.Get(i, true, 4).Equals(loaded.value().Get(i, true, 4), 4));
  auto reader = KmerSetSetReader<K, N, KeyType>::FromDirectory(dir, "txt", "", true);
  EXPECT_TRUE(reader.ok());
  EXPECT_EQ(reader.value().Size(), kss.Size());
  for (int i = 0; i < n_sets; i++) {
    auto got = reader.value().Get(i, 4);
    EXPECT_TRUE(got.ok());
    EXPECT_TRUE(got.value().Equals(sets[i], 4));
  }
  std::printf("  KmerSetSet<%d,%d>: %d -> %d nodes, %zu merges, N_proc = %lld\n", K, N, n_sets, kss.Size(),
              kss.Iterations().size(), static_cast<long long>(kss.ProcessedKmers()));
  std::filesystem::remove_all(dir);
  // the owner-sharded constructor over an RCCL communicator, as far as one process goes (one rank):
  // the same merge sequence and nodes as the plain constructor with the same bucket ids
  {
    const std::vector<int> ids = ksc::SampleBucketIds(N, ksc::BucketSeedFromEnv());
    KmerSetSet<K, N, KeyType> plain(compacts, true, 4, ids);
    unsigned char id[KSH_COMM_ID_BYTES];
    ksc::Check(ksh_comm_unique_id(id));
    ksh_comm* comm = nullptr;
    ksc::Check(ksh_comm_create_rccl(ksc::Ctx(), 0, 1, id, &comm));
    typename KmerSetSet<K, N, KeyType>::Shard shard;
    shard.comm = comm;
    KmerSetSet<K, N, KeyType> owned(compacts, true, 4, ids, -1, shard);
    EXPECT_EQ(owned.Size(), plain.Size());
    EXPECT_EQ(owned.Iterations().size(), plain.Iterations().size());
    for (std::size_t t = 0; t < plain.Iterations().size() && t < owned.Iterations().size(); t++) {
      EXPECT_EQ(owned.Iterations()[t].j, plain.Iterations()[t].j);
      EXPECT_EQ(owned.Iterations()[t].k, plain.Iterations()[t].k);
      EXPECT_EQ(owned.Iterations()[t].size_diff, plain.Iterations()[t].size_diff);
    }
    for (int i = 0; i < n_sets; i++) EXPECT_TRUE(owned.Get(i, true, 4).Equals(sets[i], 4));
    ksc::Check(ksh_comm_destroy(comm));
  }
}

int main() {
  try {
    TestKmer();
    TestKmerSet();
    TestKmerSetEditsAndThreads();
    TestSpss();
    TestCompact();
    TestCounter();
    TestSetSet<15, 14, std::uint16_t>(8, 20000);
    TestSetSet<23, 14, std::uint32_t>(6, 30000);
    TestSetSet<31, 14, std::uint64_t>(4, 20000);
  } catch (const std::exception& e) {
    std::fprintf(stderr, "exception: %s\n", e.what());
    return 2;
  }
  std::printf("%d checks, %d failed\n", g_checks, g_failed);
  return g_failed ? 1 : 0;
}
