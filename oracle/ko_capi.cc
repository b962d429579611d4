// ORACLE (test infrastructure, not product code).
//
// extern "C" surface of the CPU restatement, for ctypes (tests/, smoke(),
// bench.py cpu_baseline).  Handles are opaque; every array is caller-owned
// unless returned through a *_copy accessor.  Nothing here is linked into the
// product library.
#include <cstdint>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "ko_compact.h"
#include "ko_dsu.h"
#include "ko_kmer.h"
#include "ko_kmer_counter.h"
#include "ko_kmer_set.h"
#include "ko_kmer_set_set.h"
#include "ko_mt.h"
#include "ko_spss.h"

namespace {

struct SetH {
  ko::Geom g;
  std::unique_ptr<ko::KmerSet<std::uint32_t>> s32;
  std::unique_ptr<ko::KmerSet<std::uint64_t>> s64;
  bool wide() const { return g.key_bytes > 4; }
};

struct StringsH {
  std::vector<std::string> v;
};

struct CompactH {
  ko::Compact c;
};

struct KssH {
  ko::Geom g;
  bool canon = true;
  std::unique_ptr<ko::KmerSetSet<std::uint32_t>> k32;
  std::unique_ptr<ko::KmerSetSet<std::uint64_t>> k64;
  bool wide() const { return g.key_bytes > 4; }
};

SetH* new_set(const ko::Geom& g) {
  SetH* h = new SetH;
  h->g = g;
  if (h->wide())
    h->s64.reset(new ko::KmerSet<std::uint64_t>(g));
  else
    h->s32.reset(new ko::KmerSet<std::uint32_t>(g));
  return h;
}

template <typename KeyT>
SetH* wrap_set(ko::KmerSet<KeyT>&& s);
template <>
SetH* wrap_set(ko::KmerSet<std::uint32_t>&& s) {
  SetH* h = new SetH;
  h->g = s.geom();
  h->s32.reset(new ko::KmerSet<std::uint32_t>(std::move(s)));
  return h;
}
template <>
SetH* wrap_set(ko::KmerSet<std::uint64_t>&& s) {
  SetH* h = new SetH;
  h->g = s.geom();
  h->s64.reset(new ko::KmerSet<std::uint64_t>(std::move(s)));
  return h;
}

#define KO_SET_CALL(h, expr) ((h)->wide() ? (h)->s64->expr : (h)->s32->expr)

}  // namespace

// The reference's n_workers > 1 branches (ko_mt.h), canonical sets: for bench.py's cpu_baseline.
template <typename KeyT>
static ko::KmerSetSet<KeyT>* build_mt(std::vector<ko::Compact> cs, const std::vector<int>& ids,
                                      int max_iterations, int n_workers) {
  ko::mt::KmerSetSetMT<KeyT> m(std::move(cs), ids, max_iterations, n_workers);
  auto* k = new ko::KmerSetSet<KeyT>(std::move(m.children), std::move(m.compacts));
  k->adopt_trace(std::move(m.iterations), std::move(m.checkpoints), m.n_processed, m.initial_total_size,
                 m.final_total_size);
  return k;
}

extern "C" {

// ---- k-mer primitives -------------------------------------------------------
std::uint64_t ko_kmer_from_string(const char* s, int k) { return ko::kmer_from_string(s, k); }
void ko_kmer_to_string(std::uint64_t bits, int k, char* out) {
  std::string s = ko::kmer_to_string(bits, k);
  std::memcpy(out, s.data(), static_cast<std::size_t>(k));
  out[k] = 0;
}
std::uint64_t ko_complement(std::uint64_t bits, int k) { return ko::complement(bits, k); }
std::uint64_t ko_canonical(std::uint64_t bits, int k) { return ko::canonical(bits, k); }
std::uint64_t ko_next(std::uint64_t bits, int k, int c) { return ko::next(bits, k, c); }
std::uint64_t ko_prev(std::uint64_t bits, int k, int c) { return ko::prev(bits, k, c); }
void ko_complement_string(char* s, std::int64_t n) {
  std::string r = ko::complement_string(std::string(s, static_cast<std::size_t>(n)));
  std::memcpy(s, r.data(), static_cast<std::size_t>(n));
}
void ko_bucket_and_key(int k, int n, std::uint64_t kmer, std::int64_t* bucket, std::uint64_t* key) {
  ko::Geom g{k, n, 8};
  ko::bucket_and_key(g, kmer, bucket, key);
}
std::uint64_t ko_kmer_from_bucket_and_key(int k, int n, std::int64_t bucket, std::uint64_t key) {
  ko::Geom g{k, n, 8};
  return ko::kmer_from_bucket_and_key(g, bucket, key);
}
// Vectorised forms for fixture checks.
void ko_canonical_many(const std::uint64_t* in, std::int64_t n, int k, std::uint64_t* out) {
  for (std::int64_t i = 0; i < n; i++) out[i] = ko::canonical(in[i], k);
}
void ko_complement_many(const std::uint64_t* in, std::int64_t n, int k, std::uint64_t* out) {
  for (std::int64_t i = 0; i < n; i++) out[i] = ko::complement(in[i], k);
}

// ---- Range::Split (lib/core/range.h:52-77) ------------------------------------
void ko_range_split(std::int64_t begin, std::int64_t end, std::int64_t n, std::int64_t* out_begin,
                    std::int64_t* out_end) {
  const std::int64_t size = end - begin;
  const std::int64_t small = size / n;
  const std::int64_t large_n = size - small * n;
  const std::int64_t small_n = n - large_n;
  std::int64_t at = begin;
  for (std::int64_t i = 0; i < n; i++) {
    const std::int64_t len = i < small_n ? small : small + 1;
    out_begin[i] = at;
    out_end[i] = at + len;
    at += len;
  }
}

// ---- disjoint set ---------------------------------------------------------------
void* ko_dsu_new(int n) { return new ko::DisjointSet(n); }
void ko_dsu_free(void* d) { delete static_cast<ko::DisjointSet*>(d); }
int ko_dsu_find(void* d, int x) { return static_cast<ko::DisjointSet*>(d)->find(x); }
void ko_dsu_unite(void* d, int x, int y) { static_cast<ko::DisjointSet*>(d)->unite(x, y); }
int ko_dsu_is_same(void* d, int x, int y) { return static_cast<ko::DisjointSet*>(d)->is_same(x, y); }
// test/parallel_disjoint_set.cc: unite pairs from n_threads threads.
void ko_dsu_unite_parallel(void* d, const int* xs, const int* ys, std::int64_t n, int n_threads) {
  ko::DisjointSet* ds = static_cast<ko::DisjointSet*>(d);
  std::vector<std::thread> threads;
  for (int t = 0; t < n_threads; t++) {
    threads.emplace_back([=] {
      for (std::int64_t i = t; i < n; i += n_threads) ds->unite(xs[i], ys[i]);
    });
  }
  for (auto& th : threads) th.join();
}

// ---- KmerSet ----------------------------------------------------------------------
void* ko_set_new(int k, int n, int key_bytes) { return new_set(ko::Geom{k, n, key_bytes}); }
void ko_set_free(void* h) { delete static_cast<SetH*>(h); }
void* ko_set_copy(void* h) {
  SetH* s = static_cast<SetH*>(h);
  if (s->wide()) return wrap_set(ko::KmerSet<std::uint64_t>(*s->s64));
  return wrap_set(ko::KmerSet<std::uint32_t>(*s->s32));
}
void ko_set_add_kmers(void* h, const std::uint64_t* kmers, std::int64_t n) {
  SetH* s = static_cast<SetH*>(h);
  for (std::int64_t i = 0; i < n; i++) KO_SET_CALL(s, add(kmers[i]));
}
void ko_set_remove_kmers(void* h, const std::uint64_t* kmers, std::int64_t n) {
  SetH* s = static_cast<SetH*>(h);
  for (std::int64_t i = 0; i < n; i++) KO_SET_CALL(s, remove(kmers[i]));
}
int ko_set_contains(void* h, std::uint64_t kmer) {
  SetH* s = static_cast<SetH*>(h);
  return KO_SET_CALL(s, contains(kmer)) ? 1 : 0;
}
std::int64_t ko_set_size(void* h) {
  SetH* s = static_cast<SetH*>(h);
  return KO_SET_CALL(s, size());
}
std::uint64_t ko_set_hash(void* h) {
  SetH* s = static_cast<SetH*>(h);
  return KO_SET_CALL(s, hash());
}
void ko_set_clear(void* h) {
  SetH* s = static_cast<SetH*>(h);
  KO_SET_CALL(s, clear());
}
// All k-mers ascending; out must hold ko_set_size() values.
void ko_set_kmers(void* h, std::uint64_t* out) {
  SetH* s = static_cast<SetH*>(h);
  std::vector<std::uint64_t> v = KO_SET_CALL(s, find_all());
  std::memcpy(out, v.data(), v.size() * sizeof(std::uint64_t));
}
void ko_set_add_set(void* a, void* b) {
  SetH* x = static_cast<SetH*>(a);
  SetH* y = static_cast<SetH*>(b);
  if (x->wide())
    x->s64->add_set(*y->s64);
  else
    x->s32->add_set(*y->s32);
}
void ko_set_sub_set(void* a, void* b) {
  SetH* x = static_cast<SetH*>(a);
  SetH* y = static_cast<SetH*>(b);
  if (x->wide())
    x->s64->sub_set(*y->s64);
  else
    x->s32->sub_set(*y->s32);
}
void* ko_set_intersection(void* a, void* b) {
  SetH* x = static_cast<SetH*>(a);
  SetH* y = static_cast<SetH*>(b);
  if (x->wide()) return wrap_set(ko::set_intersection(*x->s64, *y->s64));
  return wrap_set(ko::set_intersection(*x->s32, *y->s32));
}
std::int64_t ko_set_diff(void* a, void* b) {
  SetH* x = static_cast<SetH*>(a);
  SetH* y = static_cast<SetH*>(b);
  if (x->wide()) return x->s64->diff(*y->s64);
  return x->s32->diff(*y->s32);
}

// ---- strings -----------------------------------------------------------------------
void* ko_strings_new(const char* chars, const std::int64_t* lens, std::int64_t n) {
  StringsH* h = new StringsH;
  h->v.reserve(static_cast<std::size_t>(n));
  std::int64_t at = 0;
  for (std::int64_t i = 0; i < n; i++) {
    h->v.emplace_back(chars + at, static_cast<std::size_t>(lens[i]));
    at += lens[i];
  }
  return h;
}
void ko_strings_free(void* h) { delete static_cast<StringsH*>(h); }
std::int64_t ko_strings_count(void* h) {
  return static_cast<std::int64_t>(static_cast<StringsH*>(h)->v.size());
}
std::int64_t ko_strings_total(void* h) {
  std::int64_t t = 0;
  for (const auto& s : static_cast<StringsH*>(h)->v) t += static_cast<std::int64_t>(s.size());
  return t;
}
void ko_strings_get(void* h, char* chars, std::int64_t* lens) {
  std::int64_t at = 0, i = 0;
  for (const auto& s : static_cast<StringsH*>(h)->v) {
    std::memcpy(chars + at, s.data(), s.size());
    at += static_cast<std::int64_t>(s.size());
    lens[i++] = static_cast<std::int64_t>(s.size());
  }
}

// ---- SPSS ------------------------------------------------------------------------------
void* ko_unitigs_canonical(void* set) {
  SetH* s = static_cast<SetH*>(set);
  StringsH* h = new StringsH;
  h->v = s->wide() ? ko::unitigs_canonical(*s->s64) : ko::unitigs_canonical(*s->s32);
  return h;
}
void* ko_spss_canonical(void* set) {
  SetH* s = static_cast<SetH*>(set);
  StringsH* h = new StringsH;
  h->v = s->wide() ? ko::spss_canonical(*s->s64) : ko::spss_canonical(*s->s32);
  return h;
}
// variant: 0 = canonical fast, 1 = canonical slow (fast == false), 2 = non-canonical.
void* ko_spss_variant(void* set, int variant) {
  SetH* s = static_cast<SetH*>(set);
  StringsH* h = new StringsH;
  if (variant == 2)
    h->v = s->wide() ? ko::spss_directed(*s->s64) : ko::spss_directed(*s->s32);
  else
    h->v = s->wide() ? ko::spss_canonical(*s->s64, variant == 0) : ko::spss_canonical(*s->s32, variant == 0);
  return h;
}
void* ko_unitigs_directed(void* set) {
  SetH* s = static_cast<SetH*>(set);
  StringsH* h = new StringsH;
  h->v = s->wide() ? ko::unitigs_directed(*s->s64) : ko::unitigs_directed(*s->s32);
  return h;
}
void* ko_spss_from_unitigs(void* unitigs, int k) {
  StringsH* u = static_cast<StringsH*>(unitigs);
  StringsH* h = new StringsH;
  h->v = ko::spss_canonical_from_unitigs(u->v, ko::prefixes_from_unitigs(u->v, k),
                                         ko::suffixes_from_unitigs(u->v, k), k);
  return h;
}
void* ko_set_from_spss(void* strings, int k, int n, int key_bytes, int canon) {
  StringsH* s = static_cast<StringsH*>(strings);
  ko::Geom g{k, n, key_bytes};
  if (key_bytes > 4) return wrap_set(ko::kmer_set_from_spss<std::uint64_t>(g, s->v, canon != 0));
  return wrap_set(ko::kmer_set_from_spss<std::uint32_t>(g, s->v, canon != 0));
}

// ---- StreamVByte 0124 ----------------------------------------------------------------------
std::int64_t ko_svb_max_compressed_bytes(std::uint32_t n) {
  return static_cast<std::int64_t>(ko::svb_max_compressed_bytes(n));
}
std::int64_t ko_svb_encode_0124(const std::uint32_t* in, std::uint32_t n, std::uint8_t* out) {
  return static_cast<std::int64_t>(ko::svb_encode_0124(in, n, out));
}
std::int64_t ko_svb_decode_0124(const std::uint8_t* in, std::uint32_t* out, std::uint32_t n) {
  return static_cast<std::int64_t>(ko::svb_decode_0124(in, out, n));
}

// ---- KmerSetCompact -----------------------------------------------------------------------------
void* ko_compact_from_strings(void* strings, int k, int n, int key_bytes) {
  CompactH* h = new CompactH;
  h->c = ko::Compact(ko::Geom{k, n, key_bytes}, static_cast<StringsH*>(strings)->v);
  return h;
}
void* ko_compact_from_set(void* set) {
  SetH* s = static_cast<SetH*>(set);
  CompactH* h = new CompactH;
  h->c = s->wide() ? ko::Compact::from_kmer_set(*s->s64) : ko::Compact::from_kmer_set(*s->s32);
  return h;
}
void* ko_compact_from_set_variant(void* set, int canon, int fast) {
  SetH* s = static_cast<SetH*>(set);
  CompactH* h = new CompactH;
  h->c = s->wide() ? ko::Compact::from_kmer_set(*s->s64, canon != 0, fast != 0)
                   : ko::Compact::from_kmer_set(*s->s32, canon != 0, fast != 0);
  return h;
}
void ko_compact_free(void* h) { delete static_cast<CompactH*>(h); }
void* ko_compact_to_set(void* h, int canon) {
  const ko::Compact& c = static_cast<CompactH*>(h)->c;
  if (c.geom().key_bytes > 4) return wrap_set(c.to_kmer_set<std::uint64_t>(canon != 0));
  return wrap_set(c.to_kmer_set<std::uint32_t>(canon != 0));
}
void* ko_compact_to_strings(void* h) {
  StringsH* s = new StringsH;
  s->v = static_cast<CompactH*>(h)->c.to_strings();
  return s;
}
std::int64_t ko_compact_size(void* h) { return static_cast<CompactH*>(h)->c.size(); }
std::int64_t ko_compact_weight(void* h) { return static_cast<CompactH*>(h)->c.weight(); }
std::int64_t ko_compact_n_strings(void* h) { return static_cast<CompactH*>(h)->c.n_strings(); }
std::int64_t ko_compact_n_words(void* h) {
  return static_cast<std::int64_t>(static_cast<CompactH*>(h)->c.words().size());
}
void ko_compact_words(void* h, std::uint64_t* out) {
  const auto& w = static_cast<CompactH*>(h)->c.words();
  std::memcpy(out, w.data(), w.size() * sizeof(std::uint64_t));
}
std::int64_t ko_compact_lengths_compressed_size(void* h) {
  return static_cast<std::int64_t>(static_cast<CompactH*>(h)->c.lengths_compressed().size());
}
void ko_compact_lengths_compressed(void* h, std::uint8_t* out) {
  const auto& v = static_cast<CompactH*>(h)->c.lengths_compressed();
  std::memcpy(out, v.data(), v.size());
}
// Sampled set in the device layout: offsets[n_ids + 1], keys as u64.
// First call with keys == NULL to get offsets; then with keys to fill.
void ko_compact_sampled(void* h, const int* bucket_ids, int n_ids, int canon, std::int64_t* offsets,
                        std::uint64_t* keys) {
  const ko::Compact& c = static_cast<CompactH*>(h)->c;
  std::vector<int> ids(bucket_ids, bucket_ids + n_ids);
  std::vector<std::vector<std::uint64_t>> b = c.sampled<std::uint64_t>(ids, canon != 0);
  std::int64_t at = 0;
  for (int i = 0; i < n_ids; i++) {
    offsets[i] = at;
    if (keys) std::memcpy(keys + at, b[i].data(), b[i].size() * sizeof(std::uint64_t));
    at += static_cast<std::int64_t>(b[i].size());
  }
  offsets[n_ids] = at;
}
int ko_compact_dump(void* h, const char* file_name) {
  return static_cast<CompactH*>(h)->c.dump(file_name) ? 0 : 1;
}
void* ko_compact_load(const char* file_name, int k, int n, int key_bytes) {
  CompactH* h = new CompactH;
  if (!ko::Compact::load(ko::Geom{k, n, key_bytes}, file_name, &h->c)) {
    delete h;
    return nullptr;
  }
  return h;
}

// ---- KmerSetSet ---------------------------------------------------------------------------------
void* ko_kss_build(void** compacts, int n, const int* bucket_ids, int n_ids, int canon,
                   int max_iterations) {
  std::vector<ko::Compact> cs;
  for (int i = 0; i < n; i++) cs.push_back(static_cast<CompactH*>(compacts[i])->c);
  std::vector<int> ids(bucket_ids, bucket_ids + n_ids);
  KssH* h = new KssH;
  h->g = cs.empty() ? ko::Geom{} : cs[0].geom();
  h->canon = canon != 0;
  if (h->wide())
    h->k64.reset(new ko::KmerSetSet<std::uint64_t>(std::move(cs), ids, h->canon, max_iterations));
  else
    h->k32.reset(new ko::KmerSetSet<std::uint32_t>(std::move(cs), ids, h->canon, max_iterations));
  return h;
}
void* ko_kss_build_mt(void** compacts, int n, const int* bucket_ids, int n_ids, int max_iterations,
                      int n_workers) {
  std::vector<ko::Compact> cs;
  for (int i = 0; i < n; i++) cs.push_back(static_cast<CompactH*>(compacts[i])->c);
  std::vector<int> ids(bucket_ids, bucket_ids + n_ids);
  KssH* h = new KssH;
  h->g = cs.empty() ? ko::Geom{} : cs[0].geom();
  h->canon = true;
  if (h->wide())
    h->k64.reset(build_mt<std::uint64_t>(std::move(cs), ids, max_iterations, n_workers));
  else
    h->k32.reset(build_mt<std::uint32_t>(std::move(cs), ids, max_iterations, n_workers));
  return h;
}
void ko_kss_free(void* h) { delete static_cast<KssH*>(h); }

#define KO_KSS_CALL(h, expr) ((h)->wide() ? (h)->k64->expr : (h)->k32->expr)

int ko_kss_size(void* h) {
  KssH* k = static_cast<KssH*>(h);
  return KO_KSS_CALL(k, size());
}
void* ko_kss_get(void* h, int i) {
  KssH* k = static_cast<KssH*>(h);
  if (k->wide()) return wrap_set(k->k64->get(i, k->canon));
  return wrap_set(k->k32->get(i, k->canon));
}
void* ko_kss_node(void* h, int i) {
  KssH* k = static_cast<KssH*>(h);
  CompactH* c = new CompactH;
  c->c = KO_KSS_CALL(k, compacts())[static_cast<std::size_t>(i)];
  return c;
}
int ko_kss_n_iterations(void* h) {
  KssH* k = static_cast<KssH*>(h);
  return static_cast<int>(KO_KSS_CALL(k, iterations()).size());
}
// out: n_iterations rows of {j, k, weight, original_size, size_diff}.
void ko_kss_iterations(void* h, std::int64_t* out) {
  KssH* k = static_cast<KssH*>(h);
  const auto& it = KO_KSS_CALL(k, iterations());
  for (std::size_t i = 0; i < it.size(); i++) {
    out[5 * i + 0] = it[i].j;
    out[5 * i + 1] = it[i].k;
    out[5 * i + 2] = it[i].weight;
    out[5 * i + 3] = it[i].original_size;
    out[5 * i + 4] = it[i].size_diff;
  }
}
int ko_kss_n_checkpoints(void* h) {
  KssH* k = static_cast<KssH*>(h);
  return static_cast<int>(KO_KSS_CALL(k, checkpoints()).size());
}
// out: rows of {iteration, previous, updated, stopped}; improvements as float.
void ko_kss_checkpoints(void* h, std::int64_t* out, float* improvements) {
  KssH* k = static_cast<KssH*>(h);
  const auto& cp = KO_KSS_CALL(k, checkpoints());
  for (std::size_t i = 0; i < cp.size(); i++) {
    out[4 * i + 0] = cp[i].iteration;
    out[4 * i + 1] = cp[i].previous;
    out[4 * i + 2] = cp[i].updated;
    out[4 * i + 3] = cp[i].stopped ? 1 : 0;
    improvements[i] = cp[i].improvement;
  }
}
// Initial weight table, row-major over i < j: n0 (n0 - 1) / 2 values.
void ko_kss_initial_weights(void* h, std::int64_t* out) {
  KssH* k = static_cast<KssH*>(h);
  const auto& w = KO_KSS_CALL(k, initial_weights());
  std::size_t i = 0;
  for (const auto& p : w) out[i++] = p.second;
}
std::int64_t ko_kss_stat(void* h, int which) {
  KssH* k = static_cast<KssH*>(h);
  switch (which) {
    case 0: return KO_KSS_CALL(k, initial_total_size());
    case 1: return KO_KSS_CALL(k, final_total_size());
    case 2: return KO_KSS_CALL(k, initial_total_spss_weight());
    case 3: return KO_KSS_CALL(k, n_processed());
  }
  return -1;
}
// "meta" line 0 (kmer_set_set.h:45-56) with keys ascending.
std::int64_t ko_kss_meta(void* h, char* out, std::int64_t cap) {
  KssH* k = static_cast<KssH*>(h);
  std::string s = ko::serialize_adjacency_list(KO_KSS_CALL(k, children()));
  if (out && static_cast<std::int64_t>(s.size()) < cap) std::memcpy(out, s.c_str(), s.size() + 1);
  return static_cast<std::int64_t>(s.size());
}
int ko_kss_dump(void* h, const char* dir, const char* ext) {
  KssH* k = static_cast<KssH*>(h);
  return KO_KSS_CALL(k, dump(dir, ext)) ? 0 : 1;
}
void* ko_kss_load(const char* dir, const char* ext, int k, int n, int key_bytes, int canon) {
  KssH* h = new KssH;
  h->g = ko::Geom{k, n, key_bytes};
  h->canon = canon != 0;
  bool ok;
  if (h->wide()) {
    h->k64.reset(new ko::KmerSetSet<std::uint64_t>());
    ok = ko::KmerSetSet<std::uint64_t>::load(h->g, dir, ext, h->k64.get());
  } else {
    h->k32.reset(new ko::KmerSetSet<std::uint32_t>());
    ok = ko::KmerSetSet<std::uint32_t>::load(h->g, dir, ext, h->k32.get());
  }
  if (!ok) {
    delete h;
    return nullptr;
  }
  return h;
}


// ---- k-mer counter (lib/core/kmer_counter.h) ---------------------------------------------------
void* ko_counter_new(int k, int n, int key_bytes) { return new ko::KmerCounter(ko::Geom{k, n, key_bytes}); }
void ko_counter_free(void* c) { delete static_cast<ko::KmerCounter*>(c); }
void ko_counter_add(void* c, std::uint64_t kmer, int v) {
  static_cast<ko::KmerCounter*>(c)->add(kmer, static_cast<std::uint8_t>(v));
}
int ko_counter_get(void* c, std::uint64_t kmer) { return static_cast<ko::KmerCounter*>(c)->get(kmer); }
std::int64_t ko_counter_size(void* c) { return static_cast<ko::KmerCounter*>(c)->size(); }
// FromFASTA on a whole file's bytes: 0 ok, 1 odd number of lines, 2 invalid FASTA file.
int ko_counter_from_fasta(void* c, const char* text, std::int64_t n, int canonical) {
  const std::vector<std::string> lines = ko::split_lines(std::string(text, static_cast<std::size_t>(n)));
  return static_cast<ko::KmerCounter*>(c)->from_fasta_lines(lines, canonical != 0);
}
// FromReads on '\n'-separated reads.
void ko_counter_from_reads(void* c, const char* text, std::int64_t n, int canonical) {
  static_cast<ko::KmerCounter*>(c)->from_reads(ko::split_lines(std::string(text, static_cast<std::size_t>(n))),
                                               canonical != 0);
}
void* ko_counter_to_set(void* c, int key_bytes, int cutoff, std::int64_t* cutoff_count) {
  ko::KmerCounter* k = static_cast<ko::KmerCounter*>(c);
  if (key_bytes > 4) {
    auto r = k->to_kmer_set<std::uint64_t>(cutoff);
    *cutoff_count = r.second;
    return wrap_set(std::move(r.first));
  }
  auto r = k->to_kmer_set<std::uint32_t>(cutoff);
  *cutoff_count = r.second;
  return wrap_set(std::move(r.first));
}

}  // extern "C"
