// FASTA text -> the reads' ACGT fragments as an SPSS-shaped 2-bit stream, on device: the front
// half of KmerCounter::FromFASTA (lib/core/kmer_counter.h:136-206: lines 0, 2, 4, ... are
// headers starting with '>', lines 1, 3, 5, ... are reads over ACGTN) and of FromReads
// (:64-133: a read is split at every 'N', every K-long window of a fragment is one k-mer).
// A fragment shorter than K holds no k-mer and is dropped; the fragments that remain, with
// their lengths, are exactly what ksh_spss_decode_plan + ksh_kmer_count_write consume, so
// counting reuses the decode pipeline (bucket histograms, scatter, per-bucket sort) with run
// lengths compared against the cutoff instead of plain duplicate removal.
//
// One thread per 64-byte chunk throughout; every pass is a streaming read of the text:
//   newlines per chunk -> scan            (line parity of every byte)
//   validate, count fragment starts/ends  -> scan (fragment index of every byte)
//   fragment [start, end) positions       -> length, kept?, scans (output string index / base)
//   kept bases per chunk -> scan          -> pack through LDS
// gfx950 only.
#include "ksh_bytes.h"
#include "ksh_internal.h"

#include <algorithm>

namespace ksh {

namespace {

constexpr int kFaThreads = 256;
constexpr int kFaChunk = 64;
constexpr int kFaSpan = kFaThreads * kFaChunk;

inline size_t fa256(size_t x) { return (x + 255) & ~size_t(255); }
inline unsigned fa_blocks(int64_t n, int per) { return unsigned(std::max<int64_t>(1, (n + per - 1) / per)); }

__device__ __forceinline__ bool is_acgt(unsigned char ch) { return ch == 'A' || ch == 'C' || ch == 'G' || ch == 'T'; }

__global__ __launch_bounds__(256) void k_fa_newlines(const unsigned char* __restrict__ text, int64_t n,
                                                      int64_t* __restrict__ counts) {
  __shared__ unsigned char lds[kChunkLds];
  const Chunk ch = stage_chunk(text, n, lds);
  if (ch.n == 0) return;
  const int64_t c = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  int nl = 0;
  for (int i = 0; i < ch.n; i++) nl += ch.at(i) == '\n';
  counts[c] = nl;
}

// flags bit 0: a header line that is empty or does not start with '>', or a read byte outside
// ACGTN ("invalid FASTA file", kmer_counter.h:170-190).
// starts[c] = fragments that start in chunk c; a fragment is a maximal run of ACGT bytes
// inside one read line.
__global__ __launch_bounds__(256) void k_fa_starts(const unsigned char* __restrict__ text, int64_t n,
                                                    const int64_t* __restrict__ nl_before,
                                                    int64_t* __restrict__ starts, int* __restrict__ flags) {
  __shared__ unsigned char lds[kChunkLds];
  const Chunk ch = stage_chunk(text, n, lds);
  if (ch.n == 0) return;
  const int64_t c = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  int64_t line = nl_before[c];
  unsigned char prev = ch.prev;
  int n_start = 0;
  bool bad = false;
  for (int i = 0; i < ch.n; i++) {
    const unsigned char b = ch.at(i);
    const bool read_line = line & 1;
    if (prev == '\n' && !read_line && b != '>') bad = true;  // also catches an empty header line
    if (read_line && !(is_acgt(b) || b == 'N' || b == '\n')) bad = true;
    if (read_line && is_acgt(b) && !is_acgt(prev)) n_start++;  // prev is on the same line or '\n'
    line += b == '\n';
    prev = b;
  }
  starts[c] = n_start;
  if (bad) atomicOr(flags, 1);
}

// frag_start[r] / frag_end[r]: byte range of fragment r.  The e-th end closes the e-th start,
// so ends are numbered by counting them the same way: ends before chunk c = starts before it,
// minus one if a fragment is open across the chunk's first byte.
__global__ __launch_bounds__(256) void k_fa_ranges(const unsigned char* __restrict__ text, int64_t n,
                                                    const int64_t* __restrict__ nl_before,
                                                    const int64_t* __restrict__ starts_before,
                                                    int64_t* __restrict__ frag_start,
                                                    int64_t* __restrict__ frag_end) {
  __shared__ unsigned char lds[kChunkLds];
  const Chunk ch = stage_chunk(text, n, lds);
  if (ch.n == 0) return;
  const int64_t c = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  const int64_t b0 = c * kFaChunk;
  int64_t line = nl_before[c];
  unsigned char prev = ch.prev;
  int64_t r = starts_before[c];  // index of the next fragment to start
  for (int i = 0; i < ch.n; i++) {
    const unsigned char b = ch.at(i);
    const bool read_line = line & 1;
    const bool base = read_line && is_acgt(b);
    if (base && !is_acgt(prev)) frag_start[r++] = b0 + i;
    const unsigned char next = i + 1 < ch.n ? ch.at(i + 1) : ch.next;
    if (base && !is_acgt(next)) frag_end[r - 1] = b0 + i + 1;  // r - 1: the fragment holding byte i
    line += b == '\n';
    prev = b;
  }
}

__global__ __launch_bounds__(256) void k_fa_keep(const int64_t* __restrict__ frag_start,
                                                  const int64_t* __restrict__ frag_end, int64_t n_frag, int k,
                                                  int64_t* __restrict__ kept_idx, int64_t* __restrict__ kept_len) {
  const int64_t r = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (r >= n_frag) return;
  const int64_t len = frag_end[r] - frag_start[r];
  kept_idx[r] = len >= k;
  kept_len[r] = len >= k ? len : 0;
}

// kept[c] = bases of chunk c that belong to a kept fragment
__global__ __launch_bounds__(256) void k_fa_kept_bases(const unsigned char* __restrict__ text, int64_t n,
                                                        const int64_t* __restrict__ nl_before,
                                                        const int64_t* __restrict__ starts_before,
                                                        const int64_t* __restrict__ frag_start,
                                                        const int64_t* __restrict__ frag_end, int k,
                                                        int64_t* __restrict__ kept) {
  __shared__ unsigned char lds[kChunkLds];
  const Chunk ch = stage_chunk(text, n, lds);
  if (ch.n == 0) return;
  const int64_t c = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  const int64_t b0 = c * kFaChunk;
  int64_t line = nl_before[c];
  unsigned char prev = ch.prev;
  int64_t r = starts_before[c];
  bool keep = false;
  if (b0 > 0 && r > 0) keep = frag_end[r - 1] > b0 && frag_end[r - 1] - frag_start[r - 1] >= k;  // open fragment
  int cnt = 0;
  for (int i = 0; i < ch.n; i++) {
    const unsigned char b = ch.at(i);
    const bool base = (line & 1) && is_acgt(b);
    if (base && !is_acgt(prev)) {
      keep = frag_end[r] - frag_start[r] >= k;
      r++;
    }
    cnt += base && keep;
    line += b == '\n';
    prev = b;
  }
  kept[c] = cnt;
}

__global__ __launch_bounds__(256) void k_fa_lens(const int64_t* __restrict__ frag_start,
                                                  const int64_t* __restrict__ frag_end,
                                                  const int64_t* __restrict__ kept_idx, int64_t n_frag, int k,
                                                  uint32_t* __restrict__ lens) {
  const int64_t r = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (r >= n_frag) return;
  const int64_t len = frag_end[r] - frag_start[r];
  if (len >= k) lens[kept_idx[r]] = uint32_t(len - k);
}

// The workgroup's kept bases are consecutive in the output: codes go to LDS (one byte per
// base), then 32 of them are packed per word; the first and the last word of the span may be
// shared with the neighbouring workgroups and are ORed in.
__global__ __launch_bounds__(kFaThreads) void k_fa_pack(const unsigned char* __restrict__ text, int64_t n,
                                                         const int64_t* __restrict__ nl_before,
                                                         const int64_t* __restrict__ starts_before,
                                                         const int64_t* __restrict__ frag_start,
                                                         const int64_t* __restrict__ frag_end,
                                                         const int64_t* __restrict__ kept_before, int k,
                                                         unsigned long long* __restrict__ words) {
  __shared__ unsigned char codes[kFaSpan + 32];
  __shared__ unsigned char lds[kChunkLds];
  const Chunk ch = stage_chunk(text, n, lds);
  const int64_t n_chunks = (n + kFaChunk - 1) / kFaChunk;
  const int64_t first_chunk = int64_t(blockIdx.x) * kFaThreads;
  const int64_t last_chunk = std::min<int64_t>(first_chunk + kFaThreads, n_chunks);  // exclusive
  const int64_t base0 = kept_before[first_chunk], base1 = kept_before[last_chunk];
  const int lead = int(base0 & 31);
  for (int i = threadIdx.x; i < lead; i += kFaThreads) codes[i] = 0;
  const int64_t c = first_chunk + threadIdx.x;
  if (ch.n > 0) {
    const int64_t b0 = c * kFaChunk;
    int64_t line = nl_before[c];
    unsigned char prev = ch.prev;
    int64_t r = starts_before[c];
    bool keep = false;
    if (b0 > 0 && r > 0) keep = frag_end[r - 1] > b0 && frag_end[r - 1] - frag_start[r - 1] >= k;
    int at = lead + int(kept_before[c] - base0);
    for (int i = 0; i < ch.n; i++) {
      const unsigned char b = ch.at(i);
      const bool base = (line & 1) && is_acgt(b);
      if (base && !is_acgt(prev)) {
        keep = frag_end[r] - frag_start[r] >= k;
        r++;
      }
      if (base && keep) codes[at++] = b == 'C' ? 1 : b == 'G' ? 2 : b == 'T' ? 3 : 0;
      line += b == '\n';
      prev = b;
    }
  }
  __syncthreads();
  const int n_codes = lead + int(base1 - base0);
  const int n_words = (n_codes + 31) / 32;
  unsigned long long* out = words + base0 / 32;
  for (int w = threadIdx.x; w < n_words; w += kFaThreads) {
    unsigned long long x = 0;
    const int hi = std::min(32, n_codes - w * 32);
    for (int j = (w == 0 ? lead : 0); j < hi; j++)
      x |= static_cast<unsigned long long>(codes[w * 32 + j]) << (62 - 2 * j);
    if (w == 0 || w == n_words - 1) atomicOr(&out[w], x);
    else out[w] = x;
  }
}

struct FastaPlan {
  const unsigned char* text = nullptr;
  int64_t n_bytes = 0, n_frag = 0, n_kept = 0, n_bases = 0;
  int k = 0;
  int64_t *nl_before = nullptr, *starts_before = nullptr, *kept_before = nullptr;  // n_chunks + 1 each
  int64_t *frag_start = nullptr, *frag_end = nullptr, *kept_idx = nullptr;         // n_frag (+ 1)
};

}  // namespace

}  // namespace ksh

using namespace ksh;

extern "C" {

int ksh_fasta_plan(ksh_ctx* ctx, const ksh_geom* g, const char* d_text, int64_t n_bytes, int64_t* n_fragments,
                   int64_t* n_bases) {
  if (!ctx) return fail(KSH_INVALID_ARGUMENT, "ctx is NULL");
  KSH_TRY(check_geom(g));
  if (!n_fragments || !n_bases) return fail(KSH_INVALID_ARGUMENT, "NULL output");
  if (n_bytes < 0 || (n_bytes > 0 && !d_text)) return fail(KSH_INVALID_ARGUMENT, "bad text buffer");
  FastaPlan* p = static_cast<FastaPlan*>(ctx->fasta_plan);
  if (!p) {
    p = new FastaPlan;
    ctx->fasta_plan = p;
    ctx->fasta_plan_free = [](void* q) { delete static_cast<FastaPlan*>(q); };
  }
  *p = FastaPlan();
  p->text = reinterpret_cast<const unsigned char*>(d_text);
  p->n_bytes = n_bytes;
  p->k = g->k;
  *n_fragments = 0;
  *n_bases = 0;
  if (n_bytes == 0) return KSH_OK;  // no lines: an empty counter (std::getline yields nothing)
  hipStream_t st = ctx->stream;
  const int64_t n_chunks = (n_bytes + kFaChunk - 1) / kFaChunk;
  const size_t per_chunk = fa256(size_t(n_chunks + 1) * 8);
  KSH_TRY(arena_reserve(ctx, size_t(n_chunks / 256 + 4096) * 8 + (1u << 16) + 512));
  arena_reset(ctx);
  int* flags = static_cast<int*>(arena_alloc(ctx, 256));
  if (!flags) return fail(KSH_INTERNAL, "scratch arena too small");
  KSH_HIP(hipMemsetAsync(flags, 0, sizeof(int), st));
  // pass 1: per-chunk arrays (the fragment arrays follow once their count is known)
  KSH_TRY(slot_reserve(ctx, kSlotText, 3 * per_chunk));
  ctx->text_slot_owner = 2;
  char* at = ctx->slot[kSlotText];
  p->nl_before = reinterpret_cast<int64_t*>(at);
  p->starts_before = reinterpret_cast<int64_t*>(at + per_chunk);
  p->kept_before = reinterpret_cast<int64_t*>(at + 2 * per_chunk);
  hipLaunchKernelGGL(k_fa_newlines, dim3(fa_blocks(n_chunks, 256)), dim3(256), 0, st, p->text, n_bytes,
                     p->nl_before);
  KSH_TRY(scan_exclusive_i64(ctx, p->nl_before, p->nl_before, n_chunks, p->nl_before + n_chunks));
  hipLaunchKernelGGL(k_fa_starts, dim3(fa_blocks(n_chunks, 256)), dim3(256), 0, st, p->text, n_bytes,
                     p->nl_before, p->starts_before, flags);
  KSH_TRY(scan_exclusive_i64(ctx, p->starts_before, p->starts_before, n_chunks, p->starts_before + n_chunks));
  KSH_HIP(hipMemcpyAsync(ctx->h_pinned, p->nl_before + n_chunks, 8, hipMemcpyDeviceToHost, st));
  KSH_HIP(hipMemcpyAsync(ctx->h_pinned + 1, p->starts_before + n_chunks, 8, hipMemcpyDeviceToHost, st));
  KSH_HIP(hipMemcpyAsync(ctx->h_pinned + 2, flags, sizeof(int), hipMemcpyDeviceToHost, st));
  KSH_HIP(hipMemcpyAsync(ctx->h_pinned + 3, p->text + n_bytes - 1, 1, hipMemcpyDeviceToHost, st));
  KSH_HIP(hipStreamSynchronize(st));
  const int64_t newlines = ctx->h_pinned[0];
  const bool open_line = *reinterpret_cast<unsigned char*>(ctx->h_pinned + 3) != '\n';
  const int64_t n_lines = newlines + (open_line ? 1 : 0);
  if (n_lines % 2 != 0)
    return fail(KSH_FAILED_PRECONDITION, "FASTA files should have an even number of lines");
  if (*reinterpret_cast<int*>(ctx->h_pinned + 2) & 1) return fail(KSH_FAILED_PRECONDITION, "invalid FASTA file");
  p->n_frag = ctx->h_pinned[1];
  if (p->n_frag == 0) return KSH_OK;
  // pass 2: fragments.  Their arrays go behind the per-chunk ones in the same slot; if that
  // makes the slot grow, its contents are lost and the two prefix arrays are recomputed (two
  // streaming passes).
  const size_t per_frag = fa256(size_t(p->n_frag + 1) * 8);
  const bool grows = 3 * per_chunk + 3 * per_frag > ctx->slot_bytes[kSlotText];
  KSH_TRY(slot_reserve(ctx, kSlotText, 3 * per_chunk + 3 * per_frag));
  at = ctx->slot[kSlotText];
  if (grows) {  // a fresh buffer: the two prefix arrays are gone
    p->nl_before = reinterpret_cast<int64_t*>(at);
    p->starts_before = reinterpret_cast<int64_t*>(at + per_chunk);
    p->kept_before = reinterpret_cast<int64_t*>(at + 2 * per_chunk);
    hipLaunchKernelGGL(k_fa_newlines, dim3(fa_blocks(n_chunks, 256)), dim3(256), 0, st, p->text, n_bytes,
                       p->nl_before);
    KSH_TRY(scan_exclusive_i64(ctx, p->nl_before, p->nl_before, n_chunks, p->nl_before + n_chunks));
    hipLaunchKernelGGL(k_fa_starts, dim3(fa_blocks(n_chunks, 256)), dim3(256), 0, st, p->text, n_bytes,
                       p->nl_before, p->starts_before, flags);
    KSH_TRY(scan_exclusive_i64(ctx, p->starts_before, p->starts_before, n_chunks,
                               p->starts_before + n_chunks));
  }
  p->frag_start = reinterpret_cast<int64_t*>(at + 3 * per_chunk);
  p->frag_end = reinterpret_cast<int64_t*>(at + 3 * per_chunk + per_frag);
  p->kept_idx = reinterpret_cast<int64_t*>(at + 3 * per_chunk + 2 * per_frag);
  KSH_TRY(arena_reserve(ctx, per_frag + size_t(std::max(n_chunks, p->n_frag) / 256 + 4096) * 8 + (1u << 16)));
  arena_reset(ctx);
  int64_t* kept_len = static_cast<int64_t*>(arena_alloc(ctx, size_t(p->n_frag + 1) * 8));
  if (!kept_len) return fail(KSH_INTERNAL, "scratch arena too small");
  hipLaunchKernelGGL(k_fa_ranges, dim3(fa_blocks(n_chunks, 256)), dim3(256), 0, st, p->text, n_bytes,
                     p->nl_before, p->starts_before, p->frag_start, p->frag_end);
  hipLaunchKernelGGL(k_fa_keep, dim3(fa_blocks(p->n_frag, 256)), dim3(256), 0, st, p->frag_start, p->frag_end,
                     p->n_frag, p->k, p->kept_idx, kept_len);
  KSH_TRY(scan_exclusive_i64(ctx, p->kept_idx, p->kept_idx, p->n_frag, p->kept_idx + p->n_frag));
  KSH_TRY(scan_exclusive_i64(ctx, kept_len, kept_len, p->n_frag, kept_len + p->n_frag));
  hipLaunchKernelGGL(k_fa_kept_bases, dim3(fa_blocks(n_chunks, 256)), dim3(256), 0, st, p->text, n_bytes,
                     p->nl_before, p->starts_before, p->frag_start, p->frag_end, p->k, p->kept_before);
  KSH_TRY(scan_exclusive_i64(ctx, p->kept_before, p->kept_before, n_chunks, p->kept_before + n_chunks));
  KSH_HIP(hipMemcpyAsync(ctx->h_pinned, p->kept_idx + p->n_frag, 8, hipMemcpyDeviceToHost, st));
  KSH_HIP(hipMemcpyAsync(ctx->h_pinned + 1, kept_len + p->n_frag, 8, hipMemcpyDeviceToHost, st));
  KSH_HIP(hipMemcpyAsync(ctx->h_pinned + 2, p->kept_before + n_chunks, 8, hipMemcpyDeviceToHost, st));
  KSH_HIP(hipStreamSynchronize(st));
  p->n_kept = ctx->h_pinned[0];
  p->n_bases = ctx->h_pinned[1];
  if (ctx->h_pinned[2] != p->n_bases) return fail(KSH_INTERNAL, "fragment passes disagree on the base count");
  *n_fragments = p->n_kept;
  *n_bases = p->n_bases;
  return KSH_OK;
}

int ksh_fasta_write(ksh_ctx* ctx, uint64_t* d_words, uint32_t* d_lens) {
  if (!ctx) return fail(KSH_INVALID_ARGUMENT, "ctx is NULL");
  FastaPlan* p = static_cast<FastaPlan*>(ctx->fasta_plan);
  if (!p || (p->n_kept > 0 && ctx->text_slot_owner != 2))
    return fail(KSH_FAILED_PRECONDITION, "ksh_fasta_write without ksh_fasta_plan");
  if (p->n_kept == 0) return KSH_OK;
  if (!d_words || !d_lens) return fail(KSH_INVALID_ARGUMENT, "NULL output buffer");
  hipStream_t st = ctx->stream;
  const int64_t n_chunks = (p->n_bytes + kFaChunk - 1) / kFaChunk;
  hipLaunchKernelGGL(k_fa_lens, dim3(fa_blocks(p->n_frag, 256)), dim3(256), 0, st, p->frag_start, p->frag_end,
                     p->kept_idx, p->n_frag, p->k, d_lens);
  KSH_HIP(hipMemsetAsync(d_words, 0, size_t((p->n_bases + 31) / 32) * 8, st));
  hipLaunchKernelGGL(k_fa_pack, dim3(fa_blocks(n_chunks, kFaThreads)), dim3(kFaThreads), 0, st, p->text,
                     p->n_bytes, p->nl_before, p->starts_before, p->frag_start, p->frag_end, p->kept_before, p->k,
                     reinterpret_cast<unsigned long long*>(d_words));
  KSH_HIP(hipGetLastError());
  return KSH_OK;
}

}  // extern "C"
