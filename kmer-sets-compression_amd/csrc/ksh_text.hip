// Text form of an SPSS on device: the file format of KmerSetCompact::Dump / Load
// (lib/core/kmer_set_compact.h:62-87 over WriteLines / ReadLines, lib/core/io.h:20-126): one
// string over ACGT per line, every line closed by '\n'.  The reference converts base by base
// on the host (ToStrings, kmer_set_compact.h:290-336; the private constructor, :206-266); here
// the 2-bit stream and the byte stream are converted into each other in HBM, so that a dump
// moves n_bases + n_strings bytes over PCIe once and nothing is converted on the CPU.
//
//   to text : string starts (scan of len + K) -> one end bit per base -> ends before every
//             64-base chunk (scan of popcounts) -> every thread spells one chunk into LDS
//             (a newline after each marked base), the workgroup copies its span out.
//   to SPSS : newlines before every 64-byte chunk (scan of ballots' popcounts) -> line ends ->
//             lengths, and every workgroup packs its span of bases into words through LDS
//             (the two words it shares with its neighbours are ORed in with atomics).
// gfx950 only.
#include "ksh_bytes.h"
#include "ksh_internal.h"

#include <algorithm>

namespace ksh {

namespace {

constexpr int kTextThreads = 256;
constexpr int kChunk = 64;                               // bases (to text) / bytes (to SPSS) per thread
constexpr int kSpan = kTextThreads * kChunk;             // per workgroup
constexpr int kTextLds = kSpan + kSpan / 4 + 64;         // + one newline per >= 4 bases (K >= 4) + slack

inline size_t a256(size_t x) { return (x + 255) & ~size_t(255); }
inline unsigned blocks_of(int64_t n, int per) { return unsigned(std::max<int64_t>(1, (n + per - 1) / per)); }

__global__ __launch_bounds__(256) void k_text_str_bases(const uint32_t* __restrict__ lens, int64_t n, int k,
                                                         int64_t* __restrict__ bases) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) bases[i] = int64_t(lens[i]) + k;
}

// end_bits: bit (p & 63) of word p >> 6 is set iff base p is the last base of a string.
__global__ __launch_bounds__(256) void k_text_mark_ends(const int64_t* __restrict__ str_start, int64_t n_strings,
                                                         unsigned long long* __restrict__ end_bits) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n_strings) return;
  const int64_t p = str_start[i + 1] - 1;
  atomicOr(&end_bits[p >> 6], 1ull << (p & 63));
}

__global__ __launch_bounds__(256) void k_text_popc(const unsigned long long* __restrict__ bits, int64_t n,
                                                    int64_t* __restrict__ counts) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) counts[i] = __popcll(bits[i]);
}

// Copies lds[0, n) to out[0, n) with 16-byte stores on the aligned interior.
__device__ __forceinline__ void block_copy_out(const unsigned char* lds, int64_t n, unsigned char* out) {
  const int head = int((16 - (reinterpret_cast<uintptr_t>(out) & 15)) & 15);
  const int64_t h = head < n ? head : n;
  if (int64_t(threadIdx.x) < h) out[threadIdx.x] = lds[threadIdx.x];
  const int64_t n_vec = (n - h) / 16;
  for (int64_t v = threadIdx.x; v < n_vec; v += kTextThreads) {
    uint4 x;
    unsigned char* b = reinterpret_cast<unsigned char*>(&x);
#pragma unroll
    for (int e = 0; e < 16; e++) b[e] = lds[h + v * 16 + e];
    *reinterpret_cast<uint4*>(out + h + v * 16) = x;
  }
  const int64_t done = h + n_vec * 16;
  if (int64_t(threadIdx.x) < n - done) out[done + threadIdx.x] = lds[done + threadIdx.x];
}

// One thread per 64-base chunk; a workgroup's chunks are consecutive, and so is its text.
__global__ __launch_bounds__(kTextThreads) void k_to_text(const uint64_t* __restrict__ words, int64_t n_bases,
                                                          const unsigned long long* __restrict__ end_bits,
                                                          const int64_t* __restrict__ ends_before,
                                                          unsigned char* __restrict__ text) {
  __shared__ unsigned char lds[kTextLds];
  const int64_t chunk = int64_t(blockIdx.x) * kTextThreads + threadIdx.x;
  const int64_t n_chunks = (n_bases + kChunk - 1) / kChunk;
  const int64_t first_chunk = int64_t(blockIdx.x) * kTextThreads;
  const int64_t last_chunk = std::min<int64_t>(first_chunk + kTextThreads, n_chunks);  // exclusive
  const int64_t span0 = first_chunk * kChunk + ends_before[first_chunk];
  const int64_t span1 = std::min<int64_t>(last_chunk * kChunk, n_bases) + ends_before[last_chunk];
  if (chunk < n_chunks) {
    const unsigned long long ends = end_bits[chunk];
    int at = int(chunk * kChunk + ends_before[chunk] - span0);
    const int64_t b0 = chunk * kChunk;
    const int n_here = int(std::min<int64_t>(kChunk, n_bases - b0));
    const uint64_t w0 = words[b0 / 32];
    const uint64_t w1 = n_here > 32 ? words[b0 / 32 + 1] : 0;
    for (int b = 0; b < n_here; b++) {
      const uint64_t w = b < 32 ? w0 : w1;
      const unsigned code = unsigned(w >> (62 - 2 * (b & 31))) & 3u;
      lds[at++] = "ACGT"[code];
      if ((ends >> b) & 1) lds[at++] = '\n';
    }
  }
  __syncthreads();
  block_copy_out(lds, span1 - span0, text + span0);
}

// ---- text -> SPSS ---------------------------------------------------------------------
// flags: bit 0 = a byte that is neither ACGT nor '\n'; bit 1 = a line shorter than K.
__global__ __launch_bounds__(256) void k_text_count_nl(const unsigned char* __restrict__ text, int64_t n_bytes,
                                                        int64_t* __restrict__ counts, int* __restrict__ flags) {
  __shared__ unsigned char lds[kChunkLds];
  const Chunk ch = stage_chunk(text, n_bytes, lds);
  if (ch.n == 0) return;
  const int64_t c = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  int nl = 0;
  bool bad = false;
  for (int i = 0; i < ch.n; i++) {
    const unsigned char b = ch.at(i);
    nl += b == '\n';
    bad |= !(b == 'A' || b == 'C' || b == 'G' || b == 'T' || b == '\n');
  }
  counts[c] = nl;
  if (bad) atomicOr(flags, 1);
}

// line_end[i] = byte index of the '\n' closing line i (n_bytes for an unterminated last line).
__global__ __launch_bounds__(256) void k_text_line_ends(const unsigned char* __restrict__ text, int64_t n_bytes,
                                                         const int64_t* __restrict__ nl_before,
                                                         int64_t n_lines, int64_t* __restrict__ line_end) {
  __shared__ unsigned char lds[kChunkLds];
  const Chunk ch = stage_chunk(text, n_bytes, lds);
  if (ch.n == 0) return;
  const int64_t c = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  const int64_t n_chunks = (n_bytes + kChunk - 1) / kChunk;
  const int64_t b0 = c * kChunk;
  int64_t line = nl_before[c];
  for (int i = 0; i < ch.n; i++)
    if (ch.at(i) == '\n') line_end[line++] = b0 + i;
  if (c == n_chunks - 1 && ch.at(ch.n - 1) != '\n') line_end[n_lines - 1] = n_bytes;
}

__global__ __launch_bounds__(256) void k_text_lens(const int64_t* __restrict__ line_end, int64_t n_lines, int k,
                                                    uint32_t* __restrict__ lens, int* __restrict__ flags) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n_lines) return;
  const int64_t len = line_end[i] - (i ? line_end[i - 1] + 1 : 0);
  if (len < k) {
    atomicOr(flags, 2);
    lens[i] = 0;
  } else {
    lens[i] = uint32_t(len - k);
  }
}

// One thread per 64-byte chunk; the workgroup's bases are consecutive: codes go to LDS (one
// byte per base), then 32 of them are packed per word.
__global__ __launch_bounds__(kTextThreads) void k_text_pack(const unsigned char* __restrict__ text, int64_t n_bytes,
                                                            const int64_t* __restrict__ nl_before,
                                                            unsigned long long* __restrict__ words) {
  __shared__ unsigned char codes[kSpan + 32];
  __shared__ unsigned char lds[kChunkLds];
  const Chunk ch = stage_chunk(text, n_bytes, lds);
  const int64_t n_chunks = (n_bytes + kChunk - 1) / kChunk;
  const int64_t first_chunk = int64_t(blockIdx.x) * kTextThreads;
  const int64_t last_chunk = std::min<int64_t>(first_chunk + kTextThreads, n_chunks);  // exclusive
  const int64_t base0 = first_chunk * kChunk - nl_before[first_chunk];
  const int64_t base1 = std::min<int64_t>(last_chunk * kChunk, n_bytes) - nl_before[last_chunk];
  const int lead = int(base0 & 31);  // the span's first base sits at this position of its word
  for (int i = threadIdx.x; i < lead; i += kTextThreads) codes[i] = 0;
  const int64_t c = first_chunk + threadIdx.x;
  if (ch.n > 0) {
    const int64_t b0 = c * kChunk;
    int at = lead + int(b0 - nl_before[c] - base0);
    for (int i = 0; i < ch.n; i++) {
      const unsigned char b = ch.at(i);
      if (b == '\n') continue;
      codes[at++] = b == 'C' ? 1 : b == 'G' ? 2 : b == 'T' ? 3 : 0;
    }
  }
  __syncthreads();
  const int n_codes = lead + int(base1 - base0);
  const int n_words = (n_codes + 31) / 32;
  unsigned long long* out = words + base0 / 32;
  for (int w = threadIdx.x; w < n_words; w += kTextThreads) {
    unsigned long long x = 0;
    const int hi = std::min(32, n_codes - w * 32);
    for (int j = (w == 0 ? lead : 0); j < hi; j++) x |= static_cast<unsigned long long>(codes[w * 32 + j]) << (62 - 2 * j);
    // the first and the last word of a span may be shared with the neighbouring workgroups
    if (w == 0 || w == n_words - 1) atomicOr(&out[w], x);
    else out[w] = x;
  }
}

struct TextPlan {
  const unsigned char* text = nullptr;
  int64_t n_bytes = 0, n_lines = 0, n_bases = 0;
  int k = 0;
  int64_t* nl_before = nullptr;  // n_chunks + 1
  int64_t* line_end = nullptr;   // n_lines
};

}  // namespace

}  // namespace ksh

using namespace ksh;

extern "C" {

int ksh_spss_to_text(ksh_ctx* ctx, const ksh_geom* g, const ksh_spss_view* s, char* d_text) {
  if (!ctx) return fail(KSH_INVALID_ARGUMENT, "ctx is NULL");
  KSH_TRY(check_geom(g));
  if (!s) return fail(KSH_INVALID_ARGUMENT, "spss is NULL");
  if (s->n_strings < 0 || s->n_bases < 0) return fail(KSH_INVALID_ARGUMENT, "negative size");
  if (g->k < 4) return fail(KSH_INVALID_ARGUMENT, "the text kernels need K >= 4");
  if (s->n_strings == 0 || s->n_bases == 0) return KSH_OK;
  if (!s->d_words || !s->d_lens || !d_text) return fail(KSH_INVALID_ARGUMENT, "NULL buffer");
  const int64_t n_chunks = (s->n_bases + kChunk - 1) / kChunk;
  const size_t bytes = a256(size_t(s->n_strings + 1) * 8) + a256(size_t(n_chunks) * 8) +
                       a256(size_t(n_chunks + 1) * 8) + size_t(std::max(s->n_strings, n_chunks) / 256 + 4096) * 8 +
                       (1u << 16);
  KSH_TRY(arena_reserve(ctx, bytes));
  arena_reset(ctx);
  int64_t* str_start = static_cast<int64_t*>(arena_alloc(ctx, size_t(s->n_strings + 1) * 8));
  unsigned long long* end_bits = static_cast<unsigned long long*>(arena_alloc(ctx, size_t(n_chunks) * 8));
  int64_t* ends_before = static_cast<int64_t*>(arena_alloc(ctx, size_t(n_chunks + 1) * 8));
  if (!str_start || !end_bits || !ends_before) return fail(KSH_INTERNAL, "scratch arena too small");
  hipStream_t st = ctx->stream;
  hipLaunchKernelGGL(k_text_str_bases, dim3(blocks_of(s->n_strings, 256)), dim3(256), 0, st, s->d_lens,
                     s->n_strings, g->k, str_start);
  KSH_TRY(scan_exclusive_i64(ctx, str_start, str_start, s->n_strings, str_start + s->n_strings));
  KSH_HIP(hipMemsetAsync(end_bits, 0, size_t(n_chunks) * 8, st));
  hipLaunchKernelGGL(k_text_mark_ends, dim3(blocks_of(s->n_strings, 256)), dim3(256), 0, st, str_start,
                     s->n_strings, end_bits);
  hipLaunchKernelGGL(k_text_popc, dim3(blocks_of(n_chunks, 256)), dim3(256), 0, st, end_bits, n_chunks,
                     ends_before);
  KSH_TRY(scan_exclusive_i64(ctx, ends_before, ends_before, n_chunks, ends_before + n_chunks));
  // the strings must tile the base stream exactly
  KSH_HIP(hipMemcpyAsync(ctx->h_pinned, str_start + s->n_strings, 8, hipMemcpyDeviceToHost, st));
  KSH_HIP(hipStreamSynchronize(st));
  if (ctx->h_pinned[0] != s->n_bases)
    return fail(KSH_INVALID_ARGUMENT, "sum of string lengths (%lld) != n_bases (%lld)",
                static_cast<long long>(ctx->h_pinned[0]), static_cast<long long>(s->n_bases));
  hipLaunchKernelGGL(k_to_text, dim3(blocks_of(n_chunks, kTextThreads)), dim3(kTextThreads), 0, st, s->d_words,
                     s->n_bases, end_bits, ends_before, reinterpret_cast<unsigned char*>(d_text));
  KSH_HIP(hipGetLastError());
  return KSH_OK;
}

int ksh_spss_from_text_plan(ksh_ctx* ctx, const ksh_geom* g, const char* d_text, int64_t n_bytes,
                            int64_t* n_strings, int64_t* n_bases) {
  if (!ctx) return fail(KSH_INVALID_ARGUMENT, "ctx is NULL");
  KSH_TRY(check_geom(g));
  if (!n_strings || !n_bases) return fail(KSH_INVALID_ARGUMENT, "NULL output");
  if (n_bytes < 0 || (n_bytes > 0 && !d_text)) return fail(KSH_INVALID_ARGUMENT, "bad text buffer");
  TextPlan* p = static_cast<TextPlan*>(ctx->text_plan);
  if (!p) {
    p = new TextPlan;
    ctx->text_plan = p;
    ctx->text_plan_free = [](void* q) { delete static_cast<TextPlan*>(q); };
  }
  *p = TextPlan();
  p->text = reinterpret_cast<const unsigned char*>(d_text);
  p->n_bytes = n_bytes;
  p->k = g->k;
  *n_strings = 0;
  *n_bases = 0;
  if (n_bytes == 0) return KSH_OK;
  const int64_t n_chunks = (n_bytes + kChunk - 1) / kChunk;
  KSH_TRY(arena_reserve(ctx, size_t(n_chunks / 256 + 4096) * 8 + (1u << 16) + 512));
  arena_reset(ctx);
  int* flags = static_cast<int*>(arena_alloc(ctx, 256));
  hipStream_t st = ctx->stream;
  // nl_before lives in the text slot: it is needed again by ksh_spss_from_text_write
  KSH_TRY(slot_reserve(ctx, kSlotText, a256(size_t(n_chunks + 1) * 8)));
  ctx->text_slot_owner = 1;
  p->nl_before = reinterpret_cast<int64_t*>(ctx->slot[kSlotText]);
  KSH_HIP(hipMemsetAsync(flags, 0, sizeof(int), st));
  hipLaunchKernelGGL(k_text_count_nl, dim3(blocks_of(n_chunks, 256)), dim3(256), 0, st, p->text, n_bytes,
                     p->nl_before, flags);
  KSH_TRY(scan_exclusive_i64(ctx, p->nl_before, p->nl_before, n_chunks, p->nl_before + n_chunks));
  KSH_HIP(hipMemcpyAsync(ctx->h_pinned, p->nl_before + n_chunks, 8, hipMemcpyDeviceToHost, st));
  KSH_HIP(hipMemcpyAsync(ctx->h_pinned + 1, flags, sizeof(int), hipMemcpyDeviceToHost, st));
  KSH_HIP(hipMemcpyAsync(ctx->h_pinned + 2, p->text + n_bytes - 1, 1, hipMemcpyDeviceToHost, st));
  KSH_HIP(hipStreamSynchronize(st));
  if (*reinterpret_cast<int*>(ctx->h_pinned + 1) & 1)
    return fail(KSH_INVALID_ARGUMENT, "SPSS text holds a byte that is neither A, C, G, T nor a newline");
  const int64_t newlines = ctx->h_pinned[0];
  const bool open_line = *reinterpret_cast<unsigned char*>(ctx->h_pinned + 2) != '\n';
  p->n_lines = newlines + (open_line ? 1 : 0);
  p->n_bases = n_bytes - newlines;
  *n_strings = p->n_lines;
  *n_bases = p->n_bases;
  return KSH_OK;
}

int ksh_spss_from_text_write(ksh_ctx* ctx, uint64_t* d_words, uint32_t* d_lens) {
  if (!ctx) return fail(KSH_INVALID_ARGUMENT, "ctx is NULL");
  TextPlan* p = static_cast<TextPlan*>(ctx->text_plan);
  if (!p || (p->n_bytes > 0 && (!p->nl_before || ctx->text_slot_owner != 1)))
    return fail(KSH_FAILED_PRECONDITION, "ksh_spss_from_text_write without ksh_spss_from_text_plan");
  if (p->n_lines == 0) return KSH_OK;
  if (!d_words || !d_lens) return fail(KSH_INVALID_ARGUMENT, "NULL output buffer");
  const int64_t n_chunks = (p->n_bytes + kChunk - 1) / kChunk;
  KSH_TRY(arena_reserve(ctx, a256(size_t(p->n_lines) * 8) + 512));
  arena_reset(ctx);
  int* flags = static_cast<int*>(arena_alloc(ctx, 256));
  int64_t* line_end = static_cast<int64_t*>(arena_alloc(ctx, size_t(p->n_lines) * 8));
  if (!flags || !line_end) return fail(KSH_INTERNAL, "scratch arena too small");
  hipStream_t st = ctx->stream;
  KSH_HIP(hipMemsetAsync(flags, 0, sizeof(int), st));
  hipLaunchKernelGGL(k_text_line_ends, dim3(blocks_of(n_chunks, 256)), dim3(256), 0, st, p->text, p->n_bytes,
                     p->nl_before, p->n_lines, line_end);
  hipLaunchKernelGGL(k_text_lens, dim3(blocks_of(p->n_lines, 256)), dim3(256), 0, st, line_end, p->n_lines, p->k,
                     d_lens, flags);
  KSH_HIP(hipMemsetAsync(d_words, 0, size_t((p->n_bases + 31) / 32) * 8, st));
  hipLaunchKernelGGL(k_text_pack, dim3(blocks_of(n_chunks, kTextThreads)), dim3(kTextThreads), 0, st, p->text,
                     p->n_bytes, p->nl_before, reinterpret_cast<unsigned long long*>(d_words));
  KSH_HIP(hipGetLastError());
  KSH_HIP(hipMemcpyAsync(ctx->h_pinned, flags, sizeof(int), hipMemcpyDeviceToHost, st));
  KSH_HIP(hipStreamSynchronize(st));
  if (*reinterpret_cast<int*>(ctx->h_pinned) & 2)
    return fail(KSH_INVALID_ARGUMENT, "SPSS text holds a line shorter than K = %d", p->k);
  return KSH_OK;
}

}  // extern "C"
