// Single-launch exclusive prefix sum ("chained scan") for the bucket- and tile-sized arrays of
// the plan kernels (10^4 .. 10^5 values), where a multi-launch scan is all launch latency.
// gfx950 only.
//
// Every workgroup scans its own 2048 values, publishes their sum in `state`, then adds up the
// sums published by the workgroups before it and writes its values out.  A launch has at most
// kChainMaxBlocks (128) workgroups of 256 threads, far fewer than the chip holds at once, so
// every workgroup of it becomes resident whatever the dispatch order, and a workgroup only ever
// waits for lower-numbered ones, which wait only for still lower ones: the chain always makes
// progress (and every poll loop is bounded all the same).  A sum is
// published as two words (its low 48 bits and its high 16 bits), each carrying the launch's
// epoch in its top 16 bits: a word is valid by itself, so the two need no ordering between
// them, words left by earlier launches never match, and the state array needs no clearing
// between launches.  XCD L2s are not coherent with each other, so the words are stored and
// polled with agent-scope atomics.  Sums are full 64-bit values (callers scan packed
// counters).
//
// The values come from a functor (index -> int64), which lets a caller fold the kernel that
// would have produced the input array into the scan; a second functor sees every value with
// its prefix, for callers that would otherwise read the result back in a follow-up kernel.
#ifndef KSH_SCAN_H_
#define KSH_SCAN_H_

#include "ksh_internal.h"

namespace ksh {

constexpr int kChainThreads = 256;
constexpr int kChainItems = 8;
constexpr int kChainTile = kChainThreads * kChainItems;
constexpr int64_t kChainMaxBlocks = 128;  // every workgroup reads all earlier sums: keep the chain short
constexpr int kChainValueBits = 48;       // payload bits of a published word

struct LoadArray {
  const int64_t* in;
  __device__ __forceinline__ int64_t operator()(int64_t i) const { return in[i]; }
};

// Called once per value with its index, its exclusive prefix and the value itself.
struct EmitNothing {
  __device__ __forceinline__ void operator()(int64_t, int64_t, int64_t) const {}
};

template <typename Load, typename Emit>
__global__ __launch_bounds__(kChainThreads) void k_scan_chained(Load load, Emit emit, int64_t* __restrict__ out,
                                                                 int64_t n, int64_t* __restrict__ total_out,
                                                                 unsigned long long* __restrict__ state,
                                                                 unsigned long long epoch) {
  __shared__ int64_t wave_sum[kChainThreads / 64];
  __shared__ int64_t block_prefix;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t b = blockIdx.x;
  const int64_t base = b * kChainTile + int64_t(threadIdx.x) * kChainItems;
  int64_t v[kChainItems];
  int64_t mine = 0;
#pragma unroll
  for (int i = 0; i < kChainItems; i++) {
    v[i] = base + i < n ? load(base + i) : 0;
    mine += v[i];
  }
  // inclusive scan of the thread sums across the wave, then across the four waves
  int64_t inc = mine;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int64_t o = __shfl_up(inc, d, 64);
    if (lane >= d) inc += o;
  }
  if (lane == 63) wave_sum[wave] = inc;
  __syncthreads();
  int64_t before = 0, block_total = 0;
#pragma unroll
  for (int w = 0; w < kChainThreads / 64; w++) {
    if (w < wave) before += wave_sum[w];
    block_total += wave_sum[w];
  }
  constexpr unsigned long long kValueMask = (1ull << kChainValueBits) - 1;
  const unsigned long long tag = epoch << kChainValueBits;
  if (threadIdx.x < 2) {
    const unsigned long long t = static_cast<unsigned long long>(block_total);
    const unsigned long long part = threadIdx.x == 0 ? (t & kValueMask) : (t >> kChainValueBits);
    __hip_atomic_store(&state[2 * b + threadIdx.x], tag | part, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (wave == 0) {
    unsigned long long sum_u = 0;
    for (int64_t first = b - 1; first >= 0; first -= 64) {
      const int64_t idx = first - lane;
      if (idx >= 0) {
        unsigned long long w0, w1;
        unsigned polls = 0;
        do {
          w0 = __hip_atomic_load(&state[2 * idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          w1 = __hip_atomic_load(&state[2 * idx + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          // every spin is bounded: a launch of at most kChainMaxBlocks workgroups is resident as a
          // whole, so a wait of seconds can only mean a broken launch; abort it rather than hang
          if (++polls == (1u << 28)) __builtin_trap();
        } while ((w0 >> kChainValueBits) != epoch || (w1 >> kChainValueBits) != epoch);
        sum_u += (w0 & kValueMask) | (w1 << kChainValueBits);
      }
    }
    int64_t sum = static_cast<int64_t>(sum_u);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) sum += __shfl_xor(sum, d, 64);
    if (lane == 0) block_prefix = sum;
  }
  __syncthreads();
  int64_t run = block_prefix + before + inc - mine;
  if (total_out && b == gridDim.x - 1 && threadIdx.x == kChainThreads - 1) *total_out = run + mine;
#pragma unroll
  for (int i = 0; i < kChainItems; i++) {
    if (base + i < n) {
      out[base + i] = run;
      emit(base + i, run, v[i]);
    }
    run += v[i];
  }
}

// Enqueues the chained scan when n fits it; returns false (nothing enqueued) otherwise.
template <typename Load, typename Emit = EmitNothing>
bool scan_exclusive_chained(ksh_ctx* ctx, const Load& load, int64_t* d_out, int64_t n, int64_t* d_total,
                            const Emit& emit = Emit()) {
  const int64_t blocks = (n + kChainTile - 1) / kChainTile;
  if (n <= 0 || blocks > kChainMaxBlocks || !ctx->scan_state) return false;
  ctx->scan_epoch = (ctx->scan_epoch + 1) & 0xFFFF;
  if (ctx->scan_epoch == 0) {
    // epochs are about to repeat: forget every word published so far
    (void)hipMemsetAsync(ctx->scan_state, 0, size_t(2 * kChainMaxBlocks) * sizeof(unsigned long long), ctx->stream);
    ctx->scan_epoch = 1;
  }
  hipLaunchKernelGGL((k_scan_chained<Load, Emit>), dim3(unsigned(blocks)), dim3(kChainThreads), 0, ctx->stream,
                     load, emit, d_out, n, d_total, ctx->scan_state,
                     static_cast<unsigned long long>(ctx->scan_epoch));
  return true;
}

}  // namespace ksh

#endif
