// StreamVByte "0124" pack / unpack of the SPSS string lengths (len - K), the form in which
// KmerSetCompact keeps them in memory (lib/core/kmer_set_compact.h:257-265,269-287:
// streamvbyte_encode_0124 / streamvbyte_decode_0124 of lemire/streamvbyte v0.4.1, which is
// not under /root/reference -- format restated from its published description, see
// oracle/ko_compact.h): ceil(n / 4) control bytes, then the data bytes; value i has a 2-bit
// code in control byte i / 4 at bits [2(i % 4), 2(i % 4) + 1]; code 0 / 1 / 2 / 3 = 0 / 1 / 2 / 4
// little-endian data bytes, smallest width that holds the value.
//
// One thread per group of four values; the data offset of a group is an exclusive scan of
// the groups' data sizes.
#include "ksh_internal.h"

#include <algorithm>

namespace ksh {

__device__ __forceinline__ int svb_code(uint32_t v) { return v == 0 ? 0 : (v < 256u ? 1 : (v < 65536u ? 2 : 3)); }
__device__ __forceinline__ int svb_bytes(int code) { return code == 3 ? 4 : code; }

__global__ __launch_bounds__(256) void k_svb_group_sizes(const uint32_t* __restrict__ in, int64_t n,
                                                          int64_t n_groups,
                                                          int64_t* __restrict__ sizes) {
  const int64_t g = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (g >= n_groups) return;
  int total = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int64_t idx = 4 * g + i;
    if (idx < n) total += svb_bytes(svb_code(in[idx]));
  }
  sizes[g] = total;
}

__global__ __launch_bounds__(256) void k_svb_control_sizes(const uint8_t* __restrict__ ctrl,
                                                            int64_t n, int64_t n_groups,
                                                            int64_t* __restrict__ sizes) {
  const int64_t g = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (g >= n_groups) return;
  const uint8_t c = ctrl[g];
  int total = 0;
#pragma unroll
  for (int i = 0; i < 4; i++)
    if (4 * g + i < n) total += svb_bytes((c >> (2 * i)) & 3);
  sizes[g] = total;
}

__global__ __launch_bounds__(256) void k_svb_encode(const uint32_t* __restrict__ in, int64_t n,
                                                     int64_t n_groups,
                                                     const int64_t* __restrict__ starts,
                                                     uint8_t* __restrict__ out) {
  const int64_t g = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (g >= n_groups) return;
  uint8_t* data = out + n_groups + starts[g];
  uint8_t control = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int64_t idx = 4 * g + i;
    if (idx >= n) break;
    const uint32_t v = in[idx];
    const int code = svb_code(v);
    control |= uint8_t(code << (2 * i));
    const int nb = svb_bytes(code);
    for (int b = 0; b < nb; b++) *data++ = uint8_t(v >> (8 * b));
  }
  out[g] = control;
}

__global__ __launch_bounds__(256) void k_svb_decode(const uint8_t* __restrict__ in, int64_t n,
                                                     int64_t n_groups,
                                                     const int64_t* __restrict__ starts,
                                                     uint32_t* __restrict__ out) {
  const int64_t g = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (g >= n_groups) return;
  const uint8_t c = in[g];
  const uint8_t* data = in + n_groups + starts[g];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int64_t idx = 4 * g + i;
    if (idx >= n) break;
    const int nb = svb_bytes((c >> (2 * i)) & 3);
    uint32_t v = 0;
    for (int b = 0; b < nb; b++) v |= uint32_t(*data++) << (8 * b);
    out[idx] = v;
  }
}

static int svb_plan(ksh_ctx* ctx, int64_t n_groups, int64_t** sizes, int64_t** total) {
  KSH_TRY(arena_reserve(ctx, size_t(n_groups + 8) * 8 + size_t(n_groups / 256 + 4096) * 8 + (1u << 16)));
  arena_reset(ctx);
  *sizes = static_cast<int64_t*>(arena_alloc(ctx, size_t(n_groups + 1) * 8));
  *total = *sizes + n_groups;
  if (!*sizes) return fail(KSH_INTERNAL, "scratch arena too small");
  return KSH_OK;
}

}  // namespace ksh

using namespace ksh;

extern "C" {

int ksh_svb_encode_0124(ksh_ctx* ctx, const uint32_t* d_in, int64_t n, uint8_t* d_out,
                        int64_t* bytes) {
  if (!ctx || !bytes || n < 0) return fail(KSH_INVALID_ARGUMENT, "bad argument");
  KSH_HIP(hipSetDevice(ctx->device));
  if (n == 0) {
    *bytes = 0;
    return KSH_OK;
  }
  if (!d_in) return fail(KSH_INVALID_ARGUMENT, "d_in is NULL");
  const int64_t n_groups = (n + 3) / 4;
  int64_t *sizes, *total;
  KSH_TRY(svb_plan(ctx, n_groups, &sizes, &total));
  const unsigned blocks = unsigned((n_groups + 255) / 256);
  hipLaunchKernelGGL(k_svb_group_sizes, dim3(blocks), dim3(256), 0, ctx->stream, d_in, n, n_groups,
                     sizes);
  KSH_TRY(scan_exclusive_i64(ctx, sizes, sizes, n_groups, total));
  if (d_out)
    hipLaunchKernelGGL(k_svb_encode, dim3(blocks), dim3(256), 0, ctx->stream, d_in, n, n_groups, sizes,
                       d_out);
  KSH_HIP(hipGetLastError());
  KSH_HIP(hipMemcpyAsync(ctx->h_pinned, total, 8, hipMemcpyDeviceToHost, ctx->stream));
  KSH_HIP(hipStreamSynchronize(ctx->stream));
  *bytes = n_groups + ctx->h_pinned[0];
  return KSH_OK;
}

int ksh_svb_decode_0124(ksh_ctx* ctx, const uint8_t* d_in, int64_t n, uint32_t* d_out,
                        int64_t* bytes_read) {
  if (!ctx || n < 0) return fail(KSH_INVALID_ARGUMENT, "bad argument");
  KSH_HIP(hipSetDevice(ctx->device));
  if (n == 0) {
    if (bytes_read) *bytes_read = 0;
    return KSH_OK;
  }
  if (!d_in || !d_out) return fail(KSH_INVALID_ARGUMENT, "NULL buffer");
  const int64_t n_groups = (n + 3) / 4;
  int64_t *sizes, *total;
  KSH_TRY(svb_plan(ctx, n_groups, &sizes, &total));
  const unsigned blocks = unsigned((n_groups + 255) / 256);
  hipLaunchKernelGGL(k_svb_control_sizes, dim3(blocks), dim3(256), 0, ctx->stream, d_in, n, n_groups,
                     sizes);
  KSH_TRY(scan_exclusive_i64(ctx, sizes, sizes, n_groups, total));
  hipLaunchKernelGGL(k_svb_decode, dim3(blocks), dim3(256), 0, ctx->stream, d_in, n, n_groups, sizes,
                     d_out);
  KSH_HIP(hipGetLastError());
  KSH_HIP(hipMemcpyAsync(ctx->h_pinned, total, 8, hipMemcpyDeviceToHost, ctx->stream));
  KSH_HIP(hipStreamSynchronize(ctx->stream));
  if (bytes_read) *bytes_read = n_groups + ctx->h_pinned[0];
  return KSH_OK;
}

}  // extern "C"
