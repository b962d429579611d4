// Byte streams (FASTA / SPSS text) are parsed by one thread per 64-byte chunk, 256 chunks per
// workgroup.  A thread walking its own 64 bytes straight from global memory issues 64 byte loads
// whose lanes touch 64 different cache lines each; here the workgroup first copies its 16 KB span
// into LDS with coalesced 16-byte loads and every thread then reads its chunk from LDS.
// gfx950 only.
#ifndef KSH_BYTES_H_
#define KSH_BYTES_H_

#include <hip/hip_runtime.h>

#include <cstdint>

namespace ksh {

constexpr int kChunkBytes = 64;
constexpr int kChunkThreads = 256;
constexpr int kChunkStride = 68;  // bytes between consecutive threads' chunks in LDS: odd in dwords
constexpr int kChunkLds = kChunkThreads * kChunkStride;

// A thread's view of its chunk: byte i is at(i) for 0 <= i < n; prev / next are the bytes just
// outside it ('\n' outside the text, which is what every parser here wants at both ends).
struct Chunk {
  const unsigned char* p;
  int n;
  unsigned char prev, next;
  __device__ __forceinline__ unsigned char at(int i) const { return p[i]; }
};

// Every thread of the 256-thread workgroup calls this; thread t gets the chunk starting at byte
// (blockIdx.x * 256 + t) * 64.  Threads whose chunk starts at or past n get an empty one.
__device__ __forceinline__ Chunk stage_chunk(const unsigned char* __restrict__ text, int64_t n,
                                             unsigned char* __restrict__ lds) {
  const int64_t block0 = int64_t(blockIdx.x) * (kChunkThreads * kChunkBytes);
  const int64_t block_n = min(int64_t(kChunkThreads * kChunkBytes), n - block0);
  if ((reinterpret_cast<uintptr_t>(text) & 15) == 0) {
    const int64_t n_vec = block_n > 0 ? block_n / 16 : 0;
    const uint4* src = reinterpret_cast<const uint4*>(text + block0);
    for (int v = threadIdx.x; v < n_vec; v += kChunkThreads) {
      const uint4 x = src[v];
      unsigned* d = reinterpret_cast<unsigned*>(lds + (v >> 2) * kChunkStride + (v & 3) * 16);
      d[0] = x.x;
      d[1] = x.y;
      d[2] = x.z;
      d[3] = x.w;
    }
    for (int64_t i = n_vec * 16 + threadIdx.x; i < block_n; i += kChunkThreads)
      lds[(i >> 6) * kChunkStride + (i & 63)] = text[block0 + i];
  } else {
    for (int64_t i = threadIdx.x; i < block_n; i += kChunkThreads)
      lds[(i >> 6) * kChunkStride + (i & 63)] = text[block0 + i];
  }
  __syncthreads();
  Chunk c;
  const int64_t b0 = block0 + int64_t(threadIdx.x) * kChunkBytes;
  c.p = lds + threadIdx.x * kChunkStride;
  c.n = b0 < n ? int(min(int64_t(kChunkBytes), n - b0)) : 0;
  c.prev = '\n';
  c.next = '\n';
  if (c.n > 0) {
    if (threadIdx.x > 0) c.prev = lds[(threadIdx.x - 1) * kChunkStride + kChunkBytes - 1];
    else if (b0 > 0) c.prev = text[b0 - 1];
    if (b0 + kChunkBytes < n)
      c.next = threadIdx.x + 1 < kChunkThreads ? lds[(threadIdx.x + 1) * kChunkStride] : text[b0 + kChunkBytes];
  }
  return c;
}

}  // namespace ksh

#endif
