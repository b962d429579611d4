// Random-access rates on this GPU: 4-byte reads, plain writes and atomic compare-and-swap at
// pseudo-random indices of an 800 MB array (measurement aid for the SPSS encode kernels'
// design notes; not part of the library).
#include <hip/hip_runtime.h>

#include <cstdio>

__device__ __forceinline__ unsigned long long mix(unsigned long long x) {
  x ^= x >> 33;
  x *= 0xff51afd7ed558ccdull;
  x ^= x >> 33;
  x *= 0xc4ceb9fe1a85ec53ull;
  x ^= x >> 33;
  return x;
}

template <int kOp>
__global__ __launch_bounds__(256) void k_random(unsigned* a, unsigned long long n_slots, long n_ops, unsigned* sink) {
  const long i = long(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n_ops) return;
  const unsigned long long at = mix(i) % n_slots;
  if (kOp == 0) {
    if (a[at] == 0x12345678u) sink[0] = 1;
  } else if (kOp == 1) {
    a[at] = unsigned(i);
  } else if (kOp == 2) {
    const unsigned old = atomicCAS(&a[at], 0xFFFFFFFFu, unsigned(i));
    if (old != 0xFFFFFFFFu) a[at] = 0xFFFFFFFEu;
  } else {
    atomicAdd(&a[at], 1u);
  }
}

template <typename F>
float time_it(F f, int n = 5) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  f();
  hipEventRecord(a);
  for (int i = 0; i < n; i++) f();
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  return ms / n;
}

int main() {
  const unsigned long long n_slots = 200ull * 1000 * 1000;  // 800 MB of u32
  const long n_ops = 100L * 1000 * 1000;
  unsigned *a, *sink;
  hipMalloc(&a, n_slots * 4);
  hipMalloc(&sink, 4);
  hipMemset(a, 0xFF, n_slots * 4);
  const unsigned blocks = unsigned((n_ops + 255) / 256);
  const char* names[] = {"random 4-byte reads", "random 4-byte writes", "random CAS (+ write on failure)", "random atomicAdd"};
  float ms;
  ms = time_it([&] { hipLaunchKernelGGL(k_random<0>, dim3(blocks), dim3(256), 0, 0, a, n_slots, n_ops, sink); });
  printf("%-34s %7.2f ms  %6.1f G ops/s\n", names[0], ms, n_ops / ms / 1e6);
  ms = time_it([&] { hipLaunchKernelGGL(k_random<1>, dim3(blocks), dim3(256), 0, 0, a, n_slots, n_ops, sink); });
  printf("%-34s %7.2f ms  %6.1f G ops/s\n", names[1], ms, n_ops / ms / 1e6);
  hipMemset(a, 0xFF, n_slots * 4);
  ms = time_it([&] { hipLaunchKernelGGL(k_random<2>, dim3(blocks), dim3(256), 0, 0, a, n_slots, n_ops, sink); }, 1);
  printf("%-34s %7.2f ms  %6.1f G ops/s\n", names[2], ms, n_ops / ms / 1e6);
  ms = time_it([&] { hipLaunchKernelGGL(k_random<3>, dim3(blocks), dim3(256), 0, 0, a, n_slots, n_ops, sink); });
  printf("%-34s %7.2f ms  %6.1f G ops/s\n", names[3], ms, n_ops / ms / 1e6);
  // the same reads and writes confined to smaller arrays (L2: 4 MB per XCD, Infinity Cache: 256 MB)
  for (unsigned long long mb : {8ull, 32ull, 128ull, 256ull, 400ull}) {
    const unsigned long long slots = mb * 1000 * 1000 / 4;
    ms = time_it([&] { hipLaunchKernelGGL(k_random<0>, dim3(blocks), dim3(256), 0, 0, a, slots, n_ops, sink); });
    const float mw = time_it([&] { hipLaunchKernelGGL(k_random<1>, dim3(blocks), dim3(256), 0, 0, a, slots, n_ops, sink); });
    printf("random 4-byte reads in %4llu MB %7.2f ms  %6.1f G ops/s   writes %7.2f ms  %6.1f G ops/s\n", mb, ms,
           n_ops / ms / 1e6, mw, n_ops / mw / 1e6);
  }
  return 0;
}
