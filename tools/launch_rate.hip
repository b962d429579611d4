// Workgroup dispatch rate on this GPU: empty kernels of many small workgroups (measurement
// aid for DESIGN.md 3.1; not part of the library).  hipcc --offload-arch=gfx950 -O3 launch_rate.hip
#include <hip/hip_runtime.h>

#include <cstdio>

template <int kLds>
__global__ void k_empty(int* out) {
  __shared__ int lds[kLds > 0 ? kLds : 1];
  if (kLds > 0) lds[threadIdx.x] = threadIdx.x;
  if (out && threadIdx.x == 0 && kLds > 0 && lds[0] == 12345) out[blockIdx.x] = 1;
}

// one dependent pair of loads per workgroup: a scalar "descriptor", then 1 KB of keys
__global__ void k_two_loads(const long* desc, const uint4* keys, int* out) {
  const long at = desc[blockIdx.x];
  const uint4 v = keys[at + threadIdx.x];
  if (v.x == 0x12345678u && out) out[blockIdx.x] = 1;
}

template <typename F>
float time_it(F f, int n = 20) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  for (int i = 0; i < 3; i++) f();
  hipEventRecord(a);
  for (int i = 0; i < n; i++) f();
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  return ms / n * 1e3f;
}

int main() {
  const int blocks[] = {8192, 16384, 32768, 65536};
  for (int nb : blocks) {
    printf("%6d workgroups:", nb);
    printf("  64 thr no LDS %6.1f us", time_it([&] { hipLaunchKernelGGL(k_empty<0>, dim3(nb), dim3(64), 0, 0, nullptr); }));
    printf("  64 thr 4 KB LDS %6.1f us", time_it([&] { hipLaunchKernelGGL(k_empty<1024>, dim3(nb), dim3(64), 0, 0, nullptr); }));
    printf("  256 thr 4 KB LDS %6.1f us\n", time_it([&] { hipLaunchKernelGGL(k_empty<1024>, dim3(nb), dim3(256), 0, 0, nullptr); }));
  }
  // dependent loads
  const int nb = 32768;
  long* d_desc;
  uint4* d_keys;
  const size_t n_vec = size_t(nb) * 64 * 5;
  hipMalloc(&d_desc, nb * sizeof(long));
  hipMalloc(&d_keys, n_vec * sizeof(uint4));
  hipMemset(d_keys, 0, n_vec * sizeof(uint4));
  long* h = new long[nb];
  for (int i = 0; i < nb; i++) h[i] = long(i) * 64 * 5;
  hipMemcpy(d_desc, h, nb * sizeof(long), hipMemcpyHostToDevice);
  printf("%6d workgroups, scalar load then 1 KB of keys each: %6.1f us\n", nb,
         time_it([&] { hipLaunchKernelGGL(k_two_loads, dim3(nb), dim3(64), 0, 0, d_desc, d_keys, nullptr); }));
  return 0;
}
