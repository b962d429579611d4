// How fast can single-wave workgroups pull tile-sized chunks out of HBM?  (measurement aid for
// DESIGN.md 3.1; not part of the library.)  Every workgroup loads one "tile": a descriptor, then
// `vecs` 16-byte vectors per lane from one or two places, optionally through LDS.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

struct Desc {
  long a, b;  // first vector of each range
};

template <int kVecs, bool kTwoRanges, bool kLds, bool kUseDesc>
__global__ __launch_bounds__(64) void k_tile_load(const Desc* desc, const uint4* keys, long stride_vecs,
                                                  long b_off, unsigned* out) {
  __shared__ uint4 lds[kVecs * 64];
  long a = long(blockIdx.x) * stride_vecs, b = a + b_off;
  if (kUseDesc) {
    const Desc d = desc[blockIdx.x];
    a = d.a;
    b = d.b;
  }
  uint4 r[kVecs];
#pragma unroll
  for (int j = 0; j < kVecs; j++) {
    const long at = (kTwoRanges && j >= kVecs / 2) ? b + (j - kVecs / 2) * 64 : a + j * 64;
    r[j] = keys[at + threadIdx.x];
  }
  unsigned acc = 0;
  if (kLds) {
#pragma unroll
    for (int j = 0; j < kVecs; j++) lds[j * 64 + threadIdx.x] = r[j];
    __syncthreads();
    acc = lds[(threadIdx.x * 7) % (kVecs * 64)].x;
  } else {
#pragma unroll
    for (int j = 0; j < kVecs; j++) acc ^= r[j].x ^ r[j].w;
  }
  if (acc == 0x12345678u) out[blockIdx.x] = acc;
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
  const int nb = 32768;
  const long per = 150;  // vectors between the starts of consecutive tiles' ranges (2.4 KB, config 2's shape)
  const long total_vecs = 10L * 1000 * 1000;  // 160 MB: covers nb * 256 contiguous vectors and both halves
  uint4* d_keys;
  Desc* d_desc;
  unsigned* d_out;
  hipMalloc(&d_keys, (total_vecs + 4096) * sizeof(uint4));
  hipMemset(d_keys, 1, (total_vecs + 4096) * sizeof(uint4));
  // bounds: contiguous variants touch up to nb * 256 + 512 vectors; ranged ones up to
  // total_vecs / 2 + nb * per + 6 * 64
  if (long(nb) * 256 + 512 > total_vecs || total_vecs / 2 + long(nb) * per + 6 * 64 > total_vecs) return 1;
  hipMalloc(&d_desc, nb * sizeof(Desc));
  hipMalloc(&d_out, nb * 4);
  // tiles of 2 x 2.4 KB: A range in the first 40 MB, B range in the second
  std::vector<Desc> h(nb);
  for (int i = 0; i < nb; i++) h[i] = Desc{long(i) * per, total_vecs / 2 + long(i) * per};
  hipMemcpy(d_desc, h.data(), nb * sizeof(Desc), hipMemcpyHostToDevice);
  Desc* d_desc_contig;
  hipMalloc(&d_desc_contig, nb * sizeof(Desc));
  for (int i = 0; i < nb; i++) h[i] = Desc{long(i) * 256, 0};
  hipMemcpy(d_desc_contig, h.data(), nb * sizeof(Desc), hipMemcpyHostToDevice);
  printf("per range %ld vectors (%ld B); %d workgroups\n", per, per * 16, nb);
  auto report = [&](const char* name, float us, double bytes) {
    printf("%-60s %7.1f us  %5.2f TB/s\n", name, us, bytes / us / 1e6);
  };
  // one contiguous 4 KB per workgroup, no descriptor, no LDS
  report("4 vec/lane contiguous, no desc, regs only",
         time_it([&] { hipLaunchKernelGGL((k_tile_load<4, false, false, false>), dim3(nb), dim3(64), 0, 0, d_desc, d_keys, 256L, 0L, d_out); }),
         double(nb) * 4096);
  report("4 vec/lane contiguous, desc, regs only",
         time_it([&] { hipLaunchKernelGGL((k_tile_load<4, false, false, true>), dim3(nb), dim3(64), 0, 0, d_desc_contig, d_keys, 256L, 0L, d_out); }),
         double(nb) * 4096);
  report("4 vec/lane in two ranges 40 MB apart, desc, regs only",
         time_it([&] { hipLaunchKernelGGL((k_tile_load<4, true, false, true>), dim3(nb), dim3(64), 0, 0, d_desc, d_keys, 256L, 0L, d_out); }),
         double(nb) * 4096);
  report("4 vec/lane in two ranges, desc, through LDS",
         time_it([&] { hipLaunchKernelGGL((k_tile_load<4, true, true, true>), dim3(nb), dim3(64), 0, 0, d_desc, d_keys, 256L, 0L, d_out); }),
         double(nb) * 4096);
  report("6 vec/lane in two ranges, desc, through LDS",
         time_it([&] { hipLaunchKernelGGL((k_tile_load<6, true, true, true>), dim3(nb), dim3(64), 0, 0, d_desc, d_keys, 256L, 0L, d_out); }),
         double(nb) * 6144);
  report("2 vec/lane in two ranges, desc, through LDS",
         time_it([&] { hipLaunchKernelGGL((k_tile_load<2, true, true, true>), dim3(nb), dim3(64), 0, 0, d_desc, d_keys, 256L, 0L, d_out); }),
         double(nb) * 2048);
  report("8 vec/lane contiguous, no desc, regs only (16384 wgs)",
         time_it([&] { hipLaunchKernelGGL((k_tile_load<8, false, false, false>), dim3(nb / 2), dim3(64), 0, 0, d_desc, d_keys, 512L, 0L, d_out); }),
         double(nb / 2) * 8192);
  return 0;
}
