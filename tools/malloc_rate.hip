// Cost of hipMalloc / hipFree by size on this box (the loop's phase times for "decode of the
// inputs" and "merges" vary 4x between boxes of the pool, and those are the phases that
// allocate; measurement aid, not part of the library).
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
  hipFree(nullptr);
  for (size_t mb : {4ul, 64ul, 400ul, 4000ul, 25600ul}) {
    const int reps = mb >= 4000 ? 2 : 16;
    std::vector<void*> p(reps);
    double t0 = now();
    for (int i = 0; i < reps; i++) hipMalloc(&p[i], mb << 20);
    const double t_alloc = (now() - t0) / reps;
    t0 = now();
    for (int i = 0; i < reps; i++) hipMemsetAsync(p[i], 0, mb << 20, 0);
    hipDeviceSynchronize();
    const double t_touch = (now() - t0) / reps;
    t0 = now();
    for (int i = 0; i < reps; i++) hipFree(p[i]);
    const double t_free = (now() - t0) / reps;
    printf("%6zu MB: hipMalloc %9.3f ms (%6.2f us/MB)  first memset %9.3f ms  hipFree %9.3f ms\n", mb, t_alloc * 1e3,
           t_alloc * 1e6 / mb, t_touch * 1e3, t_free * 1e3);
  }
  return 0;
}
