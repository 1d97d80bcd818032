// What does s_memtime count?  (GPU box)  One wavefront spins for a fixed number of s_memtime ticks; the host times the launch
// with events and the kernel also reads s_memrealtime (100 MHz).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void spin(unsigned long long ticks, unsigned long long* out) {
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long t = t0;
  while (t - t0 < ticks) { __builtin_amdgcn_s_sleep(8); t = __builtin_amdgcn_s_memtime(); }
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t - t0; out[1] = __builtin_amdgcn_s_memrealtime() - r0; }
}
int main() {
  unsigned long long* d; CK(hipMalloc(&d, 16));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int grid : {1, 1024}) for (unsigned long long ticks : {1000000ull, 10000000ull}) {
    hipLaunchKernelGGL(spin, dim3(grid), dim3(64), 0, 0, ticks, d); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a)); hipLaunchKernelGGL(spin, dim3(grid), dim3(64), 0, 0, ticks, d); CK(hipEventRecord(b)); CK(hipDeviceSynchronize());
    float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
    unsigned long long h[2]; CK(hipMemcpy(h, d, 16, hipMemcpyDeviceToHost));
    printf("grid %4d: %llu s_memtime ticks = %llu s_memrealtime ticks (100 MHz -> %.1f us) ; host %.1f us  -> s_memtime at %.1f MHz\n", grid, h[0], h[1], h[1] / 100.0, ms * 1e3, h[0] / (h[1] / 100.0));
  }
  return 0;
}
