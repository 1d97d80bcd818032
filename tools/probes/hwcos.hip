// Probe (build: hipcc -O2 --offload-arch=gfx950 hwcos.hip -o hwcos): accuracy of the hardware v_cos_f32 / v_sin_f32 (input in revolutions) on [-0.125, 0.125] and [-0.5, 0.5]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__global__ void k(const float* u, float* c, float* s, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { c[i] = __builtin_amdgcn_cosf(u[i]); s[i] = __builtin_amdgcn_sinf(u[i]); }
}
int main() {
  const int n = 1 << 22;
  for (double range : {0.125, 0.5}) {
    std::vector<float> u(n), c(n), s(n);
    for (int i = 0; i < n; ++i) u[i] = (float)(range * (2.0 * i / (n - 1) - 1.0));
    float *du, *dc, *ds;
    hipMalloc(&du, n * 4); hipMalloc(&dc, n * 4); hipMalloc(&ds, n * 4);
    hipMemcpy(du, u.data(), n * 4, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(du, dc, ds, n);
    hipMemcpy(c.data(), dc, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(s.data(), ds, n * 4, hipMemcpyDeviceToHost);
    double ec = 0, es = 0;
    for (int i = 0; i < n; ++i) {
      const double x = 2.0 * M_PI * (double)u[i];
      ec = fmax(ec, fabs(c[i] - cos(x))); es = fmax(es, fabs(s[i] - sin(x)));
    }
    printf("range +-%.3f rev: max abs err cos %.3e sin %.3e\n", range, ec, es);
  }
  return 0;
}
