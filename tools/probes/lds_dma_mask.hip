// Does an LDS-DMA load (global_load_lds_dwordx4) skip the lanes that EXEC masks off?  (GPU box)
//   hipcc --offload-arch=gfx950 -O3 tools/probes/lds_dma_mask.hip -o /tmp/lds_dma_mask && /tmp/lds_dma_mask
// A 2 816-byte row (176 x 16 B) is copied by three instructions; the third runs with lanes 48-63 off.  Their 256 bytes of the
// LDS image must keep the fill value.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
__global__ void k(const float* src, float* out, int nb) {
  extern __shared__ __align__(16) unsigned char s[];
  const int lane = threadIdx.x;
  for (int i = lane; i < 1024; i += 64) ((float*)s)[i] = -1.f;
  __syncthreads();
  for (int kk = 0; kk * 1024 < nb; ++kk) {
    const unsigned off = (kk * 64 + lane) * 16;
    if (off < (unsigned)nb) __builtin_amdgcn_global_load_lds((gptr_t)((const char*)src + off), (lptr_t)(s + kk * 1024), 16, 0, 0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  for (int i = lane; i < 1024; i += 64) out[i] = ((float*)s)[i];
}
int main() {
  std::vector<float> h(1024), o(1024);
  for (int i = 0; i < 1024; ++i) h[i] = (float)i;
  float *d, *dout;
  CK(hipMalloc(&d, 4096)); CK(hipMalloc(&dout, 4096));
  CK(hipMemcpy(d, h.data(), 4096, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 4096, 0, d, dout, 2816);
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(o.data(), dout, 4096, hipMemcpyDeviceToHost));
  int bad_copy = 0, bad_keep = 0;
  for (int i = 0; i < 704; ++i) bad_copy += o[i] != (float)i;
  for (int i = 704; i < 1024; ++i) bad_keep += o[i] != -1.f;
  printf("copied part: %s; bytes behind the row (masked lanes): %s\n", bad_copy ? "MISMATCH" : "ok", bad_keep ? "OVERWRITTEN" : "kept");
  return bad_copy || bad_keep;
}
