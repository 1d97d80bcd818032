// What the SHAPE of a wave's store instruction costs when a [53760 x 704] fp32 matrix (151 MB, the d ctx' of C2) is written by
// 1680 workgroups of 4 wavefronts, each wavefront owning a 32-row x 176-column tile (the epilogue of gemm_bx_areg_kernel):
//   A  16 rows x 64 B per instruction   (the 16x16 MFMA accumulator as it stands: lane (r, g) -> row r, columns 16 j + 4 g)
//   B  row-major sweep of the tile      (1 KB per instruction in 704-B row segments: what an LDS transposition would give)
//   C  4 rows x 256 B per instruction
// each with plain and nontemporal stores, and with the tile order of the kernel (column tile fastest) or row tile fastest.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/store_shape.hip -o tools/probes/store_shape && tools/probes/store_shape
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
constexpr int M = 53760, N = 704, TN = 176;
template <int SHAPE, bool NT>
__global__ __launch_bounds__(256) void store_kernel(float* C, float v) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tile_n = blockIdx.x & 3, tile_m = blockIdx.x >> 2;
  const int m0 = tile_m * 128 + wave * 32, n0 = tile_n * TN;
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  const f32x4 val = {v, v + lane, v, v};
  auto st = [&](int row, int col) {
    f32x4* p = reinterpret_cast<f32x4*>(C + (int64_t)(m0 + row) * N + n0 + col);
    if (NT) __builtin_nontemporal_store(val, p); else *p = val;
  };
  if (SHAPE == 0) {
    const int r = lane & 15, g = lane >> 4;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 11; ++j) st(16 * i + r, 16 * j + 4 * g);
  } else if (SHAPE == 1) {
#pragma unroll
    for (int it = 0; it < 22; ++it) { const int idx = it * 64 + lane; st(idx / 44, 4 * (idx % 44)); }
  } else {
    // 4 rows x 64 columns per instruction; the last 48 columns of a row as 4 rows x 48 columns (lanes 12..15 of each row idle)
    const int rr = lane >> 4, c = lane & 15;
#pragma unroll
    for (int rb = 0; rb < 8; ++rb)
#pragma unroll
      for (int cb = 0; cb < 3; ++cb) {
        if (cb < 2 || c < 12) st(4 * rb + rr, 64 * cb + 4 * c);
      }
  }
}
template <int SHAPE, bool NT>
int run(float* C, const char* name) {
  hipEvent_t t0, t1; CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
  float best = 1e9f;
  for (int rep = 0; rep < 6; ++rep) {
    CK(hipEventRecord(t0, 0));
    for (int k = 0; k < 10; ++k) hipLaunchKernelGGL((store_kernel<SHAPE, NT>), dim3(1680), dim3(256), 0, 0, C, (float)rep);
    CK(hipEventRecord(t1, 0));
    CK(hipDeviceSynchronize());
    float ms = 0; CK(hipEventElapsedTime(&ms, t0, t1));
    if (ms / 10 < best) best = ms / 10;
  }
  printf("%-46s %7.1f us   %5.2f TB/s\n", name, 1e3f * best, 4.0 * M * N / (best * 1e9));
  return 0;
}
int main() {
  float* C; CK(hipMalloc(&C, (size_t)M * N * 4));
  if (run<0, false>(C, "A 16 rows x 64 B (accumulator as it stands)")) return 1;
  if (run<0, true>(C, "A nontemporal")) return 1;
  if (run<1, false>(C, "B row-major sweep (704-B row segments)")) return 1;
  if (run<1, true>(C, "B nontemporal")) return 1;
  if (run<2, false>(C, "C 4 rows x 256 B")) return 1;
  if (run<2, true>(C, "C nontemporal")) return 1;
  CK(hipMemsetAsync(C, 0, (size_t)M * N * 4, 0)); CK(hipDeviceSynchronize());
  hipEvent_t t0, t1; CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
  CK(hipEventRecord(t0, 0)); for (int k = 0; k < 10; ++k) CK(hipMemsetAsync(C, 0, (size_t)M * N * 4, 0)); CK(hipEventRecord(t1, 0)); CK(hipDeviceSynchronize());
  float ms = 0; CK(hipEventElapsedTime(&ms, t0, t1));
  printf("%-46s %7.1f us   %5.2f TB/s\n", "hipMemsetAsync", 1e2f * ms, 4.0 * M * N / (ms / 10 * 1e9));
  return 0;
}
