// v_mfma_f32_4x4x1_16b_f32 as a "four scalars times one vector" FMA (GPU box):
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma4x4_probe.hip -o /tmp/mfma4x4 && /tmp/mfma4x4
// 1. semantics of the A-broadcast fields (cbsz = 4, abid = q): D_m[l] = C_m[l] + A[4 q + m] * B[l] for every lane l - the four
//    coefficients sit in ONE quad of the A register and the immediate picks the quad;
// 2. issue rate of independent / dependent chains beside plain v_fma_f32, at 1..4 wavefronts per SIMD (cycles per instruction
//    per SIMD from s_memtime, the median wavefront).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <cmath>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void sem_kernel(const float* a, const float* b, float* out) {
  const int l = threadIdx.x;
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  f4 d0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], acc, 0, 0, 0);     // no broadcast: the lane's own quad
  f4 d1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], acc, 4, 0, 0);     // quad 0 for every block
  f4 d2 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], acc, 4, 5, 0);     // quad 5
  f4 d3 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], acc, 4, 15, 0);    // quad 15
  f4 d4 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], d3, 4, 2, 0);      // accumulate on top
  for (int m = 0; m < 4; ++m) {
    out[(0 * 4 + m) * 64 + l] = d0[m];
    out[(1 * 4 + m) * 64 + l] = d1[m];
    out[(2 * 4 + m) * 64 + l] = d2[m];
    out[(3 * 4 + m) * 64 + l] = d3[m];
    out[(4 * 4 + m) * 64 + l] = d4[m];
  }
}

// mode 0: 16 independent v_fma_f32 per iteration; 1: 16 independent MFMAs (16 accumulator quads); 2: 4 MFMAs on ONE accumulator
// (dependent chain) x 4; 3: 12 v_fma + 4 MFMA interleaved; 4: 8 v_fma + 8 MFMA; 5: MFMA result consumed by a v_mul (dkt -> gsin shape)
template <int MODE>
__global__ __launch_bounds__(256) void rate_kernel(float* out, long long* cyc, int iters, float seed) {
  const int l = threadIdx.x & 63;
  float x[16];
  f4 acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) { x[i] = seed * (l + i); acc[i] = f4{0.f, 0.f, 0.f, 0.f}; }
  float a = seed + l, b = seed * 0.5f + l;
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < 16; ++i) x[i] = __builtin_fmaf(x[i], a, b);
    } else if (MODE == 1) {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, x[i], acc[i], 4, 3, 0);
    } else if (MODE == 2) {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i & 3] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, x[i], acc[i & 3], 4, 3, 0);
    } else if (MODE == 3) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, x[i], acc[i], 4, 1, 0);
        x[4 + 3 * i] = __builtin_fmaf(x[4 + 3 * i], a, b);
        x[5 + 3 * i] = __builtin_fmaf(x[5 + 3 * i], a, b);
        x[6 + 3 * i] = __builtin_fmaf(x[6 + 3 * i], a, b);
      }
    } else if (MODE == 4) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, x[i], acc[i], 4, 1, 0);
        x[8 + i] = __builtin_fmaf(x[8 + i], a, b);
      }
    } else if (MODE == 5) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        f4 z = f4{0.f, 0.f, 0.f, 0.f};
        z = __builtin_amdgcn_mfma_f32_4x4x1f32(a, x[4 * i], z, 4, 0, 0);
        z = __builtin_amdgcn_mfma_f32_4x4x1f32(b, x[4 * i + 1], z, 4, 1, 0);
        x[4 * i + 2] *= z[0] + z[1];
        x[4 * i + 3] *= z[2] + z[3];
      }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) asm volatile("" : "+v"(x[i]));
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += x[i] + acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 12345.678f) out[0] = s;
  if (l == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

template <int MODE>
static int run_rate(const char* name, int waves_per_simd, int per_iter_valu, int per_iter_mfma) {
  float* out; long long* cyc;
  const int blocks = 256 * waves_per_simd, waves = blocks * 4, iters = 2000;
  CK(hipMalloc(&out, 64)); CK(hipMalloc(&cyc, waves * sizeof(long long)));
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(rate_kernel<MODE>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters, 1.0e-3f);
    CK(hipDeviceSynchronize());
  }
  std::vector<long long> h(waves);
  CK(hipMemcpy(h.data(), cyc, waves * sizeof(long long), hipMemcpyDeviceToHost));
  std::sort(h.begin(), h.end());
  const double med = (double)h[waves / 2] / iters;       // cycles per iteration of ONE wavefront (s_memtime ticks = shader cycles)
  printf("%-44s waves/SIMD %d: %7.1f cycles per iteration per wave (%2d valu + %2d mfma) -> %5.2f cycles per instruction per SIMD\n", name,
         waves_per_simd, med, per_iter_valu, per_iter_mfma, med / waves_per_simd / (per_iter_valu + per_iter_mfma));
  CK(hipFree(out)); CK(hipFree(cyc));
  return 0;
}

int main() {
  std::vector<float> a(64), b(64), out(5 * 4 * 64);
  for (int i = 0; i < 64; ++i) { a[i] = 1.f + 0.37f * i; b[i] = 0.5f - 0.11f * i; }
  float *da, *db, *dout;
  CK(hipMalloc(&da, 256)); CK(hipMalloc(&db, 256)); CK(hipMalloc(&dout, out.size() * 4));
  CK(hipMemcpy(da, a.data(), 256, hipMemcpyHostToDevice)); CK(hipMemcpy(db, b.data(), 256, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(sem_kernel, dim3(1), dim3(64), 0, 0, da, db, dout);
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost));
  int bad[5] = {0, 0, 0, 0, 0};
  const int quad[5] = {-1, 0, 5, 15, 2};
  for (int t = 0; t < 5; ++t)
    for (int m = 0; m < 4; ++m)
      for (int l = 0; l < 64; ++l) {
        const int q = quad[t] < 0 ? l / 4 : quad[t];
        float want = a[4 * q + m] * b[l];
        if (t == 4) want = fmaf(a[4 * 2 + m], b[l], a[4 * 15 + m] * b[l]);
        if (out[(t * 4 + m) * 64 + l] != want) bad[t]++;
      }
  printf("semantics: own quad %s, cbsz=4 abid=0 %s, abid=5 %s, abid=15 %s, accumulate (fmaf chain, bitwise) %s\n", bad[0] ? "MISMATCH" : "ok",
         bad[1] ? "MISMATCH" : "ok", bad[2] ? "MISMATCH" : "ok", bad[3] ? "MISMATCH" : "ok", bad[4] ? "MISMATCH" : "ok");
  for (int w = 1; w <= 4; ++w) {
    if (run_rate<0>("16 independent v_fma_f32", w, 16, 0)) return 1;
    if (run_rate<1>("16 independent mfma_4x4x1 (4 FMAs each)", w, 0, 16)) return 1;
    if (run_rate<2>("16 mfma_4x4x1 on 4 accumulators (chains of 4)", w, 0, 16)) return 1;
    if (run_rate<3>("12 v_fma + 4 mfma interleaved", w, 12, 4)) return 1;
    if (run_rate<4>("8 v_fma + 8 mfma interleaved", w, 8, 8)) return 1;
    if (run_rate<5>("2 dependent mfma -> v_add -> v_mul, x4", w, 16, 8)) return 1;
  }
  return 0;
}
