// Shared host/device helpers for the gfx950 kernels.  Wave size is 64 everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdarg.h>
#include <stdio.h>
#include "../../include/pfotgn.h"

#define PFO_WAVE 64

void pfo_set_error(const char* fmt, ...);

#define PFO_REQUIRE(cond, msg)                                    \
  do {                                                            \
    if (!(cond)) {                                                \
      pfo_set_error("%s: %s", __func__, msg);                     \
      return PFO_ERR_INVALID;                                     \
    }                                                             \
  } while (0)

#define PFO_LAUNCH_CHECK()                                                        \
  do {                                                                            \
    hipError_t e__ = hipGetLastError();                                           \
    if (e__ != hipSuccess) {                                                      \
      pfo_set_error("%s: launch failed: %s", __func__, hipGetErrorString(e__));   \
      return PFO_ERR_HIP;                                                         \
    }                                                                             \
  } while (0)

// live event timing (misc.hip); no-ops unless pfo_prof_enable(1)
void pfo_prof_begin(hipStream_t s);
void pfo_prof_end(int kind, double work, hipStream_t s);

static inline int64_t pfo_ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int64_t pfo_align_up(int64_t a, int64_t b) { return pfo_ceil_div(a, b) * b; }

#ifdef __HIPCC__
// ---------------------------------------------------------------------------------------------
// sin/cos of an fp32 argument of any magnitude (time-encoder arguments reach 1.7e7 on the synthetic
// graphs and beyond on epoch-second data).  The cosine is a pure function of the exactly rounded
// fp32 argument (SURVEY §7 hard part 1), so the range reduction must be accurate, not "fast-math":
//   |x| < 2e7 : fp32 only.  k = rint(x * 2/pi) from a two-term product (k < 2^24 stays an exact integer),
//               r = x - k*pi/2 by three FMAs against a three-term pi/2 (each FMA rounds an O(1) value once);
//               |error(r)| ~ 1.2e-7, validated against fp64 over +-2e7 (max |cos err| 1.03e-7).
//   otherwise : the same reduction in fp64 (two-term pi/2), a few times slower, never taken on the benchmark.
// Then fp32 minimax polynomials on [-pi/4, pi/4] (~1 ulp).
// slow path kept out of line so that it stays a real (almost never taken) branch instead of being if-converted
__device__ __attribute__((noinline)) float pfo_reduce_f64(float x, int* quadrant) {
  const double xd = (double)x;
  const double kd = rint(xd * 0.63661977236758134308);
  double r = fma(-kd, 1.57079632679489655800e+00, xd);
  r = fma(-kd, 6.12323399573676603587e-17, r);
  *quadrant = (int)((long long)kd & 3);
  return (float)r;
}

__device__ __forceinline__ void pfo_sincosf(float x, float& s, float& c) {
#pragma clang fp contract(off)   // p below is reused as a ROUNDED product: no implicit FMA formation in this function
  float rf;
  int q;
  if (fabsf(x) < 2.0e7f) {
    const float C_HI = 0.636619746685028076171875f, C_LO = 2.5682553e-08f;
    const float P1 = 1.57079637050628662109375f, P2 = -4.37113883e-08f, P3 = -1.71512451e-15f;
    const float p = x * C_HI;
    const float e = __builtin_fmaf(x, C_HI, -p);               // exact rounding error of p
    float kf = rintf(p);
    const float f = (p - kf) + __builtin_fmaf(x, C_LO, e);     // what p missed, |f| < 1.6
    kf = kf + rintf(f);
    float r = __builtin_fmaf(-kf, P1, x);
    r = __builtin_fmaf(-kf, P2, r);
    rf = __builtin_fmaf(-kf, P3, r);
    q = ((int)kf) & 3;
  } else {
    rf = pfo_reduce_f64(x, &q);
  }
  const float r2 = rf * rf;
  const float sp = fmaf(rf * r2, fmaf(r2, fmaf(r2, -1.9515295891e-4f, 8.3321608736e-3f), -1.6666654611e-1f), rf);
  const float cp = fmaf(r2 * r2, fmaf(r2, fmaf(r2, 2.443315711809948e-5f, -1.388731625493765e-3f), 4.166664568298827e-2f),
                        fmaf(r2, -0.5f, 1.0f));
  const float ss = (q & 1) ? cp : sp;
  const float cc = (q & 1) ? sp : cp;
  s = (q & 2) ? -ss : ss;
  c = ((q + 1) & 2) ? -cc : cc;
}

__device__ __forceinline__ float pfo_cosf(float x) {
  float s, c;
  pfo_sincosf(x, s, c);
  return c;
}

// TimeEncode element: one fp32 FMA, then cosine (model/time_encoding.py:23; SURVEY §7 hard part 1)
__device__ __forceinline__ float pfo_time_arg(float t, float w, float b) { return __builtin_fmaf(t, w, b); }

// ---------------------------------------------------------------------------------------------
// Philox4x32-10 counter-based generator
struct pfo_u4 { uint32_t x, y, z, w; };
__device__ __forceinline__ pfo_u4 pfo_philox(uint64_t seed, uint64_t ctr_lo, uint64_t ctr_hi) {
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
  uint32_t c0 = (uint32_t)ctr_lo, c1 = (uint32_t)(ctr_lo >> 32), c2 = (uint32_t)ctr_hi, c3 = (uint32_t)(ctr_hi >> 32);
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return pfo_u4{c0, c1, c2, c3};
}
__device__ __forceinline__ uint32_t pfo_u4_get(const pfo_u4& v, int i) {
  return i == 0 ? v.x : (i == 1 ? v.y : (i == 2 ? v.z : v.w));
}

__device__ __forceinline__ float pfo_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float pfo_wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
#endif  // __HIPCC__
