// Shared host/device helpers for the gfx950 kernels.  Wave size is 64 everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
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
bool pfo_prof_on();   // while on, the step keeps its large launches on ONE stream so that every bracket times its kernel alone
void pfo_prof_begin(hipStream_t s);
void pfo_prof_end(int kind, double work, hipStream_t s);
// work = work_per_unit * min(*units_dev, units_cap): launches whose extent is a device-side count (read back at collect time)
void pfo_prof_end_dev(int kind, double work_per_unit, const int32_t* units_dev, int units_cap, hipStream_t s);

// Events that ride on a kernel's own completion signal (round 4).  hipEventRecord queues a marker packet, and on this part the
// kernel behind a marker starts ~6 us late (consecutive kernels of one queue start 0.0-0.1 us apart; every record on the
// caller's stream showed as a bubble in the per-queue trace).  hipExtLaunchKernelGGL takes a stop event that is bound to the
// launch itself: no extra packet.  pfo_stop_event_arm(e, skip) makes the (skip+1)-th kernel launched next through PFO_KLAUNCH
// carry `e`; pfo_stop_event_disarm(stream) records it the plain way if that launch never came (count mismatch, capture).
void pfo_stop_event_arm(hipEvent_t e, int skip);
void pfo_stop_event_disarm(hipStream_t stream);
void pfo_stop_event_cancel();
bool pfo_stop_event_take(hipEvent_t* e);
// arm -> run -> disarm as ONE statement: the error return of `expr` cancels the armed event instead of leaving it (thread-local
// state) for the next unrelated launch of this thread
#define PFO_RUN_BOUND(enable, ev, skip, stream, expr)                                            \
  do {                                                                                           \
    const bool on__ = (enable);                                                                  \
    if (on__) pfo_stop_event_arm((ev), (skip));                                                  \
    const int rcb__ = (expr);                                                                    \
    if (on__) { if (rcb__ != PFO_OK) pfo_stop_event_cancel(); else pfo_stop_event_disarm(stream); } \
    if (rcb__ != PFO_OK) return rcb__;                                                           \
  } while (0)
#define PFO_KLAUNCH(kernel, grid, block, shmem, stream, ...)                                                          \
  do {                                                                                                                \
    hipEvent_t pe__ = nullptr;                                                                                        \
    if (pfo_stop_event_take(&pe__)) hipExtLaunchKernelGGL(kernel, grid, block, shmem, stream, nullptr, pe__, 0, __VA_ARGS__); \
    else hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__);                                         \
  } while (0)

// milestones on the caller's stream (include/pfotgn.h pfo_marks_*): no-op unless enabled
bool pfo_marks_on();
void pfo_mark_at(const char* name, hipStream_t s);
#define PFO_MARK(name, s) do { if (pfo_marks_on()) pfo_mark_at(name, s); } while (0)

// Named ranges for `rocprofv3 --marker-trace` (tracing hooks): roctxRangePushA / roctxRangePop are looked up in
// the process at first use (the profiler preloads librocprofiler-sdk-roctx.so; an application may link libroctx64 itself) -
// no link dependency, and a no-op pointer test when no tracer is present.
void pfo_range_push(const char* name);
void pfo_range_pop();
struct PfoRange {
  explicit PfoRange(const char* name) { pfo_range_push(name); }
  ~PfoRange() { pfo_range_pop(); }
  PfoRange(const PfoRange&) = delete;
  PfoRange& operator=(const PfoRange&) = delete;
};

// per-TU clock accumulators (attn.hip, gemm.hip), read by misc.hip's pfo_shader_clock
int pfo_attn_clock_read(double* cycles_ticks /* [2 kernels][2] */, int reset);
int pfo_gemm_clock_read(double* cycles_ticks /* [1 kernel][2] */, int reset);

static inline int64_t pfo_ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int64_t pfo_align_up(int64_t a, int64_t b) { return pfo_ceil_div(a, b) * b; }

#ifdef __HIPCC__
// ---------------------------------------------------------------------------------------------
// sin/cos of an fp32 argument of any magnitude (time-encoder arguments reach 1.7e7 on the synthetic
// graphs and beyond on epoch-second data).  The cosine is a pure function of the exactly rounded
// fp32 argument (SURVEY §7 hard part 1), so the range reduction must be accurate, not "fast-math".
// The reduction works in REVOLUTIONS, the unit of the hardware v_sin_f32 / v_cos_f32:
//   |x| < 2e7 : fp32 only.  p = fl(x * C_HI) with C_HI + C_LO = 1/(2 pi) to 48 bits; the rounding error of p is
//               recovered exactly by one FMA, p - rint(p) is exact (p < 2^22), so
//               u = (p - rint(p)) + (x * C_LO + err) is x/(2 pi) mod 1 in [-0.5, 0.5] to ~1e-8 revolutions.
//   otherwise : the same reduction in fp64, a few times slower, never taken on the benchmark.
// v_sin_f32 / v_cos_f32 on [-0.5, 0.5] revolutions: max |error| 1.25e-7 (tools/probes/hwcos.hip), the same as
// the minimax polynomials they replace at a third of the instructions (the attention kernels evaluate 172 of
// these per neighbour: it was their largest VALU item).
// slow path kept out of line so that it stays a real (almost never taken) branch instead of being if-converted
__device__ __attribute__((noinline)) float pfo_revolutions_f64(float x) {
  const double xd = (double)x;
  const double kd = rint(xd * 0.15915494309189533577);
  double r = fma(-kd, 6.283185307179586232, xd);            // x - k * 2 pi against a two-term 2 pi (k < 2^53 / 2^53: exact products)
  r = fma(-kd, 2.4492935982947064e-16, r);
  return (float)(r * 0.15915494309189533577);
}

__device__ __forceinline__ float pfo_revolutions_fast(float x) {
#pragma clang fp contract(off)   // p below is reused as a ROUNDED product: no implicit FMA formation in this function
  const float C_HI = 0.15915493667125702f, C_LO = 6.4206382432985265e-09f;
  const float p = x * C_HI;
  const float e = __builtin_fmaf(x, C_HI, -p);               // exact rounding error of p
  // v_fract_f32: p - floor(p), exact; the result lies in [0, 1) (+ the correction) instead of [-0.5, 0.5] - the hardware sine /
  // cosine take revolutions of either sign - and costs one instruction where p - rint(p) costs two (r3: -0.15 % per step, the
  // time-encoding accuracy tests unchanged)
  return __builtin_amdgcn_fractf(p) + __builtin_fmaf(x, C_LO, e);
}
// The out-of-range test is taken wave-wide (one scalar branch; a per-lane branch around the call costs a dozen scalar
// instructions per evaluation in the attention kernels): only a wavefront that holds such a lane runs the fp64 path.
__device__ __forceinline__ float pfo_revolutions(float x) {
  const bool big = !(fabsf(x) < 2.0e7f);
  float u = pfo_revolutions_fast(x);
  if (__builtin_expect(__ballot(big) != 0ull, 0)) {
    if (big) u = pfo_revolutions_f64(x);
  }
  return u;
}

__device__ __forceinline__ void pfo_sincosf(float x, float& s, float& c) {
  const float u = pfo_revolutions(x);
  s = __builtin_amdgcn_sinf(u);
  c = __builtin_amdgcn_cosf(u);
}

__device__ __forceinline__ float pfo_cosf(float x) { return __builtin_amdgcn_cosf(pfo_revolutions(x)); }

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

// Wave-wide sum delivered as a wave-uniform scalar: 6 DPP adds (quad xor 1/2, row_ror 4/8, row_bcast 15/31) and
// one v_readlane instead of 6 ds_bpermute round trips through the LDS crossbar.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float pfo_dpp_add(float v) {
  return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, false));
}
__device__ __forceinline__ float pfo_wave_sum_scalar(float v) {
  v = pfo_dpp_add<0xB1, 0xF>(v);     // quad_perm [1,0,3,2]
  v = pfo_dpp_add<0x4E, 0xF>(v);     // quad_perm [2,3,0,1]
  v = pfo_dpp_add<0x124, 0xF>(v);    // row_ror:4
  v = pfo_dpp_add<0x128, 0xF>(v);    // row_ror:8   -> every lane holds its 16-lane row sum
  v = pfo_dpp_add<0x142, 0xA>(v);    // row_bcast:15 into rows 1 and 3
  v = pfo_dpp_add<0x143, 0xC>(v);    // row_bcast:31 into rows 2 and 3 -> lane 63 holds the total
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
// N independent wave-wide sums with their DPP stages interleaved: a dependent DPP chain needs wait states between its
// steps (the compiler pads them with s_nop); stage by stage over N values there is always an independent instruction
// Halving form for N = 4 and N = 2 (gfx950 lane swaps): v_permlane32_swap exchanges the upper half of one register with the
// lower half of another, so ONE swap + ONE add folds TWO values across lanes i / i + 32, each surviving in one half of the
// wavefront; v_permlane16_swap does the same across rows of 16 - after two such steps every row holds the partials of one
// sum, and four row-local DPP steps finish all four sums at once: 3 swaps + 7 adds instead of 24 DPP adds.
__device__ __forceinline__ float pfo_swap_add32(float a, float b) {       // lanes 0-31: a[i] + a[i + 32], lanes 32-63: b[i - 32] + b[i]
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float pfo_swap_add16(float a, float b) {       // rows 0, 2: a's row pair sums, rows 1, 3: b's
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float pfo_row_sum(float v) {                    // every lane: the sum over its row of 16
  v = pfo_dpp_add<0xB1, 0xF>(v);
  v = pfo_dpp_add<0x4E, 0xF>(v);
  v = pfo_dpp_add<0x124, 0xF>(v);
  return pfo_dpp_add<0x128, 0xF>(v);
}
template <int N>
__device__ __forceinline__ void pfo_wave_sum_scalar_n(float (&v)[N]) {
  if constexpr (N == 4) {
    // after the two steps: row 0 holds sum 0, row 1 sum 2, row 2 sum 1, row 3 sum 3 (partials over 16 lanes each)
    const float x = pfo_row_sum(pfo_swap_add16(pfo_swap_add32(v[0], v[1]), pfo_swap_add32(v[2], v[3])));
    v[0] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 0));
    v[2] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 16));
    v[1] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 32));
    v[3] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 48));
    return;
  }
  if constexpr (N == 8) {
    // two rounds of the four-sum form (20 vector instructions against 56 for eight interleaved DPP trees): four heads x two keys
    float lo[4] = {v[0], v[1], v[2], v[3]}, hi[4] = {v[4], v[5], v[6], v[7]};
    pfo_wave_sum_scalar_n<4>(lo);
    pfo_wave_sum_scalar_n<4>(hi);
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[i] = lo[i]; v[4 + i] = hi[i]; }
    return;
  }
  if constexpr (N == 2) {
    // rows 0-1: sum 0, rows 2-3: sum 1; the row pairs are folded by a second swap of the register with itself
    const float y = pfo_swap_add32(v[0], v[1]);
    const float x = pfo_row_sum(pfo_swap_add16(y, y));
    v[0] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 0));
    v[1] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 32));
    return;
  }
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = pfo_dpp_add<0xB1, 0xF>(v[i]);
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = pfo_dpp_add<0x4E, 0xF>(v[i]);
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = pfo_dpp_add<0x124, 0xF>(v[i]);
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = pfo_dpp_add<0x128, 0xF>(v[i]);
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = pfo_dpp_add<0x142, 0xA>(v[i]);
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = pfo_dpp_add<0x143, 0xC>(v[i]);
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v[i]), 63));
}
// exp(x) for x <= 0 on the hardware exponential (v_exp_f32 = 2^x, ~1 ulp): softmax weights of wave-uniform scores.
// expf()'s full range reduction costs ~10 instructions per call, and the attention kernels make 4-8 calls per key pair.
__device__ __forceinline__ float pfo_exp_neg(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }

// wave-wide sum / maximum, the result in EVERY lane: two lane swaps fold the halves and the row pairs, four row-local DPP
// steps the rows - eight vector instructions, no trip through the LDS crossbar (six ds_bpermute round trips before)
template <int CTRL>
__device__ __forceinline__ float pfo_dpp_max(float v) {
  return fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false)));
}
__device__ __forceinline__ float pfo_wave_sum(float v) {
  v = pfo_swap_add32(v, v);
  v = pfo_swap_add16(v, v);
  return pfo_row_sum(v);
}
__device__ __forceinline__ float pfo_wave_max(float v) {
  const auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
  const auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
  v = pfo_dpp_max<0xB1>(v);
  v = pfo_dpp_max<0x4E>(v);
  v = pfo_dpp_max<0x124>(v);
  return pfo_dpp_max<0x128>(v);
}

// ---------------------------------------------------------------------------------------------
// Shader clock as the product kernels see it (include/pfotgn.h pfo_shader_clock): the first wavefront of workgroup 0 of a few
// large kernels reads the shader cycle counter (s_memtime) and the constant 100 MHz counter (s_memrealtime) at its start and its
// end and adds both differences to a per-kernel pair - two scalar reads and two atomics per LAUNCH.  cycles / (ticks * 10 ns)
// = the clock that wavefront ran at, under that kernel's own power draw.
struct PfoClockStamp { unsigned long long c0, r0; };
__device__ __forceinline__ PfoClockStamp pfo_clock_begin() {
  PfoClockStamp s;
  s.c0 = __builtin_amdgcn_s_memtime();
  s.r0 = __builtin_amdgcn_s_memrealtime();
  return s;
}
__device__ __forceinline__ void pfo_clock_end(const PfoClockStamp& s, unsigned long long* acc /* [2]: cycles, ticks */) {
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  atomicAdd(acc, c1 - s.c0);
  atomicAdd(acc + 1, r1 - s.r0);
}
#endif  // __HIPCC__
