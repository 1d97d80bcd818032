// Neighbour-tile temporal attention (K4/K5 neighbour side).
//
// One wavefront per instance; the 64 lanes stride the feature dimension so every neighbour row is
// read as contiguous 256-byte pieces (rows of the layer-0 table are L2/Infinity-Cache resident).
// Keys are never materialised: key_j = [gathered row | edge feature | cos(fma(dt, w, b))] is formed
// in registers, used for the score, folded into the running context by an online softmax, and
// recomputed in the backward (the reference's [N,K,C] key tensor is 1.4 GB at C2).
// The K/V projections are folded: the kernel consumes qk_h = Wk_h^T Q_h and emits
// ctx_h = sum_j a_jh key_j; both projections become plain GEMMs on [N, C] operands.
#include "attn.hpp"
#include <algorithm>
#include <type_traits>
#include <string.h>

struct AttnDev {
  int N, K, D, Ef, H, Cp;
  const float* QK; const int32_t* qk_row; int64_t qk_ld; const float* nbr_tab; int64_t nbr_ld; const int32_t* nbr_row; int64_t nbr_row_base; int nbr_relu;
  const int32_t* nbr_ids; const float* edge_feat; const int32_t* eidx; const float* dt; const float* tw; const float* tb;
  float scale, dropout_p; uint64_t seed, offset; const uint64_t* offset_dev; const uint8_t* keep_inject;
  float* ctx; float* attw; uint8_t* inv;
  const float* dctx; float* dQK; float* d_nbr; int64_t d_nbr_ld; double* dtime_part;
  int64_t d_nbr_rep;  // DMODE 1: floats between the per-XCD replicas of the gradient table (0: one table)
  int d_nbr_nrep;     // ... and how many of them are in use (a power of two)
  // run-merged layer-1 backward: instances in (table row, run key) order, seg_ptr[*n_rows] of them
  const int32_t* members; const int32_t* seg_ptr; const int32_t* n_rows; const int32_t* run_cnt;
  int det; double* dtime_slab;   // deterministic mode (attn.hpp)
  uint8_t* dqk_live;             // run-merged kernel: [members] 1 = dQK row m holds a sum, 0 = folded into a later row / nothing
  float* dq_rows; int64_t dq_ld; // run-merged kernel: per-table-row sums of dQK, added to atomically (attn.hpp)
  int xcd_g;    // run-merged backward: G consecutive chunks of members per XCD turn (0 = chunks round-robin over the XCDs)
  int abl;      // timing-only ablation switch (PFO_ATTN_ABL): 1 = spread the atomic destinations (wrong results)
};

// shader-clock pairs (common.hpp pfo_clock_*): 0 = attention forward (ring form), 1 = run-merged backward
__device__ unsigned long long g_attn_clock[2][2];
int pfo_attn_clock_read(double* out, int reset) {
  unsigned long long h[2][2];
  if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_attn_clock), sizeof(h)) != hipSuccess) return PFO_ERR_HIP;
  for (int i = 0; i < 2; ++i) { out[2 * i] = (double)h[i][0]; out[2 * i + 1] = (double)h[i][1]; }
  if (reset) { memset(h, 0, sizeof(h)); if (hipMemcpyToSymbol(HIP_SYMBOL(g_attn_clock), h, sizeof(h)) != hipSuccess) return PFO_ERR_HIP; }
  return PFO_OK;
}

// deterministic mode: a gradient element is added as 2^-40 fixed point - integer addition does not depend on the order
__device__ __forceinline__ void det_add(float* table, int64_t idx, float v) {
  atomicAdd(reinterpret_cast<unsigned long long*>(table) + idx, (unsigned long long)__double2ll_rn((double)v * PFO_DET_SCALE));
}

// dropout keep-bits for slot `lane` of instance n: bit h = keep for head h (H <= 4)
__device__ __forceinline__ unsigned attn_keep_bits(uint64_t seed, uint64_t offset, int64_t n, int lane, float p) {
  if (p <= 0.f) return 0xFu;
  const pfo_u4 r = pfo_philox(seed, (uint64_t)n * 64ull + (uint64_t)lane, offset);
  const uint32_t thr = (uint32_t)fminf(p * 4294967296.0f, 4294967040.0f);
  return (r.x >= thr ? 1u : 0u) | (r.y >= thr ? 2u : 0u) | (r.z >= thr ? 4u : 0u) | (r.w >= thr ? 8u : 0u);
}

// ... or the caller's injected decisions (parity tests): one wave-uniform test per instance
__device__ __forceinline__ unsigned attn_keep_for(const AttnDev& a, uint64_t rng_off, int64_t n, int lane) {
  if (a.keep_inject && a.dropout_p > 0.f) return lane < a.K ? (unsigned)a.keep_inject[n * a.K + lane] : 0xFu;
  return attn_keep_bits(a.seed, rng_off, n, lane, a.dropout_p);
}

// Keys are processed in chunks of KC: all gathers of a chunk are issued back to back (one memory latency per
// chunk instead of one per key), the chunk's KC*H dot-product butterflies are interleaved (6 dependent shuffle
// stages per chunk instead of per key), then the online-softmax state is advanced key by key.  Per-slot metadata
// (row, edge id, dt, validity) is loaded once, one slot per lane, and broadcast with v_readlane.
#ifndef KC_FWD
#define KC_FWD 2
#endif
#ifndef KC_BWD
#define KC_BWD 2
#endif

__device__ __forceinline__ int rl_i(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ float rl_f(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }

#ifndef FWD_WAVES
#define FWD_WAVES(NR, H) ((NR) * (H) <= 6 ? 5 : 2)      // registers per lane the compiler may use: 96 for D <= 192 with two heads
#endif
template <int NR, int H>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(FWD_WAVES(NR, H)))) void attn_fwd_kernel(const AttnDev a) {
  // the time-encoder parameters live in LDS, not in six registers per lane: with them the kernel fits 96 registers = 5
  // wavefronts per SIMD instead of 4
  __shared__ float s_tw[NR * 64], s_tb[NR * 64];
  for (int c = threadIdx.x; c < NR * 64; c += 256) {
    s_tw[c] = c < a.D ? a.tw[c] : 0.f;
    s_tb[c] = c < a.D ? a.tb[c] : 0.f;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t n = (int64_t)blockIdx.x * 4 + wave;     // one instance per wavefront (a grid-stride form was measured: the
  if (n >= a.N) return;                                 // loop-carried state cost 32 VGPRs = 2 waves/SIMD and 25 % of the time)
  const int D = a.D, Ef = a.Ef, K = a.K, C = 2 * D + Ef, Cp = a.Cp;   // Cp: per-head row stride (C + 2 extra columns, padded)

  float qn[H][NR], qt[H][NR], qe[H];
  const float* qk = a.QK + (a.qk_row ? (int64_t)a.qk_row[n] : n) * a.qk_ld;
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const int c = lane + 64 * r;
#pragma unroll
    for (int h = 0; h < H; ++h) {
      qn[h][r] = c < D ? qk[h * Cp + c] : 0.f;
      qt[h][r] = c < D ? qk[h * Cp + D + Ef + c] : 0.f;
    }
  }
#pragma unroll
  for (int h = 0; h < H; ++h) qe[h] = lane < Ef ? qk[h * Cp + D + lane] : 0.f;

  const int64_t slot0 = n * K;
  const bool inK = lane < K;
  const int my_id = inK ? a.nbr_ids[slot0 + lane] : 0;
  const int my_row = inK ? (a.nbr_row ? a.nbr_row[slot0 + lane] : (int)(a.nbr_row_base + slot0 + lane)) : 0;
  const int my_e = inK ? a.eidx[slot0 + lane] : 0;
  const float my_dt = inK ? a.dt[slot0 + lane] : 0.f;
  const unsigned long long valid = __ballot(inK && my_id != 0);
  float* ctx = a.ctx + n * H * Cp;
  if (valid == 0ull) {
    // no valid neighbour: the reference attends to padded slot 0 and then zero-fills the attention
    // output (temporal_attention.py:60-65,84), so nothing computed here is ever observed
    if (lane == 0) a.inv[n] = 1;
    for (int c = lane; c < H * Cp; c += 64) ctx[c] = 0.f;        // includes the Σa' and valid-flag columns
    for (int c = lane; c < H * K; c += 64) a.attw[n * H * K + c] = 0.f;
    return;
  }
  if (lane == 0) a.inv[n] = 0;

  const unsigned keep = attn_keep_for(a, a.offset + (a.offset_dev ? *a.offset_dev : 0ull), n, lane);
  const float keep_scale = a.dropout_p > 0.f ? 1.f / (1.f - a.dropout_p) : 1.f;

  float m[H], l[H], ld[H], my_s[H];
  float an[H][NR], at[H][NR], ae[H];
#pragma unroll
  for (int h = 0; h < H; ++h) {
    m[h] = -INFINITY; l[h] = 0.f; ld[h] = 0.f; my_s[h] = -INFINITY; ae[h] = 0.f;
#pragma unroll
    for (int r = 0; r < NR; ++r) { an[h][r] = 0.f; at[h][r] = 0.f; }
  }

  // Software pipeline over chunks of KC_FWD keys: the rows of chunk c+1 are gathered while chunk c is scored, so a
  // wavefront pays the gather latency once per instance, not once per chunk.  Two register sets used alternately
  // (A, B): copying the prefetched set over the current one cost 38 v_mov per iteration.
  unsigned long long vm = valid;
  int jsA[KC_FWD], jsB[KC_FWD];
  float knA[KC_FWD][NR], keA[KC_FWD], knB[KC_FWD][NR], keB[KC_FWD];
  auto pick = [&](int (&jj)[KC_FWD]) {
#pragma unroll
    for (int c = 0; c < KC_FWD; ++c) {
      jj[c] = vm ? (__ffsll((long long)vm) - 1) : -1;
      vm &= vm - 1ull;
    }
  };
  auto gather = [&](const int (&jj)[KC_FWD], float (&kk)[KC_FWD][NR], float (&ee)[KC_FWD]) {
#pragma unroll
    for (int c = 0; c < KC_FWD; ++c) {
      const int j = jj[c] < 0 ? 0 : jj[c];
      const float* src = a.nbr_tab + (uint32_t)rl_i(my_row, j) * (uint32_t)a.nbr_ld;        // 32-bit element offsets (checked on the host): one scalar multiply
      const int e = rl_i(my_e, j);
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        const int cc = lane + 64 * r;
        kk[c][r] = (jj[c] >= 0 && (r < NR - 1 || cc < D)) ? src[cc] : 0.f;
      }
      ee[c] = (jj[c] >= 0 && lane < Ef) ? a.edge_feat[(uint32_t)e * (uint32_t)Ef + (uint32_t)lane] : 0.f;
    }
  };
  auto process = [&](const int (&js)[KC_FWD], const float (&kn)[KC_FWD][NR], const float (&ke)[KC_FWD]) {
    float kt[KC_FWD][NR], dtv[KC_FWD], arg[KC_FWD][NR];
#pragma unroll
    for (int c = 0; c < KC_FWD; ++c) dtv[c] = rl_f(my_dt, js[c] < 0 ? 0 : js[c]);
    // time encoding of the chunk: ONE out-of-range test for all its arguments (wave-wide), then the fast reduction
    bool big = false;
#pragma unroll
    for (int c = 0; c < KC_FWD; ++c)
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        arg[c][r] = pfo_time_arg(dtv[c], s_tw[lane + 64 * r], s_tb[lane + 64 * r]);
        big = big || !(fabsf(arg[c][r]) < 2.0e7f);
      }
    const bool any_big = __ballot(big) != 0ull;
    float part[KC_FWD * H];
#pragma unroll
    for (int c = 0; c < KC_FWD; ++c) {
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        const int cc = lane + 64 * r;
        // lanes beyond D (last r only) evaluate a harmless cosine and are masked by a select, not a branch
        const float cv = __builtin_expect(any_big, 0) ? pfo_cosf(arg[c][r]) : __builtin_amdgcn_cosf(pfo_revolutions_fast(arg[c][r]));
        kt[c][r] = (js[c] >= 0 && (r < NR - 1 || cc < D)) ? cv : 0.f;
      }
#pragma unroll
      for (int h = 0; h < H; ++h) {
        float pp = ke[c] * qe[h];
#pragma unroll
        for (int r = 0; r < NR; ++r) pp = fmaf(kn[c][r], qn[h][r], fmaf(kt[c][r], qt[h][r], pp));
        part[c * H + h] = pp;
      }
    }
    pfo_wave_sum_scalar_n<KC_FWD * H>(part);
#pragma unroll
    for (int c = 0; c < KC_FWD; ++c) {
      if (js[c] < 0) continue;
      const unsigned kb = (unsigned)rl_i((int)keep, js[c]);
#pragma unroll
      for (int h = 0; h < H; ++h) {
        const float sc = part[c * H + h] * a.scale;
        if (lane == js[c]) my_s[h] = sc;
        // the score is wave-uniform: the running sums are rescaled only when the maximum actually moves (a scalar
        // branch, taken for the first few keys of a row), otherwise a key costs one FMA per accumulator
        if (sc > m[h]) {
          const float corr = pfo_exp_neg(m[h] - sc);
          l[h] *= corr; ld[h] *= corr; ae[h] *= corr;
#pragma unroll
          for (int r = 0; r < NR; ++r) { an[h][r] *= corr; at[h][r] *= corr; }
          m[h] = sc;
        }
        const float pr = pfo_exp_neg(sc - m[h]);
        const float pd = ((kb >> h) & 1u) ? pr * keep_scale : 0.f;
        l[h] += pr;
        ld[h] += pd;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
          an[h][r] = fmaf(pd, kn[c][r], an[h][r]);
          at[h][r] = fmaf(pd, kt[c][r], at[h][r]);
        }
        ae[h] = fmaf(pd, ke[c], ae[h]);
      }
    }
  };
  pick(jsA);
  gather(jsA, knA, keA);
  while (true) {
    pick(jsB);
    if (jsB[0] >= 0) gather(jsB, knB, keB);
    process(jsA, knA, keA);
    if (jsB[0] < 0) break;
    pick(jsA);
    if (jsA[0] >= 0) gather(jsA, knA, keA);
    process(jsB, knB, keB);
    if (jsA[0] < 0) break;
  }
#pragma unroll
  for (int h = 0; h < H; ++h) {
    const float il = 1.f / l[h];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const int c = lane + 64 * r;
      if (c < D) {
        ctx[h * Cp + c] = an[h][r] * il;
        ctx[h * Cp + D + Ef + c] = at[h][r] * il;
      }
    }
    if (lane < Ef) ctx[h * Cp + D + lane] = ae[h] * il;
    if (lane < K) a.attw[(n * H + h) * K + lane] = ((valid >> lane) & 1ull) ? pfo_exp_neg(my_s[h] - m[h]) * il : 0.f;
    // extra columns consumed by the merged value/out/fc1 projection: C = Σ_j a'_jh (multiplies the folded value
    // bias), C+1 = 1 on head 0 (multiplies the folded out_proj bias; absent on rows without a valid neighbour)
    if (lane < Cp - C) ctx[h * Cp + C + lane] = lane == 0 ? ld[h] * il : ((lane == 1 && h == 0) ? 1.f : 0.f);
  }
}

// ---------------------------------------------------------------------------------------------
// KEY RING (round 6).  The kernel above keeps two register sets of KC_FWD keys: with five wavefronts per SIMD that is ten gathered
// rows in flight per SIMD, and a wavefront waits out one L2 round trip per pair of keys (ten per instance: its ~13 us life is
// mostly that).  Here the gathered rows never pass through registers: ONE global_load_lds_dwordx4 per key moves [row | edge
// feature] (lanes < D/4 read the row, the next Ef/4 lanes the edge feature - the source address of an LDS-DMA load is per lane)
// into a wavefront-private ring of FWD_RING slots in LDS, FWD_RING - 2 keys ahead of the pair being scored; the pair is read
// back with three ds_read_b32 per key.  No registers are held by data in flight, so the depth is set by LDS, not by the
// register file.  Node and edge columns are one contiguous [0, D + Ef) vector on both sides (key slot and qk' row), the time
// half is formed in registers as before.  vmcnt counts the DMAs exactly: pairs are issued two at a time (an odd tail re-reads
// its last key), so the pair at the head of the ring has landed when at most 2 (pairs issued behind it) are outstanding.
#ifndef FWD_RING
#define FWD_RING 4      // keys per wavefront's ring (even; 8 and 6 measure the same: the depth is not what binds)
#endif
#ifndef FWD_RING_WAVES
#define FWD_RING_WAVES(NR, H) ((NR) * (H) <= 6 ? 6 : ((NR) * (H) <= 8 ? 4 : ((NR) * (H) <= 12 ? 3 : 2)))   // wavefronts per SIMD the register budget is cut for
#endif
#ifndef FWD_RING_MAX_NRH
#define FWD_RING_MAX_NRH 12   // column groups x heads the ring form takes (beyond: the register form)
#endif
#ifndef FWD_Q_DMA
#define FWD_Q_DMA 0    // the qk' row by LDS-DMA (1: 0.176 ms per step) or by register-bound loads in front of the ring's DMAs (0: 0.168)
#endif
template <int N> __device__ __forceinline__ void pfo_wait_vm() {
  static_assert(N >= 0 && N <= 48, "vmcnt");
  if constexpr (N == 48) asm volatile("s_waitcnt vmcnt(48)" ::: "memory");
  else if constexpr (N == 32) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
  else if constexpr (N == 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
  else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  else if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
  else if constexpr (N == 14) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
  else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else if constexpr (N == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
  else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
// wait until at most 2 * pairs_behind DMAs are outstanding (pairs_behind <= FWD_RING / 2 - 1, wave-uniform)
template <int P = FWD_RING / 2 - 1>
__device__ __forceinline__ void pfo_wait_pairs(int pairs_behind) {
  if constexpr (P <= 0) { pfo_wait_vm<0>(); }
  else {
    if (pairs_behind >= P) pfo_wait_vm<2 * P>();
    else pfo_wait_pairs<P - 1>(pairs_behind);
  }
}
// FWD_STAMPS (diagnostic build only, -DFWD_STAMPS=1): shader cycles a wavefront spends per section of attn_fwd_ring_kernel, summed
// over the launch into pfo_fwd_stamps: 0 whole wavefront, 1 prologue up to the barrier, 2 query row + first DMAs issued, 3 waiting
// for a pair's DMA, 4 LDS reads + time encoding + scores (up to the reduced scalars), 5 softmax + context update, 6 epilogue,
// 7 wavefronts, 8 pairs
#ifndef FWD_STAMPS
#define FWD_STAMPS 0
#endif
#if FWD_STAMPS
// (one private row per wavefront, summed on the host: atomics of every wavefront into one row serialise in one L2 line and
//  stretch exactly the sections that touch memory)
#define FWD_STAMP_ROWS 65536
__device__ unsigned long long pfo_fwd_stamps[FWD_STAMP_ROWS * 10];
extern "C" int pfo_attn_fwd_stamps(unsigned long long* out, int reset) {
  static unsigned long long host[FWD_STAMP_ROWS * 10];
  if (out) {
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(pfo_fwd_stamps), sizeof(host)) != hipSuccess) return PFO_ERR_HIP;
    for (int i = 0; i < 10; ++i) out[i] = 0;
    for (int r = 0; r < FWD_STAMP_ROWS; ++r) for (int i = 0; i < 10; ++i) out[i] += host[r * 10 + i];
  }
  if (reset) { memset(host, 0, sizeof(host)); if (hipMemcpyToSymbol(HIP_SYMBOL(pfo_fwd_stamps), host, sizeof(host)) != hipSuccess) return PFO_ERR_HIP; }
  return PFO_OK;
}
#define FWD_STAMP_BEGIN() unsigned long long fst_[10] = {}; const unsigned long long fst_begin = __builtin_amdgcn_s_memtime(); unsigned long long fst_last = fst_begin
#define FWD_STAMP(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); fst_[i] += t_ - fst_last; fst_last = t_; } while (0)
#define FWD_STAMP_COUNT(i) do { fst_[i] += 1; } while (0)
#define FWD_STAMP_END(row, nw) do { fst_[0] = __builtin_amdgcn_s_memtime() - fst_begin; fst_[7] = (nw); if (lane < 10) { unsigned long long v_ = 0; for (int i_ = 0; i_ < 10; ++i_) v_ = lane == i_ ? fst_[i_] : v_; pfo_fwd_stamps[((row) % FWD_STAMP_ROWS) * 10 + lane] += v_; } } while (0)
#else
#define FWD_STAMP_BEGIN() do {} while (0)
#define FWD_STAMP(i) do {} while (0)
#define FWD_STAMP_COUNT(i) do {} while (0)
#define FWD_STAMP_END(row, nw) do {} while (0)
#endif
template <int NR, int H>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(FWD_RING_WAVES(NR, H)))) void attn_fwd_ring_kernel(const AttnDev a) {
  constexpr int SLOTF = NR * 64, RP = FWD_RING / 2;      // floats per ring slot; pairs the ring holds
  static_assert(FWD_RING % 2 == 0 && FWD_RING >= 2, "ring");
  __shared__ float s_tw[NR * 64], s_tb[NR * 64];
  __shared__ __align__(16) float s_ring[4][FWD_RING][SLOTF];
  extern __shared__ __align__(16) unsigned char s_q[];   // [4 wavefronts][H Cp floats]: the instance's qk' row (LDS-DMA)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int D = a.D, Ef = a.Ef, K = a.K, DE = D + Ef, C = 2 * D + Ef, Cp = a.Cp;
  const uint32_t q_bytes = (uint32_t)(H * Cp) * 4u;
  FWD_STAMP_BEGIN();
  const bool clk_on = blockIdx.x == 0 && threadIdx.x < 64;       // (wave-uniform)
  PfoClockStamp clk;
  if (clk_on) clk = pfo_clock_begin();
  // Every first-level load of the wavefront leaves before the barrier - the time-encoder parameters for the workgroup's LDS
  // copy, this instance's slot metadata and its query row's index (clamped instance for the wavefronts behind N): ONE round
  // trip where the barrier used to separate two.
  const int64_t n = (int64_t)blockIdx.x * 4 + wave;
  const int64_t nc = n < a.N ? n : (int64_t)a.N - 1;
  const int64_t slot0 = nc * K;
  const bool inK = lane < K;
  // (clamped addresses, not predicated loads: a predicated load is its own basic block with the wait for its result at the
  //  end - the six loads then queue as six dependent round trips; straight-line, they leave together)
  const int64_t si = slot0 + min(lane, K - 1);
  const int tcol = min((int)threadIdx.x, D - 1);
  const float twr = a.tw[tcol], tbr = a.tb[tcol];
  const int id_r = a.nbr_ids[si];
  // (an optional array is read through a pointer select - any valid address when it is absent - and its value selected away
  //  behind the loads: a load under a pointer test is a basic block of its own, too)
  const int row_l = (a.nbr_row ? a.nbr_row : a.nbr_ids)[si];
  const int e_r = a.eidx[si];
  const float dt_r = a.dt[si];
  const int qrow_l = (a.qk_row ? a.qk_row : a.nbr_ids)[nc];
  const uint32_t* const offp = a.offset_dev ? reinterpret_cast<const uint32_t*>(a.offset_dev) : reinterpret_cast<const uint32_t*>(a.tw);
  const uint32_t off_lo = offp[0], off_hi = offp[1];
  const int row_r = a.nbr_row ? row_l : (int)(a.nbr_row_base + si);
  const int qrow = a.qk_row ? qrow_l : (int)nc;
  const uint64_t rng_off = a.offset + (a.offset_dev ? (((uint64_t)off_hi << 32) | off_lo) : 0ull);
  int my_id = inK ? id_r : 0, my_row = inK ? row_r : 0, my_e = inK ? e_r : 0;
  float my_dt = inK ? dt_r : 0.f;
  if (threadIdx.x < NR * 64) { const bool on = (int)threadIdx.x < D; s_tw[threadIdx.x] = on ? twr : 0.f; s_tb[threadIdx.x] = on ? tbr : 0.f; }
  // columns [D + Ef, 64 NR) of a slot are never written by a DMA (lanes behind the row's end are off): cleared once, they
  // read as zero and the dot products need no select
  float* const ring = &s_ring[wave][0][0];
  if (lane + 64 * (NR - 1) >= DE)
#pragma unroll
    for (int s = 0; s < FWD_RING; ++s) ring[s * SLOTF + lane + 64 * (NR - 1)] = 0.f;
  __syncthreads();
  if (n >= a.N) return;
  FWD_STAMP(1);
  // one range test per instance (as in the run-merged backward): |fma(dt, w, b)| <= max|dt| max|w| + max|b| < 2e7 -> the fp32
  // range reduction holds for every key of the instance
  float wmax = 0.f, bmax = 0.f;
#pragma unroll
  for (int r = 0; r < NR; ++r) { wmax = fmaxf(wmax, fabsf(s_tw[lane + 64 * r])); bmax = fmaxf(bmax, fabsf(s_tb[lane + 64 * r])); }
  wmax = pfo_wave_max(wmax); bmax = pfo_wave_max(bmax);

  const unsigned long long valid = __ballot(inK && my_id != 0);
  float* ctx = a.ctx + n * H * Cp;
  if (valid == 0ull) {
    if (lane == 0) a.inv[n] = 1;
    for (int c = lane; c < H * Cp; c += 64) ctx[c] = 0.f;
    for (int c = lane; c < H * K; c += 64) a.attw[n * H * K + c] = 0.f;
    return;
  }
  const bool fast = fmaf(pfo_wave_max(fabsf(my_dt)), wmax, bmax) < 2.0e7f;     // (every register-bound load is consumed before the first DMA)
  // ---- round trip 2, all of it LDS-DMA: the instance's qk' row (no register-bound load is in flight beside the ring: the
  // compiler waits for vmcnt(0) at the first use of such a load, i.e. for every DMA behind it) and the first pairs of keys.
  // Pair p = valid keys 2p, 2p + 1 in slot order; slots (2p) % RING, (2p + 1) % RING.
  typedef __attribute__((address_space(1))) const void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  unsigned char* const qb = s_q + (size_t)wave * q_bytes;
  // (every value that came from a register-bound load is pinned HERE: sunk behind the first DMA, its use would wait for vmcnt(0))
  asm volatile("" : "+v"(my_row), "+v"(my_e), "+v"(my_dt), "+v"(my_id));
#if FWD_Q_DMA
  {
    const char* src = reinterpret_cast<const char*>(a.QK + (int64_t)qrow * a.qk_ld);
    const uint32_t lo = (uint32_t)lane * 16u;
    for (uint32_t k = 0; k * 1024u < q_bytes; ++k)
      if (k * 1024u + lo < q_bytes) __builtin_amdgcn_global_load_lds((gptr_t)(src + k * 1024u + lo), (lptr_t)(qb + k * 1024u), 16, 0, 0);
  }
#else
  // (register-bound loads of the qk' row in front of the DMAs, clamped columns: the compiler's vmcnt(0) at their first use -
  //  the top of the walk - is the wait for the first pair anyway)
  float r1[H][NR], rt[H][NR];
  {
    const float* qk = a.QK + (int64_t)qrow * a.qk_ld;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const int c1 = min(lane + 64 * r, DE - 1), ct = min(lane + 64 * r, D - 1);
#pragma unroll
      for (int h = 0; h < H; ++h) { r1[h][r] = qk[h * Cp + c1]; rt[h][r] = qk[h * Cp + DE + ct]; }
    }
  }
#endif
  const int D4 = D >> 2, DE4 = DE >> 2;
  // per lane: the table it reads (row table or edge features), its row stride in bytes and its 16-byte piece of the row; the
  // per-key address is then ONE 64-bit multiply-add (base + index * stride) on a select of two wave-uniform indices
  const bool is_node = lane < D4;
  const uint64_t base_l = (is_node ? (uint64_t)(uintptr_t)a.nbr_tab : (uint64_t)(uintptr_t)a.edge_feat) + (uint64_t)((is_node ? lane : lane - D4) * 16);
  const uint32_t mul_l = is_node ? (uint32_t)a.nbr_ld * 4u : (uint32_t)Ef * 4u;
  const uint32_t node_mask = is_node ? 0xFFFFFFFFu : 0u;        // index select without a branch: e + ((row - e) & mask)
  unsigned long long vm = valid;                               // keys not yet issued
  const int nv = __popcll(valid);
  const int n_pairs = (nv + 1) >> 1;
  int issued = 0;                                              // pairs issued
  auto issue_pair = [&]() {
    int j0 = __ffsll((long long)vm) - 1;
    vm &= vm - 1ull;
    int j1 = vm ? (__ffsll((long long)vm) - 1) : j0;           // an odd tail re-reads its last key (its weight is forced to zero)
    vm &= vm - 1ull;
    const int sl = (2 * issued) % FWD_RING;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int j = c ? j1 : j0;
      const uint32_t e_s = (uint32_t)rl_i(my_e, j);
      const uint32_t idx = e_s + (((uint32_t)rl_i(my_row, j) - e_s) & node_mask);
      const uint64_t src = base_l + (uint64_t)idx * (uint64_t)mul_l;
      if (lane < DE4) __builtin_amdgcn_global_load_lds((gptr_t)(uintptr_t)src, (lptr_t)(ring + (sl + c) * SLOTF), 16, 0, 0);
    }
    issued += 1;
  };
#pragma unroll
  for (int p = 0; p < RP; ++p)
    if (p < n_pairs) issue_pair();
  FWD_STAMP(2);
  const unsigned keep = attn_keep_for(a, rng_off, n, lane);
  const float keep_scale = a.dropout_p > 0.f ? 1.f / (1.f - a.dropout_p) : 1.f;
  // the first pair has landed, and with it (older) the qk' row: [node | edge] columns as one vector, the time columns as
  // another, both pre-multiplied by scale * log2(e) - a score then leaves the reduction ready for v_exp_f32 (2^x)
  pfo_wait_pairs(issued - 1);
  float q1[H][NR], qt[H][NR];
  {
    const float* qk = reinterpret_cast<const float*>(qb);
    const float qs = a.scale * 1.44269504088896340736f;
#if FWD_Q_DMA
    // (clamped columns + a select behind the reads: predicated, each of the 4 NR H reads waited for its own LDS round trip)
    float r1[H][NR], rt[H][NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const int c1 = min(lane + 64 * r, DE - 1), ct = min(lane + 64 * r, D - 1);
#pragma unroll
      for (int h = 0; h < H; ++h) { r1[h][r] = qk[h * Cp + c1]; rt[h][r] = qk[h * Cp + DE + ct]; }
    }
#else
    (void)qk;
#endif
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const int c = lane + 64 * r;
#pragma unroll
      for (int h = 0; h < H; ++h) {
        q1[h][r] = c < DE ? r1[h][r] * qs : 0.f;
        qt[h][r] = c < D ? rt[h][r] * qs : 0.f;
      }
    }
  }

  float m[H], l[H], ld[H], my_s[H];                            // m, my_s: scores in log2 units
  float a1[H][NR], at[H][NR];
#pragma unroll
  for (int h = 0; h < H; ++h) {
    m[h] = -INFINITY; l[h] = 0.f; ld[h] = 0.f; my_s[h] = -INFINITY;
#pragma unroll
    for (int r = 0; r < NR; ++r) { a1[h][r] = 0.f; at[h][r] = 0.f; }
  }
  auto walk = [&](auto fast_c) {
  constexpr bool FAST = decltype(fast_c)::value;
  unsigned long long wm = valid;                               // keys not yet scored
  for (int p = 0; p < n_pairs; ++p) {
    const int js0 = __ffsll((long long)wm) - 1;
    wm &= wm - 1ull;
    const bool has1 = wm != 0ull;
    const int js1 = has1 ? (__ffsll((long long)wm) - 1) : js0;
    wm &= wm - 1ull;
    const bool more = p + RP < n_pairs;                        // the pair RP ahead exists: RP - 1 pairs stay in flight behind this one
    FWD_STAMP_COUNT(8);
    if (more) pfo_wait_vm<2 * (RP - 1)>(); else pfo_wait_vm<0>();
    FWD_STAMP(3);
    const float* sl = ring + ((2 * p) % FWD_RING) * SLOTF;
    float kn[2][NR], kt[2][NR], dtv[2];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int r = 0; r < NR; ++r) kn[c][r] = sl[c * SLOTF + lane + 64 * r];
    dtv[0] = rl_f(my_dt, js0); dtv[1] = rl_f(my_dt, js1);
    float part[2 * H];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        // (lanes beyond D carry w = b = 0: cos(0) = 1 there, against a zero query column - no select)
        const float arg = pfo_time_arg(dtv[c], s_tw[lane + 64 * r], s_tb[lane + 64 * r]);
        kt[c][r] = __builtin_amdgcn_cosf(FAST ? pfo_revolutions_fast(arg) : pfo_revolutions(arg));
      }
#pragma unroll
      for (int h = 0; h < H; ++h) {
        float pp = 0.f;
#pragma unroll
        for (int r = 0; r < NR; ++r) pp = fmaf(kn[c][r], q1[h][r], fmaf(kt[c][r], qt[h][r], pp));
        part[c * H + h] = pp;
      }
    }
    // the slots are free again (their values sit in registers): the pair RP ahead starts its trip
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (more) issue_pair();
    pfo_wave_sum_scalar_n<2 * H>(part);
    FWD_STAMP(4);
    // an absent second key (odd tail) scores -inf: weight 0 in every sum, no branch
    float sc[2][H];
#pragma unroll
    for (int h = 0; h < H; ++h) { sc[0][h] = part[h]; sc[1][h] = has1 ? part[H + h] : -INFINITY; }
    // The scores are wave-uniform: the running sums are rescaled only when a maximum actually moves - ONE scalar branch per
    // pair (taken for the first pairs of a row), all heads inside it
    bool up = false;
#pragma unroll
    for (int h = 0; h < H; ++h) up = up || sc[0][h] > m[h] || sc[1][h] > m[h];
    if (up) {
#pragma unroll
      for (int h = 0; h < H; ++h) {
        const float mn = fmaxf(m[h], fmaxf(sc[0][h], sc[1][h]));
        const float corr = __builtin_amdgcn_exp2f(m[h] - mn);     // (first pair: 2^-inf = 0 against zero sums)
        l[h] *= corr; ld[h] *= corr;
#pragma unroll
        for (int r = 0; r < NR; ++r) { a1[h][r] *= corr; at[h][r] *= corr; }
        m[h] = mn;
      }
    }
    const unsigned kb0 = (unsigned)rl_i((int)keep, js0), kb1 = (unsigned)rl_i((int)keep, js1);
    const int js1c = has1 ? js1 : -1;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const unsigned kb = c ? kb1 : kb0;
#pragma unroll
      for (int h = 0; h < H; ++h) {
        if (lane == (c ? js1c : js0)) my_s[h] = sc[c][h];
        const float pr = __builtin_amdgcn_exp2f(sc[c][h] - m[h]);
        const float pd = ((kb >> h) & 1u) ? pr * keep_scale : 0.f;
        l[h] += pr;
        ld[h] += pd;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
          a1[h][r] = fmaf(pd, kn[c][r], a1[h][r]);
          at[h][r] = fmaf(pd, kt[c][r], at[h][r]);
        }
      }
    }
#if FWD_STAMPS
    asm volatile("s_nop 0" :: "v"(a1[0][0]), "v"(at[H - 1][NR - 1]) : "memory");
#endif
    FWD_STAMP(5);
  }
  };
  if (fast) walk(std::true_type{}); else walk(std::false_type{});
  if (lane == 0) a.inv[n] = 0;       // (here, not in front of the DMAs: the compiler drains a store before the first LDS-DMA load)
#pragma unroll
  for (int h = 0; h < H; ++h) {
    const float il = 1.f / l[h];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const int c = lane + 64 * r;
      if (c < DE) ctx[h * Cp + c] = a1[h][r] * il;
      if (c < D) ctx[h * Cp + DE + c] = at[h][r] * il;
    }
    if (lane < K) a.attw[(n * H + h) * K + lane] = ((valid >> lane) & 1ull) ? __builtin_amdgcn_exp2f(my_s[h] - m[h]) * il : 0.f;
    if (lane < Cp - C) ctx[h * Cp + C + lane] = lane == 0 ? ld[h] * il : ((lane == 1 && h == 0) ? 1.f : 0.f);
  }
  FWD_STAMP(6);
  FWD_STAMP_END(n, 1);
  if (clk_on && lane == 0) pfo_clock_end(clk, g_attn_clock[0]);
}
// ---------------------------------------------------------------------------------------------
// INSTANCE PIPELINE (round 6).  In-kernel stamps of the ring form (tools/probes/fwd_stamps.py, profiles/r6_fwd_stamps.txt) show
// where a wavefront's life goes: 14 % waiting for its first-level loads (slot metadata, query row index), 24 % for the query row
// behind them, 14 % for the first pair of keys behind THAT, 40 % scoring, 8 % storing - three dependent round trips per instance
// in front of ~12 k cycles of arithmetic, with every wavefront of a SIMD in the same phases.  Here a wavefront takes FWD_IPW
// consecutive instances and nothing it needs is loaded into registers: the metadata of all its instances arrives by LDS-DMA in
// ONE round trip at the start; query rows and key pairs then flow through LDS in consumption order - the query row of instance
// i + 1 and its first key pairs are requested while instance i is scored (the key ring is one FIFO across instances).  Every
// later round trip hides behind the arithmetic of the instance before.  vmcnt is kept exact by a software sequence counter:
// `seq` counts the vector-memory instructions issued so far (DMAs and the epilogue's stores, each statement one instruction
// that always has active lanes), every FIFO unit remembers seq at its last DMA, and the consumer waits for
// vmcnt(seq - unit's seq) - the operations issued behind the unit may stay in flight.  Counting too few younger operations only
// makes a wait stricter; nothing is counted that might not be issued (the zero-neighbour path's stores are not).
#ifndef FWD_IPW
#define FWD_IPW 4       // instances per wavefront
#endif
#ifndef FWD_PIPE_RING
#define FWD_PIPE_RING 4 // key slots per wavefront (even)
#endif
#ifndef FWD_PIPE_WAVES
#define FWD_PIPE_WAVES 5
#endif
// ... with the counts the run-merged backward meets exact at its benchmark shapes (one pair's gathers: 8 or 6, the staging
// DMAs: 15 / 16, both: 23 / 24) and the nearest lower count otherwise - waiting for fewer outstanding operations is always safe
__device__ __forceinline__ void pfo_wait_allowed_exact(int allowed) {
#define PFO_WVM(n) if (allowed >= n) { asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory"); return; }
  PFO_WVM(24) PFO_WVM(23) PFO_WVM(22) PFO_WVM(21) PFO_WVM(16) PFO_WVM(15) PFO_WVM(14) PFO_WVM(13) PFO_WVM(8) PFO_WVM(6) PFO_WVM(4) PFO_WVM(2)
#undef PFO_WVM
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
__device__ __forceinline__ void pfo_wait_allowed(int allowed) {   // wave-uniform: wait until at most `allowed` vector-memory operations are outstanding
  if (allowed >= 48) pfo_wait_vm<48>();
  else if (allowed >= 32) pfo_wait_vm<32>();
  else if (allowed >= 24) pfo_wait_vm<24>();
  else if (allowed >= 16) pfo_wait_vm<16>();
  else if (allowed >= 12) pfo_wait_vm<12>();
  else if (allowed >= 8) pfo_wait_vm<8>();
  else if (allowed >= 6) pfo_wait_vm<6>();
  else if (allowed >= 4) pfo_wait_vm<4>();
  else if (allowed >= 3) pfo_wait_vm<3>();
  else if (allowed >= 2) pfo_wait_vm<2>();
  else if (allowed >= 1) pfo_wait_vm<1>();
  else pfo_wait_vm<0>();
}
static size_t attn_fwd_pipe_wave_bytes(int NR, int H, int Cp, int K) {
  return (size_t)FWD_PIPE_RING * NR * 64 * 4 + (size_t)H * Cp * 4 + (size_t)pfo_align_up(5 * FWD_IPW * K * 4, 16);
}
template <int NR, int H>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(FWD_PIPE_WAVES))) void attn_fwd_pipe_kernel(const AttnDev a) {
  constexpr int SLOTF = NR * 64, R = FWD_PIPE_RING, P = FWD_IPW, RP = R / 2;
  static_assert(R % 2 == 0 && R >= 2, "ring");
  __shared__ float s_tw[NR * 64], s_tb[NR * 64];
  __shared__ float s_sc[4][H][64];                       // per wavefront: the raw scores of the instance being walked, by slot
  extern __shared__ __align__(16) unsigned char s_dyn[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int D = a.D, Ef = a.Ef, K = a.K, DE = D + Ef, C = 2 * D + Ef, Cp = a.Cp, PK = P * K;
  const uint32_t q_bytes = (uint32_t)(H * Cp) * 4u;
  const uint32_t meta_bytes = (uint32_t)((5 * PK * 4 + 15) & ~15);
  unsigned char* const wbase = s_dyn + (size_t)wave * ((size_t)R * SLOTF * 4 + q_bytes + meta_bytes);
  float* const ring = reinterpret_cast<float*>(wbase);
  const float* const qbuf = reinterpret_cast<const float*>(wbase + R * SLOTF * 4);
  int* const meta = reinterpret_cast<int*>(wbase + R * SLOTF * 4 + q_bytes);     // [ids | rows | edge ids | dt | query rows][P K]
  typedef __attribute__((address_space(1))) const void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  const int64_t n0 = ((int64_t)blockIdx.x * 4 + wave) * P;
  const int cnt = (int)min((int64_t)P, (int64_t)a.N - n0);                        // this wavefront's instances (<= 0: none)
  FWD_STAMP_BEGIN();
  // ---- round trip 1: the metadata of all the wavefront's instances, one dword per lane and instruction
  if (cnt > 0) {
    const int live = cnt * K;
    const int64_t s0 = n0 * K;
    for (int k = 0; k * 64 < live; ++k) {
      const int i = k * 64 + lane;
      if (i < live) {
        __builtin_amdgcn_global_load_lds((gptr_t)(a.nbr_ids + s0 + i), (lptr_t)(meta + 0 * PK + k * 64), 4, 0, 0);
        if (a.nbr_row) __builtin_amdgcn_global_load_lds((gptr_t)(a.nbr_row + s0 + i), (lptr_t)(meta + 1 * PK + k * 64), 4, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)(a.eidx + s0 + i), (lptr_t)(meta + 2 * PK + k * 64), 4, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)(a.dt + s0 + i), (lptr_t)(meta + 3 * PK + k * 64), 4, 0, 0);
      }
    }
    if (a.qk_row && lane < cnt) __builtin_amdgcn_global_load_lds((gptr_t)(a.qk_row + n0 + lane), (lptr_t)(meta + 4 * PK), 4, 0, 0);
  }
  {
    float twv = 0.f, tbv = 0.f;
    if (threadIdx.x < NR * 64 && (int)threadIdx.x < D) { twv = a.tw[threadIdx.x]; tbv = a.tb[threadIdx.x]; }
    if (threadIdx.x < NR * 64) { s_tw[threadIdx.x] = twv; s_tb[threadIdx.x] = tbv; }
  }
  if (lane + 64 * (NR - 1) >= DE)
#pragma unroll
    for (int s = 0; s < R; ++s) ring[s * SLOTF + lane + 64 * (NR - 1)] = 0.f;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (cnt <= 0) return;
  float wmax = 0.f, bmax = 0.f;
#pragma unroll
  for (int r = 0; r < NR; ++r) { wmax = fmaxf(wmax, fabsf(s_tw[lane + 64 * r])); bmax = fmaxf(bmax, fabsf(s_tb[lane + 64 * r])); }
  // (wave-uniform bounds, kept as scalars)
  const float wmax_s = rl_f(pfo_wave_max(wmax), 0), bmax_s = rl_f(pfo_wave_max(bmax), 0);
  const int D4 = D >> 2, DE4 = DE >> 2;
  const uint64_t rng_off = a.offset + (a.offset_dev ? *a.offset_dev : 0ull);
  const float keep_scale = a.dropout_p > 0.f ? 1.f / (1.f - a.dropout_p) : 1.f;

  int seq = 0;                        // vector-memory instructions issued since the barrier (those that are counted)
  int q_seq = 0;                      // seq at the last DMA of the query row in flight
  int ring_seq = 0;                   // lane s: seq at the last DMA of the pair in ring position s
  int pairs_issued = 0, pairs_consumed = 0;
  // ---- the issue cursor: instance iss_i, its slots' table rows / edge ids, the keys not yet requested
  int iss_i = 0, is_row = 0, is_e = 0;
  unsigned long long is_vm = 0ull;
  auto load_issue_meta = [&](int i) {
    int ln = lane;
    asm volatile("" : "+v"(ln));      // (opaque: no per-lane LDS address of this rarely executed block is kept across the walk)
    const bool ok = ln < K;
    const int id = ok ? meta[0 * PK + i * K + ln] : 0;
    is_row = ok ? (a.nbr_row ? meta[1 * PK + i * K + ln] : (int)(a.nbr_row_base + (n0 + i) * K + ln)) : 0;
    is_e = ok ? meta[2 * PK + i * K + ln] : 0;
    is_vm = __ballot(ok && id != 0);
  };
  auto issue_q = [&](int i) {
    const int64_t qrow = a.qk_row ? (int64_t)__builtin_amdgcn_readfirstlane(meta[4 * PK + i]) : n0 + i;
    const char* src = reinterpret_cast<const char*>(a.QK + qrow * a.qk_ld);
    uint32_t lo = (uint32_t)lane * 16u;
    asm volatile("" : "+v"(lo));
    for (uint32_t k = 0; k * 1024u < q_bytes; ++k) {
      if (k * 1024u + lo < q_bytes) __builtin_amdgcn_global_load_lds((gptr_t)(src + k * 1024u + lo), (lptr_t)(wbase + R * SLOTF * 4 + k * 1024u), 16, 0, 0);
      seq += 1;
    }
    q_seq = seq;
  };
  auto try_issue = [&]() {
    if (pairs_issued - pairs_consumed >= RP) return;
    while (is_vm == 0ull && iss_i + 1 < cnt) { iss_i += 1; load_issue_meta(iss_i); }
    if (is_vm == 0ull) return;
    const int j0 = __ffsll((long long)is_vm) - 1;
    is_vm &= is_vm - 1ull;
    const int j1 = is_vm ? (__ffsll((long long)is_vm) - 1) : j0;     // an odd tail re-reads its last key (its score is never used)
    is_vm &= is_vm - 1ull;
    const int pos = pairs_issued % RP;
    // per lane: the table it reads (rows: lanes < D / 4, edge features: the next Ef / 4), its stride and its 16-byte piece -
    // recomputed per pair from the lane id (a handful of vector instructions) instead of held across the walk
    int ln = lane;
    asm volatile("" : "+v"(ln));
    const bool is_node = ln < D4;
    const uint64_t base_l = (is_node ? (uint64_t)(uintptr_t)a.nbr_tab : (uint64_t)(uintptr_t)a.edge_feat) + (uint64_t)((is_node ? ln : ln - D4) * 16);
    const uint32_t mul_l = is_node ? (uint32_t)a.nbr_ld * 4u : (uint32_t)Ef * 4u;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int j = c ? j1 : j0;
      const uint32_t e_s = (uint32_t)rl_i(is_e, j), r_s = (uint32_t)rl_i(is_row, j);
      const uint32_t idx = is_node ? r_s : e_s;
      const uint64_t src = base_l + (uint64_t)idx * (uint64_t)mul_l;
      if (ln < DE4) __builtin_amdgcn_global_load_lds((gptr_t)(uintptr_t)src, (lptr_t)(ring + (2 * pos + c) * SLOTF), 16, 0, 0);
    }
    seq += 2;
    ring_seq = ln == pos ? seq : ring_seq;
    pairs_issued += 1;
  };
  load_issue_meta(0);
  issue_q(0);
#pragma unroll 1
  for (int p = 0; p < RP; ++p) try_issue();
  FWD_STAMP(1);

#pragma unroll 1
  for (int ci = 0; ci < cnt; ++ci) {
    const int64_t n = n0 + ci;
    int li = lane;
    asm volatile("" : "+v"(li));                                 // (per-instance addresses are formed per instance, not hoisted)
    const bool inK = li < K;
    const int c_id = inK ? meta[0 * PK + ci * K + li] : 0;
    const float my_dt = inK ? __int_as_float(meta[3 * PK + ci * K + li]) : 0.f;
    const unsigned long long valid = __ballot(inK && c_id != 0);
    float* ctx = a.ctx + n * H * Cp;
    if (valid == 0ull) {
      // no valid neighbour (temporal_attention.py:60-65,84): zero rows; the query row in flight is dropped, the next one follows it
      if (li == 0) a.inv[n] = 1;
      for (int c = li; c < H * Cp; c += 64) ctx[c] = 0.f;
      for (int c = li; c < H * K; c += 64) a.attw[n * H * K + c] = 0.f;
      if (ci + 1 < cnt) issue_q(ci + 1);
      continue;
    }
    // ---- this instance's qk' row: landed behind everything issued before it; the buffer then takes the next instance's
    pfo_wait_allowed(seq - q_seq);
    float q1[H][NR], qt[H][NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const int c = li + 64 * r;
#pragma unroll
      for (int h = 0; h < H; ++h) {
        q1[h][r] = c < DE ? qbuf[h * Cp + c] : 0.f;
        qt[h][r] = c < D ? qbuf[h * Cp + DE + c] : 0.f;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (ci + 1 < cnt) issue_q(ci + 1);
    FWD_STAMP(2);
    const unsigned keep = attn_keep_for(a, rng_off, n, li);
    float m[H], l[H], ld[H];
    float a1[H][NR], at[H][NR];
#pragma unroll
    for (int h = 0; h < H; ++h) {
      m[h] = -INFINITY; l[h] = 0.f; ld[h] = 0.f;
#pragma unroll
      for (int r = 0; r < NR; ++r) { a1[h][r] = 0.f; at[h][r] = 0.f; }
    }
    const int n_pairs = (__popcll(valid) + 1) >> 1;
    const bool fast = fmaf(rl_f(pfo_wave_max(fabsf(my_dt)), 0), wmax_s, bmax_s) < 2.0e7f;
    unsigned long long wm = valid;
#pragma unroll 1
    for (int p = 0; p < n_pairs; ++p) {
      int js[2];
      js[0] = __ffsll((long long)wm) - 1;
      wm &= wm - 1ull;
      js[1] = wm ? (__ffsll((long long)wm) - 1) : -1;
      wm &= wm - 1ull;
      const int pos = pairs_consumed % RP;
      FWD_STAMP_COUNT(8);
      pfo_wait_allowed(seq - rl_i(ring_seq, pos));             // the pair at the head of the FIFO has landed
      FWD_STAMP(3);
      int lp = li;
      asm volatile("" : "+v"(lp));
      const float* sl = ring + (2 * pos) * SLOTF + lp;
      float kn[2][NR], kt[2][NR], dtv[2], arg[2][NR];
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int r = 0; r < NR; ++r) kn[c][r] = sl[c * SLOTF + 64 * r];
#pragma unroll
      for (int c = 0; c < 2; ++c) dtv[c] = rl_f(my_dt, js[c] < 0 ? js[0] : js[c]);
      float part[2 * H];
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int r = 0; r < NR; ++r) {
          arg[c][r] = pfo_time_arg(dtv[c], s_tw[lp + 64 * r], s_tb[lp + 64 * r]);
          kt[c][r] = pfo_revolutions_fast(arg[c][r]);
        }
      // the two slots are free (their values sit in registers): the FIFO takes its next pair - of this instance or the next
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      pairs_consumed += 1;
      try_issue();
      if (__builtin_expect(!fast, 0)) {
        // some argument of this instance may leave the range of the fp32 reduction: those lanes take the fp64 one
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int r = 0; r < NR; ++r)
            if (!(fabsf(arg[c][r]) < 2.0e7f)) kt[c][r] = pfo_revolutions_f64(arg[c][r]);
      }
#pragma unroll
      for (int c = 0; c < 2; ++c) {
#pragma unroll
        for (int r = 0; r < NR; ++r) kt[c][r] = __builtin_amdgcn_cosf(kt[c][r]);   // (lanes beyond D: w = b = 0, cos(0) against a zero query column)
#pragma unroll
        for (int h = 0; h < H; ++h) {
          float pp = 0.f;
#pragma unroll
          for (int r = 0; r < NR; ++r) pp = fmaf(kn[c][r], q1[h][r], fmaf(kt[c][r], qt[h][r], pp));
          part[c * H + h] = pp;
        }
      }
      pfo_wave_sum_scalar_n<2 * H>(part);
      FWD_STAMP(4);
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        if (js[c] < 0) continue;
        const unsigned kb = (unsigned)rl_i((int)keep, js[c]);
#pragma unroll
        for (int h = 0; h < H; ++h) {
          const float sc = part[c * H + h] * a.scale;
          if (lp == 0) s_sc[wave][h][js[c]] = sc;
          if (sc > m[h]) {
            const float corr = pfo_exp_neg(m[h] - sc);
            l[h] *= corr; ld[h] *= corr;
#pragma unroll
            for (int r = 0; r < NR; ++r) { a1[h][r] *= corr; at[h][r] *= corr; }
            m[h] = sc;
          }
          const float pr = pfo_exp_neg(sc - m[h]);
          const float pd = ((kb >> h) & 1u) ? pr * keep_scale : 0.f;
          l[h] += pr;
          ld[h] += pd;
#pragma unroll
          for (int r = 0; r < NR; ++r) {
            a1[h][r] = fmaf(pd, kn[c][r], a1[h][r]);
            at[h][r] = fmaf(pd, kt[c][r], at[h][r]);
          }
        }
      }
#if FWD_STAMPS
      asm volatile("s_nop 0" :: "v"(a1[0][0]), "v"(at[H - 1][NR - 1]) : "memory");
#endif
      FWD_STAMP(5);
    }
    // ---- the instance's rows: H (2 NR + 2) + 1 store instructions, every one with active lanes (64 (NR - 1) < D, K >= 1, Cp - C >= 2)
    if (li == 0) a.inv[n] = 0;
#pragma unroll
    for (int h = 0; h < H; ++h) {
      const float il = 1.f / l[h];
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        const int c = li + 64 * r;
        if (c < DE) ctx[h * Cp + c] = a1[h][r] * il;
        if (c < D) ctx[h * Cp + DE + c] = at[h][r] * il;
      }
      if (li < K) a.attw[(n * H + h) * K + li] = ((valid >> li) & 1ull) ? pfo_exp_neg(s_sc[wave][h][li] - m[h]) * il : 0.f;
      if (li < Cp - C) ctx[h * Cp + C + li] = li == 0 ? ld[h] * il : ((li == 1 && h == 0) ? 1.f : 0.f);
    }
    seq += H * (2 * NR + 2) + 1;
    FWD_STAMP(6);
  }
  FWD_STAMP_END(n0 / P, cnt);
}
static bool attn_fwd_pipe_ok(const PfoAttn& a) {
  static const int on = getenv("PFO_ATTN_FWD_PIPE") ? atoi(getenv("PFO_ATTN_FWD_PIPE")) : 0;      // A/B switch; off: measured 0.245-0.263 ms per step against 0.172 for the ring form (profiles/r6_experiments.txt)
  static const int min_n = getenv("PFO_ATTN_FWD_PIPE_MIN") ? atoi(getenv("PFO_ATTN_FWD_PIPE_MIN")) : 16384;
  const int64_t qk_ld = a.qk_ld > 0 ? a.qk_ld : (int64_t)a.H * a.Cp;
  // (query rows travel 16 bytes per lane; metadata as FWD_IPW K dwords per array; small launches keep one instance per wavefront)
  return on && a.N >= min_n && (qk_ld % 4) == 0 && (((uintptr_t)a.QK) & 15u) == 0 && FWD_IPW * a.K <= 1024 && FWD_IPW <= 64 &&
         ((a.D + 63) / 64) * a.H <= 6;                             // (the instantiations its launcher holds)
}

// the ring form takes rows it can move 16 bytes at a time, with [node | edge] inside NR column groups and one DMA per key
static bool attn_fwd_ring_ok(const PfoAttn& a) {
  static const int on = getenv("PFO_ATTN_FWD_RING") ? atoi(getenv("PFO_ATTN_FWD_RING")) : 1;     // A/B switch
  const int NRv = (a.D + 63) / 64;
  return on && (a.D % 4) == 0 && (a.Ef % 4) == 0 && a.D + a.Ef <= 64 * NRv && a.D + a.Ef <= 256 && (a.nbr_ld % 4) == 0 &&
         (((uintptr_t)a.nbr_tab | (uintptr_t)a.edge_feat | (uintptr_t)a.QK) & 15u) == 0 && NRv * a.H <= FWD_RING_MAX_NRH &&
         ((a.qk_ld > 0 ? a.qk_ld : (int64_t)a.H * a.Cp) % 4) == 0;
}

#ifndef ATTN_BWD_MAX_BLOCKS
#define ATTN_BWD_MAX_BLOCKS 4096
#endif
int pfo_attn_bwd_max_parts() { return ATTN_TIME_BINS; }

// (DMODE 3: the deterministic form of 1 - int64 fixed-point atomics into one table, attn.hpp)
// DMODE: what happens to the neighbour-row gradients - 0 none (layer 1 without memory: level-0 rows are constants),
// 1 float atomics into the rows `nbr_row` names (layer 1 over the touched-node table), 2 plain stores (layers >= 2,
// where every (instance, slot) owns its row).  A template parameter: as a run-time test it cost three scalar branch
// sequences per key inside the inner loop.
template <int NR, int H, int DMODE>
__device__ __forceinline__ void attn_bwd_body(const AttnDev& a) {
  __shared__ double s_red[4][2][NR * 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int D = a.D, Ef = a.Ef, K = a.K, C = 2 * D + Ef, Cp = a.Cp;
  constexpr bool wdirect = (DMODE == 2);                // ... and their gradients are written, not accumulated
  const bool direct = (a.nbr_row == nullptr);
  float tw[NR], tb[NR];
  double dw[NR], db[NR];     // sums of terms scaled by dt ~ 1e7 with heavy cancellation: accumulate in fp64
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const int c = lane + 64 * r;
    tw[r] = c < D ? a.tw[c] : 0.f;
    tb[r] = c < D ? a.tb[c] : 0.f;
    dw[r] = 0.0; db[r] = 0.0;
  }
  const float keep_scale = a.dropout_p > 0.f ? 1.f / (1.f - a.dropout_p) : 1.f;
  // DMODE 1: every XCD adds into ITS replica of the gradient table.  The eight L2s are kept coherent by hardware, so
  // float atomics from all XCDs on one table make each cache line migrate between L2s (the 500 item rows take 77 % of
  // the adds); with one replica per XCD the atomics stay in the local L2.  The XCC id only picks the replica: a wrong
  // value would cost speed, never correctness (the atomics are device-coherent either way).
  float* const d_nbr_x = (DMODE == 1) ? a.d_nbr + (int64_t)(__builtin_amdgcn_s_getreg(6164) & (a.d_nbr_nrep - 1)) * a.d_nbr_rep : a.d_nbr;   // hwreg(HW_REG_XCC_ID, 0, 4)

  for (int64_t n = (int64_t)blockIdx.x * 4 + wave; n < a.N; n += (int64_t)gridDim.x * 4) {
    float* dqk_out = a.dQK + n * H * Cp;
    const int64_t slot0 = n * K;
    const bool inK = lane < K;
    const int my_id = inK ? a.nbr_ids[slot0 + lane] : 0;
    const int my_row = inK ? (direct ? (int)(a.nbr_row_base + slot0 + lane) : a.nbr_row[slot0 + lane]) : 0;
    const int my_e = inK ? a.eidx[slot0 + lane] : 0;
    const float my_dt = inK ? a.dt[slot0 + lane] : 0.f;
    const unsigned long long valid = __ballot(inK && my_id != 0);
    if (wdirect) {
      // padded slots own a gradient row too (the buffer is reused every step): zero it
      unsigned long long im = ~valid & (K >= 64 ? ~0ull : ((1ull << K) - 1ull));
      while (im) {
        const int j = __ffsll((long long)im) - 1;
        im &= im - 1ull;
        float* dst = a.d_nbr + (a.nbr_row_base + slot0 + j) * a.d_nbr_ld;
        for (int c = lane; c < D; c += 64) dst[c] = 0.f;
      }
    }
    if (valid == 0ull) {
      for (int c = lane; c < H * Cp; c += 64) dqk_out[c] = 0.f;
      continue;
    }
    float qn[H][NR], qt[H][NR], qe[H], gn[H][NR], gt[H][NR], ge[H], t[H], dsb[H];
    float dqn[H][NR], dqt[H][NR], dqe[H];
    const float* qk = a.QK + (a.qk_row ? (int64_t)a.qk_row[n] : n) * a.qk_ld;
    const float* dc = a.dctx + n * H * Cp;
    const float* cx = a.ctx + n * H * Cp;
#pragma unroll
    for (int h = 0; h < H; ++h) {
      float part = 0.f;
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        const int c = lane + 64 * r;
        const bool ok = c < D;
        qn[h][r] = ok ? qk[h * Cp + c] : 0.f;
        qt[h][r] = ok ? qk[h * Cp + D + Ef + c] : 0.f;
        gn[h][r] = ok ? dc[h * Cp + c] : 0.f;
        gt[h][r] = ok ? dc[h * Cp + D + Ef + c] : 0.f;
        if (ok) part = fmaf(gn[h][r], cx[h * Cp + c], fmaf(gt[h][r], cx[h * Cp + D + Ef + c], part));
        dqn[h][r] = 0.f; dqt[h][r] = 0.f;
      }
      qe[h] = lane < Ef ? qk[h * Cp + D + lane] : 0.f;
      ge[h] = lane < Ef ? dc[h * Cp + D + lane] : 0.f;
      if (lane < Ef) part = fmaf(ge[h], cx[h * Cp + D + lane], part);
      dqe[h] = 0.f;
      t[h] = part;
      dsb[h] = dc[h * Cp + C];          // d loss / d (Σ_j a'_jh): the gradient of the context's extra column
    }
    // delta_h = sum_j a_jh * da_jh = dctx_h . ctx_h + d(Σa')_h * (Σa')_h; interleaved butterflies
    pfo_wave_sum_scalar_n<H>(t);
#pragma unroll
    for (int h = 0; h < H; ++h) t[h] = fmaf(dsb[h], cx[h * Cp + C], t[h]);

    const unsigned keep = attn_keep_for(a, a.offset + (a.offset_dev ? *a.offset_dev : 0ull), n, lane);
    float my_a[H];
#pragma unroll
    for (int h = 0; h < H; ++h) my_a[h] = inK ? a.attw[(n * H + h) * K + lane] : 0.f;

    unsigned long long vm = valid;
    while (vm) {
      int js[KC_BWD];
#pragma unroll
      for (int c = 0; c < KC_BWD; ++c) {
        js[c] = vm ? (__ffsll((long long)vm) - 1) : -1;
        vm &= vm - 1ull;
      }
      float kn[KC_BWD][NR], kt[KC_BWD][NR], ks[KC_BWD][NR], ke[KC_BWD], dtv[KC_BWD];
      int rows[KC_BWD];
#pragma unroll
      for (int c = 0; c < KC_BWD; ++c) {
        const int j = js[c] < 0 ? 0 : js[c];
        rows[c] = rl_i(my_row, j);
        const float* src = a.nbr_tab + (int64_t)rows[c] * a.nbr_ld;
        const int e = rl_i(my_e, j);
        dtv[c] = rl_f(my_dt, j);
#pragma unroll
        for (int r = 0; r < NR; ++r) {
          const int cc = lane + 64 * r;
          kn[c][r] = (js[c] >= 0 && (r < NR - 1 || cc < D)) ? src[cc] : 0.f;
        }
        ke[c] = (js[c] >= 0 && lane < Ef) ? a.edge_feat[(uint32_t)e * (uint32_t)Ef + (uint32_t)lane] : 0.f;
      }
      float part[KC_BWD][H];
#pragma unroll
      for (int c = 0; c < KC_BWD; ++c) {
#pragma unroll
        for (int r = 0; r < NR; ++r) {
          const int cc = lane + 64 * r;
          float sv, cv;
          pfo_sincosf(pfo_time_arg(dtv[c], tw[r], tb[r]), sv, cv);
          const bool on = js[c] >= 0 && (r < NR - 1 || cc < D);     // select, not a branch (only the last r can be off)
          kt[c][r] = on ? cv : 0.f;
          ks[c][r] = on ? sv : 0.f;
        }
#pragma unroll
        for (int h = 0; h < H; ++h) {
          float pp = ke[c] * ge[h];
#pragma unroll
          for (int r = 0; r < NR; ++r) pp = fmaf(kn[c][r], gn[h][r], fmaf(kt[c][r], gt[h][r], pp));
          part[c][h] = pp;
        }
      }
#pragma unroll
      for (int c = 0; c < KC_BWD; ++c)
#pragma unroll
        for (int h = 0; h < H; ++h) part[c][h] = pfo_wave_sum_scalar(part[c][h]);
#pragma unroll
      for (int c = 0; c < KC_BWD; ++c) {
        if (js[c] < 0) continue;
        const unsigned kb = (unsigned)rl_i((int)keep, js[c]);
        float cA[H], cB[H];
#pragma unroll
        for (int h = 0; h < H; ++h) {
          const float ks_h = ((kb >> h) & 1u) ? keep_scale : 0.f;
          const float da = (part[c][h] + dsb[h]) * ks_h;               // d loss / d a_jh (through dropout)
          const float aj = rl_f(my_a[h], js[c]);
          const float dscore = aj * (da - t[h]);                       // softmax backward
          cA[h] = aj * ks_h;                                           // a'_jh multiplies dctx_h
          cB[h] = dscore * a.scale;                                    // multiplies qk_h
#pragma unroll
          for (int r = 0; r < NR; ++r) {
            dqn[h][r] = fmaf(cB[h], kn[c][r], dqn[h][r]);
            dqt[h][r] = fmaf(cB[h], kt[c][r], dqt[h][r]);
          }
          dqe[h] = fmaf(cB[h], ke[c], dqe[h]);
        }
        float* dst = (DMODE == 1 || DMODE == 2) ? d_nbr_x + (int64_t)rows[c] * a.d_nbr_ld : nullptr;
        if (DMODE == 1 && a.abl == 1) dst = d_nbr_x + (int64_t)((((unsigned)rows[c] * 2654435761u) + (unsigned)n * 40503u) % 8192u) * a.d_nbr_ld;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
          const int cc = lane + 64 * r;
          float dkn = 0.f, dkt = 0.f;
#pragma unroll
          for (int h = 0; h < H; ++h) {
            dkn = fmaf(cA[h], gn[h][r], fmaf(cB[h], qn[h][r], dkn));
            dkt = fmaf(cA[h], gt[h][r], fmaf(cB[h], qt[h][r], dkt));
          }
          if (r < NR - 1 || cc < D) {                    // 64 * (NR - 1) < D: only the last r needs the lane test
            if (DMODE == 2) dst[cc] = (a.nbr_relu && !(kn[c][r] > 0.f)) ? 0.f : dkn;   // the row is a ReLU output of the layer below
            else if (DMODE == 1 && a.abl != 2) atomicAdd(dst + cc, dkn);
            else if (DMODE == 3) det_add(a.d_nbr, (int64_t)rows[c] * a.d_nbr_ld + cc, dkn);
          }
          const float gsin = -ks[c][r] * dkt;            // d/d(arg) cos(arg) = -sin(arg); ks = 0 on lanes beyond D
          dw[r] += (double)gsin * (double)dtv[c];
          db[r] += (double)gsin;
        }
      }
    }
#pragma unroll
    for (int h = 0; h < H; ++h) {
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        const int c = lane + 64 * r;
        if (c < D) {
          dqk_out[h * Cp + c] = dqn[h][r];
          dqk_out[h * Cp + D + Ef + c] = dqt[h][r];
        }
      }
      if (lane < Ef) dqk_out[h * Cp + D + lane] = dqe[h];
      if (lane < Cp - C) dqk_out[h * Cp + C + lane] = 0.f;      // padding columns feed GEMMs: keep them finite
    }
  }
  // time-encoder partials: fold the four wavefronts, one slab row per workgroup (deterministic)
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    s_red[wave][0][lane + 64 * r] = dw[r];
    s_red[wave][1][lane + 64 * r] = db[r];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < 2 * D; c += 256) {
    const int which = c / D, cc = c - which * D;
    const double v = s_red[0][which][cc] + s_red[1][which][cc] + s_red[2][which][cc] + s_red[3][which][cc];
    // fp64 atomics into one of ATTN_TIME_BINS accumulator rows (all layers of a step share them; folded once at the end);
    // deterministic mode: this workgroup's own slab row (folded in row order)
    if (a.det) a.dtime_slab[(int64_t)blockIdx.x * 2 * D + c] = v;     // (once per workgroup: a run-time test costs nothing here)
    else atomicAdd(&a.dtime_part[(int64_t)(blockIdx.x & (ATTN_TIME_BINS - 1)) * 2 * D + c], v);
  }
}

// ---------------------------------------------------------------------------------------------
// KEY RING for the per-instance backward (round 6): the form above gathers a pair of keys into registers and uses them at once -
// one exposed round trip per pair, ten per instance - and its predicated first-level loads queue as dependent round trips
// (ISA notes at attn_fwd_ring_kernel).  Here, as in the ring forward: every first-level load clamped and in one batch, one
// LDS-DMA per key into a wavefront-private ring BWD_RING - 2 keys ahead of the pair being differentiated, [node | edge] columns
// as one vector on every side (key slot, qk' row, d ctx' row, d qk' row), counted vmcnt.  DMODE 0 (no key-side gradients:
// layer 1 without memory - C5) and 2 (plain stores: layers >= 2); the atomic forms keep the register kernel (layer 1 with
// memory takes the run-merged kernel anyway).  Same arithmetic per element as attn_bwd_body.
#ifndef BWD_RING
#define BWD_RING 4
#endif
template <int NR, int H, int DMODE>
__device__ __forceinline__ void attn_bwd_ring_body(const AttnDev& a) {
  static_assert(DMODE == 0 || DMODE == 2, "ring form: no key-side gradients, or plain stores");
  constexpr int SLOTF = NR * 64, RP = BWD_RING / 2;
  __shared__ double s_red[4][2][NR * 64];
  __shared__ __align__(16) float s_ring[4][BWD_RING][SLOTF];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int D = a.D, Ef = a.Ef, K = a.K, DE = D + Ef, C = 2 * D + Ef, Cp = a.Cp;
  const bool direct = (a.nbr_row == nullptr);
  float tw[NR], tb[NR];
  double dw[NR], db[NR];     // sums of terms scaled by dt ~ 1e7 with heavy cancellation: accumulate in fp64
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const int c = min(lane + 64 * r, D - 1);
    const float w = a.tw[c], b = a.tb[c];
    tw[r] = lane + 64 * r < D ? w : 0.f;
    tb[r] = lane + 64 * r < D ? b : 0.f;
    dw[r] = 0.0; db[r] = 0.0;
  }
  float* const ring = &s_ring[wave][0][0];
  if (lane + 64 * (NR - 1) >= DE)
#pragma unroll
    for (int s = 0; s < BWD_RING; ++s) ring[s * SLOTF + lane + 64 * (NR - 1)] = 0.f;     // (columns no DMA writes read as zero)
  const float keep_scale = a.dropout_p > 0.f ? 1.f / (1.f - a.dropout_p) : 1.f;
  const uint64_t rng_off = a.offset + (a.offset_dev ? *a.offset_dev : 0ull);
  typedef __attribute__((address_space(1))) const void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  const int D4 = D >> 2, DE4 = DE >> 2;
  const bool is_node = lane < D4;
  const uint64_t base_l = (is_node ? (uint64_t)(uintptr_t)a.nbr_tab : (uint64_t)(uintptr_t)a.edge_feat) + (uint64_t)((is_node ? lane : lane - D4) * 16);
  const uint32_t mul_l = is_node ? (uint32_t)a.nbr_ld * 4u : (uint32_t)Ef * 4u;
  const uint32_t node_mask = is_node ? 0xFFFFFFFFu : 0u;
  const bool inK = lane < K;

  for (int64_t n = (int64_t)blockIdx.x * 4 + wave; n < a.N; n += (int64_t)gridDim.x * 4) {
    float* dqk_out = a.dQK + n * H * Cp;
    const int64_t slot0 = n * K;
    // ---- first-level loads, clamped, one batch
    const int64_t si = slot0 + min(lane, K - 1);
    const int id_r = a.nbr_ids[si];
    const int row_l = (direct ? a.nbr_ids : a.nbr_row)[si];
    const int e_r = a.eidx[si];
    const float dt_r = a.dt[si];
    const int qrow_l = (a.qk_row ? a.qk_row : a.nbr_ids)[n];
    float a_r[H];
#pragma unroll
    for (int h = 0; h < H; ++h) a_r[h] = a.attw[(n * H + h) * K + min(lane, K - 1)];
    int my_id = inK ? id_r : 0, my_row = inK ? (direct ? (int)(a.nbr_row_base + si) : row_l) : 0, my_e = inK ? e_r : 0;
    float my_dt = inK ? dt_r : 0.f;
    float my_a[H];
#pragma unroll
    for (int h = 0; h < H; ++h) my_a[h] = inK ? a_r[h] : 0.f;
    const int qrow = a.qk_row ? qrow_l : (int)n;
    const unsigned long long valid = __ballot(inK && my_id != 0);
    if (DMODE == 2) {
      // padded slots own a gradient row too (the buffer is reused every step): zero it
      unsigned long long im = ~valid & (K >= 64 ? ~0ull : ((1ull << K) - 1ull));
      while (im) {
        const int j = __ffsll((long long)im) - 1;
        im &= im - 1ull;
        float* dst = a.d_nbr + (a.nbr_row_base + slot0 + j) * a.d_nbr_ld;
        for (int c = lane; c < D; c += 64) dst[c] = 0.f;
      }
    }
    if (valid == 0ull) {
      for (int c = lane; c < H * Cp; c += 64) dqk_out[c] = 0.f;
      continue;
    }
    // ---- second level: the instance's qk', d ctx' and ctx' rows (register-bound, clamped, in front of the DMAs)
    float q1[H][NR], qt[H][NR], g1[H][NR], gt[H][NR], t[H], dsb[H];
    {
      const float* qk = a.QK + (int64_t)qrow * a.qk_ld;
      const float* dc = a.dctx + n * H * Cp;
      const float* cx = a.ctx + n * H * Cp;
      float c1[H][NR], ct[H][NR], cs[H];
#pragma unroll
      for (int h = 0; h < H; ++h) {
#pragma unroll
        for (int r = 0; r < NR; ++r) {
          const int k1 = min(lane + 64 * r, DE - 1), kt_ = min(lane + 64 * r, D - 1);
          q1[h][r] = qk[h * Cp + k1]; qt[h][r] = qk[h * Cp + DE + kt_];
          g1[h][r] = dc[h * Cp + k1]; gt[h][r] = dc[h * Cp + DE + kt_];
          c1[h][r] = cx[h * Cp + k1]; ct[h][r] = cx[h * Cp + DE + kt_];
        }
        dsb[h] = dc[h * Cp + C];          // d loss / d (sum_j a'_jh): the gradient of the context's extra column
        cs[h] = cx[h * Cp + C];
      }
#pragma unroll
      for (int h = 0; h < H; ++h) {
        float part = 0.f;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
          const int c = lane + 64 * r;
          if (!(c < DE)) { q1[h][r] = 0.f; g1[h][r] = 0.f; c1[h][r] = 0.f; }
          if (!(c < D)) { qt[h][r] = 0.f; gt[h][r] = 0.f; ct[h][r] = 0.f; }
          part = fmaf(g1[h][r], c1[h][r], fmaf(gt[h][r], ct[h][r], part));
        }
        t[h] = part;
      }
      // delta_h = sum_j a_jh da_jh = d ctx'_h . ctx'_h + d(sum a')_h (sum a')_h
      pfo_wave_sum_scalar_n<H>(t);
#pragma unroll
      for (int h = 0; h < H; ++h) t[h] = fmaf(dsb[h], cs[h], t[h]);
    }
    const unsigned keep = attn_keep_for(a, rng_off, n, lane);
    // (every register-bound value is pinned in front of the first DMA: the compiler's wait for it must not sit behind one)
    asm volatile("" : "+v"(my_row), "+v"(my_e), "+v"(my_dt), "+v"(my_id));
#pragma unroll
    for (int h = 0; h < H; ++h) { asm volatile("" : "+v"(my_a[h]), "+v"(t[h]), "+v"(dsb[h])); }
    unsigned kp = keep;
    asm volatile("" : "+v"(kp));
    // ---- the ring: pair p = valid keys 2p, 2p + 1
    unsigned long long vm = valid;
    const int n_pairs = (__popcll(valid) + 1) >> 1;
    int issued = 0;
    auto issue_pair = [&]() {
      int j0 = __ffsll((long long)vm) - 1;
      vm &= vm - 1ull;
      int j1 = vm ? (__ffsll((long long)vm) - 1) : j0;
      vm &= vm - 1ull;
      const int sl = (2 * issued) % BWD_RING;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int j = c ? j1 : j0;
        const uint32_t e_s = (uint32_t)rl_i(my_e, j);
        const uint32_t idx = e_s + (((uint32_t)rl_i(my_row, j) - e_s) & node_mask);
        const uint64_t src = base_l + (uint64_t)idx * (uint64_t)mul_l;
        if (lane < DE4) __builtin_amdgcn_global_load_lds((gptr_t)(uintptr_t)src, (lptr_t)(ring + (sl + c) * SLOTF), 16, 0, 0);
      }
      issued += 1;
    };
#pragma unroll
    for (int p = 0; p < RP; ++p)
      if (p < n_pairs) issue_pair();
    float dq1[H][NR], dqt[H][NR];
#pragma unroll
    for (int h = 0; h < H; ++h)
#pragma unroll
      for (int r = 0; r < NR; ++r) { dq1[h][r] = 0.f; dqt[h][r] = 0.f; }
    unsigned long long wm = valid;
    for (int p = 0; p < n_pairs; ++p) {
      int js[2];
      js[0] = __ffsll((long long)wm) - 1;
      wm &= wm - 1ull;
      js[1] = wm ? (__ffsll((long long)wm) - 1) : -1;
      wm &= wm - 1ull;
      const bool more = p + RP < n_pairs;
      if (more) pfo_wait_vm<2 * (RP - 1)>(); else pfo_wait_vm<0>();
      const float* sl = ring + ((2 * p) % BWD_RING) * SLOTF;
      float kn[2][NR], kt[2][NR], ks[2][NR], dtv[2];
      int rows[2];
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int j = js[c] < 0 ? js[0] : js[c];
        rows[c] = rl_i(my_row, j);
        dtv[c] = rl_f(my_dt, j);
#pragma unroll
        for (int r = 0; r < NR; ++r) kn[c][r] = sl[c * SLOTF + lane + 64 * r];
      }
      float part[2 * H];
#pragma unroll
      for (int c = 0; c < 2; ++c) {
#pragma unroll
        for (int r = 0; r < NR; ++r) {
          float sv, cv;
          pfo_sincosf(pfo_time_arg(dtv[c], tw[r], tb[r]), sv, cv);
          const bool on = r < NR - 1 || lane + 64 * r < D;
          kt[c][r] = on ? cv : 0.f;
          ks[c][r] = on ? sv : 0.f;
        }
#pragma unroll
        for (int h = 0; h < H; ++h) {
          float pp = 0.f;
#pragma unroll
          for (int r = 0; r < NR; ++r) pp = fmaf(kn[c][r], g1[h][r], fmaf(kt[c][r], gt[h][r], pp));
          part[c * H + h] = pp;
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // the slots' values sit in registers: the pair RP ahead starts its trip
      if (more) issue_pair();
      pfo_wave_sum_scalar_n<2 * H>(part);
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        if (js[c] < 0) continue;
        const unsigned kb = (unsigned)rl_i((int)kp, js[c]);
        float cA[H], cB[H];
#pragma unroll
        for (int h = 0; h < H; ++h) {
          const float ks_h = ((kb >> h) & 1u) ? keep_scale : 0.f;
          const float da = (part[c * H + h] + dsb[h]) * ks_h;            // d loss / d a_jh (through dropout)
          const float aj = rl_f(my_a[h], js[c]);
          const float dscore = aj * (da - t[h]);                         // softmax backward
          cA[h] = aj * ks_h;                                             // a'_jh multiplies d ctx'_h
          cB[h] = dscore * a.scale;                                      // multiplies qk'_h
#pragma unroll
          for (int r = 0; r < NR; ++r) {
            dq1[h][r] = fmaf(cB[h], kn[c][r], dq1[h][r]);
            dqt[h][r] = fmaf(cB[h], kt[c][r], dqt[h][r]);
          }
        }
        float* dst = (DMODE == 2) ? a.d_nbr + (int64_t)rows[c] * a.d_nbr_ld : nullptr;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
          const int cc = lane + 64 * r;
          float dkn = 0.f, dkt = 0.f;
#pragma unroll
          for (int h = 0; h < H; ++h) {
            dkn = fmaf(cA[h], g1[h][r], fmaf(cB[h], q1[h][r], dkn));
            dkt = fmaf(cA[h], gt[h][r], fmaf(cB[h], qt[h][r], dkt));
          }
          if (DMODE == 2 && cc < D) dst[cc] = (a.nbr_relu && !(kn[c][r] > 0.f)) ? 0.f : dkn;   // the row is a ReLU output of the layer below
          const float gsin = -ks[c][r] * dkt;            // d/d(arg) cos(arg) = -sin(arg); ks = 0 on lanes beyond D
          dw[r] += (double)gsin * (double)dtv[c];
          db[r] += (double)gsin;
        }
      }
    }
#pragma unroll
    for (int h = 0; h < H; ++h) {
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        const int c = lane + 64 * r;
        if (c < DE) dqk_out[h * Cp + c] = dq1[h][r];
        if (c < D) dqk_out[h * Cp + DE + c] = dqt[h][r];
      }
      if (lane < Cp - C) dqk_out[h * Cp + C + lane] = 0.f;      // padding columns feed GEMMs: keep them finite
    }
    // (the stores above are drained before the next instance's DMAs by the compiler: a store round trip per instance, behind ~20 keys of arithmetic)
  }
  // time-encoder partials: fold the four wavefronts, one slab row per workgroup (deterministic)
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    s_red[wave][0][lane + 64 * r] = dw[r];
    s_red[wave][1][lane + 64 * r] = db[r];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < 2 * D; c += 256) {
    const int which = c / D, cc = c - which * D;
    const double v = s_red[0][which][cc] + s_red[1][which][cc] + s_red[2][which][cc] + s_red[3][which][cc];
    if (a.det) a.dtime_slab[(int64_t)blockIdx.x * 2 * D + c] = v;
    else atomicAdd(&a.dtime_part[(int64_t)(blockIdx.x & (ATTN_TIME_BINS - 1)) * 2 * D + c], v);
  }
}
static bool attn_bwd_ring_ok(const PfoAttn& a, int dmode) {
  static const int on = getenv("PFO_ATTN_BWD_RING") ? atoi(getenv("PFO_ATTN_BWD_RING")) : 1;      // A/B switch
  const int NRv = (a.D + 63) / 64;
  return on && (dmode == 0 || dmode == 2) && (a.D % 4) == 0 && (a.Ef % 4) == 0 && a.D + a.Ef <= 64 * NRv && a.D + a.Ef <= 256 &&
         (a.nbr_ld % 4) == 0 && (((uintptr_t)a.nbr_tab | (uintptr_t)a.edge_feat) & 15u) == 0 && NRv * a.H <= 12;
}

// ---------------------------------------------------------------------------------------------
// Layer-1 backward with SHIFT MERGING.  The level-0 gradient scatter is bound by the float-atomic rate of the part (~1.3 TB/s of
// added bytes chip-wide, MI355X_MICROARCH.md: one 256-byte wave instruction per ~50 ns per CU); everything else in this kernel
// hides behind it.  Instances arrive ordered by (touched-table row, entries of the row's history before the instance's time) -
// memory.hpp pfo_seg_build_launch - so consecutive members of a row walk its history forwards, and under most-recent sampling
// slot j of an instance holds history entry cnt - K + j: the neighbour lists of two instances of one node are SHIFTS of each
// other by the difference of their counts (identical when the counts are equal - every ~4 instances of a user in a batch;
// shifted by one or two entries for the instances of an item, whose history grows inside the batch window).
// One wavefront (a 64-thread workgroup) walks a chunk of RUN_CHUNK consecutive members.  Lane q stands for history entry
// cnt0 - K + q of the group's first instance; an instance whose count is cnt0 + d puts its slot j on lane j + d.  The key-side
// gradient rows of a group leave as ONE set of float atomics, one row per distinct history entry, when the row, the lane range
// (K + d <= 64) or the chunk ends.  Round 2 merged only identical lists (d = 0): at C2 that left the ~12 k item instances, 55 %
// of the atomic instructions, unmerged.
//
// The rows are not accumulated while the instances are walked.  The gradient of a neighbour row summed over a group is
//     sum_i sum_h ( cA_ih * g_ih  +  cB_ih * q_h )        g_ih = d ctx'_h (node part) of instance i,  q_h = the node's query
// with wave-uniform scalars cA (post-dropout weight) and cB (d score * scale), and q_h is the same for every instance of the
// node.  So the walk only keeps the SCALARS - per instance and slot cA (LDS), per lane the running sum of cB (a register) - and
// the rows are formed once per group from the re-read g rows (just used: cache hits).
#ifndef RUN_CHUNK
#define RUN_CHUNK 4     // measured at C2 (round 2): 2: 399, 3: 370, 4: 365, 6: 402, 8: 417 us (a wavefront walks its chunk serially:
#endif                  // long chunks merge more atomics but leave a tail)

#ifndef RUN_CPW
#define RUN_CPW 1      // consecutive chunks per wavefront: one prologue (member ids -> query rows / counts) for all their members
#endif
#ifndef KC_RUNS
#define KC_RUNS 2      // keys in flight per wavefront
#endif
// RUNS_LATE (round 6): the rarely executed blocks of attn_bwd_runs_kernel (staging, row-sum store, flush) read their pointers and
// strides from the kernel-argument segment WHERE THEY USE THEM, through a pointer the optimiser cannot see through - hoisted to the
// kernel's entry, two dozen scalar pairs lived across the key walk and were parked in vector lanes (149 SGPR spills: a v_readlane /
// v_writelane per access on a kernel that is short of vector issue slots).
#ifndef RUNS_LATE
#define RUNS_LATE 0      // (measured: 0.329-0.333 against 0.269 ms per step - scalar loads on the flush / staging paths wait on lgkmcnt with the LDS traffic; off)
#endif
#if RUNS_LATE && defined(__HIP_DEVICE_COMPILE__)      // (the host pass of this file only parses the kernel)
#define RUNS_LATE_ARGS(ka) const AttnDev* ka = reinterpret_cast<const AttnDev*>(__builtin_amdgcn_kernarg_segment_ptr()); asm volatile("" : "+s"(ka))
#else
#define RUNS_LATE_ARGS(ka) const AttnDev* const ka = &a
#endif
#ifndef RUNS_ASM_GATHER
#define RUNS_ASM_GATHER 0   // 1: key gathers by inline asm + counted vmcnt, staging behind the first two pairs (attn_bwd_runs_kernel) - measured 0.297 against 0.267 ms per step (profiles/r6_experiments.txt 10), off
#endif
#ifndef RUNS_CW
#define RUNS_CW 0      // counted vmcnt at a member's start (attn_bwd_runs_kernel)
#endif
#ifndef RUNS_WAVES
#define RUNS_WAVES(NR, H) ((NR) * (H) <= 6 ? 3 : 2)
#endif
// MEMBER STAGING (round 5).  In-kernel stamps (tools/probes/runs_stamps.py, -DRUNS_STAMPS=1) showed where a wavefront's life
// went: 47 % in the member set-up, 32 % walking the 20 keys, 12 % in the flush - the set-up is a chain of dependent round trips
// (members[m] -> qk_row[n] / run_cnt[n] -> the query row; then d ctx' and ctx', streamed from HBM exactly once by this kernel,
// ~9 us per member against ~6 us for the walk), paid once per member with three wavefronts per SIMD to hide it.  Now:
//  * a chunk prologue resolves the indirections of ALL its members at once (one lane per member: two dependent loads per chunk);
//  * the rows of member i+1 (d ctx', ctx', the node's query row: 3 x H Cp floats) and its per-slot metadata (neighbour ids, table
//    rows, edge ids, dt, the attention weights: (4 + H) x K words) travel by LDS-DMA (global_load_lds: no registers, no wait)
//    into a wavefront-private staging image while member i is walked; the set-up of member i+1 reads LDS.  EXEC masks the
//    lanes behind a row's end (tools/probes/lds_dma_mask.hip: masked lanes write nothing);
//  * the flush takes the run's member ids from the prologue's register instead of re-loading members[].
// The r4 one-dword-per-line touch of the next rows (-2 %, +36 % fetched bytes) is gone.
// How a wavefront spends its cycles (round 3; measured against the round-2 loop, 357 vs 363 us at equal atomics - the atomic
// rate, not the issue rate, bounds this kernel):
//  * the gathers of key chunk c+1 are issued before chunk c is scored (two register sets used alternately, as in the forward);
//  * no exec-mask region and no select in the key loop: gathers read clamped addresses (the last 64-column group re-reads
//    column D-1, the edge lanes column Ef-1) instead of being predicated, and the lanes they feed carry zero query / gradient
//    values or are never stored; the time-encoder parameters are zero beyond D, so the encodings there are cos(0), sin(0);
//  * the "argument too large for the fp32 range reduction" test is taken once per INSTANCE from max|dt| * max|w| + max|b|
//    (wave-uniform), not per chunk on every argument;
//  * the per-key softmax-backward scalars of a chunk are reduced together (one interleaved DPP tree for KC*H sums), the running
//    sum of cB per key lives in a register (select on lane == key) instead of an LDS read-modify-write.
// RUNS_STAMPS (diagnostic build only, -DRUNS_STAMPS=1): shader cycles a wavefront spends per section, summed over the launch into
// pfo_runs_stamps (0 whole wavefront, 1 member set-up, 2 key walk, 3 flush, 4 row-sum store, 5 members, 6 chunks); the stamps
// go to a buffer of their own and no output is computed from them (MI355X_MICROARCH.md, DVFS give-back item 6)
#ifndef RUNS_STAMPS
#define RUNS_STAMPS 0
#endif
#if RUNS_STAMPS
// (one private row per workgroup, summed on the host - round 6: the round-5 form added every wavefront's sums into ONE row
//  with atomics, which serialise in one L2 line and stretch the sections that touch memory)
#define RUNS_STAMP_ROWS 16384
__device__ unsigned long long pfo_runs_stamps[RUNS_STAMP_ROWS * 8];
extern "C" int pfo_attn_runs_stamps(unsigned long long* out, int reset) {
  static unsigned long long host[RUNS_STAMP_ROWS * 8];
  if (out) {
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(pfo_runs_stamps), sizeof(host)) != hipSuccess) return PFO_ERR_HIP;
    for (int i = 0; i < 8; ++i) out[i] = 0;
    for (int r = 0; r < RUNS_STAMP_ROWS; ++r) for (int i = 0; i < 8; ++i) out[i] += host[r * 8 + i];
  }
  if (reset) { memset(host, 0, sizeof(host)); if (hipMemcpyToSymbol(HIP_SYMBOL(pfo_runs_stamps), host, sizeof(host)) != hipSuccess) return PFO_ERR_HIP; }
  return PFO_OK;
}
#define STAMP() ((unsigned long long)__builtin_amdgcn_s_memtime())
#endif
template <int NR, int H, bool DET>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(RUNS_WAVES(NR, H)))) void attn_bwd_runs_kernel(const AttnDev a) {
#if RUNS_STAMPS
  const unsigned long long st_begin = STAMP();
  unsigned long long st_setup = 0, st_walk = 0, st_flush = 0, st_store = 0, st_members = 0, st_chunks = 0;
#endif
  __shared__ float s_tw[NR * 64], s_tb[NR * 64];
  __shared__ float s_cA[RUN_CHUNK][H][64];     // [instance of the group][head][slot]: cA of that key; zero where the slot is empty
  __shared__ int s_delta[RUN_CHUNK];            // [instance of the group]: its shift (count - the group's first count)
  __shared__ int s_ch[3][RUN_CPW * RUN_CHUNK];  // the wavefront's members: instance id, query row, history count (prologue)
  // staging image (dynamic LDS, sized by the launcher: pfo_attn_runs_lds_bytes): [d ctx' row | ctx' row | query row | metadata]
  extern __shared__ __align__(16) unsigned char s_stage[];
  const int lane = threadIdx.x;
  const int D = a.D, Ef = a.Ef, K = a.K, C = 2 * D + Ef, Cp = a.Cp;
  const bool clk_on = blockIdx.x == 0;
  PfoClockStamp clk;
  if (clk_on) clk = pfo_clock_begin();
  // Two chains of dependent loads open a wavefront's life: (A) *n_rows -> seg_ptr[.] = the members' count M, (B) the
  // wavefront's member ids -> their query rows / history counts.  (B) is issued speculatively - clamped indices, no dependence
  // on M - so the two chains overlap instead of queueing (a wavefront lives ~35 us: every round trip is 3-4 % of it).
  constexpr int UM = RUN_CPW * RUN_CHUNK;                          // members per wavefront
  const int G8 = 8 * a.xcd_g;
  auto unit_of = [&](int blk) {
    if (G8 <= 0) return blk;
    const int grp = blk / G8, r = blk - grp * G8;
    return grp * G8 + (r & 7) * a.xcd_g + (r >> 3);
  };
  auto spec_member = [&](int unit) { return a.members[min(unit * UM + min(lane, UM - 1), a.N - 1)]; };
  int sp_raw = spec_member(unit_of((int)blockIdx.x));
  const int nr_rows = *a.n_rows;
  float wmax = 0.f, bmax = 0.f;
  {
    // (clamped addresses, not predicated loads: the six loads leave together - predicated, each 64-column group waited for
    // its own round trip at the head of every wavefront's life)
    float wv[NR], bv[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) { const int cc = min(lane + 64 * r, D - 1); wv[r] = a.tw[cc]; bv[r] = a.tb[cc]; }
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const int c = lane + 64 * r;
      const float w = c < D ? wv[r] : 0.f, b = c < D ? bv[r] : 0.f;
      s_tw[c] = w; s_tb[c] = b;
      wmax = fmaxf(wmax, fabsf(w)); bmax = fmaxf(bmax, fabsf(b));
    }
  }
  sp_raw = min(max(sp_raw, 0), a.N - 1);                           // (behind the members' count the list holds anything)
  int sp_slot = a.qk_row[sp_raw], sp_cnt = a.run_cnt[sp_raw];
  const int M = a.seg_ptr[nr_rows];                                // members = instances that sit on a real node
  const int n_chunks = (M + RUN_CHUNK - 1) / RUN_CHUNK;
  wmax = pfo_wave_max(wmax); bmax = pfo_wave_max(bmax);
  // clamped column of this lane in each 64-column group, and the edge column
  int colr[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) colr[r] = min(lane + 64 * r, D - 1);
  const int cole = min(lane, max(Ef, 1) - 1);
  const float keep_scale = a.dropout_p > 0.f ? 1.f / (1.f - a.dropout_p) : 1.f;
  float* const d_nbr_x = a.d_nbr + (int64_t)(__builtin_amdgcn_s_getreg(6164) & (a.d_nbr_nrep - 1)) * a.d_nbr_rep;
  const uint64_t rng_off = a.offset + (a.offset_dev ? *a.offset_dev : 0ull);
  const float* const nbr_tab = a.nbr_tab;
  const float* const edge_feat = a.edge_feat;
  const uint32_t nbr_ld = (uint32_t)a.nbr_ld;
  // the staging image and the LDS-DMA that fills it (one member at a time; the issuing wavefront's vmcnt covers it)
  typedef __attribute__((address_space(1))) const void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  const uint32_t row_bytes = (uint32_t)(H * Cp) * 4u;             // a multiple of 16 (Cp is a multiple of 4)
  const float* const st_dc = reinterpret_cast<const float*>(s_stage);
  const float* const st_cx = reinterpret_cast<const float*>(s_stage + row_bytes);
  const float* const st_qk = reinterpret_cast<const float*>(s_stage + 2 * row_bytes);
  unsigned char* const st_meta = s_stage + 3 * row_bytes;        // (4 + H) arrays of K words: ids, table rows, edge ids, dt, weights
  // COUNTED WAIT (round 6): vmcnt retires in issue order, so "this member's image has landed" needs only the operations issued
  // BEFORE its DMAs to be done; the row atomics and row-sum stores of a flush behind them (an atomic stays in vmcnt for ~3 000
  // cycles with every CU adding) may stay in flight across the next member's set-up and walk.  since_stage counts them - only
  // statements that surely issue one instruction each; too low a count only makes the wait stricter.
  int since_stage = 0;
  int stage_count = 0;          // DMA instructions issued by stage() so far (every one has active lanes)
  const bool inject = a.keep_inject != nullptr && a.dropout_p > 0.f;
  auto stage = [&](int64_t n, int slot) {
    RUNS_LATE_ARGS(ka);
    const char* g_dc = reinterpret_cast<const char*>(ka->dctx + n * H * Cp);
    const char* g_cx = reinterpret_cast<const char*>(ka->ctx + n * H * Cp);
    const char* g_qk = reinterpret_cast<const char*>(ka->QK + (int64_t)slot * ka->qk_ld);
    // (the lane offset is made opaque: hoisted out of the member loop, the per-lane addresses of nine loads would be kept alive
    // as 64-bit register pairs across the key walk - the kernel sits at its register limit, they went to scratch)
    uint32_t lo = (uint32_t)lane * 4u;
    asm volatile("" : "+v"(lo));
    for (uint32_t k = 0; k * 1024u < row_bytes; ++k) {
      const uint32_t off = k * 1024u + lo * 4u;
      if (off < row_bytes) {                                     // lanes behind the row's end stay off: they write nothing
        __builtin_amdgcn_global_load_lds((gptr_t)(g_dc + off), (lptr_t)(s_stage + k * 1024u), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)(g_cx + off), (lptr_t)(s_stage + row_bytes + k * 1024u), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)(g_qk + off), (lptr_t)(s_stage + 2 * row_bytes + k * 1024u), 16, 0, 0);
      }
      stage_count += 3;                                          // (k * 1024 < row_bytes: lane 0 is on)
    }
    if (lane < K) {
      // wave-uniform row starts + ONE 32-bit lane offset (the scalar-base form of the load: no per-lane 64-bit pointers to keep)
      const uint32_t kb = (uint32_t)K * 4u;
      const int64_t s0 = n * K;
      __builtin_amdgcn_global_load_lds((gptr_t)(reinterpret_cast<const char*>(ka->nbr_ids + s0) + lo), (lptr_t)(st_meta), 4, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)(reinterpret_cast<const char*>(ka->nbr_row + s0) + lo), (lptr_t)(st_meta + kb), 4, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)(reinterpret_cast<const char*>(ka->eidx + s0) + lo), (lptr_t)(st_meta + 2 * kb), 4, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)(reinterpret_cast<const char*>(ka->dt + s0) + lo), (lptr_t)(st_meta + 3 * kb), 4, 0, 0);
#pragma unroll
      for (int h = 0; h < H; ++h)
        __builtin_amdgcn_global_load_lds((gptr_t)(reinterpret_cast<const char*>(ka->attw + (n * H + h) * K) + lo), (lptr_t)(st_meta + (4 + h) * kb), 4, 0, 0);
      // injected dropout decisions (parity tests) ride in the image too, one byte per slot (a dword of LDS each): a register-bound load of them in
      // the member's set-up put the compiler's s_waitcnt vmcnt(0) for it - taken whether or not the load ran - right behind
      // these DMAs: every member waited out the staging round trip it was meant to walk beside
      if (inject) __builtin_amdgcn_global_load_lds((gptr_t)(reinterpret_cast<const char*>(ka->keep_inject + s0) + (lo >> 2)), (lptr_t)(st_meta + (4 + H) * kb), 1, 0, 0);
    }
    stage_count += 4 + H + (inject ? 1 : 0);                     // (K >= 1: lane 0 is on)
    since_stage = 0;
  };

  // Workgroups go to the XCDs round-robin by their id.  Consecutive chunks hold members of the same table row or of
  // neighbouring ones (the list is ordered by row): with xcd_g = G > 0 every XCD takes G consecutive chunks at a time (block b ->
  // chunk (b / 8G) 8G + (b mod 8) G + (b mod 8G) / 8), so the shifted neighbour lists of one node's members, its query row and
  // the rows its atomics land on stay in ONE L2 instead of being fetched by up to eight.
  // A wavefront takes a UNIT of RUN_CPW consecutive chunks.  The chunk stays the scope of a run (its rows leave at the chunk's
  // end at the latest) and of the time-encoder partial sums; the unit is the scope of the prologue and of the member staging.
  const int n_units = (n_chunks + RUN_CPW - 1) / RUN_CPW;
  const int n_walk = G8 > 0 ? (n_units + G8 - 1) / G8 * G8 : n_units;
  for (int blk = blockIdx.x; blk < n_walk; blk += gridDim.x) {
    const int unit = unit_of(blk);
    if (blk != (int)blockIdx.x) {                                  // (a capped grid only: later units load their prologue here)
      sp_raw = min(max(spec_member(unit), 0), a.N - 1);
      sp_slot = a.qk_row[sp_raw]; sp_cnt = a.run_cnt[sp_raw];
    }
    if (unit >= n_units) continue;
    // prologue: the members' instance ids, query rows and history counts, one lane per member, and the first member's rows
    // on their way
    const int u0 = unit * UM;
    const int u_end = min(M, u0 + UM);
    {
      const bool ch_on = lane < UM && u0 + lane < u_end;
      const int ch_n = ch_on ? sp_raw : 0;
      const int ch_slot = ch_on ? sp_slot : -1;
      const int ch_cnt = ch_on ? sp_cnt : 0;
      if (lane < UM) { s_ch[0][lane] = ch_n; s_ch[1][lane] = ch_slot; s_ch[2][lane] = ch_cnt; }   // (one wavefront: no barrier)
      if (u0 < u_end) stage(rl_i(ch_n, 0), rl_i(ch_slot, 0));
    }
    auto ch_get = [&](int which, int i) { return __builtin_amdgcn_readfirstlane(s_ch[which][i]); };
    for (int chunk = unit * RUN_CPW; chunk < min(n_chunks, (unit + 1) * RUN_CPW); ++chunk) {
    const int m0 = chunk * RUN_CHUNK;
    const int m_end = min(M, m0 + RUN_CHUNK);
    float dwc[NR], dbc[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) { dwc[r] = 0.f; dbc[r] = 0.f; }
    // the current group: lane q <-> history entry cnt0 - K + q of the node run_slot
    int run_slot = -1, run_cnt0 = 0, run_rows = 0, run_len = 0, run_first = 0;
    unsigned long long run_valid = 0ull;                           // lanes whose history entry exists and was seen
    // The query-side gradient rows [dqk' (node | edge | time)] of consecutive members that sit on ONE table row are summed
    // here, in registers, and stored once (row acc_m, flagged live): the per-row sum pass that follows (segsum) adds the
    // members of a row anyway, and a chunk of 4 members holds ~2.5 per row at C2 - less than half the rows are written / re-read
    float dqn[H][NR], dqt[H][NR], dqe[H];
    int acc_m = -1;
    auto acc_reset = [&]() {
#pragma unroll
      for (int h = 0; h < H; ++h) {
#pragma unroll
        for (int r = 0; r < NR; ++r) { dqn[h][r] = 0.f; dqt[h][r] = 0.f; }
        dqe[h] = 0.f;
      }
    };
    // (acc_store, flush and stage address their rows as a wave-uniform base + ONE 32-bit lane offset that is opaque to the
    // optimiser: hoisted out of the member loop, their per-lane 64-bit element offsets lived across the key walk - a dozen
    // register pairs at a kernel that sits on its register limit, i.e. scratch reloads in front of every store)
    auto acc_store = [&]() {
      RUNS_LATE_ARGS(ka);
      if (acc_m >= 0 && !DET && ka->dq_rows) {
        // added straight into the table row's sum (the members of a row sit in several chunks: float atomics, ~20 k rows of
        // H Cp floats per launch at C2 beside the ~275 k neighbour rows)
        char* out = reinterpret_cast<char*>(ka->dq_rows + (int64_t)run_slot * ka->dq_ld);
        uint32_t lo = (uint32_t)lane * 4u;
        asm volatile("" : "+v"(lo));
#pragma unroll
        for (int h = 0; h < H; ++h) {
#pragma unroll
          for (int r = 0; r < NR; ++r) {
            if (lo < (uint32_t)(D - 64 * r) * 4u) {
              atomicAdd(reinterpret_cast<float*>(out + (uint32_t)(h * Cp + 64 * r) * 4u + lo), dqn[h][r]);
              atomicAdd(reinterpret_cast<float*>(out + (uint32_t)(h * Cp + D + Ef + 64 * r) * 4u + lo), dqt[h][r]);
            }
          }
          if (lo < (uint32_t)Ef * 4u) atomicAdd(reinterpret_cast<float*>(out + (uint32_t)(h * Cp + D) * 4u + lo), dqe[h]);
        }
        since_stage += H * 2 * NR + (Ef > 0 ? H : 0);
      } else if (acc_m >= 0) {
        char* out = reinterpret_cast<char*>(ka->dQK + (int64_t)acc_m * H * Cp);      // row m, not n: the per-row sums then stream contiguous rows
        uint32_t lo = (uint32_t)lane * 4u;
        asm volatile("" : "+v"(lo));
#pragma unroll
        for (int h = 0; h < H; ++h) {
#pragma unroll
          for (int r = 0; r < NR; ++r) {
            if (lo < (uint32_t)(D - 64 * r) * 4u) {              // column lane + 64 r < D (64 r < D for every r < NR)
              *reinterpret_cast<float*>(out + (uint32_t)(h * Cp + 64 * r) * 4u + lo) = dqn[h][r];
              *reinterpret_cast<float*>(out + (uint32_t)(h * Cp + D + Ef + 64 * r) * 4u + lo) = dqt[h][r];
            }
          }
          if (lo < (uint32_t)Ef * 4u) *reinterpret_cast<float*>(out + (uint32_t)(h * Cp + D) * 4u + lo) = dqe[h];
          if (lo < (uint32_t)(Cp - C) * 4u) *reinterpret_cast<float*>(out + (uint32_t)(h * Cp + C) * 4u + lo) = 0.f;
        }
        if (lane == 0) ka->dqk_live[acc_m] = 1;
        since_stage += H * (2 * NR + 1) + (Ef > 0 ? H : 0) + 1;
      }
      acc_m = -1;
      acc_reset();
    };
    acc_reset();
    float sBr[H];                                                  // lane q: sum of cB over the group's instances that hold entry q
#pragma unroll
    for (int h = 0; h < H; ++h) sBr[h] = 0.f;
    auto flush = [&]() {                                           // the run's rows: one float atomic per element
      if (run_valid != 0ull && run_len > 0) {
        RUNS_LATE_ARGS(ka);
        const char* qk = reinterpret_cast<const char*>(ka->QK + (int64_t)run_slot * ka->qk_ld);
        uint32_t lo = (uint32_t)lane * 4u;
        asm volatile("" : "+v"(lo));
        float qn[H][NR], g[RUN_CHUNK][H][NR];
#pragma unroll
        for (int h = 0; h < H; ++h)
#pragma unroll
          for (int r = 0; r < NR; ++r)
            qn[h][r] = lo < (uint32_t)(D - 64 * r) * 4u ? *reinterpret_cast<const float*>(qk + (uint32_t)(h * Cp + 64 * r) * 4u + lo) : 0.f;
        float cAr[RUN_CHUNK][H];                                   // lane q: cA of the slot instance i has on history entry q
#pragma unroll
        for (int i = 0; i < RUN_CHUNK; ++i) {
          const bool on = i < run_len;                             // wave-uniform
          const char* dc = reinterpret_cast<const char*>(ka->dctx + (int64_t)ch_get(0, (on ? run_first + i : run_first) - u0) * H * Cp);
          const int sl = lane - (on ? s_delta[i] : 0);             // the instance's slot on this lane (may lie outside [0, K): zero)
#pragma unroll
          for (int h = 0; h < H; ++h) {
            cAr[i][h] = (on && sl >= 0) ? s_cA[i][h][sl < 0 ? 0 : sl] : 0.f;
#pragma unroll
            for (int r = 0; r < NR; ++r)
              g[i][h][r] = (on && lo < (uint32_t)(D - 64 * r) * 4u) ? *reinterpret_cast<const float*>(dc + (uint32_t)(h * Cp + 64 * r) * 4u + lo) : 0.f;
          }
        }
        unsigned long long vmask = run_valid;
        while (vmask) {
          const int j = __ffsll((long long)vmask) - 1;
          vmask &= vmask - 1ull;
          const int64_t drow = (int64_t)rl_i(run_rows, j) * ka->d_nbr_ld;
          char* dst = reinterpret_cast<char*>(d_nbr_x + drow);
          float row[NR];
#pragma unroll
          for (int r = 0; r < NR; ++r) row[r] = 0.f;
#pragma unroll
          for (int h = 0; h < H; ++h) {
            const float sb = rl_f(sBr[h], j);
#pragma unroll
            for (int r = 0; r < NR; ++r) row[r] = fmaf(sb, qn[h][r], row[r]);
#pragma unroll
            for (int i = 0; i < RUN_CHUNK; ++i) {
              const float ca = rl_f(cAr[i][h], j);                 // 0 beyond the run's length
#pragma unroll
              for (int r = 0; r < NR; ++r) row[r] = fmaf(ca, g[i][h][r], row[r]);
            }
          }
#pragma unroll
          for (int r = 0; r < NR; ++r) {
            if ((r < NR - 1 || lo < (uint32_t)(D - 64 * r) * 4u) && a.abl != 2) {
              if (DET) det_add(ka->d_nbr, drow + lane + 64 * r, row[r]); else atomicAdd(reinterpret_cast<float*>(dst + (uint32_t)(256 * r) + lo), row[r]);
            }
          }
          if (a.abl != 2) since_stage += NR;
        }
      }
#pragma unroll
      for (int h = 0; h < H; ++h) sBr[h] = 0.f;
      run_len = 0;
    };

#if RUNS_STAMPS
    st_chunks += 1;
#endif
    for (int m = m0; m < m_end; ++m) {
#if RUNS_STAMPS
      unsigned long long st0 = STAMP();
      st_members += 1;
#endif
      const int64_t n = ch_get(0, m - u0);
      const int slot = ch_get(1, m - u0);
      const int cnt_n = ch_get(2, m - u0);
#if RUNS_CW
      pfo_wait_allowed(min(since_stage, 48));                    // this member's staging image has landed (counted wait, above)
#else
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this member's staging image has landed
#endif
      const bool inK = lane < K;
      const int* const mi = reinterpret_cast<const int*>(st_meta);
      const int my_id = inK ? mi[lane] : 0;
      const int my_row = inK ? mi[K + lane] : 0;
      const int my_e = inK ? mi[2 * K + lane] : 0;
      const float my_dt = inK ? __int_as_float(mi[3 * K + lane]) : 0.f;
      const unsigned long long valid = __ballot(inK && my_id != 0);
      int delta = cnt_n - run_cnt0;                              // members of a row arrive by ascending count
      if (slot != run_slot || delta < 0 || delta > 64 - K) {     // another node, or the lanes run out: the group's rows leave
#if RUNS_STAMPS
        const unsigned long long sa = STAMP();
#endif
        acc_store();                                             // (the query-side sums too: never live across a flush)
#if RUNS_STAMPS
        const unsigned long long sb = STAMP();
#endif
        flush();
#if RUNS_STAMPS
        const unsigned long long sc = STAMP();
        st_store += sb - sa; st_flush += sc - sb; st0 += sc - sa;
#endif
        run_slot = slot; run_cnt0 = cnt_n; run_rows = 0; run_valid = 0ull;
        delta = 0;
      }
      {
        // this instance's slots move to lanes j + delta: row indices and validity join the group's
        const int rows_sh = __builtin_amdgcn_ds_bpermute((lane - delta) << 2, my_row);
        const unsigned long long vs = valid << delta;
        run_rows = ((vs >> lane) & 1ull) ? rows_sh : run_rows;
        run_valid |= vs;
      }
      if (lane == 0 && (DET || !a.dq_rows)) a.dqk_live[m] = 0;  // (set when this member's position receives a stored sum)
      if (valid == 0ull) {                                       // no neighbour: nothing to add to the row's sums
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (m + 1 < u_end) stage(ch_get(0, m + 1 - u0), ch_get(1, m + 1 - u0));
        continue;
      }
      acc_m = m;
      if (run_len == 0) run_first = m;                           // the group's instances with a neighbour are consecutive members
      const int run_i = run_len;                                 // (an instance without history has count 0: first of its row)
      run_len += 1;
      s_delta[run_i] = delta;
      float qt[H][NR], gn[H][NR], gt[H][NR], ge[H], tds[2 * H];
      float dsb[H], cxs[H], my_a[H];
#pragma unroll
      for (int h = 0; h < H; ++h) {
        float part = 0.f;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
          const int c = lane + 64 * r;
          const bool ok = c < D;
          qt[h][r] = ok ? st_qk[h * Cp + D + Ef + c] : 0.f;
          gn[h][r] = ok ? st_dc[h * Cp + c] : 0.f;
          gt[h][r] = ok ? st_dc[h * Cp + D + Ef + c] : 0.f;
          if (ok) part = fmaf(gn[h][r], st_cx[h * Cp + c], fmaf(gt[h][r], st_cx[h * Cp + D + Ef + c], part));
        }
        ge[h] = lane < Ef ? st_dc[h * Cp + D + lane] : 0.f;
        if (lane < Ef) part = fmaf(ge[h], st_cx[h * Cp + D + lane], part);
        tds[h] = part;
        dsb[h] = st_dc[h * Cp + C];          // d loss / d (sum_j a'_jh): one address for the wavefront (an LDS broadcast)
        cxs[h] = st_cx[h * Cp + C];
        my_a[h] = inK ? __int_as_float(mi[(4 + h) * K + lane]) : 0.f;
      }
      // injected dropout decisions: read HERE, with the rest of the image, in front of the next member's staging (a sub-dword
      // LDS-DMA load writes one zero-extended DWORD per lane: the bytes sit at a stride of four)
      const unsigned keep_img = (inject && inK) ? ((unsigned)mi[(4 + H) * K + lane] & 0xFFu) : 0xFu;
      // the image is free again: the next member's rows start their trip from inside the walk (RUNS_ASM_GATHER: behind the
      // first two pairs' gathers) or right here, and land while this member is walked
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#if !RUNS_ASM_GATHER
      if (m + 1 < u_end) stage(ch_get(0, m + 1 - u0), ch_get(1, m + 1 - u0));
#endif
      pfo_wave_sum_scalar_n<H>(reinterpret_cast<float(&)[H]>(tds));     // delta_h = dctx_h . ctx_h (+ the extra column below)
      float t[H];
#pragma unroll
      for (int h = 0; h < H; ++h) {
        dsb[h] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(dsb[h])));
        t[h] = fmaf(dsb[h], __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(cxs[h]))), tds[h]);
      }
      // Per slot, once per member: the dropout multiplier ks_jh (0 or 1 / (1 - p)) and with it cA_jh = a_jh ks_jh - the
      // coefficient of d ctx'_h in the key-side rows does not depend on the backward at all: it goes to the group's LDS table
      // here (one store per head) instead of from inside the key loop (compare + exec-masked store per key), and the loop reads
      // ks with one v_readlane per head instead of unpacking a bit mask (scalar and / compare / select + a vector select)
      float my_ks[H];
      {
        const unsigned keep = inject ? keep_img : attn_keep_bits(a.seed, rng_off, n, lane, a.dropout_p);
#pragma unroll
        for (int h = 0; h < H; ++h) {
          my_ks[h] = ((keep >> h) & 1u) ? keep_scale : 0.f;
          s_cA[run_i][h][lane] = ((valid >> lane) & 1ull) ? my_a[h] * my_ks[h] : 0.f;
        }
      }
      // one range test per instance: |fma(dt, w, b)| <= max|dt| max|w| + max|b| < 2e7 -> the fp32 reduction holds for every key
      const bool fast = fmaf(pfo_wave_max(fabsf(my_dt)), wmax, bmax) < 2.0e7f;

      auto walk = [&](auto fast_c) {
      constexpr bool FAST = decltype(fast_c)::value;
      unsigned long long vm = valid;
      int jsA[KC_RUNS], jsB[KC_RUNS];
      float knA[KC_RUNS][NR], keA[KC_RUNS], knB[KC_RUNS][NR], keB[KC_RUNS];
      auto pick = [&](int (&jj)[KC_RUNS]) {
#pragma unroll
        for (int c = 0; c < KC_RUNS; ++c) {
          jj[c] = vm ? (__ffsll((long long)vm) - 1) : -1;
          vm &= vm - 1ull;
        }
      };
#if RUNS_ASM_GATHER
      // ASM GATHERS (round 6).  While an LDS-DMA is in flight the compiler waits for vmcnt(0) at the first use of any
      // register-bound load result - the first pair of every walk waited out the staging DMAs issued in front of it, i.e.
      // the round trip the staging was built to hide (ISA of the round-5 kernel).  The key gathers are issued by inline
      // asm instead, invisible to the compiler's counter, and waited for by a COUNTED vmcnt: vseq counts the gather loads
      // and staging DMAs issued; a pair is ready when at most (vseq - its vseq) younger operations are outstanding.  The
      // staging of the next member is issued BEHIND the first two pairs' gathers: vmcnt retires in order, so only pairs
      // gathered behind it wait for it - two pairs of arithmetic later.
      int vseq = 0;
      const int L_pair = KC_RUNS * (NR + (Ef > 0 ? 1 : 0));
      auto gather = [&](const int (&jj)[KC_RUNS], float (&kk)[KC_RUNS][NR], float (&ee)[KC_RUNS]) {
#pragma unroll
        for (int c = 0; c < KC_RUNS; ++c) {
          const int j = jj[c] < 0 ? 0 : jj[c];                   // an absent key re-reads slot 0's row: never used
          const float* src = nbr_tab + (uint32_t)rl_i(my_row, j) * nbr_ld;
          const uint32_t e = (uint32_t)rl_i(my_e, j);
#pragma unroll
          for (int r = 0; r < NR; ++r) {
            const float* pr = src + colr[r];
            asm volatile("global_load_dword %0, %1, off" : "=v"(kk[c][r]) : "v"(pr));
          }
          if (Ef > 0) {
            const float* pe = edge_feat + e * (uint32_t)Ef + (uint32_t)cole;
            asm volatile("global_load_dword %0, %1, off" : "=v"(ee[c]) : "v"(pe));
          } else {
            ee[c] = 0.f;
          }
        }
        vseq += L_pair;
      };
      // the pair gathered when vseq stood at `at` has landed; its registers are tied to the wait (no use is scheduled above it)
      auto landed = [&](int at, float (&kk)[KC_RUNS][NR], float (&ee)[KC_RUNS]) {
        pfo_wait_allowed_exact(vseq - at);
#pragma unroll
        for (int c = 0; c < KC_RUNS; ++c) {
#pragma unroll
          for (int r = 0; r < NR; ++r) asm volatile("" : "+v"(kk[c][r]));
          asm volatile("" : "+v"(ee[c]));
        }
      };
#else
      auto gather = [&](const int (&jj)[KC_RUNS], float (&kk)[KC_RUNS][NR], float (&ee)[KC_RUNS]) {
#pragma unroll
        for (int c = 0; c < KC_RUNS; ++c) {
          const int j = jj[c] < 0 ? 0 : jj[c];                   // an absent key re-reads slot 0's row: never used
          const float* src = nbr_tab + (uint32_t)rl_i(my_row, j) * nbr_ld;
          const uint32_t e = (uint32_t)rl_i(my_e, j);
#pragma unroll
          for (int r = 0; r < NR; ++r) kk[c][r] = src[colr[r]];
          ee[c] = Ef > 0 ? edge_feat[e * (uint32_t)Ef + (uint32_t)cole] : 0.f;
        }
      };
#endif
      auto process = [&](const int (&js)[KC_RUNS], const float (&kn)[KC_RUNS][NR], const float (&ke)[KC_RUNS]) {
        float kt[KC_RUNS][NR], ks[KC_RUNS][NR], dtv[KC_RUNS], part[KC_RUNS * H];
#pragma unroll
        for (int c = 0; c < KC_RUNS; ++c) {
          dtv[c] = rl_f(my_dt, js[c] < 0 ? 0 : js[c]);
#pragma unroll
          for (int r = 0; r < NR; ++r) {
            const float arg = pfo_time_arg(dtv[c], s_tw[lane + 64 * r], s_tb[lane + 64 * r]);
            const float u = FAST ? pfo_revolutions_fast(arg) : pfo_revolutions(arg);
            ks[c][r] = __builtin_amdgcn_sinf(u);
            kt[c][r] = __builtin_amdgcn_cosf(u);
          }
#pragma unroll
          for (int h = 0; h < H; ++h) {
            float pp = ke[c] * ge[h];
#pragma unroll
            for (int r = 0; r < NR; ++r) pp = fmaf(kn[c][r], gn[h][r], fmaf(kt[c][r], gt[h][r], pp));
            part[c * H + h] = pp;
          }
        }
        pfo_wave_sum_scalar_n<KC_RUNS * H>(part);
#pragma unroll
        for (int c = 0; c < KC_RUNS; ++c) {
          if (js[c] < 0) continue;
          const bool mine = lane == js[c] + delta;              // the lane of this key's history entry
          float cA[H], cB[H];
#pragma unroll
          for (int h = 0; h < H; ++h) {
            const float ks_h = rl_f(my_ks[h], js[c]);
            const float da = (part[c * H + h] + dsb[h]) * ks_h;
            const float aj = rl_f(my_a[h], js[c]);
            const float dscore = aj * (da - t[h]);
            cA[h] = aj * ks_h;
            cB[h] = dscore * a.scale;
            sBr[h] = mine ? sBr[h] + cB[h] : sBr[h];               // key js[c]'s share of the run's rows (flush)
#pragma unroll
            for (int r = 0; r < NR; ++r) {
              dqn[h][r] = fmaf(cB[h], kn[c][r], dqn[h][r]);
              dqt[h][r] = fmaf(cB[h], kt[c][r], dqt[h][r]);
            }
            dqe[h] = fmaf(cB[h], ke[c], dqe[h]);
          }
#pragma unroll
          for (int r = 0; r < NR; ++r) {
            float dkt = 0.f;
#pragma unroll
            for (int h = 0; h < H; ++h) dkt = fmaf(cA[h], gt[h][r], fmaf(cB[h], qt[h][r], dkt));
            const float gsin = -ks[c][r] * dkt;                 // d/d(arg) cos(arg) = -sin(arg); sin(0) = 0 on lanes beyond D
            dwc[r] = fmaf(gsin, dtv[c], dwc[r]);
            dbc[r] += gsin;
          }
        }
      };
#if RUNS_ASM_GATHER
      pick(jsA);
      gather(jsA, knA, keA);
      int atA = vseq, atB = 0;
      bool first = true;
      while (true) {
        pick(jsB);
        if (jsB[0] >= 0) { gather(jsB, knB, keB); atB = vseq; }
        if (first) {
          // the next member's image, behind the gathers of this member's first two pairs (stage() counts its DMAs into vseq)
          first = false;
          if (m + 1 < u_end) { const int before = stage_count; stage(ch_get(0, m + 1 - u0), ch_get(1, m + 1 - u0)); vseq += stage_count - before; }
        }
        landed(atA, knA, keA);
        process(jsA, knA, keA);
        if (jsB[0] < 0) break;
        pick(jsA);
        if (jsA[0] >= 0) { gather(jsA, knA, keA); atA = vseq; }
        landed(atB, knB, keB);
        process(jsB, knB, keB);
        if (jsA[0] < 0) break;
      }
#else
      pick(jsA);
      gather(jsA, knA, keA);
      while (true) {
        pick(jsB);
        if (jsB[0] >= 0) gather(jsB, knB, keB);
        process(jsA, knA, keA);
        if (jsB[0] < 0) break;
        pick(jsA);
        if (jsA[0] >= 0) gather(jsA, knA, keA);
        process(jsB, knB, keB);
        if (jsA[0] < 0) break;
      }
#endif
      };
#if RUNS_STAMPS
      const unsigned long long st1 = STAMP();
#endif
      if (fast) walk(std::true_type{}); else walk(std::false_type{});
#if RUNS_STAMPS
      const unsigned long long st2 = STAMP();
      st_setup += st1 - st0; st_walk += st2 - st1;
#endif
    }
#if RUNS_STAMPS
    const unsigned long long se0 = STAMP();
#endif
    acc_store();                                                 // the chunk's last row sums
#if RUNS_STAMPS
    const unsigned long long se1 = STAMP();
#endif
    flush();                                                     // the chunk's last run
#if RUNS_STAMPS
    const unsigned long long se2 = STAMP();
    st_store += se1 - se0; st_flush += se2 - se1;
#endif
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const int c = lane + 64 * r;
      if (c < D) {
        if (DET) {
          a.dtime_slab[(int64_t)chunk * 2 * D + c] = (double)dwc[r];
          a.dtime_slab[(int64_t)chunk * 2 * D + D + c] = (double)dbc[r];
        } else {
          double* bin = a.dtime_part + (int64_t)(chunk & (ATTN_TIME_BINS - 1)) * 2 * D;
          atomicAdd(bin + c, (double)dwc[r]);
          atomicAdd(bin + D + c, (double)dbc[r]);
        }
      }
    }
    }   // chunks of the unit
    // deterministic mode: every slab row is written - the chunks of this unit beyond the members' count own zero rows
    if (DET)
      for (int chunk = max(n_chunks, unit * RUN_CPW); chunk < (unit + 1) * RUN_CPW; ++chunk)
        for (int c = lane; c < 2 * D; c += 64) a.dtime_slab[(int64_t)chunk * 2 * D + c] = 0.0;
  }
  // (the deterministic launch has one workgroup per POSSIBLE unit; those beyond the members' count own zero rows too)
  if (DET && (int)blockIdx.x >= n_units)
    for (int chunk = (int)blockIdx.x * RUN_CPW; chunk < ((int)blockIdx.x + 1) * RUN_CPW; ++chunk)
      for (int c = lane; c < 2 * D; c += 64) a.dtime_slab[(int64_t)chunk * 2 * D + c] = 0.0;
  if (clk_on && lane == 0) pfo_clock_end(clk, g_attn_clock[1]);
#if RUNS_STAMPS
  if (lane < 7) {
    const unsigned long long tot_ = STAMP() - st_begin;
    const unsigned long long v_ = lane == 0 ? tot_ : lane == 1 ? st_setup : lane == 2 ? st_walk : lane == 3 ? st_flush : lane == 4 ? st_store : lane == 5 ? st_members : st_chunks;
    pfo_runs_stamps[(blockIdx.x % RUNS_STAMP_ROWS) * 8 + lane] += v_;
  }
#endif
}

int64_t pfo_attn_bwd_det_parts(int64_t N) {
  return std::max<int64_t>(pfo_align_up(pfo_ceil_div(N, RUN_CHUNK), RUN_CPW), std::min<int64_t>(ATTN_BWD_MAX_BLOCKS, pfo_ceil_div(N, 4)));
}

static void to_dev(const PfoAttn& a, AttnDev& d) {
  d.N = a.N; d.K = a.K; d.D = a.D; d.Ef = a.Ef; d.H = a.H; d.Cp = a.Cp;
  d.QK = a.QK; d.qk_row = a.qk_row; d.qk_ld = a.qk_ld > 0 ? a.qk_ld : (int64_t)a.H * a.Cp; d.nbr_tab = a.nbr_tab; d.nbr_ld = a.nbr_ld; d.nbr_row = a.nbr_row; d.nbr_row_base = a.nbr_row_base; d.nbr_relu = a.nbr_relu;
  d.nbr_ids = a.nbr_ids; d.edge_feat = a.edge_feat; d.eidx = a.eidx; d.dt = a.dt; d.tw = a.tw; d.tb = a.tb;
  d.scale = a.scale; d.dropout_p = a.dropout_p; d.seed = a.seed; d.offset = a.offset; d.offset_dev = a.offset_dev; d.keep_inject = a.keep_inject;
  static const int abl = getenv("PFO_ATTN_ABL") ? atoi(getenv("PFO_ATTN_ABL")) : 0;
  d.abl = abl;
  d.xcd_g = 0;
  d.ctx = a.ctx; d.attw = a.attw; d.inv = a.inv;
  d.dctx = a.dctx; d.dQK = a.dQK; d.d_nbr = a.d_nbr; d.d_nbr_ld = a.d_nbr_ld; d.d_nbr_rep = a.d_nbr_rep; d.d_nbr_nrep = a.d_nbr_nrep > 0 ? a.d_nbr_nrep : 1;
  d.dtime_part = a.dtime_part;
  d.det = a.det; d.dtime_slab = a.dtime_slab; d.dqk_live = a.dqk_live;
  d.dq_rows = a.det ? nullptr : a.dq_rows; d.dq_ld = a.dq_ld;
  d.members = a.members; d.seg_ptr = a.seg_ptr; d.n_rows = a.n_rows; d.run_cnt = a.run_cnt;
}

static int check_common(const PfoAttn& a) {
  PFO_REQUIRE(a.N > 0 && a.K >= 1 && a.K <= PFO_MAX_NEIGHBORS, "bad N / K");
  PFO_REQUIRE(a.D >= 1 && a.D <= 256, "D must be <= 256");
  PFO_REQUIRE(a.Ef >= 0 && a.Ef <= 64, "Ef must be <= 64");
  PFO_REQUIRE(a.H == 1 || a.H == 2 || a.H == 4, "n_heads must be 1, 2 or 4");
  PFO_REQUIRE(a.QK && a.nbr_tab && a.nbr_ids && a.eidx && a.dt && a.tw && a.tb && a.ctx && a.attw && a.inv, "null input");
  PFO_REQUIRE(a.Ef == 0 || a.edge_feat, "null edge features");
  PFO_REQUIRE(a.dropout_p >= 0.f && a.dropout_p < 1.f, "dropout must be in [0, 1)");
  PFO_REQUIRE(a.Cp >= 2 * a.D + a.Ef + 2 && (a.Cp % 4) == 0, "Cp must hold C + 2 columns and be a multiple of 4");
  // the kernels address the gathered rows with 32-bit element offsets
  PFO_REQUIRE(a.nbr_rows > 0 && (uint64_t)a.nbr_rows * (uint64_t)a.nbr_ld < (1ull << 32), "neighbour table too large for 32-bit row offsets");
  PFO_REQUIRE(a.Ef == 0 || (a.edge_rows > 0 && (uint64_t)a.edge_rows * (uint64_t)a.Ef < (1ull << 32)), "edge feature table too large for 32-bit row offsets");
  return PFO_OK;
}

#define ATTN_DISPATCH(KERNEL, grid)                                                                           \
  {                                                                                                           \
    const int NRv = (a.D + 63) / 64;                                                                          \
    const dim3 g((unsigned)(grid)), b(256);                                                                   \
    bool done = true;                                                                                         \
    switch (NRv * 8 + a.H) {                                                                                  \
      case 1 * 8 + 1: PFO_KLAUNCH((KERNEL<1, 1>), g, b, 0, stream, d); break;                          \
      case 1 * 8 + 2: PFO_KLAUNCH((KERNEL<1, 2>), g, b, 0, stream, d); break;                          \
      case 1 * 8 + 4: PFO_KLAUNCH((KERNEL<1, 4>), g, b, 0, stream, d); break;                          \
      case 2 * 8 + 1: PFO_KLAUNCH((KERNEL<2, 1>), g, b, 0, stream, d); break;                          \
      case 2 * 8 + 2: PFO_KLAUNCH((KERNEL<2, 2>), g, b, 0, stream, d); break;                          \
      case 2 * 8 + 4: PFO_KLAUNCH((KERNEL<2, 4>), g, b, 0, stream, d); break;                          \
      case 3 * 8 + 1: PFO_KLAUNCH((KERNEL<3, 1>), g, b, 0, stream, d); break;                          \
      case 3 * 8 + 2: PFO_KLAUNCH((KERNEL<3, 2>), g, b, 0, stream, d); break;                          \
      case 3 * 8 + 4: PFO_KLAUNCH((KERNEL<3, 4>), g, b, 0, stream, d); break;                          \
      case 4 * 8 + 1: PFO_KLAUNCH((KERNEL<4, 1>), g, b, 0, stream, d); break;                          \
      case 4 * 8 + 2: PFO_KLAUNCH((KERNEL<4, 2>), g, b, 0, stream, d); break;                          \
      case 4 * 8 + 4: PFO_KLAUNCH((KERNEL<4, 4>), g, b, 0, stream, d); break;                          \
      default: done = false;                                                                                  \
    }                                                                                                         \
    PFO_REQUIRE(done, "unsupported (D, H) combination");                                                      \
  }

// the ring forms: NR H <= 12 (attn_bwd_ring_ok)
#define ATTN_DISPATCH_RING(KERNEL, grid)                                                                      \
  {                                                                                                           \
    const int NRv = (a.D + 63) / 64;                                                                          \
    const dim3 g((unsigned)(grid)), b(256);                                                                   \
    bool done = true;                                                                                         \
    switch (NRv * 8 + a.H) {                                                                                  \
      case 1 * 8 + 1: PFO_KLAUNCH((KERNEL<1, 1>), g, b, 0, stream, d); break;                          \
      case 1 * 8 + 2: PFO_KLAUNCH((KERNEL<1, 2>), g, b, 0, stream, d); break;                          \
      case 1 * 8 + 4: PFO_KLAUNCH((KERNEL<1, 4>), g, b, 0, stream, d); break;                          \
      case 2 * 8 + 1: PFO_KLAUNCH((KERNEL<2, 1>), g, b, 0, stream, d); break;                          \
      case 2 * 8 + 2: PFO_KLAUNCH((KERNEL<2, 2>), g, b, 0, stream, d); break;                          \
      case 2 * 8 + 4: PFO_KLAUNCH((KERNEL<2, 4>), g, b, 0, stream, d); break;                          \
      case 3 * 8 + 1: PFO_KLAUNCH((KERNEL<3, 1>), g, b, 0, stream, d); break;                          \
      case 3 * 8 + 2: PFO_KLAUNCH((KERNEL<3, 2>), g, b, 0, stream, d); break;                          \
      case 3 * 8 + 4: PFO_KLAUNCH((KERNEL<3, 4>), g, b, 0, stream, d); break;                          \
      case 4 * 8 + 1: PFO_KLAUNCH((KERNEL<4, 1>), g, b, 0, stream, d); break;                          \
      case 4 * 8 + 2: PFO_KLAUNCH((KERNEL<4, 2>), g, b, 0, stream, d); break;                          \
      default: done = false;                                                                                  \
    }                                                                                                         \
    PFO_REQUIRE(done, "unsupported (D, H) combination");                                                      \
  }

// minimum wavefronts per SIMD the register allocation must allow: 3 where the kernel fits 168 VGPRs without spilling
#ifndef BWD_WAVES
#define BWD_WAVES(NR, H, DMODE) (((H) <= 2 && (NR) <= 3) ? ((DMODE) == 1 ? 2 : 3) : (((H) == 4 && (NR) == 4) ? 1 : 2))
#endif
template <int NR, int H> __global__ __launch_bounds__(256, BWD_WAVES(NR, H, 0)) void attn_bwd_ring_kernel_none(const AttnDev a) { attn_bwd_ring_body<NR, H, 0>(a); }
template <int NR, int H> __global__ __launch_bounds__(256, BWD_WAVES(NR, H, 2)) void attn_bwd_ring_kernel_direct(const AttnDev a) { attn_bwd_ring_body<NR, H, 2>(a); }
template <int NR, int H> __global__ __launch_bounds__(256, BWD_WAVES(NR, H, 0)) void attn_bwd_kernel_none(const AttnDev a) { attn_bwd_body<NR, H, 0>(a); }
template <int NR, int H> __global__ __launch_bounds__(256, BWD_WAVES(NR, H, 1)) void attn_bwd_kernel(const AttnDev a) { attn_bwd_body<NR, H, 1>(a); }
template <int NR, int H> __global__ __launch_bounds__(256, BWD_WAVES(NR, H, 2)) void attn_bwd_kernel_direct(const AttnDev a) { attn_bwd_body<NR, H, 2>(a); }
template <int NR, int H> __global__ __launch_bounds__(256, BWD_WAVES(NR, H, 1)) void attn_bwd_kernel_det(const AttnDev a) { attn_bwd_body<NR, H, 3>(a); }
int pfo_attn_fwd_launch(const PfoAttn& a, hipStream_t stream) {
  if (int rc = check_common(a)) return rc;
  AttnDev d;
  to_dev(a, d);
  // algorithmic bytes per instance: K gathered rows + edge feature + (eidx, dt, id), qk in, ctx + weights out
  const double C = 2.0 * a.D + a.Ef;
  const double bytes = (double)a.N * (a.K * (4.0 * a.D + 4.0 * a.Ef + 12.0) + 2.0 * a.H * C * 4.0 + 4.0 * a.H * a.K);
  pfo_prof_begin(stream);
  if (attn_fwd_ring_ok(a) && attn_fwd_pipe_ok(a)) {
    const int NRp = (a.D + 63) / 64;
    const dim3 g((unsigned)pfo_ceil_div(a.N, 4 * FWD_IPW)), b(256);
    const size_t lds = 4 * attn_fwd_pipe_wave_bytes(NRp, a.H, a.Cp, a.K);
    switch (NRp * 8 + a.H) {
      case 1 * 8 + 1: PFO_KLAUNCH((attn_fwd_pipe_kernel<1, 1>), g, b, lds, stream, d); break;
      case 1 * 8 + 2: PFO_KLAUNCH((attn_fwd_pipe_kernel<1, 2>), g, b, lds, stream, d); break;
      case 1 * 8 + 4: PFO_KLAUNCH((attn_fwd_pipe_kernel<1, 4>), g, b, lds, stream, d); break;
      case 2 * 8 + 1: PFO_KLAUNCH((attn_fwd_pipe_kernel<2, 1>), g, b, lds, stream, d); break;
      case 2 * 8 + 2: PFO_KLAUNCH((attn_fwd_pipe_kernel<2, 2>), g, b, lds, stream, d); break;
      case 3 * 8 + 1: PFO_KLAUNCH((attn_fwd_pipe_kernel<3, 1>), g, b, lds, stream, d); break;
      default: PFO_KLAUNCH((attn_fwd_pipe_kernel<3, 2>), g, b, lds, stream, d); break;
    }
  } else if (attn_fwd_ring_ok(a)) {
    const dim3 g((unsigned)pfo_ceil_div(a.N, 4)), b(256);
    const size_t qlds = FWD_Q_DMA ? 4 * (size_t)a.H * a.Cp * 4 : 0;   // the four wavefronts' qk' rows
    switch (((a.D + 63) / 64) * 8 + a.H) {
      case 1 * 8 + 1: PFO_KLAUNCH((attn_fwd_ring_kernel<1, 1>), g, b, qlds, stream, d); break;
      case 1 * 8 + 2: PFO_KLAUNCH((attn_fwd_ring_kernel<1, 2>), g, b, qlds, stream, d); break;
      case 1 * 8 + 4: PFO_KLAUNCH((attn_fwd_ring_kernel<1, 4>), g, b, qlds, stream, d); break;
      case 2 * 8 + 1: PFO_KLAUNCH((attn_fwd_ring_kernel<2, 1>), g, b, qlds, stream, d); break;
      case 2 * 8 + 2: PFO_KLAUNCH((attn_fwd_ring_kernel<2, 2>), g, b, qlds, stream, d); break;
      case 3 * 8 + 1: PFO_KLAUNCH((attn_fwd_ring_kernel<3, 1>), g, b, qlds, stream, d); break;
      case 3 * 8 + 2: PFO_KLAUNCH((attn_fwd_ring_kernel<3, 2>), g, b, qlds, stream, d); break;
      case 2 * 8 + 4: PFO_KLAUNCH((attn_fwd_ring_kernel<2, 4>), g, b, qlds, stream, d); break;
      case 3 * 8 + 4: PFO_KLAUNCH((attn_fwd_ring_kernel<3, 4>), g, b, qlds, stream, d); break;
      case 4 * 8 + 1: PFO_KLAUNCH((attn_fwd_ring_kernel<4, 1>), g, b, qlds, stream, d); break;
      case 4 * 8 + 2: PFO_KLAUNCH((attn_fwd_ring_kernel<4, 2>), g, b, qlds, stream, d); break;
      default: PFO_REQUIRE(false, "unsupported (D, H) combination");       // (attn_fwd_ring_ok admits NR H <= 12)
    }
  } else {
    ATTN_DISPATCH(attn_fwd_kernel, pfo_ceil_div(a.N, 4));
  }
  PFO_LAUNCH_CHECK();
  pfo_prof_end(PFO_PROF_ATTN_FWD, bytes, stream);
  return PFO_OK;
}

// the step's dropout multipliers, written out (include/pfotgn.h pfo_attn_dropout_mask): same function, same counters
__global__ void attn_dropout_mask_kernel(uint64_t seed, uint64_t offset, int64_t N, int K, int H, float p, float* __restrict__ out) {
  const float keep_scale = p > 0.f ? 1.f / (1.f - p) : 1.f;
  const int64_t total = N * K;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t n = e / K;
    const int j = (int)(e - n * K);
    const unsigned keep = attn_keep_bits(seed, offset, n, j, p);
    for (int h = 0; h < H; ++h) out[(n * H + h) * K + j] = ((keep >> h) & 1u) ? keep_scale : 0.f;
  }
}
extern "C" int pfo_attn_dropout_mask(uint64_t seed, uint64_t offset, int64_t N, int32_t K, int32_t H, float p, float* out,
                                     void* stream) {
  PFO_REQUIRE(out != nullptr && N >= 0, "bad arguments");
  PFO_REQUIRE(K >= 1 && K <= 64, "K must be in [1, 64] (one keep word per lane of a wavefront)");
  PFO_REQUIRE(H == 1 || H == 2 || H == 4, "n_heads must be 1, 2 or 4");
  PFO_REQUIRE(p >= 0.f && p < 1.f, "dropout must be in [0, 1)");
  if (N == 0) return PFO_OK;
  const int64_t total = N * K;
  const unsigned grid = (unsigned)std::min<int64_t>(4096, pfo_ceil_div(total, 256));
  PFO_KLAUNCH(attn_dropout_mask_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, seed, offset, N, (int)K, (int)H, p, out);
  PFO_LAUNCH_CHECK();
  return PFO_OK;
}

bool pfo_attn_bwd_runs_possible(int K, int D, int H) {
  static const int runs_on = getenv("PFO_ATTN_RUNS") ? atoi(getenv("PFO_ATTN_RUNS")) : 1;                    // A/B switch
  // D > 192 with four heads (NR = 4, H = 4) needs more than the 256 registers a lane can have: the run-merged kernel would spill
  // 160-176 B per lane to scratch there - that shape takes the per-instance kernel (no spill), like uniform sampling does
  const bool fits = !(D > 192 && H == 4);
  return runs_on && K <= 64 && fits;
}
bool pfo_attn_bwd_uses_runs(const PfoAttn& a) {
  // (the staging LDS-DMA moves 16 bytes per lane: rows must start on 16-byte boundaries)
  const int64_t qk_ld = a.qk_ld > 0 ? a.qk_ld : (int64_t)a.H * a.Cp;
  const bool aligned = (qk_ld % 4) == 0 && (((uintptr_t)a.QK | (uintptr_t)a.dctx | (uintptr_t)a.ctx) & 15u) == 0;
  return a.d_nbr && a.nbr_row && pfo_attn_bwd_runs_possible(a.K, a.D, a.H) && a.members && a.seg_ptr && a.n_rows && a.qk_row && a.run_cnt &&
         (a.dqk_live || (a.dq_rows && !a.det)) && aligned;
}

int pfo_attn_bwd_launch(const PfoAttn& a, int* n_parts, hipStream_t stream) {
  if (int rc = check_common(a)) return rc;
  PFO_REQUIRE(a.dctx && a.dQK && a.dtime_part, "null backward buffers");
  PFO_REQUIRE(!a.det || a.dtime_slab, "deterministic mode needs the slab");
  AttnDev d;
  to_dev(a, d);
  static const int bwd_blocks_env = getenv("PFO_ATTN_BWD_BLOCKS") ? atoi(getenv("PFO_ATTN_BWD_BLOCKS")) : ATTN_BWD_MAX_BLOCKS;
  const int bwd_blocks = a.det ? ATTN_BWD_MAX_BLOCKS : bwd_blocks_env;          // (deterministic: the slab's row count is fixed)
  const int grid = (int)std::min<int64_t>(bwd_blocks, pfo_ceil_div(a.N, 4));
  // rows read again + their gradient rows written/added, qk + dctx + ctx in, dqk out
  const double C = 2.0 * a.D + a.Ef;
  const double bytes = (double)a.N * (a.K * (8.0 * a.D + 4.0 * a.Ef + 12.0) + 4.0 * a.H * C * 4.0 + 4.0 * a.H * a.K);
  const int dmode = !a.d_nbr ? 0 : (a.nbr_row ? 1 : 2);
  static const int lds_pad = getenv("PFO_ATTN_RUNS_LDSPAD") ? atoi(getenv("PFO_ATTN_RUNS_LDSPAD")) : 0;   // occupancy probe
  // the run-merged kernel's staging image: three rows of H Cp floats + (4 + H) metadata arrays of K words (attn_bwd_runs_kernel)
  const size_t run_lds = (size_t)lds_pad + 3 * (size_t)a.H * a.Cp * 4 + (size_t)(4 + a.H) * a.K * 4 + (size_t)pfo_align_up(4 * a.K, 16);   // (+ K injected keep bytes, one dword each)
  if (pfo_attn_bwd_uses_runs(a)) {
    // run-merged form: single-wavefront workgroups, one chunk of members each (the grid-stride loop only matters when the
    // grid is capped for an experiment)
    static const int rblocks = getenv("PFO_ATTN_RUNS_BLOCKS") ? atoi(getenv("PFO_ATTN_RUNS_BLOCKS")) : 0;
    static const int xcd_g = getenv("PFO_ATTN_XCD_G") ? atoi(getenv("PFO_ATTN_XCD_G")) : 16;              // A/B switch (counter pass: FETCH 625 -> 487 MB per launch at 16, 512 at 4; the launch time does not move)
    d.xcd_g = a.det ? 0 : std::max(0, xcd_g);                  // (deterministic mode: slab row = chunk = workgroup id)
    int64_t all_units = pfo_ceil_div(pfo_ceil_div(a.N, RUN_CHUNK), RUN_CPW);       // a wavefront takes RUN_CPW chunks
    if (d.xcd_g > 0) all_units = pfo_align_up(all_units, 8 * d.xcd_g);
    const int rgrid = (int)((rblocks > 0 && !a.det) ? std::min<int64_t>(rblocks, all_units) : all_units);
    pfo_prof_begin(stream);
    const int NRv = (a.D + 63) / 64;
    const dim3 g((unsigned)rgrid), b(64);
    bool done = true;
#define RUNS_GO(NRc, Hc)                                                                                              \
  case NRc * 8 + Hc:                                                                                                  \
    if (a.det) PFO_KLAUNCH((attn_bwd_runs_kernel<NRc, Hc, true>), g, b, run_lds, stream, d);                   \
    else PFO_KLAUNCH((attn_bwd_runs_kernel<NRc, Hc, false>), g, b, run_lds, stream, d);                        \
    break;
    switch (NRv * 8 + a.H) {
      RUNS_GO(1, 1) RUNS_GO(1, 2) RUNS_GO(1, 4) RUNS_GO(2, 1) RUNS_GO(2, 2) RUNS_GO(2, 4)
      RUNS_GO(3, 1) RUNS_GO(3, 2) RUNS_GO(3, 4) RUNS_GO(4, 1) RUNS_GO(4, 2)      // (4, 4): pfo_attn_bwd_runs_possible says no
      default: done = false;
    }
#undef RUNS_GO
    PFO_REQUIRE(done, "unsupported (D, H) combination");
    PFO_LAUNCH_CHECK();
    pfo_prof_end(PFO_PROF_ATTN_BWD_RUNS, bytes, stream);
    if (n_parts) *n_parts = a.det ? rgrid * RUN_CPW : ATTN_TIME_BINS;   // deterministic: slab rows written (one per chunk, RUN_CPW per workgroup)
    return PFO_OK;
  }
  pfo_prof_begin(stream);
  if (attn_bwd_ring_ok(a, dmode) && dmode == 0) { ATTN_DISPATCH_RING(attn_bwd_ring_kernel_none, grid); }
  else if (attn_bwd_ring_ok(a, dmode)) { ATTN_DISPATCH_RING(attn_bwd_ring_kernel_direct, grid); }
  else if (dmode == 0) { ATTN_DISPATCH(attn_bwd_kernel_none, grid); }
  else if (dmode == 1 && a.det) { ATTN_DISPATCH(attn_bwd_kernel_det, grid); }
  else if (dmode == 1) { ATTN_DISPATCH(attn_bwd_kernel, grid); }
  else { ATTN_DISPATCH(attn_bwd_kernel_direct, grid); }
  PFO_LAUNCH_CHECK();
  pfo_prof_end(PFO_PROF_ATTN_BWD, bytes, stream);
  if (n_parts) *n_parts = a.det ? grid : ATTN_TIME_BINS;
  return PFO_OK;
}
