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
  int xcd_g;    // run-merged backward: G consecutive chunks of members per XCD turn (0 = chunks round-robin over the XCDs)
  int abl;      // timing-only ablation switch (PFO_ATTN_ABL): 1 = spread the atomic destinations (wrong results)
};

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

#ifndef KC_RUNS
#define KC_RUNS 2      // keys in flight per wavefront
#endif
#ifndef RUNS_WAVES
#define RUNS_WAVES(NR, H) ((NR) * (H) <= 6 ? 3 : 2)
#endif
// RUNS_PREFETCH (round 4): a member's d ctx' and ctx' rows (2 x 2.8 KB at C2) are streamed from HBM exactly once, by this
// kernel, and the member's setup needs them before its first key can be scored: with three wavefronts per SIMD that round trip
// was exposed once per member (~4 us of work each).  While member m is processed one LDS-DMA instruction touches every
// 128-byte line of member m+1's two rows (lanes 0-31: d ctx', 32-63: ctx'; one dword per line into a landing pad nobody
// reads - no registers, no wait): the rows are on their way when m+1's setup asks for them.  Measured: the kernel 322-324 ->
// 315-318 us (-2 %), but the counter pass shows FETCH_SIZE +36 % (the touched 64-byte sectors are fetched again by the real
// loads: the pad warms the Infinity Cache, not the L2) - 6 us for 224 MB of extra fabric reads.  OFF by default.
#ifndef RUNS_PREFETCH
#define RUNS_PREFETCH 0
#endif
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
template <int NR, int H, bool DET>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(RUNS_WAVES(NR, H)))) void attn_bwd_runs_kernel(const AttnDev a) {
  __shared__ float s_tw[NR * 64], s_tb[NR * 64];
  __shared__ float s_cA[RUN_CHUNK][H][64];     // [instance of the group][head][slot]: cA of that key; zero where the slot is empty
  __shared__ int s_delta[RUN_CHUNK];            // [instance of the group]: its shift (count - the group's first count)
#if RUNS_PREFETCH
  __shared__ float s_pf[64];                    // landing pad of the next member's row prefetch (never read)
#endif
  const int lane = threadIdx.x;
  const int D = a.D, Ef = a.Ef, K = a.K, C = 2 * D + Ef, Cp = a.Cp;
  float wmax = 0.f, bmax = 0.f;
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const int c = lane + 64 * r;
    const float w = c < D ? a.tw[c] : 0.f, b = c < D ? a.tb[c] : 0.f;
    s_tw[c] = w; s_tb[c] = b;
    wmax = fmaxf(wmax, fabsf(w)); bmax = fmaxf(bmax, fabsf(b));
  }
  wmax = pfo_wave_max(wmax); bmax = pfo_wave_max(bmax);
  // clamped column of this lane in each 64-column group, and the edge column
  int colr[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) colr[r] = min(lane + 64 * r, D - 1);
  const int cole = min(lane, max(Ef, 1) - 1);
  const float keep_scale = a.dropout_p > 0.f ? 1.f / (1.f - a.dropout_p) : 1.f;
  float* const d_nbr_x = a.d_nbr + (int64_t)(__builtin_amdgcn_s_getreg(6164) & (a.d_nbr_nrep - 1)) * a.d_nbr_rep;
  const int M = a.seg_ptr[*a.n_rows];                            // members = instances that sit on a real node
  const int n_chunks = (M + RUN_CHUNK - 1) / RUN_CHUNK;
  const uint64_t rng_off = a.offset + (a.offset_dev ? *a.offset_dev : 0ull);
  const float* const nbr_tab = a.nbr_tab;
  const float* const edge_feat = a.edge_feat;
  const uint32_t nbr_ld = (uint32_t)a.nbr_ld;

  // Workgroups go to the XCDs round-robin by their id.  Consecutive chunks hold members of the same table row or of
  // neighbouring ones (the list is ordered by row): with xcd_g = G > 0 every XCD takes G consecutive chunks at a time (block b ->
  // chunk (b / 8G) 8G + (b mod 8) G + (b mod 8G) / 8), so the shifted neighbour lists of one node's members, its query row and
  // the rows its atomics land on stay in ONE L2 instead of being fetched by up to eight.
  const int G8 = 8 * a.xcd_g;
  const int n_walk = G8 > 0 ? (n_chunks + G8 - 1) / G8 * G8 : n_chunks;
  for (int blk = blockIdx.x; blk < n_walk; blk += gridDim.x) {
    int chunk = blk;
    if (G8 > 0) {
      const int grp = blk / G8, r = blk - grp * G8;
      chunk = grp * G8 + (r & 7) * a.xcd_g + (r >> 3);
      if (chunk >= n_chunks) continue;
    }
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
    auto acc_store = [&]() {
      if (acc_m >= 0) {
        float* dqk_out = a.dQK + (int64_t)acc_m * H * Cp;          // row m, not n: the per-row sums then stream contiguous rows
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
          if (lane < Cp - C) dqk_out[h * Cp + C + lane] = 0.f;
        }
        if (lane == 0) a.dqk_live[acc_m] = 1;
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
        const float* qk = a.QK + (int64_t)run_slot * a.qk_ld;
        float qn[H][NR], g[RUN_CHUNK][H][NR];
#pragma unroll
        for (int h = 0; h < H; ++h)
#pragma unroll
          for (int r = 0; r < NR; ++r) {
            const int c = lane + 64 * r;
            qn[h][r] = c < D ? qk[h * Cp + c] : 0.f;
          }
        float cAr[RUN_CHUNK][H];                                   // lane q: cA of the slot instance i has on history entry q
#pragma unroll
        for (int i = 0; i < RUN_CHUNK; ++i) {
          const bool on = i < run_len;                             // wave-uniform
          const float* dc = a.dctx + (int64_t)a.members[on ? run_first + i : run_first] * H * Cp;
          const int sl = lane - (on ? s_delta[i] : 0);             // the instance's slot on this lane (may lie outside [0, K): zero)
#pragma unroll
          for (int h = 0; h < H; ++h) {
            cAr[i][h] = (on && sl >= 0) ? s_cA[i][h][sl < 0 ? 0 : sl] : 0.f;
#pragma unroll
            for (int r = 0; r < NR; ++r) {
              const int c = lane + 64 * r;
              g[i][h][r] = (on && c < D) ? dc[h * Cp + c] : 0.f;
            }
          }
        }
        unsigned long long vmask = run_valid;
        while (vmask) {
          const int j = __ffsll((long long)vmask) - 1;
          vmask &= vmask - 1ull;
          const int64_t drow = (int64_t)rl_i(run_rows, j) * a.d_nbr_ld;
          float* dst = d_nbr_x + drow;
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
            const int cc = lane + 64 * r;
            if ((r < NR - 1 || cc < D) && a.abl != 2) {
              if (DET) det_add(a.d_nbr, drow + cc, row[r]); else atomicAdd(dst + cc, row[r]);
            }
          }
        }
      }
#pragma unroll
      for (int h = 0; h < H; ++h) sBr[h] = 0.f;
      run_len = 0;
    };

    const int m_end = min(M, (chunk + 1) * RUN_CHUNK);
#if RUNS_PREFETCH
    // the chunk's member ids in lanes 0 .. RUN_CHUNK-1 (one load per chunk instead of a dependent scalar load per member)
    const int mem_ids = (lane < RUN_CHUNK && chunk * RUN_CHUNK + lane < m_end) ? a.members[chunk * RUN_CHUNK + lane] : 0;
    const uint32_t row_bytes = (uint32_t)(H * Cp) * 4u;
    const uint32_t pf_off = min((uint32_t)(lane & 31) * 128u, row_bytes - 4u);
#endif
    for (int m = chunk * RUN_CHUNK; m < m_end; ++m) {
#if RUNS_PREFETCH
      const int64_t n = rl_i(mem_ids, m - chunk * RUN_CHUNK);
      if (m + 1 < m_end) {
        typedef __attribute__((address_space(1))) const void* gptr_t;
        typedef __attribute__((address_space(3))) void* lptr_t;
        const int64_t n2 = rl_i(mem_ids, m + 1 - chunk * RUN_CHUNK);
        const char* src = reinterpret_cast<const char*>(lane < 32 ? a.dctx : a.ctx) + n2 * (int64_t)row_bytes + pf_off;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)s_pf, 4, 0, 0);
      }
#else
      const int64_t n = a.members[m];
#endif
      const int64_t slot0 = n * K;
      const bool inK = lane < K;
      const int my_id = inK ? a.nbr_ids[slot0 + lane] : 0;
      const int my_row = inK ? a.nbr_row[slot0 + lane] : 0;
      const int my_e = inK ? a.eidx[slot0 + lane] : 0;
      const float my_dt = inK ? a.dt[slot0 + lane] : 0.f;
      const unsigned long long valid = __ballot(inK && my_id != 0);
      const int slot = a.qk_row[n];
      const int cnt_n = a.run_cnt[n];
      int delta = cnt_n - run_cnt0;                              // members of a row arrive by ascending count
      if (slot != run_slot || delta < 0 || delta > 64 - K) {     // another node, or the lanes run out: the group's rows leave
        acc_store();                                             // (the query-side sums too: never live across a flush)
        flush();
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
      if (lane == 0) a.dqk_live[m] = 0;                         // (set when this member's position receives a stored sum)
      if (valid == 0ull) continue;                               // no neighbour: nothing to add to the row's sums
      acc_m = m;
      if (run_len == 0) run_first = m;                           // the group's instances with a neighbour are consecutive members
      const int run_i = run_len;                                 // (an instance without history has count 0: first of its row)
      run_len += 1;
      s_delta[run_i] = delta;
#pragma unroll
      for (int h = 0; h < H; ++h) s_cA[run_i][h][lane] = 0.f;
      float qt[H][NR], gn[H][NR], gt[H][NR], ge[H], tds[2 * H];
      const float* qk = a.QK + (int64_t)slot * a.qk_ld;
      const float* dc = a.dctx + n * H * Cp;
      const float* cx = a.ctx + n * H * Cp;
#pragma unroll
      for (int h = 0; h < H; ++h) {
        float part = 0.f;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
          const int c = lane + 64 * r;
          const bool ok = c < D;
          qt[h][r] = ok ? qk[h * Cp + D + Ef + c] : 0.f;
          gn[h][r] = ok ? dc[h * Cp + c] : 0.f;
          gt[h][r] = ok ? dc[h * Cp + D + Ef + c] : 0.f;
          if (ok) part = fmaf(gn[h][r], cx[h * Cp + c], fmaf(gt[h][r], cx[h * Cp + D + Ef + c], part));
        }
        ge[h] = lane < Ef ? dc[h * Cp + D + lane] : 0.f;
        if (lane < Ef) part = fmaf(ge[h], cx[h * Cp + D + lane], part);
        tds[h] = part;
      }
      pfo_wave_sum_scalar_n<H>(reinterpret_cast<float(&)[H]>(tds));     // delta_h = dctx_h . ctx_h (+ the extra column below)
      float t[H], dsb[H];
#pragma unroll
      for (int h = 0; h < H; ++h) {
        dsb[h] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(dc[h * Cp + C])));
        t[h] = fmaf(dsb[h], __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(cx[h * Cp + C]))), tds[h]);
      }
      const unsigned keep = attn_keep_for(a, rng_off, n, lane);
      float my_a[H];
#pragma unroll
      for (int h = 0; h < H; ++h) my_a[h] = inK ? a.attw[(n * H + h) * K + lane] : 0.f;
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
          const unsigned kb = (unsigned)rl_i((int)keep, js[c]);
          const bool mine = lane == js[c] + delta;              // the lane of this key's history entry
          float cA[H], cB[H];
#pragma unroll
          for (int h = 0; h < H; ++h) {
            const float ks_h = ((kb >> h) & 1u) ? keep_scale : 0.f;
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
          if (lane == js[c]) {
#pragma unroll
            for (int h = 0; h < H; ++h) s_cA[run_i][h][lane] = cA[h];
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
      };
      if (fast) walk(std::true_type{}); else walk(std::false_type{});
    }
    acc_store();                                                 // the chunk's last row sums
    flush();                                                     // the chunk's last run
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
  }
  // deterministic mode: the launch has one workgroup per POSSIBLE chunk; those beyond the members' count own a zero row
  if (DET && (int)blockIdx.x >= n_chunks)
    for (int c = lane; c < 2 * D; c += 64) a.dtime_slab[(int64_t)blockIdx.x * 2 * D + c] = 0.0;
}

int64_t pfo_attn_bwd_det_parts(int64_t N) { return std::max<int64_t>(pfo_ceil_div(N, RUN_CHUNK), std::min<int64_t>(ATTN_BWD_MAX_BLOCKS, pfo_ceil_div(N, 4))); }

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

// minimum wavefronts per SIMD the register allocation must allow: 3 where the kernel fits 168 VGPRs without spilling
#ifndef BWD_WAVES
#define BWD_WAVES(NR, H, DMODE) (((H) <= 2 && (NR) <= 3) ? ((DMODE) == 1 ? 2 : 3) : (((H) == 4 && (NR) == 4) ? 1 : 2))
#endif
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
  ATTN_DISPATCH(attn_fwd_kernel, pfo_ceil_div(a.N, 4));
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
  return a.d_nbr && a.nbr_row && pfo_attn_bwd_runs_possible(a.K, a.D, a.H) && a.members && a.seg_ptr && a.n_rows && a.qk_row && a.run_cnt &&
         a.dqk_live;
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
  const size_t run_lds = (size_t)lds_pad;
  if (pfo_attn_bwd_uses_runs(a)) {
    // run-merged form: single-wavefront workgroups, one chunk of members each (the grid-stride loop only matters when the
    // grid is capped for an experiment)
    static const int rblocks = getenv("PFO_ATTN_RUNS_BLOCKS") ? atoi(getenv("PFO_ATTN_RUNS_BLOCKS")) : 0;
    static const int xcd_g = getenv("PFO_ATTN_XCD_G") ? atoi(getenv("PFO_ATTN_XCD_G")) : 16;              // A/B switch (counter pass: FETCH 625 -> 487 MB per launch at 16, 512 at 4; the launch time does not move)
    d.xcd_g = a.det ? 0 : std::max(0, xcd_g);                  // (deterministic mode: slab row = chunk = workgroup id)
    int64_t all_chunks = pfo_ceil_div(a.N, RUN_CHUNK);
    if (d.xcd_g > 0) all_chunks = pfo_align_up(all_chunks, 8 * d.xcd_g);
    const int rgrid = (int)((rblocks > 0 && !a.det) ? std::min<int64_t>(rblocks, all_chunks) : all_chunks);
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
    if (n_parts) *n_parts = a.det ? rgrid : ATTN_TIME_BINS;       // deterministic: slab rows written (one per workgroup)
    return PFO_OK;
  }
  pfo_prof_begin(stream);
  if (dmode == 0) { ATTN_DISPATCH(attn_bwd_kernel_none, grid); }
  else if (dmode == 1 && a.det) { ATTN_DISPATCH(attn_bwd_kernel_det, grid); }
  else if (dmode == 1) { ATTN_DISPATCH(attn_bwd_kernel, grid); }
  else { ATTN_DISPATCH(attn_bwd_kernel_direct, grid); }
  PFO_LAUNCH_CHECK();
  pfo_prof_end(PFO_PROF_ATTN_BWD, bytes, stream);
  if (n_parts) *n_parts = a.det ? grid : ATTN_TIME_BINS;
  return PFO_OK;
}
