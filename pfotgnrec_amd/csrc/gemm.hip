// fp32 contraction on the gfx950 matrix cores: v_mfma_f32_16x16x4_f32 (exact fp32, 256 FLOP/clk/CU).
//
// Why fp32 MFMA: the parity bar is 1e-4 relative on embeddings (BASELINE.json north_star), which
// rules out bf16/fp16 operands; gfx950 has no xf32.  Tile = 128 x 176 x 32, 4 wavefronts, each
// wavefront owning a 32 x 176 strip as 2 x 11 MFMA tiles (88 accumulator VGPRs).  176 = 11*16 is
// chosen for THIS model: every N it sees (D=172, 2D=344, 2D+Ef=348, 3D=516) is k*176 minus a few
// columns, so column padding waste is ~2% where a 128/256-wide tile would waste 10-25%.
// Operands are staged global -> registers -> LDS (one LDS buffer, next tile's global loads in
// flight during the MFMAs of the current one); LDS row strides (34 / 144 / 176 floats) make every
// fragment read conflict-free for the two 32-lane halves of ds_read_b32.
#include "gemm.hpp"
#include <algorithm>
#include <string.h>
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define BM 128
#define BN 176
#define BK 32
#define GEMM_THREADS 256
#define LDA_RM 34     // row-major A/B tile row stride (floats): 34 mod 32 = 2 -> banks 2r+g distinct
#define LDA_KM 144    // k-major A tile row stride: 144 mod 32 = 16
#define LDB_KM 176    // k-major B tile row stride: 176 mod 32 = 16
#define AS_FLOATS 4608   // max(128*34, 32*144)
#define BS_FLOATS 5984   // max(176*34, 32*176)

// XCD-aware tile order of a (row tiles x column tiles) launch flattened into a 1-D grid: workgroups go to the XCDs round-robin
// by id, so id = (t / 8) * 8 * tiles_n + c * 8 + (t mod 8) puts every column tile c of row tile t on XCD t & 7 - the A rows
// the column tiles share come from HBM once and from that XCD's L2 for the others.  false: no such tile (the last group).
__device__ __forceinline__ bool xcd_tile(int id, int tiles_m, int tiles_n, int& tm, int& tn) {
  const int per = 8 * tiles_n, grp = id / per, r = id - grp * per;
  tn = r >> 3;
  tm = grp * 8 + (r & 7);
  return tm < tiles_m;
}
struct GemmDev {
  int xcd_tm, xcd_tn;              // > 0: 1-D grid in xcd_tile order (row tiles, column tiles); 0: (row tile, column tile) = blockIdx.(x, y)
  const float* A[2]; int64_t lda[2]; const int32_t* a_idx[2];
  const float* B[2]; int64_t ldb[2]; const int32_t* b_idx;
  int K[2];
  float* C; int64_t ldc;
  const float* bias; const float* row_scale; int64_t rs_ld; const uint8_t* row_zero;
  const float* relu_src; int64_t relu_ld;
  const float* add_src; int64_t add_ld; const int32_t* add_idx;
  int M, N; const int32_t* m_dev;
  int relu, accumulate;
  int nsplit; int split_chunk;     // k-major split-K
  int64_t a_bs[2], b_bs[2], c_bs, bias_bs, rs_bs;
  int n_real;                      // columns of B that exist in memory; column n_real (if < N) is the bias column
  float* slab_base;                // split-K: this problem's slab region
  int dyn_chunk;                   // split-K chunk = f(device-side K) instead of split_chunk
  const void* b_img; int b_img_rows;   // pre-split bf16x3 image of B (source 0) and its padded row count
  const void* b_img2;                  // ... of the second K-concatenated source (same padded row count)
  // GRU gate backward epilogue of the 32-row kernel (gemm.hpp PfoGemm::gg_*)
  const float* gg_gates; const float* gg_h; const uint8_t* gg_hm; const float* gg_dh0; float* gg_dgi; float* gg_dgh;
};

static bool aligned4(const void* p) { return (((uintptr_t)p) & 15) == 0; }

#ifndef PFO_DEFAULT_TILE
#define PFO_DEFAULT_TILE 0
#endif
#ifndef PFO_DEFAULT_BF16X3
#define PFO_DEFAULT_BF16X3 1
#endif
#ifndef BX_EXP
#define BX_EXP 0      // timing-only ablations of the bf16x3 kernel (wrong results): 1 = first tile only, 2 = no MFMA, 3 = no LDS refill
#endif
#ifndef PFO_DEFAULT_AREG
#define PFO_DEFAULT_AREG 1
#endif
#ifndef BXA_NT
#define BXA_NT 0         // A/B: nontemporal stores in the image kernels' epilogue
#endif
#ifndef BXA_STAGGER
#define BXA_STAGGER 0    // A/B: the second workgroup of every CU's first round starts BXA_STAGGER x 3.4 us late (phases of the two differ)
#endif
#ifndef PFO_DEFAULT_TN8
#define PFO_DEFAULT_TN8 500      // weight-gradient launches whose problems all have >= this many rows take 256-row tiles (0 = never)
#endif
#ifndef PFO_DEFAULT_ASTAT
#define PFO_DEFAULT_ASTAT 1
#endif
#ifndef PFO_DEFAULT_AREG8
#define PFO_DEFAULT_AREG8 512 // the image kernel's large form as eight single-strip wavefronts (gemm_bx_areg8_kernel) from this many workgroups on
#endif
#ifndef BXA_EPI
#define BXA_EPI 1        // gemm_bx_areg_kernel: loads of the epilogue in front of its stores (0 = the round-3 form, kept for the A/B)
#endif
#ifndef BXA_STAMPS
#define BXA_STAMPS 0     // diagnostic build: per-workgroup wall-clock stamps (100 MHz) of gemm_bx_areg_kernel, read by pfo_debug_bxa_stamps
#endif
#if BXA_STAMPS
__device__ uint64_t g_bxa_stamps[4096 * 8];
extern "C" int pfo_debug_bxa_stamps(uint64_t* host_out, int n_words) {
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_bxa_stamps), (size_t)n_words * 8, 0, hipMemcpyDeviceToHost) == hipSuccess ? 0 : 1;
}
#define BXA_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.x < 4096) g_bxa_stamps[blockIdx.x * 8 + (k)] = wall_clock64(); } while (0)
#else
#define BXA_STAMP(k) do { } while (0)
#endif
#ifndef BXA_ABL
#define BXA_ABL 0        // timing-only ablations of gemm_bx_areg_kernel (wrong results), bits: 1 no MFMA, 2 no stores, 4 A rows of tile 0 only
#endif                   //   (cache hits), 8 no image DMA after the first tile, 16 no row split after the first tile
#ifndef PFO_BX_MIN_TILES
#define PFO_BX_MIN_TILES 400
#endif
#ifndef GEMM_EXP
#define GEMM_EXP 0       // timing-only ablations (wrong results): 1 = no global loads / LDS stores after the first tile, 2 = no MFMA,
#endif                   //                                          3 = no barriers + no reloads

// four consecutive floats of which `nv` lie inside the matrix; `safe` is any valid 16-byte aligned address,
// loaded instead of an out-of-range one so that the load itself needs no branch
template <bool VEC>
__device__ __forceinline__ float4 ld4(const float* p, int nv, const float* safe) {
  if (VEC) {
    const float4 v = *reinterpret_cast<const float4*>(nv > 0 ? p : safe);
    return nv > 0 ? v : float4{0.f, 0.f, 0.f, 0.f};
  } else {
    float4 r;
    r.x = nv > 0 ? p[0] : 0.f;
    r.y = nv > 1 ? p[1] : 0.f;
    r.z = nv > 2 ? p[2] : 0.f;
    r.w = nv > 3 ? p[3] : 0.f;
    return r;
  }
}

// Two workgroup shapes, both 256 threads = 4 wavefronts:
//   BIG   : 128 x 176 output tile, wavefront w owns rows [32w, 32w+32) x all 11 column tiles   (2 x 11 MFMA tiles)
//   SMALL :  32 x 176 output tile, wavefront w owns all 32 rows x column tiles {w, w+4, w+8}    (2 x 3 MFMA tiles)
// SMALL exists for the layer-2 launches (M = 2560 rows): 4x the workgroups, so the chip is not left idle.
// The K loop runs over a flattened list of 32-deep tiles of up to two K-concatenated sources; the global loads of
// tile t+1 are in flight (registers) while the MFMAs of tile t run from LDS.
template <bool A_KM, bool B_KM, int TILE, bool VEC>
__device__ __forceinline__ void gemm_tile(const GemmDev& p, const int bx, const int by, int zb, const int split_in,
                                          float* __restrict__ lds_a, float* __restrict__ lds_b) {
  constexpr bool TINY = (TILE == 2);           // 32 x 64 tile: the multi-problem launches (see gemm_multi_kernel)
  constexpr bool SMALL = (TILE == 1) || TINY;
  constexpr int TBM = SMALL ? 32 : BM;
  constexpr int TBN = TINY ? 64 : BN;          // columns of the tile
  constexpr int NB4 = TBN / 4;                 // float4 per k-row of a k-major B tile
  constexpr int NBL = (TBN * 8 + 255) / 256;   // float4 of the B tile per thread
  constexpr int NJ = TINY ? 1 : (SMALL ? 3 : 11);
  constexpr int NI = 2;                        // 16-row strips per wavefront
  constexpr int NA = TBM / 32;                 // float4 per thread for the A tile
  constexpr int LDAK = TINY ? 48 : LDA_KM;     // k-major A row stride, = 16 mod 32
  constexpr int LDBK = TINY ? 80 : LDB_KM;     // k-major B row stride, = 16 mod 32
  float* const As[1] = {lds_a};
  float* const Bs[1] = {lds_b};
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int m0 = bx * TBM, n0 = by * TBN;
  const int split = split_in;

  int Mlim = p.M;
  int Kext0 = p.K[0];
  if (p.m_dev) {
    const int md = *p.m_dev;
    if (A_KM) Kext0 = min(Kext0, md); else Mlim = min(Mlim, md);
  }
  if (!A_KM && m0 >= Mlim) return;
  int kbeg = 0, kend0 = Kext0;
  if (A_KM && p.nsplit > 1) {
    // with a device-side K bound (GRU rows) the chunks are cut from the ACTUAL extent so every split has work
    const int chunk = p.dyn_chunk ? ((Kext0 + p.nsplit - 1) / p.nsplit + BK - 1) / BK * BK : p.split_chunk;
    kbeg = split * chunk;
    kend0 = min(Kext0, kbeg + chunk);
  }
  const int wrow = SMALL ? 0 : 32 * wave;      // first tile row of this wavefront
  const int wcol = SMALL ? 16 * wave : 0;      // first tile column; SMALL strides columns by 64

  f32x4 acc[NI][NJ];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // flattened tile list
  const int T0 = kend0 > kbeg ? (kend0 - kbeg + BK - 1) / BK : 0;
  const int T1 = (p.K[1] > 0 && p.A[1]) ? (p.K[1] + BK - 1) / BK : 0;
  const int T = T0 + T1;

  float4 a_reg[NA], b_reg[NBL];
  float4 a_reg2[TINY ? NA : 1], b_reg2[TINY ? NBL : 1];   // TINY: second staging set (loads two tiles ahead)
  const float* a_row[NA];
  const float* b_row[NBL];
  bool a_ok[NA], b_ok[NBL];
  const float *Ab = nullptr, *Bb = nullptr;
  int64_t lda = 0, ldb = 0;
  int Ks = 0, cur_src = -1;
  const float* safe = p.A[0];

  auto bind_src = [&](int src) __attribute__((always_inline)) {
    cur_src = src;
    Ks = (src == 0) ? kend0 : p.K[1];
    Ab = p.A[src] + zb * p.a_bs[src];
    Bb = p.B[src] + zb * p.b_bs[src];
    lda = p.lda[src]; ldb = p.ldb[src];
    if (!A_KM) {
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        const int gm = m0 + ((tid + 256 * i) >> 3);
        a_ok[i] = gm < Mlim;
        int64_t ridx = a_ok[i] ? gm : 0;
        if (a_ok[i] && p.a_idx[src]) ridx = p.a_idx[src][gm];
        a_row[i] = Ab + ridx * lda;
      }
    }
    if (!B_KM) {
#pragma unroll
      for (int i = 0; i < NBL; ++i) {
        const int f = tid + 256 * i;
        const int gn = n0 + (f >> 3);
        b_ok[i] = (f < TBN * 8) && gn < p.N;
        b_row[i] = Bb + (int64_t)(b_ok[i] ? gn : 0) * ldb;
      }
    }
  };
  auto load_tile = [&](int t, auto& a_reg, auto& b_reg) __attribute__((always_inline)) {
    const int src = t < T0 ? 0 : 1;
    if (src != cur_src) bind_src(src);
    const int k0 = (src == 0) ? kbeg + t * BK : (t - T0) * BK;
    if (!A_KM) {
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        const int k = k0 + 4 * ((tid + 256 * i) & 7);
        a_reg[i] = ld4<VEC>(a_row[i] + k, a_ok[i] ? Ks - k : 0, safe);
      }
    } else {
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        const int f = tid + 256 * i;
        const int k = k0 + (f / (TBM / 4));
        const int m = m0 + 4 * (f % (TBM / 4));
        a_reg[i] = ld4<VEC>(Ab + (int64_t)k * lda + m, k < Ks ? p.M - m : 0, safe);
      }
    }
    if (!B_KM) {
#pragma unroll
      for (int i = 0; i < NBL; ++i) {
        const int k = k0 + 4 * ((tid + 256 * i) & 7);
        b_reg[i] = ld4<VEC>(b_row[i] + k, b_ok[i] ? Ks - k : 0, safe);
      }
    } else {
#pragma unroll
      for (int i = 0; i < NBL; ++i) {
        const int f = tid + 256 * i;
        const int kr = f / NB4;
        const int k = k0 + kr;
        const int n = n0 + 4 * (f - kr * NB4);
        const bool ok = (f < BK * NB4) && k < Ks && n < p.N;
        int64_t krow = ok ? k : 0;
        if (ok && p.b_idx && src == 0) krow = p.b_idx[k];
        b_reg[i] = ld4<VEC>(Bb + krow * ldb + n, ok ? p.n_real - n : 0, safe);
        if (A_KM && ok && p.n_real < p.N) {
          // bias column: dW's extra column accumulates sum_k dY[k][m] * (scale[k] or 1)
          const int e = p.n_real - n;
          if (e >= 0 && e < 4) {
            const float one = 1.f;
            if (e == 0) b_reg[i].x = one; else if (e == 1) b_reg[i].y = one; else if (e == 2) b_reg[i].z = one; else b_reg[i].w = one;
          }
        }
      }
    }
  };
  auto store_tile = [&](int buf, const auto& a_reg, const auto& b_reg) __attribute__((always_inline)) {
    float* as = As[buf];
    float* bs = Bs[buf];
    if (!A_KM) {
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        const int f = tid + 256 * i;
        float* d = as + (f >> 3) * LDA_RM + 4 * (f & 7);
        *reinterpret_cast<float2*>(d) = float2{a_reg[i].x, a_reg[i].y};
        *reinterpret_cast<float2*>(d + 2) = float2{a_reg[i].z, a_reg[i].w};
      }
    } else {
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        const int f = tid + 256 * i;
        *reinterpret_cast<float4*>(as + (f / (TBM / 4)) * LDAK + 4 * (f % (TBM / 4))) = a_reg[i];
      }
    }
    if (!B_KM) {
#pragma unroll
      for (int i = 0; i < NBL; ++i) {
        const int f = tid + 256 * i;
        if (f < TBN * 8) {
          float* d = bs + (f >> 3) * LDA_RM + 4 * (f & 7);
          *reinterpret_cast<float2*>(d) = float2{b_reg[i].x, b_reg[i].y};
          *reinterpret_cast<float2*>(d + 2) = float2{b_reg[i].z, b_reg[i].w};
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < NBL; ++i) {
        const int f = tid + 256 * i;
        if (f < BK * NB4) {
          const int kr = f / NB4;
          *reinterpret_cast<float4*>(bs + kr * LDBK + 4 * (f - kr * NB4)) = b_reg[i];
        }
      }
    }
  };
  // fragments of k-step s (lane (r, g) holds A[row r][k = 4s + g], B[k = 4s + g][col r])
  auto load_frags = [&](const float* as, const float* bs, int s, float (&a)[NI], float (&b)[NJ]) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NI; ++i)
      a[i] = A_KM ? as[(4 * s + g) * LDAK + wrow + 16 * i + r] : as[(wrow + 16 * i + r) * LDA_RM + 4 * s + g];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int col = wcol + (SMALL ? 64 : 16) * j + r;
      b[j] = B_KM ? bs[(4 * s + g) * LDBK + col] : bs[col * LDA_RM + 4 * s + g];
    }
  };
  // a wavefront whose 16-row strip lies entirely beyond M issues no MFMAs for it (172-row weight-gradient tiles:
  // the matrix pipe of its SIMD is left to the co-resident workgroup)
  const bool strip_on[2] = {m0 + wrow < p.M, m0 + wrow + 16 < p.M};
  auto mma = [&](const float (&a)[NI], const float (&b)[NJ]) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NI; ++i)
      if (strip_on[i])
#pragma unroll
      for (int j = 0; j < NJ; ++j)
        if (!SMALL || wcol + 64 * j < TBN)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
  };
  // all 8 k-steps of a tile; LDS beyond K holds zeros, so the K tail needs no guard.  Fragment reads of step
  // s+1 are issued ahead of the MFMAs of step s.
  auto compute_tile = [&](int buf) __attribute__((always_inline)) {
    const float* as = As[buf];
    const float* bs = Bs[buf];
    float a0[NI], b0[NJ], a1[NI], b1[NJ];
    load_frags(as, bs, 0, a0, b0);
#pragma unroll
    for (int s = 0; s < BK / 4; s += 2) {
      load_frags(as, bs, s + 1, a1, b1);
      mma(a0, b0);
      if (s + 2 < BK / 4) load_frags(as, bs, s + 2, a0, b0);
      mma(a1, b1);
    }
  };

  if constexpr (TINY && GEMM_EXP == 0) {
    // The small problems (composite weights, their gradient chains) are a handful of workgroups beside the step's large
    // launches: a k-tile costs one memory round trip, not its 16 MFMAs per wavefront.  Two tiles of loads stay in flight
    // (two register sets; the loop is unrolled by two so that the sets are compile-time names).
    if (T > 0) {
      load_tile(0, a_reg, b_reg);
      if (T > 1) load_tile(1, a_reg2, b_reg2);
      store_tile(0, a_reg, b_reg);
      __syncthreads();
      for (int t = 0; t < T; t += 2) {
        if (t + 2 < T) load_tile(t + 2, a_reg, b_reg);
        compute_tile(0);
        __syncthreads();
        if (t + 1 < T) store_tile(0, a_reg2, b_reg2);
        __syncthreads();
        if (t + 1 < T) {
          if (t + 3 < T) load_tile(t + 3, a_reg2, b_reg2);
          compute_tile(0);
          __syncthreads();
          if (t + 2 < T) store_tile(0, a_reg, b_reg);
          __syncthreads();
        }
      }
    }
  } else if (T > 0) {
    load_tile(0, a_reg, b_reg);
    store_tile(0, a_reg, b_reg);
    __syncthreads();
    for (int t = 0; t < T; ++t) {
      const bool more = (GEMM_EXP == 1 || GEMM_EXP == 3) ? false : (t + 1 < T);
      if (more) load_tile(t + 1, a_reg, b_reg);  // global loads in flight during the MFMAs below
      if (GEMM_EXP != 2) compute_tile(0);
      if (GEMM_EXP != 3) __syncthreads();        // every wavefront is done reading the tile
      if (more) store_tile(0, a_reg, b_reg);
      if (GEMM_EXP != 3) __syncthreads();
    }
  }

  // epilogue.  C/D layout of 16x16x4: col = lane & 15, row = 4*(lane >> 4) + reg
  float* Cb;
  int64_t ldc;
  const bool plain = (A_KM && p.nsplit > 1);
  if (plain) {
    Cb = p.slab_base + ((int64_t)zb * p.nsplit + split) * (int64_t)p.M * p.N;
    ldc = p.N;
  } else {
    Cb = p.C + zb * p.c_bs;
    ldc = p.ldc;
  }
  const float* bias = (p.bias && !plain) ? p.bias + zb * p.bias_bs : nullptr;
  const float* rs = (p.row_scale && !plain) ? p.row_scale + zb * p.rs_bs : nullptr;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int row = m0 + wrow + 16 * i + 4 * g + reg;
      if (row >= Mlim) continue;
      const float rscale = rs ? rs[(int64_t)row * p.rs_ld] : 1.f;
      const bool zero = (!plain && p.row_zero) ? (p.row_zero[row] != 0) : false;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int col = n0 + wcol + (SMALL ? 64 : 16) * j + r;
        if (col >= p.N || (SMALL && wcol + 64 * j >= TBN)) continue;
        float v = acc[i][j][reg];
        if (!plain) {
          if (p.accumulate) v += Cb[(int64_t)row * ldc + col];
          if (bias) v = fmaf(bias[col], rscale, v);
          if (zero) v = 0.f;
          if (p.relu) v = fmaxf(v, 0.f);
          if (p.relu_src) v = (p.relu_src[(int64_t)row * p.relu_ld + col] > 0.f) ? v : 0.f;
        }
        Cb[(int64_t)row * ldc + col] = v;
      }
    }
  }
}

#define GEMM_LDS_DECL                                                                    \
  __shared__ __attribute__((aligned(16))) float lds_a[AS_FLOATS]; \
  __shared__ __attribute__((aligned(16))) float lds_b[BS_FLOATS]

template <bool A_KM, bool B_KM, int TILE, bool VEC>
__global__ __launch_bounds__(GEMM_THREADS) void gemm_f32_kernel(const GemmDev p) {
  GEMM_LDS_DECL;
  int zb = blockIdx.z, split = 0;
  if (A_KM && p.nsplit > 1) { split = zb % p.nsplit; zb /= p.nsplit; }
  gemm_tile<A_KM, B_KM, TILE, VEC>(p, blockIdx.x, blockIdx.y, zb, split, lds_a, lds_b);
}

// ---------------------------------------------------------------------------------------------
// fp32 contraction on the BF16 matrix cores by a 3-way split ("bf16x3"): x = x1 + x2 + x3 with x1 = hi16(x),
// x2 = hi16(x - x1), x3 = hi16(x - x1 - x2) (24 mantissa bits in three bf16 pieces, residuals exact), and
//   a.b ~= a1b1 + (a1b2 + a2b1) + (a2b2 + a1b3 + a3b1)          (dropped terms <= 2^-24 |a||b|)
// accumulated in fp32 by v_mfma_f32_16x16x32_bf16.  Six bf16 MFMAs (16 cycles each) replace eight fp32 MFMAs
// (32 cycles each) per 16x16x32 block: 2.7x the matrix rate at fp32-level accuracy (validated against fp64 in
// tests/test_gpu_kernels.py).  Row-major A and B only (the nn.Linear layout); same tile, staging and epilogue as
// the fp32 kernel.  LDS image per piece: [row][32 bf16] = 64-byte rows, 16-byte k-chunks XOR-swizzled with
// f(row >> 2) = {0,3,2,1} so that every ds_read_b128 fragment read covers all 64 banks exactly once.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define BX_A_PIECE (BM * 64)     // bytes per piece of the A tile
#define BX_B_PIECE (BN * 64)

template <int N, typename F, int I = 0>
__device__ __forceinline__ void bx_for(F&& f) {
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); bx_for<N, F, I + 1>(static_cast<F&&>(f)); }
}
__device__ __forceinline__ int bx_swz(int row) { return (4 - ((row >> 2) & 3)) & 3; }   // {0,3,2,1}

// 3-way bf16 split of four fp32 values, packed two per dword (element e in the low half of dword e/2):
// 4 VALU per value (and, sub, and, sub: the residuals are exact) + one v_perm_b32 per pair and piece
__device__ __forceinline__ void bx_split4(const float4 v, uint2& o1, uint2& o2, uint2& o3) {
  const float x[4] = {v.x, v.y, v.z, v.w};
  uint32_t p1[4], p2[4], p3[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    p1[e] = __float_as_uint(x[e]);
    const float r1 = x[e] - __uint_as_float(p1[e] & 0xFFFF0000u);                 // exact
    p2[e] = __float_as_uint(r1);
    const float r2 = r1 - __uint_as_float(p2[e] & 0xFFFF0000u);                   // exact
    p3[e] = __float_as_uint(r2);
  }
  constexpr uint32_t HI2 = 0x07060302u;                                            // {hi16(a), hi16(b)} -> a in the upper half
  o1 = uint2{__builtin_amdgcn_perm(p1[1], p1[0], HI2), __builtin_amdgcn_perm(p1[3], p1[2], HI2)};
  o2 = uint2{__builtin_amdgcn_perm(p2[1], p2[0], HI2), __builtin_amdgcn_perm(p2[3], p2[2], HI2)};
  o3 = uint2{__builtin_amdgcn_perm(p3[1], p3[0], HI2), __builtin_amdgcn_perm(p3[3], p3[2], HI2)};
}

__device__ __forceinline__ void bx_split_store(char* base, int piece_bytes, int row, int c4, const float4 v) {
  // float4 = 4 consecutive k of one row: k = 4*c4 .. 4*c4+3 -> 16-byte chunk c4 >> 1, half c4 & 1
  const int off = row * 64 + (((c4 >> 1) ^ bx_swz(row)) << 4) + ((c4 & 1) << 3);
  uint2 o1, o2, o3;
  bx_split4(v, o1, o2, o3);
  *reinterpret_cast<uint2*>(base + off) = o1;
  *reinterpret_cast<uint2*>(base + piece_bytes + off) = o2;
  *reinterpret_cast<uint2*>(base + 2 * piece_bytes + off) = o3;
}

// ---------------------------------------------------------------------------------------------
// Second split format (FMT 1, "fp16x2"): every fp32 operand element is the sum of TWO fp16 pieces of its value times a power of
// two, x * 2^s = h + l with h = fp16(x 2^s), l = fp16(x 2^s - h) (22 significant bits), and a 16x16x32 block takes THREE
// v_mfma_f32_16x16x32_f16 (h.l + l.h + h.h) instead of the six bf16 products: half the matrix-pipe work and two thirds of the
// staging traffic at the same accuracy on the fp64 test (tests/test_gpu_kernels.py).  fp16 has 5 exponent bits, so the power of
// two is chosen per ROW of each operand, from the row's largest magnitude, so that the largest scaled value lies in
// [2^12, 2^15) (weight rows at the top, activation rows with HX_GROW binades of headroom): elements down to 2^-16 of their
// row's maximum keep all 22 bits, smaller ones an absolute error of 2^-38 of the
// row maximum - far below the rounding of an fp32 accumulation, but a NORM-wise bound where bf16x3 (fp32's own exponent range)
// gives a component-wise one.  Weight images carry the exponent of each image row behind the tiles (bimg_h_kernel); activation
// rows are scaled by the consuming kernel from a RUNNING maximum over the k-tiles it has seen: when a later tile holds a larger
// value the row's accumulators are multiplied by the (exact) power of two between the old and the new scale, the way an online
// softmax rescales its sums.  Powers of two throughout: scaling and unscaling are exact.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define HX_EMIN 1             // clamp of the biased exponent of a row maximum: rows of zeros and denormals scale like 2^-126
#define HX_EMAX 254
#define HX_TOP 141            // a row whose exponent is E is scaled by 2^(HX_TOP - E): maximum in [2^14, 2^15)
#define HX_GROW 2             // binades of headroom a row takes when its running maximum outgrows its scale (see bx_split_rows)
template <int FMT> struct BxFmt;
template <> struct BxFmt<0> { static constexpr int NP = 3; };
template <> struct BxFmt<1> { static constexpr int NP = 2; };
// process-wide choice of the format of every pre-split image and of the kernels that read them (A/B switch: PFO_BX_FMT=0)
#define PFO_DEFAULT_BX_FMT 1
#define PFO_DEFAULT_TN_FMT 1     // the weight-gradient tile: 1 = two fp16 pieces with one scale per operand and K-slab, 0 = bf16x3
#ifndef BX_AREG_OCC
#define BX_AREG_OCC 2
#endif
int pfo_bx_fmt() {
  static const int f = getenv("PFO_BX_FMT") ? (atoi(getenv("PFO_BX_FMT")) != 0 ? 1 : 0) : PFO_DEFAULT_BX_FMT;
  return f;
}
__device__ __forceinline__ int hx_exp_of(float m) {           // biased exponent of |m| (m >= 0), clamped
  return min(max((int)(__float_as_uint(m) >> 23), HX_EMIN), HX_EMAX);
}
__device__ __forceinline__ int hx_exp_of_bits(uint32_t b) { return min(max((int)(b >> 23), HX_EMIN), HX_EMAX); }
// maximum over the four lanes r, r + 16, r + 32, r + 48 (the four k-groups of one fragment row), in every one of them: two
// lane-swap VALU instructions (gfx950), no LDS round trip.  Bits of non-negative floats order like unsigned integers.
__device__ __forceinline__ uint32_t hx_max_over_g(uint32_t u) {
  const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  u = max(a[0], a[1]);
  const auto b = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return max(b[0], b[1]);
}
__device__ __forceinline__ float hx_absmax8(const float4 a, const float4 b) {
  return fmaxf(fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w))),
               fmaxf(fmaxf(fabsf(b.x), fabsf(b.y)), fmaxf(fabsf(b.z), fabsf(b.w))));
}
// four values times 2^se -> the two fp16 pieces, packed two per dword (element e in the low half of dword e/2)
__device__ __forceinline__ void hx_split4(const float4 v, int se, uint2& oh, uint2& ol) {
  const f32x2 p0 = {__builtin_amdgcn_ldexpf(v.x, se), __builtin_amdgcn_ldexpf(v.y, se)};
  const f32x2 p1 = {__builtin_amdgcn_ldexpf(v.z, se), __builtin_amdgcn_ldexpf(v.w, se)};
  const f16x2 h0 = __builtin_convertvector(p0, f16x2), h1 = __builtin_convertvector(p1, f16x2);
  const f16x2 l0 = __builtin_convertvector(p0 - __builtin_convertvector(h0, f32x2), f16x2);   // residual exact in fp32
  const f16x2 l1 = __builtin_convertvector(p1 - __builtin_convertvector(h1, f32x2), f16x2);
  oh = uint2{__builtin_bit_cast(uint32_t, h0), __builtin_bit_cast(uint32_t, h1)};
  ol = uint2{__builtin_bit_cast(uint32_t, l0), __builtin_bit_cast(uint32_t, l1)};
}
// the products of one 16x16x32 block, smallest terms first; x = the operand whose rows become accumulator COLUMNS (first MFMA
// operand), y = the other one.  FMT 0: pieces hi | mid | lo (bf16), FMT 1: hi | lo (fp16)
template <int FMT>
__device__ __forceinline__ f32x4 bx_mma(const u32x4 (&x)[BxFmt<FMT>::NP], const u32x4 (&y)[BxFmt<FMT>::NP], f32x4 c) {
  if constexpr (FMT == 0) {
    typedef __bf16 v8 __attribute__((ext_vector_type(8)));
#define BXM(q, w) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8, x[q]), __builtin_bit_cast(v8, y[w]), c, 0, 0, 0)
    BXM(0, 2); BXM(2, 0); BXM(1, 1); BXM(0, 1); BXM(1, 0); BXM(0, 0);
#undef BXM
  } else {
#define BXM(q, w) c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, x[q]), __builtin_bit_cast(f16x8, y[w]), c, 0, 0, 0)
    BXM(0, 1); BXM(1, 0); BXM(0, 0);
#undef BXM
  }
  return c;
}
// product Q of bx_mma's list alone (two accumulators interleaved: a dependent MFMA waits for its predecessor's result)
template <int FMT, int Q>
__device__ __forceinline__ f32x4 bx_mma_q(const u32x4 (&x)[BxFmt<FMT>::NP], const u32x4 (&y)[BxFmt<FMT>::NP], f32x4 c) {
  if constexpr (FMT == 0) {
    typedef __bf16 v8 __attribute__((ext_vector_type(8)));
    constexpr int qx[6] = {0, 2, 1, 0, 1, 0}, qy[6] = {2, 0, 1, 1, 0, 0};
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8, x[qx[Q]]), __builtin_bit_cast(v8, y[qy[Q]]), c, 0, 0, 0);
  } else {
    constexpr int qx[3] = {0, 1, 0}, qy[3] = {1, 0, 0};
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, x[qx[Q]]), __builtin_bit_cast(f16x8, y[qy[Q]]), c, 0, 0, 0);
  }
}
__device__ __forceinline__ f32x4 hx_scale4(const f32x4 a, const int4 e, int base) {          // a[c] * 2^(base + e[c])
  return f32x4{__builtin_amdgcn_ldexpf(a[0], base + e.x), __builtin_amdgcn_ldexpf(a[1], base + e.y),
               __builtin_amdgcn_ldexpf(a[2], base + e.z), __builtin_amdgcn_ldexpf(a[3], base + e.w)};
}

// The A fragments of one k-tile from the raw rows a lane holds ([strip][half]: 8 consecutive k of one row per strip).  FMT 1
// keeps the running row scale (rowE) and moves the row's accumulators when a larger value arrives.
template <int FMT, int NJ, int S = 2>
__device__ __forceinline__ void bx_split_rows(const float4 (&a_raw)[S][2], u32x4 (&a)[S][BxFmt<FMT>::NP], int (&rowE)[S], f32x4 (&acc)[S][NJ],
                                              const bool first) {
#pragma unroll
  for (int i = 0; i < S; ++i) {
    if constexpr (FMT == 0) {
      uint2 lo[3], hi[3];
      bx_split4(a_raw[i][0], lo[0], lo[1], lo[2]);
      bx_split4(a_raw[i][1], hi[0], hi[1], hi[2]);
#pragma unroll
      for (int q = 0; q < 3; ++q) a[i][q] = u32x4{lo[q].x, lo[q].y, hi[q].x, hi[q].y};
    } else {
      // the row's maximum over this k-tile: 8 values here, the other 24 in the lanes 16 / 32 / 48 further on
      const int e = hx_exp_of_bits(hx_max_over_g(__float_as_uint(hx_absmax8(a_raw[i][0], a_raw[i][1]))));
      if (e > rowE[i]) {                                     // a larger value than any before: the sums so far move to the new scale
        // HX_GROW binades of headroom with every move (a maximum that creeps up does not move the row at every tile); on the
        // first tile the sums are still zero and nothing has to move
        const int en = min(e + HX_GROW, HX_EMAX);
        if (!first) {
          const int d = rowE[i] - en;
#pragma unroll
          for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[i][j][c] = __builtin_amdgcn_ldexpf(acc[i][j][c], d);
        }
        rowE[i] = en;
      }
      const int se = HX_TOP - rowE[i];
      uint2 h0, l0, h1, l1;
      hx_split4(a_raw[i][0], se, h0, l0);
      hx_split4(a_raw[i][1], se, h1, l1);
      a[i][0] = u32x4{h0.x, h0.y, h1.x, h1.y};
      a[i][1] = u32x4{l0.x, l0.y, l1.x, l1.y};
    }
  }
}

// epilogue of the operand-swapped bf16x3 kernels: four consecutive columns col..col+3 of one output row
// GRU gate backward for hidden units col .. col+3 of table row `row` (PfoGemm::gg_*): v = the row's query-side gradient
__device__ __forceinline__ void bx_gru_gates4(const GemmDev& p, int row, int col, const f32x4 v) {
  const int D = p.N;
  float o[4][4] = {};                                  // [dpr | dpz | dpn | dpn r][unit]
  if (p.gg_hm[row]) {
    const int64_t e = (int64_t)row * D + col;
    const float* gs = p.gg_gates + (int64_t)row * 4 * D + col;
    const float4 h4 = *reinterpret_cast<const float4*>(p.gg_h + e), k4 = *reinterpret_cast<const float4*>(p.gg_dh0 + e);
    const float4 r4 = *reinterpret_cast<const float4*>(gs), z4 = *reinterpret_cast<const float4*>(gs + D);
    const float4 n4 = *reinterpret_cast<const float4*>(gs + 2 * D), g4 = *reinterpret_cast<const float4*>(gs + 3 * D);
    const float dh[4] = {k4.x + v[0], k4.y + v[1], k4.z + v[2], k4.w + v[3]};      // key side (scattered by the attention backward) + query side
    const float hh[4] = {h4.x, h4.y, h4.z, h4.w}, rr[4] = {r4.x, r4.y, r4.z, r4.w}, zz[4] = {z4.x, z4.y, z4.z, z4.w};
    const float nn[4] = {n4.x, n4.y, n4.z, n4.w}, gg[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float dn = dh[u] * (1.f - zz[u]);
      const float dz = dh[u] * (hh[u] - nn[u]);
      const float dpn = dn * (1.f - nn[u] * nn[u]);
      const float dr = dpn * gg[u];
      o[0][u] = dr * rr[u] * (1.f - rr[u]);
      o[1][u] = dz * zz[u] * (1.f - zz[u]);
      o[2][u] = dpn;
      o[3][u] = dpn * rr[u];
    }
  }
  float* gis = p.gg_dgi + (int64_t)row * 3 * D + col;
  float* ghs = p.gg_dgh + (int64_t)row * 3 * D + col;
  const float4 a = {o[0][0], o[0][1], o[0][2], o[0][3]}, b = {o[1][0], o[1][1], o[1][2], o[1][3]};
  const float4 c = {o[2][0], o[2][1], o[2][2], o[2][3]}, c2 = {o[3][0], o[3][1], o[3][2], o[3][3]};
  *reinterpret_cast<float4*>(gis) = a; *reinterpret_cast<float4*>(gis + D) = b; *reinterpret_cast<float4*>(gis + 2 * D) = c;
  *reinterpret_cast<float4*>(ghs) = a; *reinterpret_cast<float4*>(ghs + D) = b; *reinterpret_cast<float4*>(ghs + 2 * D) = c2;
}

__device__ __forceinline__ void bx_store4(const GemmDev& p, float* Cb, int64_t ldc, const float* bias, float rscale, bool zero,
                                          bool n4, int row, int col, const f32x4 a, const float* addrow = nullptr) {
  float v[4] = {a[0], a[1], a[2], a[3]};
  float* cp = Cb + (int64_t)row * ldc + col;
  if (n4) {
    if (p.accumulate) { const float4 o = *reinterpret_cast<const float4*>(cp); v[0] += o.x; v[1] += o.y; v[2] += o.z; v[3] += o.w; }
    if (bias) {
      const float4 bv = *reinterpret_cast<const float4*>(bias + col);
      v[0] = fmaf(bv.x, rscale, v[0]); v[1] = fmaf(bv.y, rscale, v[1]); v[2] = fmaf(bv.z, rscale, v[2]); v[3] = fmaf(bv.w, rscale, v[3]);
    }
    if (addrow) { const float4 o = *reinterpret_cast<const float4*>(addrow + col); v[0] += o.x; v[1] += o.y; v[2] += o.z; v[3] += o.w; }
    if (zero) { v[0] = v[1] = v[2] = v[3] = 0.f; }
    if (p.relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
    if (p.relu_src) {
      const float4 m = *reinterpret_cast<const float4*>(p.relu_src + (int64_t)row * p.relu_ld + col);
      v[0] = m.x > 0.f ? v[0] : 0.f; v[1] = m.y > 0.f ? v[1] : 0.f; v[2] = m.z > 0.f ? v[2] : 0.f; v[3] = m.w > 0.f ? v[3] : 0.f;
    }
    if (BXA_NT) __builtin_nontemporal_store(f32x4{v[0], v[1], v[2], v[3]}, reinterpret_cast<f32x4*>(cp));
    else *reinterpret_cast<float4*>(cp) = float4{v[0], v[1], v[2], v[3]};
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (col + e >= p.N) continue;
      float x = v[e];
      if (p.accumulate) x += cp[e];
      if (bias) x = fmaf(bias[col + e], rscale, x);
      if (addrow) x += addrow[col + e];
      if (zero) x = 0.f;
      if (p.relu) x = fmaxf(x, 0.f);
      if (p.relu_src) x = (p.relu_src[(int64_t)row * p.relu_ld + col + e] > 0.f) ? x : 0.f;
      cp[e] = x;
    }
  }
}
__device__ __forceinline__ bool bx_n4(const GemmDev& p, const float* Cb, int64_t ldc, const float* bias) {
  return (p.N & 3) == 0 && (ldc & 3) == 0 && (((uintptr_t)Cb) & 15) == 0 &&
         (!p.add_src || ((p.add_ld & 3) == 0 && (((uintptr_t)p.add_src) & 15) == 0)) &&
         (!p.relu_src || ((p.relu_ld & 3) == 0 && (((uintptr_t)p.relu_src) & 15) == 0)) && (!bias || (((uintptr_t)bias) & 15) == 0);
}

template <bool BSPLIT>
__global__ __launch_bounds__(GEMM_THREADS, 2) void gemm_bf16x3_kernel(const GemmDev p) {
  __shared__ __attribute__((aligned(16))) char lds[3 * BX_A_PIECE + 3 * BX_B_PIECE];
  char* const As = lds;
  char* const Bs = lds + 3 * BX_A_PIECE;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int zb = blockIdx.z;
  int Mlim = p.M;
  if (p.m_dev) Mlim = min(Mlim, *p.m_dev);
  if (m0 >= Mlim) return;
  const int wrow = 32 * wave;

  f32x4 acc[2][11];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 11; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int T0 = (p.K[0] + BK - 1) / BK;
  const int T1 = (p.K[1] > 0 && p.A[1]) ? (p.K[1] + BK - 1) / BK : 0;
  const int T = T0 + T1;
  constexpr int NB = BSPLIT ? 9 : 6;                  // staged 16-byte units of B per thread
  float4 a_reg[4], b_reg[6];
  f32x4 i_reg[NB];                                    // (vector type: a struct copy here would pin the array in scratch)
  const float* a_row[4];
  const float* b_row[6];
  bool a_ok[4], b_ok[6];
  int Ks = 0, cur_src = -1;
  const float* safe = p.A[0];
  // pre-split B: tile t, piece q of the rows n0.. is one contiguous BX_B_PIECE-byte run of the image
  const char* img = nullptr;                           // bound per source in bind_src
  const int64_t img_piece = (int64_t)p.b_img_rows * 64;

  auto bind_src = [&](int src) {
    cur_src = src;
    Ks = p.K[src];
    const float* Ab = p.A[src] + zb * p.a_bs[src];
    const float* Bb = p.B[src] + zb * p.b_bs[src];
    if constexpr (BSPLIT) img = reinterpret_cast<const char*>(src == 0 ? p.b_img : p.b_img2) + (int64_t)n0 * 64;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int gm = m0 + ((tid + 256 * i) >> 3);
      a_ok[i] = gm < Mlim;
      int64_t ridx = a_ok[i] ? gm : 0;
      if (a_ok[i] && p.a_idx[src]) ridx = p.a_idx[src][gm];
      a_row[i] = Ab + ridx * p.lda[src];
    }
    if constexpr (!BSPLIT) {
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const int f = tid + 256 * i;
        const int gn = n0 + (f >> 3);
        b_ok[i] = (f < BN * 8) && gn < p.N;
        b_row[i] = Bb + (int64_t)(b_ok[i] ? gn : 0) * p.ldb[src];
      }
    }
  };
  auto load_tile = [&](int t) {
    const int src = t < T0 ? 0 : 1;
    if (src != cur_src) bind_src(src);
    const int k0 = (src == 0) ? t * BK : (t - T0) * BK;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = k0 + 4 * ((tid + 256 * i) & 7);
      a_reg[i] = ld4<true>(a_row[i] + k, a_ok[i] ? Ks - k : 0, safe);
    }
    if constexpr (BSPLIT) {
      const char* tile = img + (int64_t)(src == 0 ? t : t - T0) * 3 * img_piece;
      // 3 pieces x 704 units of 16 bytes = 8.25 units per thread (tail clamped); piece boundaries fall at units 704, 1408
      bx_for<9>([&](auto uc) {
        constexpr int u = decltype(uc)::value;
        const int unit = min(tid + 256 * u, 3 * (BX_B_PIECE / 16) - 1);
        const int q = (unit >= 2 * (BX_B_PIECE / 16)) ? 2 : (unit >= BX_B_PIECE / 16 ? 1 : 0);
        i_reg[u] = *reinterpret_cast<const f32x4*>(tile + q * img_piece + (unit - q * (BX_B_PIECE / 16)) * 16);
      });
    } else {
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const int k = k0 + 4 * ((tid + 256 * i) & 7);
        b_reg[i] = ld4<true>(b_row[i] + k, b_ok[i] ? Ks - k : 0, safe);
      }
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int f = tid + 256 * i;
      bx_split_store(As, BX_A_PIECE, f >> 3, f & 7, a_reg[i]);
    }
    if constexpr (BSPLIT) {
      bx_for<9>([&](auto uc) {
        constexpr int u = decltype(uc)::value;
        const int unit = tid + 256 * u;
        if (u < 8 || unit < 3 * (BX_B_PIECE / 16)) *reinterpret_cast<f32x4*>(Bs + unit * 16) = i_reg[u];
      });
    } else {
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const int f = tid + 256 * i;
        if (f < BN * 8) bx_split_store(Bs, BX_B_PIECE, f >> 3, f & 7, b_reg[i]);
      }
    }
  };
  const int frag_off = r * 64 + ((g ^ bx_swz(r)) << 4);     // rows of a strip start at multiples of 16: same swizzle
  const bool strip_on[2] = {m0 + wrow < p.M, m0 + wrow + 16 < p.M};
  auto compute_tile = [&]() {
    bf16x8 a[2][3];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int q = 0; q < 3; ++q) a[i][q] = *reinterpret_cast<const bf16x8*>(As + q * BX_A_PIECE + (wrow + 16 * i) * 64 + frag_off);
    // B fragments are double-buffered in registers: the LDS reads of column tile j+1 are in flight during the 12
    // MFMAs of tile j (with one register set the reads could only issue after those MFMAs, latency fully exposed)
    auto ldb = [&](bf16x8 (&b)[3], int j) {
#pragma unroll
      for (int q = 0; q < 3; ++q) b[q] = *reinterpret_cast<const bf16x8*>(Bs + q * BX_B_PIECE + (16 * j) * 64 + frag_off);
    };
    auto mma = [&](const bf16x8 (&b)[3], int j) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        if (!strip_on[i]) continue;
        f32x4 c = acc[i][j];
        // operands swapped (weights as the MFMA "A" side): the accumulator holds the TRANSPOSED tile, lane (r, g)
        // owns row r and the four consecutive columns 4g..4g+3 - 16-byte epilogue accesses.  Smallest terms first.
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[0], a[i][2], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[2], a[i][0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[1], a[i][1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[0], a[i][1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[1], a[i][0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[0], a[i][0], c, 0, 0, 0);
        acc[i][j] = c;
      }
    };
    bf16x8 b0[3], b1[3];
    ldb(b0, 0);
#pragma unroll
    for (int j = 0; j < 11; j += 2) {
      if (j + 1 < 11) ldb(b1, j + 1);
      mma(b0, j);
      if (j + 2 < 11) ldb(b0, j + 2);
      if (j + 1 < 11) mma(b1, j + 1);
    }
  };

  if (T > 0) {
    load_tile(0);
    store_tile();
    __syncthreads();
    for (int t = 0; t < T; ++t) {
      const bool more = (BX_EXP == 1) ? false : (t + 1 < T);
      if (more) load_tile(t + 1);
      if (BX_EXP != 2) compute_tile();
      __syncthreads();
      if (more && BX_EXP != 3) store_tile();
      __syncthreads();
    }
  }

  float* Cb = p.C + zb * p.c_bs;
  const int64_t ldc = p.ldc;
  const float* bias = p.bias ? p.bias + zb * p.bias_bs : nullptr;
  const float* rs = p.row_scale ? p.row_scale + zb * p.rs_bs : nullptr;
  const bool n4 = bx_n4(p, Cb, ldc, bias);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = m0 + wrow + 16 * i + r;
    if (row >= Mlim) continue;
    const float rscale = rs ? rs[(int64_t)row * p.rs_ld] : 1.f;
    const bool zero = p.row_zero ? (p.row_zero[row] != 0) : false;
    const float* addrow = p.add_src ? p.add_src + (int64_t)(p.add_idx ? p.add_idx[row] : row) * p.add_ld : nullptr;
#pragma unroll
    for (int j = 0; j < 11; ++j) {
      const int col = n0 + 16 * j + 4 * g;
      if (col < p.N) bx_store4(p, Cb, ldc, bias, rscale, zero, n4, row, col, acc[i][j], addrow);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Second form of the pre-split-image kernel: the A tile never touches LDS.  Every wavefront loads the 32 rows it owns
// straight in MFMA fragment order (lane (r, g): row r, k = 8g..8g+7 = two float4), splits them in registers into the
// three bf16x8 fragments, and only the shared B image goes through LDS - double-buffered, so a k-tile costs ONE barrier
// and the copy of tile t+1 into the other buffer runs beside the MFMAs of tile t.
template <int WAVES, int FMT, int S = 2>
__device__ __forceinline__ void bx_areg_body(const GemmDev& p, char* lds) {
  constexpr int NP = BxFmt<FMT>::NP;
  constexpr int NTHR = 64 * WAVES;                   // threads per workgroup; S strips of 16 rows per wavefront
  constexpr int UNITS = NP * (BX_B_PIECE / 16);      // 16-byte units of one B image tile (a multiple of 64)
  constexpr int NDMA = (UNITS + NTHR - 1) / NTHR;    // rounds of 16-byte units per thread
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  int tile_m = blockIdx.x, tile_n = blockIdx.y;
  BXA_STAMP(0);
#if BXA_STAMPS
  if (threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.x < 4096) {
    uint32_t hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    g_bxa_stamps[blockIdx.x * 8 + 7] = ((uint64_t)xcc << 32) | hw;
  }
#endif
  if (BXA_STAGGER > 0) {
    const int lin = blockIdx.x + blockIdx.y * gridDim.x;
    if (lin < 512 && ((lin >> 8) & 1)) {
#pragma unroll 1
      for (int k = 0; k < BXA_STAGGER; ++k) __builtin_amdgcn_s_sleep(127);
    }
  }
  if (p.xcd_tn > 0 && !xcd_tile(blockIdx.x, p.xcd_tm, p.xcd_tn, tile_m, tile_n)) return;
  const int m0 = tile_m * (16 * S * WAVES), n0 = tile_n * BN;
  int Mlim = p.M;
  if (p.m_dev) Mlim = min(Mlim, *p.m_dev);
  if (m0 >= Mlim || n0 >= p.N) return;
  const int wrow = 16 * S * wave;

  f32x4 acc[S][11];
#pragma unroll
  for (int i = 0; i < S; ++i)
#pragma unroll
    for (int j = 0; j < 11; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int T0 = (p.K[0] + BK - 1) / BK;
  const int T1 = (p.K[1] > 0 && p.A[1]) ? (p.K[1] + BK - 1) / BK : 0;
  const int T = T0 + T1;
  const float* safe = p.A[0];
  const int64_t img_piece = (int64_t)p.b_img_rows * 64;
  // rows of this lane (one per strip) for both sources
  const float* a_row[2][S];
  bool a_ok[S];
#pragma unroll
  for (int i = 0; i < S; ++i) {
    const int gm = m0 + wrow + 16 * i + r;
    a_ok[i] = gm < Mlim;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      int64_t ridx = a_ok[i] ? gm : 0;
      if (a_ok[i] && (s2 == 0 || T1 > 0) && p.a_idx[s2]) ridx = p.a_idx[s2][gm];
      a_row[s2][i] = (s2 == 0 || T1 > 0) ? p.A[s2] + ridx * p.lda[s2] + 8 * g : p.A[0];
    }
  }
  const char* img[2] = {reinterpret_cast<const char*>(p.b_img) + (int64_t)n0 * 64,
                        T1 > 0 ? reinterpret_cast<const char*>(p.b_img2) + (int64_t)n0 * 64 : nullptr};
  // FMT 1: exponents of the image rows (this block's columns) behind the tiles of each image
  const int32_t* bexp[2] = {reinterpret_cast<const int32_t*>(reinterpret_cast<const char*>(p.b_img) + (int64_t)T0 * NP * img_piece) + n0,
                            T1 > 0 ? reinterpret_cast<const int32_t*>(reinterpret_cast<const char*>(p.b_img2) + (int64_t)T1 * NP * img_piece) + n0
                                   : nullptr};

  float4 a_raw[2][S][2];              // [set][strip][half]: k = k0 + 8g + 4*half ..; tile t lives in set t & 1
  typedef __attribute__((address_space(1))) const void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  // A rows of tile t -> registers, two tiles ahead of the MFMAs (HBM latency under load exceeds one k-tile of MFMAs now that a
  // tile is 66 of them)
  auto load_a = [&](int t, auto setc) {
    constexpr int set = decltype(setc)::value;
    const int src = t < T0 ? 0 : 1;
    const int ts = src == 0 ? t : t - T0;
    const int k = ts * BK + 8 * g;
    const int Ks = p.K[src];
    const int toff = (BXA_ABL & 4) ? 0 : ts * BK;
#pragma unroll
    for (int i = 0; i < S; ++i)
#pragma unroll
      for (int h = 0; h < 2; ++h)
        a_raw[set][i][h] = ld4<true>(a_row[src][i] + toff + 4 * h, a_ok[i] ? Ks - (k + 4 * h) : 0, safe);
  };
  // B image tile -> LDS by asynchronous global->LDS loads (the LDS image IS the global image: a lane-linear copy, no
  // staging registers, no ds_write); `buf` is the buffer being filled for tile t
  auto load_b = [&](int t, int buf) {
    if ((BXA_ABL & 8) && t > 0) return;
    const int src = t < T0 ? 0 : 1;
    const int ts = src == 0 ? t : t - T0;
    const char* tile = img[src] + (int64_t)ts * NP * img_piece;
    char* Bs = lds + buf * NP * BX_B_PIECE;
    bx_for<NDMA>([&](auto uc) {
      constexpr int u = decltype(uc)::value;
      const int unit0 = 64 * wave + NTHR * u;                  // the last round covers some of the wavefronts only (wave-uniform)
      if (unit0 < UNITS) {
        const int unit = unit0 + lane;
        const int q = unit / (BX_B_PIECE / 16);
        __builtin_amdgcn_global_load_lds((gptr_t)(tile + q * img_piece + (unit - q * (BX_B_PIECE / 16)) * 16),
                                         (lptr_t)(Bs + unit0 * 16), 16, 0, 0);
      }
    });
  };
  u32x4 a[S][NP];
  int rowE[S];
#pragma unroll
  for (int i = 0; i < S; ++i) rowE[i] = HX_EMIN;    // FMT 1: biased exponent of the running maximum of this lane's two rows
  auto split_a = [&](auto setc, bool first) {
    if ((BXA_ABL & 16) && !first) {
#pragma unroll
      for (int i = 0; i < S; ++i)
#pragma unroll
        for (int h = 0; h < 2; ++h) asm volatile("" :: "v"(a_raw[decltype(setc)::value][i][h].x), "v"(a_raw[decltype(setc)::value][i][h].w));
      return;
    }
    bx_split_rows<FMT, 11, S>(a_raw[decltype(setc)::value], a, rowE, acc, first);
  };
  const int frag_off = r * 64 + ((g ^ bx_swz(r)) << 4);
  auto compute_tile = [&](int buf) {
    const char* Bs = lds + buf * NP * BX_B_PIECE;
    auto ldb = [&](u32x4 (&b)[NP], int j) {
#pragma unroll
      for (int q = 0; q < NP; ++q) b[q] = *reinterpret_cast<const u32x4*>(Bs + q * BX_B_PIECE + (16 * j) * 64 + frag_off);
    };
    auto mma = [&](const u32x4 (&b)[NP], int j) {
      if (BXA_ABL & 1) {
#pragma unroll
        for (int q = 0; q < NP; ++q) asm volatile("" :: "v"(b[q]), "v"(a[0][q]), "v"(a[1][q]));
        return;
      }
#pragma unroll
      for (int i = 0; i < S; ++i) acc[i][j] = bx_mma<FMT>(b, a[i], acc[i][j]);     // operands swapped: image rows = accumulator columns
    };
    if constexpr (S == 1) {
      // one strip: the products of a column tile are ONE dependent chain - two column tiles are multiplied product by product
      // in turn, and the fragments of the next pair are read meanwhile (four register sets)
      constexpr int NQ = FMT == 0 ? 6 : 3;
      u32x4 bb[4][NP];
      ldb(bb[0], 0); ldb(bb[1], 1);
      bx_for<6>([&](auto pc) {
        constexpr int j = 2 * decltype(pc)::value, cs = (decltype(pc)::value & 1) * 2;
        if constexpr (j + 2 < 11) ldb(bb[2 - cs], j + 2);
        if constexpr (j + 3 < 11) ldb(bb[3 - cs], j + 3);
        if (BXA_ABL & 1) {
#pragma unroll
          for (int q = 0; q < NP; ++q) asm volatile("" :: "v"(bb[cs][q]), "v"(bb[cs + 1][q]), "v"(a[0][q]));
        } else {
          bx_for<NQ>([&](auto qc) {
            constexpr int Q = decltype(qc)::value;
            acc[0][j] = bx_mma_q<FMT, Q>(bb[cs], a[0], acc[0][j]);
            if constexpr (j + 1 < 11) acc[0][j + 1] = bx_mma_q<FMT, Q>(bb[cs + 1], a[0], acc[0][j + 1]);
          });
        }
      });
    } else {
    u32x4 b0[NP], b1[NP];
    ldb(b0, 0);
#pragma unroll
    for (int j = 0; j < 11; j += 2) {
      if (j + 1 < 11) ldb(b1, j + 1);
      mma(b0, j);
      if (j + 2 < 11) ldb(b0, j + 2);
      if (j + 1 < 11) mma(b1, j + 1);
    }
    }
  };

  using C0 = std::integral_constant<int, 0>;
  using C1 = std::integral_constant<int, 1>;
#if BXA_STAMPS
  uint32_t st_acc[4] = {0, 0, 0, 0};      // shader cycles of wavefront 0 in: MFMA + LDS reads | wait for the loads | row split | barrier
  uint64_t st_loop0 = 0;
#endif
  // one k-tile: the fragments of tile t are in `a`; queue the image of t+1 and the rows of t+2, multiply, split the rows of t+1
  auto step = [&](int t, auto curc) {
    constexpr int cur = decltype(curc)::value;                 // == t & 1
    const bool more = t + 1 < T, more2 = t + 2 < T;
    if (more) load_b(t + 1, (t + 1) & 1);
    if (more2) load_a(t + 2, curc);                            // (the raw rows of tile t were split before this call)
    if constexpr (FMT == 1) {
      if (t == T0 && T1 > 0) {                     // second source: its image rows have their own scales - move the sums over
#pragma unroll
        for (int j = 0; j < 11; ++j) {
          const int4 e1 = *reinterpret_cast<const int4*>(bexp[0] + 16 * j + 4 * g);
          const int4 e2 = *reinterpret_cast<const int4*>(bexp[1] + 16 * j + 4 * g);
          const int4 d = {e1.x - e2.x, e1.y - e2.y, e1.z - e2.z, e1.w - e2.w};
#pragma unroll
          for (int i = 0; i < S; ++i) acc[i][j] = hx_scale4(acc[i][j], d, 0);
        }
      }
    }
#if BXA_STAMPS
    const uint64_t c0 = __builtin_amdgcn_s_memtime();
#endif
    compute_tile(cur);
#if BXA_STAMPS
    asm volatile("s_nop 0" ::: "memory");
    const uint64_t c1 = __builtin_amdgcn_s_memtime();
    uint64_t c2 = c1, c3 = c1;
#endif
    if (more) {
      // the DMA is ordered only by the issuing wave's vmcnt + the barrier; the four row loads of tile t+2 were issued last
      if (more2) { if constexpr (S == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); }
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#if BXA_STAMPS
      c2 = __builtin_amdgcn_s_memtime();
#endif
      split_a(std::integral_constant<int, 1 - cur>{}, false);  // this wavefront's own fragments for tile t+1
#if BXA_STAMPS
      asm volatile("s_nop 0" ::: "memory");
      c3 = __builtin_amdgcn_s_memtime();
#endif
    }
    __syncthreads();
#if BXA_STAMPS
    const uint64_t c4 = __builtin_amdgcn_s_memtime();
    st_acc[0] += (uint32_t)(c1 - c0); st_acc[1] += (uint32_t)(c2 - c1); st_acc[2] += (uint32_t)(c3 - c2); st_acc[3] += (uint32_t)(c4 - c3);
#endif
  };
  if (T > 0) {
    load_a(0, C0{});
    load_b(0, 0);
    if (T > 1) { load_a(1, C1{}); if constexpr (S == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    split_a(C0{}, true);
    __syncthreads();
    BXA_STAMP(1);
#if BXA_STAMPS
    st_loop0 = __builtin_amdgcn_s_memtime();
#endif
    for (int t = 0; t < T; t += 2) {
      step(t, C0{});
      if (t + 1 < T) step(t + 1, C1{});
    }
  }
  BXA_STAMP(2);
#if BXA_STAMPS
  if (threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.x < 4096) {
    g_bxa_stamps[blockIdx.x * 8 + 5] = ((uint64_t)st_acc[1] << 32) | st_acc[0];
    g_bxa_stamps[blockIdx.x * 8 + 6] = ((uint64_t)st_acc[3] << 32) | st_acc[2];
    g_bxa_stamps[blockIdx.x * 8 + 3] = __builtin_amdgcn_s_memtime() - st_loop0;      // (overwritten by stamp 3 unless BXA_STAMPS == 2)
  }
#endif

  float* Cb = p.C;
  const int64_t ldc = p.ldc;
  const float* bias = p.bias;
  const float* rs = p.row_scale;
  const bool n4 = bx_n4(p, Cb, ldc, bias);
  const int32_t* bexp_last = T1 > 0 ? bexp[1] : bexp[0];
  if ((BXA_ABL & 2) && p.K[0] >= 0) return;
  if (BXA_EPI && n4) {
    // Every load of the epilogue BEFORE the first store.  The compiler may not move a load above a store that could alias it,
    // so "load the column's exponents, scale, store" 22 times over was a chain of 22 load latencies per wavefront, each behind
    // the previous store (in-kernel stamps, profiles/r5_areg_stamps.txt: 11 of a workgroup's 25 us at the d ctx' shape).
    // Per-COLUMN values (image-row exponents, bias) go through LDS - the image buffers are free after the last k-tile's
    // barrier; per-ROW sources (accumulate / addend rows / ReLU source) are read one source at a time for both strips.
    int* s_exp = reinterpret_cast<int*>(lds);
    float* s_bias = reinterpret_cast<float*>(lds) + BN;
    for (int c = tid; c < BN; c += NTHR) {
      const bool in = n0 + c < p.N;
      if constexpr (FMT == 1) s_exp[c] = in ? bexp_last[c] : 0;
      s_bias[c] = (bias && in) ? bias[n0 + c] : 0.f;
    }
    __syncthreads();
    int row[S]; bool ok[S];
#pragma unroll
    for (int i = 0; i < S; ++i) { row[i] = m0 + wrow + 16 * i + r; ok[i] = row[i] < Mlim; if (!ok[i]) row[i] = m0; }
    float rscale[S];
#pragma unroll
    for (int i = 0; i < S; ++i) rscale[i] = rs ? rs[(int64_t)row[i] * p.rs_ld] : 1.f;
#pragma unroll
    for (int i = 0; i < S; ++i)
#pragma unroll
      for (int j = 0; j < 11; ++j) {
        const int c = 16 * j + 4 * g;
        if constexpr (FMT == 1) acc[i][j] = hx_scale4(acc[i][j], *reinterpret_cast<const int4*>(s_exp + c), rowE[i] - 2 * HX_TOP);
      }
    auto each_row_source = [&](const float* src, int64_t ld, const int32_t* idx, auto&& apply) {
      const float* rp[S];
#pragma unroll
      for (int i = 0; i < S; ++i) rp[i] = src + (int64_t)(idx ? idx[row[i]] : row[i]) * ld + n0 + 4 * g;
      const int lastc = p.N - n0 - 4 - 4 * g;                  // offset of the last in-range float4 from rp (n4: N is a multiple of 4)
      float4 t[S][11];                                         // (columns past N: the row's last four instead - never stored)
#pragma unroll
      for (int i = 0; i < S; ++i)
#pragma unroll
        for (int j = 0; j < 11; ++j) t[i][j] = *reinterpret_cast<const float4*>(rp[i] + min(16 * j, lastc));
#pragma unroll
      for (int i = 0; i < S; ++i)
#pragma unroll
        for (int j = 0; j < 11; ++j) apply(acc[i][j], t[i][j]);
    };
    if (p.accumulate) each_row_source(Cb, ldc, nullptr, [](f32x4& a, const float4 o) { a[0] += o.x; a[1] += o.y; a[2] += o.z; a[3] += o.w; });
    if (bias) {
#pragma unroll
      for (int i = 0; i < S; ++i)
#pragma unroll
        for (int j = 0; j < 11; ++j) {
          const float4 bv = *reinterpret_cast<const float4*>(s_bias + 16 * j + 4 * g);
          acc[i][j][0] = fmaf(bv.x, rscale[i], acc[i][j][0]); acc[i][j][1] = fmaf(bv.y, rscale[i], acc[i][j][1]);
          acc[i][j][2] = fmaf(bv.z, rscale[i], acc[i][j][2]); acc[i][j][3] = fmaf(bv.w, rscale[i], acc[i][j][3]);
        }
    }
    if (p.add_src) each_row_source(p.add_src, p.add_ld, p.add_idx, [](f32x4& a, const float4 o) { a[0] += o.x; a[1] += o.y; a[2] += o.z; a[3] += o.w; });
    if (p.row_zero) {
#pragma unroll
      for (int i = 0; i < S; ++i)
        if (p.row_zero[row[i]] != 0) {
#pragma unroll
          for (int j = 0; j < 11; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    if (p.relu) {
#pragma unroll
      for (int i = 0; i < S; ++i)
#pragma unroll
        for (int j = 0; j < 11; ++j)
#pragma unroll
          for (int c = 0; c < 4; ++c) acc[i][j][c] = fmaxf(acc[i][j][c], 0.f);
    }
    if (p.relu_src) each_row_source(p.relu_src, p.relu_ld, nullptr, [](f32x4& a, const float4 m) {
      a[0] = m.x > 0.f ? a[0] : 0.f; a[1] = m.y > 0.f ? a[1] : 0.f; a[2] = m.z > 0.f ? a[2] : 0.f; a[3] = m.w > 0.f ? a[3] : 0.f; });
#pragma unroll
    for (int i = 0; i < S; ++i) {
      if (!ok[i]) continue;
      float* cp = Cb + (int64_t)row[i] * ldc + n0 + 4 * g;
#pragma unroll
      for (int j = 0; j < 11; ++j)
        if (n0 + 16 * j + 4 * g < p.N) *reinterpret_cast<f32x4*>(cp + 16 * j) = acc[i][j];
    }
  } else {
#pragma unroll
  for (int i = 0; i < S; ++i) {
    const int row = m0 + wrow + 16 * i + r;
    if (row >= Mlim) continue;
    const float rscale = rs ? rs[(int64_t)row * p.rs_ld] : 1.f;
    const bool zero = p.row_zero ? (p.row_zero[row] != 0) : false;
    const float* addrow = p.add_src ? p.add_src + (int64_t)(p.add_idx ? p.add_idx[row] : row) * p.add_ld : nullptr;
#pragma unroll
    for (int j = 0; j < 11; ++j) {
      const int col = n0 + 16 * j + 4 * g;
      if (col < p.N) {
        f32x4 v = acc[i][j];
        if constexpr (FMT == 1) v = hx_scale4(v, *reinterpret_cast<const int4*>(bexp_last + 16 * j + 4 * g), rowE[i] - 2 * HX_TOP);
        bx_store4(p, Cb, ldc, bias, rscale, zero, n4, row, col, v, addrow);
      }
    }
  }
  }
#if BXA_STAMPS
  if (BXA_STAMPS != 2) BXA_STAMP(3);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  BXA_STAMP(4);
#endif
}
template <int FMT>
__global__ __launch_bounds__(GEMM_THREADS, FMT == 1 ? BX_AREG_OCC : 2) void gemm_bx_areg_kernel(const GemmDev p) {
  __shared__ __attribute__((aligned(16))) char lds[2 * BxFmt<FMT>::NP * BX_B_PIECE];
  bx_areg_body<4, FMT>(p, lds);
}
// The same 128-row workgroup as EIGHT wavefronts of one 16-row strip each: half the registers per wavefront (<= 128), so four
// wavefronts per SIMD instead of two - the row split (vector pipe) of one wavefront runs beside the MFMAs of another and a
// wavefront waiting for its rows leaves three to issue.  In-kernel stamps of the four-wavefront form (profiles/r5_areg_stamps.txt):
// per k-tile 1 370 cycles of MFMAs, 330 waiting for rows, 650-740 splitting rows, 150-180 at the barrier, one after the other.
template <int FMT>
__global__ __launch_bounds__(512, 2) void gemm_bx_areg8_kernel(const GemmDev p) {
  __shared__ __attribute__((aligned(16))) char lds[2 * BxFmt<FMT>::NP * BX_B_PIECE];
  bx_areg_body<8, FMT, 1>(p, lds);
}
// ---------------------------------------------------------------------------------------------
// A-STATIONARY form of the image kernel for SHORT contractions with MANY output columns (the d ctx' contraction: K = 172,
// N = 704): a wavefront loads its 32 rows ONCE, finds each row's scale from the whole row (no running maximum, no rescaling),
// splits them into the fp16 fragments of ALL k-tiles (<= 6 tiles: 96 registers) and keeps them; the image of B then streams
// through an LDS ring, two 16-column tiles per slot, and every pair of column tiles is multiplied, scaled and STORED while the
// next ones arrive.  Against gemm_bx_areg_kernel on this shape (in-kernel stamps, profiles/r5_areg_stamps.txt): the rows are
// read and split once instead of once per 176-column block (4 x), the 3 us of load latency in front of a workgroup's first MFMA
// is paid once per 128 x 704 instead of per 128 x 176 output, and the stores leave evenly over the workgroup's life instead of
// in a burst behind its last k-tile (workgroups of one launch run in lockstep: everyone stored at once, nobody multiplied).
// Plain stores only (no bias / addend / ReLU epilogue), N a multiple of 32, K <= 192, fp16x2 format; a workgroup whose 128 rows
// are not all inside M takes conservative waits (the exact vmcnt bookkeeping below assumes no store is skipped).
#define AS_TMAX 6
#define AS_RING 3
template <bool FULL, int T>
__device__ __forceinline__ void bx_astat_body(const GemmDev& p, char* lds) {
  constexpr int NP = 2;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int m0 = blockIdx.x * 128;
  int Mlim = p.M;
  if (p.m_dev) Mlim = min(Mlim, *p.m_dev);
  const int wrow = 32 * wave;
  const int K = p.K[0];
  const int NS = p.N >> 5;                                         // slots: pairs of column tiles
  const int64_t img_piece = (int64_t)p.b_img_rows * 64;
  const char* img = reinterpret_cast<const char*>(p.b_img);
  const int slot_bytes = 4 * T * 1024;                             // [column tile 0 | 1][k-tile][piece] x (16 image rows x 64 B)
  char* ring = lds;
  const int32_t* bexp = reinterpret_cast<const int32_t*>(img + (int64_t)T * NP * img_piece);   // the image rows' exponents
  typedef __attribute__((address_space(1))) const void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  // chunk c of a slot = (column tile jj, k-tile t, piece q), c = (jj * T + t) * 2 + q: 1 KB, contiguous in the image; wavefront w
  // moves the chunks w, w + 4, ... (T of them)
  auto dma_slot = [&](int s) {
    char* dst = ring + (s % AS_RING) * (4 * AS_TMAX * 1024);
    bx_for<T>([&](auto uc) {                                        // exactly T instructions per wavefront (the vmcnt bookkeeping counts them)
      const int c = wave + 4 * decltype(uc)::value;
      const int q = c & 1, h = c >> 1, jj = h >= T ? 1 : 0, t = h - jj * T;
      const char* src = img + ((int64_t)t * NP + q) * img_piece + (int64_t)(32 * s + 16 * jj) * 64;
      __builtin_amdgcn_global_load_lds((gptr_t)(src + lane * 16), (lptr_t)(dst + c * 1024), 16, 0, 0);
    });
  };
  // ---- the rows: loaded once, scaled by the row's own maximum, split for every k-tile
  u32x4 a[AS_TMAX][2][NP];
  int rowE[2];
  {
    float4 raw[AS_TMAX][2][2];
    const float* a_row[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int gm = m0 + wrow + 16 * i + r;
      int64_t ridx = (FULL || gm < Mlim) ? gm : m0;
      if (p.a_idx[0]) ridx = p.a_idx[0][ridx];
      a_row[i] = p.A[0] + ridx * p.lda[0] + 8 * g;
    }
#pragma unroll
    for (int t = 0; t < AS_TMAX; ++t)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int k = t * BK + 8 * g + 4 * h;
          raw[t][i][h] = (t < T && k < K) ? *reinterpret_cast<const float4*>(a_row[i] + t * BK + 4 * h) : float4{0.f, 0.f, 0.f, 0.f};
        }
    // (behind the row loads in program order: the first MFMA needs both, and the rows come from HBM)
    dma_slot(0);
    if (NS > 1) dma_slot(1);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      float mx = 0.f;
#pragma unroll
      for (int t = 0; t < AS_TMAX; ++t) mx = fmaxf(mx, hx_absmax8(raw[t][i][0], raw[t][i][1]));
      rowE[i] = hx_exp_of_bits(hx_max_over_g(__float_as_uint(mx)));
      const int se = HX_TOP - rowE[i];
#pragma unroll
      for (int t = 0; t < AS_TMAX; ++t) {
        uint2 h0, l0, h1, l1;
        hx_split4(raw[t][i][0], se, h0, l0);
        hx_split4(raw[t][i][1], se, h1, l1);
        a[t][i][0] = u32x4{h0.x, h0.y, h1.x, h1.y};
        a[t][i][1] = u32x4{l0.x, l0.y, l1.x, l1.y};
      }
    }
  }
  const int frag_off = r * 64 + ((g ^ bx_swz(r)) << 4);
  float* crow[2];
  bool ok[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = m0 + wrow + 16 * i + r;
    ok[i] = FULL || row < Mlim;
    crow[i] = p.C + (int64_t)(ok[i] ? row : m0) * p.ldc + 4 * g;
  }
  f32x4 prev[2][2];                                                // the pair of column tiles multiplied one slot ago, stored in this one
  int4 pexp[2];                                                    // ... and its columns' exponents, fetched one slot ahead (2 loads)
  // vmcnt bookkeeping of one wavefront, oldest first, at the top of slot s:
  //   ... DMA(s) [T] | stores(s-2) [4] | exponents(s-1) [2] | DMA(s+1) [T]
  // (iteration s-2 ended with DMA(s); iteration s-1 issued stores(s-2), the exponent loads of slot s-1, DMA(s+1)): DMA(s) has
  // landed when at most 4 + 2 + T younger operations are outstanding.  Members that do not exist (no stores before slot 2, no
  // DMA behind the last slot) are not counted: counting something that was never issued would make the wait too lax.
  auto wait_vm = [&](auto nc) {
    constexpr int n = (FULL && BXA_ABL == 0) ? decltype(nc)::value : 0;
    static_assert(n <= 12, "vmcnt");
    if constexpr (n == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if constexpr (n == 11) asm volatile("s_waitcnt vmcnt(11)" ::: "memory");
    else if constexpr (n == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else if constexpr (n == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
    else if constexpr (n == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (n == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    else if constexpr (n == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (n == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else if constexpr (n == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (n == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if constexpr (n == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if constexpr (n == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
#if BXA_STAMPS
  uint32_t st_acc[4] = {0, 0, 0, 0};
  BXA_STAMP(1);
  const uint64_t st_loop0 = __builtin_amdgcn_s_memtime();
#endif
  // One slot: wait, barrier, then ONE straight-line block in which the 12 memory instructions of the slot (4 stores of the pair
  // multiplied one slot ago, 2 exponent loads, T DMA chunks of slot s+2) are spread between the k-tiles' MFMAs.  Issued in a
  // burst behind the barrier - by all four wavefronts at once - they queued in the CU's one texture-address unit for 1 430
  // cycles per slot while no wavefront multiplied (in-kernel stamps, profiles/r5_areg_stamps.txt).
  auto slot_body = [&](int s, auto hsc, auto hdc, auto nc) {
    constexpr bool HS = decltype(hsc)::value, HD = decltype(hdc)::value;
#if BXA_STAMPS
    const uint64_t c0 = __builtin_amdgcn_s_memtime();
#endif
    wait_vm(nc);
#if BXA_STAMPS
    const uint64_t c1 = __builtin_amdgcn_s_memtime();
#endif
    __syncthreads();                                               // DMA(s) of every wavefront has landed; slot (s-1) % RING is free
#if BXA_STAMPS
    const uint64_t c2 = __builtin_amdgcn_s_memtime();
    const uint64_t c3 = c2;
#endif
    const char* Bs = ring + (s % AS_RING) * (4 * AS_TMAX * 1024);
    char* dst = ring + ((s + 2) % AS_RING) * (4 * AS_TMAX * 1024);
    f32x4 acc[2][2];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[jj][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    u32x4 b0[2][NP], b1[2][NP];
    auto ldb = [&](u32x4 (&b)[2][NP], int t) {
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int q = 0; q < NP; ++q) b[jj][q] = *reinterpret_cast<const u32x4*>(Bs + ((jj * T + t) * 2 + q) * 1024 + frag_off);
    };
    auto mma = [&](const u32x4 (&b)[2][NP], auto tc) {
      constexpr int t = decltype(tc)::value;
      // four independent accumulators, product by product: a dependent MFMA never follows its predecessor directly
      bx_for<3>([&](auto qc) {
        constexpr int Q = decltype(qc)::value;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
          for (int i = 0; i < 2; ++i) acc[jj][i] = bx_mma_q<1, Q>(b[jj], a[t][i], acc[jj][i]);
      });
    };
    // memory instruction m of the slot, in the order the bookkeeping above assumes: 0..3 stores (jj, i) | 4, 5 exponents | 6.. DMA
    int4 nexp[2];
    auto mem_op = [&](auto mc) {
      constexpr int m = decltype(mc)::value;
      if constexpr (m < 4) {
        if constexpr (HS) {
          constexpr int jj = m >> 1, i = m & 1;
          const f32x4 v = hx_scale4(prev[jj][i], pexp[jj], rowE[i] - 2 * HX_TOP);
          if ((FULL || ok[i]) && !((BXA_ABL & 2) && K > 0)) *reinterpret_cast<f32x4*>(crow[i] + 32 * (s - 1) + 16 * jj) = v;
        }
      } else if constexpr (m < 6) {
        nexp[m - 4] = *reinterpret_cast<const int4*>(bexp + 32 * s + 16 * (m - 4) + 4 * g);
      } else if constexpr (m - 6 < T) {
        if constexpr (HD) {
          if (!((BXA_ABL & 8) && K > 0)) {
            const int c = wave + 4 * (m - 6);
            const int q = c & 1, h = c >> 1, jj = h >= T ? 1 : 0, t = h - jj * T;
            const char* src = img + ((int64_t)t * NP + q) * img_piece + (int64_t)(32 * (s + 2) + 16 * jj) * 64;
            __builtin_amdgcn_global_load_lds((gptr_t)(src + lane * 16), (lptr_t)(dst + c * 1024), 16, 0, 0);
          }
        }
      }
    };
    constexpr int NMEM = 6 + T, PER = (NMEM + T - 1) / T;         // memory instructions per k-tile
    ldb(b0, 0);
    bx_for<T>([&](auto tc) {
      constexpr int t = decltype(tc)::value;
      if constexpr ((t & 1) == 0) { if constexpr (t + 1 < T) ldb(b1, t + 1); mma(b0, tc); }
      else                        { if constexpr (t + 1 < T) ldb(b0, t + 1); mma(b1, tc); }
      bx_for<PER>([&](auto kc) { mem_op(std::integral_constant<int, t * PER + decltype(kc)::value>{}); });
    });
    // (the order the scheduler must keep: per k-tile the fragment reads of the next tile, 12 MFMAs, PER memory instructions)
    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
    bx_for<T>([&](auto tc) {
      if constexpr (decltype(tc)::value + 1 < T) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);
      __builtin_amdgcn_sched_group_barrier(0x010, PER, 0);
    });
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      pexp[jj] = nexp[jj];
#pragma unroll
      for (int i = 0; i < 2; ++i) prev[jj][i] = acc[jj][i];
    }
#if BXA_STAMPS
    asm volatile("s_nop 0" :: "v"(prev[0][0]), "v"(prev[1][1]) : "memory");
    const uint64_t c4 = __builtin_amdgcn_s_memtime();
    st_acc[0] += (uint32_t)(c1 - c0); st_acc[1] += (uint32_t)(c2 - c1); st_acc[2] += (uint32_t)(c3 - c2); st_acc[3] += (uint32_t)(c4 - c3);
#endif
  };
  using TT_ = std::true_type; using FF_ = std::false_type;
  // NS >= 4 (launcher: N >= 352).  Slot 0: nothing to store; slot 1: no stores are outstanding yet; the last two: no DMA
  slot_body(0, FF_{}, TT_{}, std::integral_constant<int, T>{});                       // (both first DMAs came from the prologue)
  slot_body(1, TT_{}, TT_{}, std::integral_constant<int, 2 + T>{});
  for (int s = 2; s < NS - 2; ++s) slot_body(s, TT_{}, TT_{}, std::integral_constant<int, 4 + 2 + T>{});
  slot_body(NS - 2, TT_{}, FF_{}, std::integral_constant<int, 4 + 2 + T>{});
  slot_body(NS - 1, TT_{}, FF_{}, std::integral_constant<int, 4 + 2>{});
  {                                                                // the last pair
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const f32x4 v = hx_scale4(prev[jj][i], pexp[jj], rowE[i] - 2 * HX_TOP);
        if (FULL || ok[i]) *reinterpret_cast<f32x4*>(crow[i] + 32 * (NS - 1) + 16 * jj) = v;
      }
  }
#if BXA_STAMPS
  BXA_STAMP(2);
  if (threadIdx.x == 0 && blockIdx.x < 4096) {
    g_bxa_stamps[blockIdx.x * 8 + 5] = ((uint64_t)st_acc[1] << 32) | st_acc[0];
    g_bxa_stamps[blockIdx.x * 8 + 6] = ((uint64_t)st_acc[3] << 32) | st_acc[2];
    g_bxa_stamps[blockIdx.x * 8 + 3] = (uint64_t)NS;
    g_bxa_stamps[blockIdx.x * 8 + 7] = __builtin_amdgcn_s_memtime() - st_loop0;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  BXA_STAMP(4);
#endif
}
#define AS_NMAX 1024
__global__ __launch_bounds__(GEMM_THREADS, 2) void gemm_bx_astat_kernel(const GemmDev p) {
  __shared__ __attribute__((aligned(16))) char lds_dyn[AS_RING * 4 * AS_TMAX * 1024];      // 72 KB: two workgroups per CU
  int Mlim = p.M;
  if (p.m_dev) Mlim = min(Mlim, *p.m_dev);
  const int m0 = blockIdx.x * 128;
  if (m0 >= Mlim) return;
  BXA_STAMP(0);
  const int T = (p.K[0] + BK - 1) / BK;                             // 1 .. AS_TMAX (launcher)
  const bool full = m0 + 128 <= Mlim;
  bx_for<AS_TMAX>([&](auto tc) {
    constexpr int TT = decltype(tc)::value + 1;
    if (T == TT) { if (full) bx_astat_body<true, TT>(p, lds_dyn); else bx_astat_body<false, TT>(p, lds_dyn); }
  });
}
// ---------------------------------------------------------------------------------------------
// The lazy GRU of the touched rows in ONE launch (memory_updater.py:18-61: torch.nn.GRUCell on [message | memory]): both
// contractions - message rows x W_ih^T (K = 3D + Ef) and memory rows x W_hh^T (K = D), K-concatenated - with the gate math in
// the epilogue, so the pre-activations gi / gh (2 x 3D floats per row) never go to HBM.  What makes that possible is the ORDER
// of the image rows (bimg_gate_row): a column block holds, for two groups of 16 hidden units, the four tiles r | z | n_i | n_h.
// In the operand-swapped accumulator layout lane (r, g) owns columns 4g .. 4g+3 of EVERY tile of row r: the r, z, n_i and n_h
// pre-activations of hidden units 16 q + 4g .. + 3 sit in one lane, in four accumulator tiles.  n_i takes only the message
// part and n_h only the memory part (n = tanh(n_i + r * n_h), torch GRUCell): their tiles skip the other source's k-tiles
// instead of multiplying zero rows, so no MFMA is spent that the two separate launches did not spend.
// Same staging as gemm_bx_areg_kernel: A rows straight to registers in fragment order and split there, the B image tile (128
// rows x 3 pieces = 24 KB) DMA'd into a double-buffered LDS image, one barrier per k-tile.
#define GF_NJ 8
#define GF_BN (GF_NJ * 16)
#define GF_B_PIECE (GF_BN * 64)
struct GruFusedDev {
  const float* msg_rows; int64_t ld_msg; int K0;
  const float* h_rows; int64_t ld_h; int K1;
  const void* img0; const void* img1; int img_rows;
  int xcd_tm, xcd_tn;              // as in GemmDev
  const float* b_ih; const float* b_hh;
  const uint8_t* hm; const int32_t* touched; const float* node_feat;
  float* upd_mem; float* h0_tab; float* gates;
  int D, M; const int32_t* m_dev;
  int gather;
};
int pfo_gru_img_rows(int D) { return (int)pfo_ceil_div(D, 32) * GF_BN; }
int64_t pfo_gru_img_bytes(int D, int K) { return (int64_t)pfo_ceil_div(K, 32) * 3 * pfo_gru_img_rows(D) * 64; }

__device__ __forceinline__ float gf_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.44269504088896340736f * x)); }
__device__ __forceinline__ float gf_tanh(float x) {
  const float e = __builtin_amdgcn_exp2f(-2.88539008177792681472f * fabsf(x));
  return copysignf((1.f - e) * __builtin_amdgcn_rcpf(1.f + e), x);
}

template <int FMT>
__global__ __launch_bounds__(GEMM_THREADS, 2) void gru_fused_kernel(const GruFusedDev p) {
  constexpr int NP = BxFmt<FMT>::NP;
  __shared__ __attribute__((aligned(16))) char lds[2 * NP * GF_B_PIECE];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  int tile_m = blockIdx.x, tile_n = blockIdx.y;
  if (p.xcd_tn > 0 && !xcd_tile(blockIdx.x, p.xcd_tm, p.xcd_tn, tile_m, tile_n)) return;
  const int m0 = tile_m * BM, n0 = tile_n * GF_BN;
  const int Mlim = min(p.M, p.m_dev ? *p.m_dev : p.M);
  if (m0 >= Mlim) return;
  const int wrow = 32 * wave;
  f32x4 acc[2][GF_NJ];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < GF_NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int T0 = (p.K0 + BK - 1) / BK, T1 = (p.K1 + BK - 1) / BK, T = T0 + T1;
  const float* safe = p.msg_rows;
  const int64_t img_piece = (int64_t)p.img_rows * 64;
  const float* a_row[2][2];
  bool a_ok[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int gm = m0 + wrow + 16 * i + r;
    a_ok[i] = gm < Mlim;
    int64_t ridx = a_ok[i] ? gm : 0;
    if (p.gather) ridx = a_ok[i] ? p.touched[gm] : 0;          // straight from the per-node tables
    a_row[0][i] = p.msg_rows + ridx * p.ld_msg + 8 * g;
    a_row[1][i] = p.h_rows + ridx * p.ld_h + 8 * g;
  }
  const char* img[2] = {reinterpret_cast<const char*>(p.img0) + (int64_t)n0 * 64, reinterpret_cast<const char*>(p.img1) + (int64_t)n0 * 64};
  // FMT 1: exponents of the image rows of this block, behind the tiles of each image
  const int32_t* bexp[2] = {reinterpret_cast<const int32_t*>(reinterpret_cast<const char*>(p.img0) + (int64_t)T0 * NP * img_piece) + n0,
                            reinterpret_cast<const int32_t*>(reinterpret_cast<const char*>(p.img1) + (int64_t)T1 * NP * img_piece) + n0};
  // the rows of tile t live in register set t & 1 and are fetched TWO tiles ahead (as in bx_areg_body: with one set the loads of
  // tile t+1 had the 36 MFMAs of tile t to arrive in and every k-tile ended in a wait; 67-69 -> 60-65 us at C2)
  float4 a_raw[2][2][2];
  typedef __attribute__((address_space(1))) const void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  auto load_a = [&](int t, auto setc) {
    constexpr int set = decltype(setc)::value;
    const int src = t < T0 ? 0 : 1;
    const int ts = src == 0 ? t : t - T0;
    const int k = ts * BK + 8 * g;
    const int Ks = src == 0 ? p.K0 : p.K1;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int h = 0; h < 2; ++h)
        a_raw[set][i][h] = ld4<true>(a_row[src][i] + ts * BK + 4 * h, a_ok[i] ? Ks - (k + 4 * h) : 0, safe);
  };
  auto load_b = [&](int t, int buf) {
    const int src = t < T0 ? 0 : 1;
    const int ts = src == 0 ? t : t - T0;
    const char* tile = img[src] + (int64_t)ts * NP * img_piece;
    char* Bs = lds + buf * NP * GF_B_PIECE;
    bx_for<2 * NP>([&](auto uc) {                              // NP pieces x 512 units of 16 bytes = 2 NP rounds of 256 threads
      constexpr int u = decltype(uc)::value;
      const int unit = tid + 256 * u;
      const int q = unit >> 9;                                 // 512 units per piece
      __builtin_amdgcn_global_load_lds((gptr_t)(tile + q * img_piece + (unit - (q << 9)) * 16),
                                       (lptr_t)(Bs + (64 * wave + 256 * u) * 16), 16, 0, 0);
    });
  };
  u32x4 a[2][NP];
  int rowE[2] = {HX_EMIN, HX_EMIN};
  auto split_a = [&](auto setc, bool first) { bx_split_rows<FMT, GF_NJ>(a_raw[decltype(setc)::value], a, rowE, acc, first); };
  const int frag_off = r * 64 + ((g ^ bx_swz(r)) << 4);
  // tiles that take source SRC: the message part feeds r, z, n_i (tiles 0 1 2 | 4 5 6), the memory part r, z, n_h (0 1 3 | 4 5 7)
  auto compute_tile = [&](int buf, auto src_c) {
    constexpr int SRC = decltype(src_c)::value;
    constexpr int act[6] = {0, 1, SRC == 0 ? 2 : 3, 4, 5, SRC == 0 ? 6 : 7};
    const char* Bs = lds + buf * NP * GF_B_PIECE;
    auto ldb = [&](u32x4 (&b)[NP], int j) {
#pragma unroll
      for (int q = 0; q < NP; ++q) b[q] = *reinterpret_cast<const u32x4*>(Bs + q * GF_B_PIECE + (16 * j) * 64 + frag_off);
    };
    auto mma = [&](const u32x4 (&b)[NP], int j) {
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[i][j] = bx_mma<FMT>(b, a[i], acc[i][j]);
    };
    u32x4 b0[NP], b1[NP];
    ldb(b0, act[0]);
#pragma unroll
    for (int q = 0; q < 6; q += 2) {
      ldb(b1, act[q + 1]);
      mma(b0, act[q]);
      if (q + 2 < 6) ldb(b0, act[q + 2]);
      mma(b1, act[q + 1]);
    }
  };
  using C0 = std::integral_constant<int, 0>;
  using C1 = std::integral_constant<int, 1>;
  load_a(0, C0{});
  load_b(0, 0);
  if (T > 1) { load_a(1, C1{}); asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  split_a(C0{}, true);
  __syncthreads();
  auto step = [&](int t, auto curc) {
    constexpr int cur = decltype(curc)::value;                 // t & 1
    const bool more = t + 1 < T, more2 = t + 2 < T;
    if (more) load_b(t + 1, (t + 1) & 1);
    if (more2) load_a(t + 2, curc);                            // (the rows of tile t were split before this call)
    if constexpr (FMT == 1) {
      if (t == T0) {                                // the memory part: the r and z tiles move from W_ih's row scales to W_hh's
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int tt = 0; tt < 2; ++tt) {
            const int j = 4 * q + tt;
            const int4 e1 = *reinterpret_cast<const int4*>(bexp[0] + 16 * j + 4 * g);
            const int4 e2 = *reinterpret_cast<const int4*>(bexp[1] + 16 * j + 4 * g);
            const int4 d = {e1.x - e2.x, e1.y - e2.y, e1.z - e2.z, e1.w - e2.w};
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[i][j] = hx_scale4(acc[i][j], d, 0);
          }
      }
    }
    if (t < T0) compute_tile(t & 1, std::integral_constant<int, 0>{}); else compute_tile(t & 1, std::integral_constant<int, 1>{});
    if (more) {
      // (the DMA is ordered only by the issuing wavefront's vmcnt + the barrier; the four row loads of tile t+2 were issued last)
      if (more2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      split_a(std::integral_constant<int, 1 - cur>{}, false);
    }
    __syncthreads();
  };
  for (int t = 0; t < T; t += 2) {
    step(t, C0{});
    if (t + 1 < T) step(t + 1, C1{});
  }
  // gates.  lane (r, g), strip i: row m0 + wrow + 16 i + r; group q: hidden units u0 .. u0 + 3, u0 = 32 by + 16 q + 4 g
  // Two passes: every load and the gate math of all four (strip, group) pairs first, the 24 stores after them - a load may not
  // pass a store that could alias it, so "load, compute, store" four times over was four load latencies one after the other.
  const int D = p.D;
  float hn_[2][2][4], h0_[2][2][4];
  bool live[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = m0 + wrow + 16 * i + r;
    const bool rok = row < Mlim;
    const int64_t id = p.touched[rok ? row : m0];
    const int64_t srow = p.gather ? id : (rok ? row : m0);     // row of the state operands (tables or packed copies)
    const bool has = p.hm[srow] != 0;
    if constexpr (FMT == 1) {                         // back to plain fp32: r, z and n_h carry W_hh's row scales, n_i W_ih's
#pragma unroll
      for (int j = 0; j < GF_NJ; ++j)
        acc[i][j] = hx_scale4(acc[i][j], *reinterpret_cast<const int4*>(bexp[(j & 3) == 2 ? 0 : 1] + 16 * j + 4 * g), rowE[i] - 2 * HX_TOP);
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int u0 = 32 * tile_n + 16 * q + 4 * g;
      live[i][q] = rok && u0 < D;                               // D % 4 == 0: a lane's four units are all inside or all outside
      const int uc = u0 < D ? u0 : 0;
      const float4 h4 = *reinterpret_cast<const float4*>(p.h_rows + srow * p.ld_h + uc);
      const float4 nf = *reinterpret_cast<const float4*>(p.node_feat + id * D + uc);
      const float hv[4] = {h4.x, h4.y, h4.z, h4.w}, nfv[4] = {nf.x, nf.y, nf.z, nf.w};
      float b_r[4], b_z[4], b_n[4], b_g[4];                     // (the bias vectors sit anywhere in the parameter buffer: scalar loads)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        b_r[e] = p.b_ih[uc + e] + p.b_hh[uc + e];
        b_z[e] = p.b_ih[D + uc + e] + p.b_hh[D + uc + e];
        b_n[e] = p.b_ih[2 * D + uc + e];
        b_g[e] = p.b_hh[2 * D + uc + e];
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float pr = acc[i][4 * q + 0][e] + b_r[e];
        const float pz = acc[i][4 * q + 1][e] + b_z[e];
        const float pn = acc[i][4 * q + 2][e] + b_n[e];
        const float gh = acc[i][4 * q + 3][e] + b_g[e];
        const float rr = gf_sigmoid(pr);
        const float zz = gf_sigmoid(pz);
        const float nn = gf_tanh(pn + rr * gh);
        const float hn = has ? (1.f - zz) * nn + zz * hv[e] : hv[e];      // no pending message: the memory row is kept
        acc[i][4 * q + 0][e] = rr; acc[i][4 * q + 1][e] = zz; acc[i][4 * q + 2][e] = nn; acc[i][4 * q + 3][e] = gh;
        hn_[i][q][e] = hn; h0_[i][q][e] = hn + nfv[e];
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = m0 + wrow + 16 * i + r;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      if (!live[i][q]) continue;
      const int u0 = 32 * tile_n + 16 * q + 4 * g;
      *reinterpret_cast<float4*>(p.upd_mem + (int64_t)row * D + u0) = float4{hn_[i][q][0], hn_[i][q][1], hn_[i][q][2], hn_[i][q][3]};
      *reinterpret_cast<float4*>(p.h0_tab + (int64_t)row * D + u0) = float4{h0_[i][q][0], h0_[i][q][1], h0_[i][q][2], h0_[i][q][3]};
      float* gs = p.gates + (int64_t)row * 4 * D + u0;           // kept for the backward: r | z | n | gh_n
      *reinterpret_cast<f32x4*>(gs) = acc[i][4 * q + 0];
      *reinterpret_cast<f32x4*>(gs + D) = acc[i][4 * q + 1];
      *reinterpret_cast<f32x4*>(gs + 2 * D) = acc[i][4 * q + 2];
      *reinterpret_cast<f32x4*>(gs + 3 * D) = acc[i][4 * q + 3];
    }
  }
}

int pfo_gru_fused_launch(const PfoGruFused& f, hipStream_t stream) {
  PFO_REQUIRE(f.msg_rows && f.h_rows && f.img_ih && f.img_hh && f.b_ih && f.b_hh && f.hm && f.touched && f.node_feat && f.upd_mem &&
              f.h0_tab && f.gates, "null argument");
  PFO_REQUIRE(f.D > 0 && (f.D % 4) == 0 && f.K_msg > 0 && (f.K_msg % 4) == 0 && f.cap_rows > 0, "bad sizes");
  PFO_REQUIRE(aligned4(f.msg_rows) && aligned4(f.h_rows) && aligned4(f.node_feat) && aligned4(f.upd_mem) && aligned4(f.h0_tab) &&
              aligned4(f.gates), "operands must be 16-byte aligned");
  GruFusedDev d;
  d.msg_rows = f.msg_rows; d.ld_msg = f.K_msg; d.K0 = f.K_msg; d.h_rows = f.h_rows; d.ld_h = f.D; d.K1 = f.D;
  d.img0 = f.img_ih; d.img1 = f.img_hh; d.img_rows = pfo_gru_img_rows(f.D);
  d.b_ih = f.b_ih; d.b_hh = f.b_hh; d.hm = f.hm; d.touched = f.touched; d.node_feat = f.node_feat;
  d.upd_mem = f.upd_mem; d.h0_tab = f.h0_tab; d.gates = f.gates; d.D = f.D; d.M = f.cap_rows; d.m_dev = f.n_rows;
  d.gather = f.gather;
  pfo_prof_begin(stream);
  static const int xcd = getenv("PFO_GEMM_XCD") ? atoi(getenv("PFO_GEMM_XCD")) : 1;                        // A/B switch
  const int tmr = (int)pfo_ceil_div(f.cap_rows, BM), tnc = (int)pfo_ceil_div(f.D, 32);
  dim3 grid((unsigned)tmr, (unsigned)tnc, 1);
  d.xcd_tm = d.xcd_tn = 0;
  if (xcd && tnc > 1) { d.xcd_tm = tmr; d.xcd_tn = tnc; grid = dim3((unsigned)(pfo_ceil_div(tmr, 8) * 8 * tnc), 1, 1); }
  if (pfo_bx_fmt()) PFO_KLAUNCH(gru_fused_kernel<1>, grid, dim3(GEMM_THREADS), 0, stream, d);
  else PFO_KLAUNCH(gru_fused_kernel<0>, grid, dim3(GEMM_THREADS), 0, stream, d);
  PFO_LAUNCH_CHECK();
  pfo_prof_end_dev(PFO_PROF_GRU_FUSED, 2.0 * 3 * f.D * ((double)f.K_msg + f.D), f.n_rows, f.cap_rows, stream);   // per-row FLOPs of the two contractions
  return PFO_OK;
}

// ---------------------------------------------------------------------------------------------
// The same contraction for SHORT operands (the layer-2 launches: a few thousand rows, where 128-row tiles leave most
// CUs idle and every workgroup is latency-bound): 32 rows x 176 columns per workgroup, wavefront w owns the column
// tiles w, w+4, w+8 of all 32 rows; two LDS buffers, ONE barrier per k-tile, and two register stages so that the
// global loads of k-tiles t+2 and t+3 are in flight while tile t is multiplied.  Pre-split B image only.
// NT = column tiles (of 16) per workgroup: 11 (the whole 176-column block of the image) or 4 - launches with a few dozen
// row tiles (layer 2 at C2: 80) put three times the workgroups on the chip, each staging a third of the image per k-tile.
#define SK_ROWS 32
#define SK_A_PIECE (SK_ROWS * 64)
template <int NT, int FMT>
__global__ __launch_bounds__(GEMM_THREADS, 2) void gemm_bx_skinny_kernel(const GemmDev p) {
  constexpr int NP = BxFmt<FMT>::NP;
  constexpr int SK_B_PIECE = NT * 16 * 64;                     // bytes of one piece of the staged B slice
  constexpr int SK_BUF_BYTES = NP * SK_A_PIECE + NP * SK_B_PIECE + SK_ROWS * 4;   // + the row exponents of the A tile (FMT 1)
  constexpr int NU = (NP * (SK_B_PIECE / 16) + 255) / 256;     // 16-byte units of the B slice per thread
  constexpr int NJW = (NT + 3) / 4;                            // column tiles per wavefront
  __shared__ __attribute__((aligned(16))) char lds[2 * SK_BUF_BYTES];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int m0 = blockIdx.x * SK_ROWS, n0 = blockIdx.y * (NT * 16);
  int Mlim = p.M;
  if (p.m_dev) Mlim = min(Mlim, *p.m_dev);
  if (m0 >= Mlim) return;
  const int T0 = (p.K[0] + BK - 1) / BK;
  const int T1 = (p.K[1] > 0 && p.A[1]) ? (p.K[1] + BK - 1) / BK : 0;     // optional second K-concatenated source
  const int T = T0 + T1;
  f32x4 acc[2][NJW];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int jj = 0; jj < NJW; ++jj) acc[i][jj] = f32x4{0.f, 0.f, 0.f, 0.f};

  // staging: one float4 of A (row tid >> 3, k quad tid & 7) and NU 16-byte units of the B image per thread and stage
  float4 a_st[2];
  f32x4 i_st[2][NU];
  const int a_r = tid >> 3, a_c4 = tid & 7;
  const bool a_ok = m0 + a_r < Mlim;
  int64_t ridx0 = a_ok ? m0 + a_r : 0, ridx1 = ridx0;
  if (a_ok && p.a_idx[0]) ridx0 = p.a_idx[0][m0 + a_r];
  if (a_ok && T1 > 0 && p.a_idx[1]) ridx1 = p.a_idx[1][m0 + a_r];
  const float* a_row0 = p.A[0] + ridx0 * p.lda[0];
  const float* a_row1 = T1 > 0 ? p.A[1] + ridx1 * p.lda[1] : a_row0;
  const float* safe = p.A[0];
  const char* img0 = reinterpret_cast<const char*>(p.b_img) + (int64_t)n0 * 64;
  const char* img1 = T1 > 0 ? reinterpret_cast<const char*>(p.b_img2) + (int64_t)n0 * 64 : img0;
  const int64_t img_piece = (int64_t)p.b_img_rows * 64;
  const int img_units_left = min(SK_B_PIECE / 16, (p.b_img_rows - n0) * 4);      // 16-byte units of this slice inside the image
  // FMT 1: exponents of the image rows behind the tiles of each image; the A rows' running exponents travel with the tile
  const int32_t* bexp0 = reinterpret_cast<const int32_t*>(reinterpret_cast<const char*>(p.b_img) + (int64_t)T0 * NP * img_piece);
  const int32_t* bexp1 = T1 > 0 ? reinterpret_cast<const int32_t*>(reinterpret_cast<const char*>(p.b_img2) + (int64_t)T1 * NP * img_piece) : bexp0;
  int stE = HX_EMIN;                   // staging side: running exponent of row a_r (the same in its 8 staging lanes)
  int accE[2] = {HX_EMIN, HX_EMIN};    // accumulator side: the exponent the sums of this lane's two rows are scaled with
  auto load_global = [&](int t, auto sc) {
    constexpr int st = decltype(sc)::value;
    const bool second = t >= T0;
    const int ts = second ? t - T0 : t;
    const int k = ts * BK + 4 * a_c4;
    a_st[st] = ld4<true>((second ? a_row1 : a_row0) + k, a_ok ? (second ? p.K[1] : p.K[0]) - k : 0, safe);
    const char* tile = (second ? img1 : img0) + (int64_t)ts * NP * img_piece;
    bx_for<NU>([&](auto uc) {
      constexpr int u = decltype(uc)::value;
      const int unit = min(tid + 256 * u, NP * (SK_B_PIECE / 16) - 1);
      const int q = unit / (SK_B_PIECE / 16);
      // the last 64-column slice of a 176-row image block is 48 rows: rows beyond the image are clamped (their columns are never stored)
      const int in_piece = min(unit - q * (SK_B_PIECE / 16), img_units_left - 1);
      i_st[st][u] = *reinterpret_cast<const f32x4*>(tile + q * img_piece + in_piece * 16);
    });
  };
  auto store_lds = [&](int buf, auto sc) {
    constexpr int st = decltype(sc)::value;
    char* As = lds + buf * SK_BUF_BYTES;
    char* Bs = As + NP * SK_A_PIECE;
    if constexpr (FMT == 0) bx_split_store(As, SK_A_PIECE, a_r, a_c4, a_st[st]);
    else {
      // the row's maximum over this k-tile sits in 8 neighbouring lanes (4 values each): quad swaps + half-row mirror
      const float4 v = a_st[st];
      uint32_t m = __float_as_uint(fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
      m = max(m, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)m, 0xB1, 0xF, 0xF, false));     // quad_perm [1,0,3,2]
      m = max(m, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)m, 0x4E, 0xF, 0xF, false));     // quad_perm [2,3,0,1]
      m = max(m, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)m, 0x141, 0xF, 0xF, false));    // row_half_mirror
      { const int e = hx_exp_of_bits(m); if (e > stE) stE = min(e + HX_GROW, HX_EMAX); }      // (headroom: see bx_split_rows)
      const int off = a_r * 64 + (((a_c4 >> 1) ^ bx_swz(a_r)) << 4) + ((a_c4 & 1) << 3);
      uint2 oh, ol;
      hx_split4(v, HX_TOP - stE, oh, ol);
      *reinterpret_cast<uint2*>(As + off) = oh;
      *reinterpret_cast<uint2*>(As + SK_A_PIECE + off) = ol;
      if (a_c4 == 0) reinterpret_cast<int32_t*>(As + NP * SK_A_PIECE + NP * SK_B_PIECE)[a_r] = stE;
    }
    bx_for<NU>([&](auto uc) {
      constexpr int u = decltype(uc)::value;
      const int unit = tid + 256 * u;
      if (256 * (u + 1) <= NP * (SK_B_PIECE / 16) || unit < NP * (SK_B_PIECE / 16)) *reinterpret_cast<f32x4*>(Bs + unit * 16) = i_st[st][u];
    });
  };
  const int frag_off = r * 64 + ((g ^ bx_swz(r)) << 4);
  auto compute = [&](int buf, int t) {
    const char* As = lds + buf * SK_BUF_BYTES;
    const char* Bs = As + NP * SK_A_PIECE;
    if constexpr (FMT == 1) {
      const int32_t* Es = reinterpret_cast<const int32_t*>(Bs + NP * SK_B_PIECE);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int e = Es[16 * i + r];
        if (e > accE[i]) {                                     // the row's scale moved with this tile: the sums follow
#pragma unroll
          for (int jj = 0; jj < NJW; ++jj)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[i][jj][c] = __builtin_amdgcn_ldexpf(acc[i][jj][c], accE[i] - e);
          accE[i] = e;
        }
      }
      if (T1 > 0 && t == T0) {                                 // second source: from the first image's row scales to the second's
#pragma unroll
        for (int jj = 0; jj < NJW; ++jj) {
          const int col = min(n0 + 16 * min(wave + 4 * jj, NT - 1) + 4 * g, p.b_img_rows - 4);
          const int4 e1 = *reinterpret_cast<const int4*>(bexp0 + col), e2 = *reinterpret_cast<const int4*>(bexp1 + col);
          const int4 d = {e1.x - e2.x, e1.y - e2.y, e1.z - e2.z, e1.w - e2.w};
#pragma unroll
          for (int i = 0; i < 2; ++i) acc[i][jj] = hx_scale4(acc[i][jj], d, 0);
        }
      }
    }
    u32x4 a[2][NP], b[NJW][NP];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int q = 0; q < NP; ++q) a[i][q] = *reinterpret_cast<const u32x4*>(As + q * SK_A_PIECE + (16 * i) * 64 + frag_off);
#pragma unroll
    for (int jj = 0; jj < NJW; ++jj)
#pragma unroll
      for (int q = 0; q < NP; ++q) {
        const int j = min(wave + 4 * jj, NT - 1);             // NT = 11: wave 3 has no third tile, re-reads tile 10, result unused
        b[jj][q] = *reinterpret_cast<const u32x4*>(Bs + q * SK_B_PIECE + (16 * j) * 64 + frag_off);
      }
    if constexpr (FMT == 0) {
      typedef __bf16 v8 __attribute__((ext_vector_type(8)));
      constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};     // smallest terms first
#pragma unroll
      for (int t6 = 0; t6 < 6; ++t6)
#pragma unroll
        for (int jj = 0; jj < NJW; ++jj)
#pragma unroll
          for (int i = 0; i < 2; ++i)
            acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8, b[jj][PB[t6]]), __builtin_bit_cast(v8, a[i][PA[t6]]), acc[i][jj], 0, 0, 0);
    } else {
      constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};
#pragma unroll
      for (int t3 = 0; t3 < 3; ++t3)
#pragma unroll
        for (int jj = 0; jj < NJW; ++jj)
#pragma unroll
          for (int i = 0; i < 2; ++i)
            acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, b[jj][PB[t3]]), __builtin_bit_cast(f16x8, a[i][PA[t3]]), acc[i][jj], 0, 0, 0);
    }
  };
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  load_global(0, S0{});
  if (T > 1) load_global(1, S1{});
  store_lds(0, S0{});
  if (T > 2) load_global(2, S0{});
  __syncthreads();
  for (int t = 0; t < T; t += 2) {
    if (t + 1 < T) {                       // tile t in buffer 0; stage 1 holds t+1, stage 0 has t+2 in flight
      store_lds(1, S1{});
      if (t + 3 < T) load_global(t + 3, S1{});
    }
    compute(0, t);
    __syncthreads();
    if (t + 1 >= T) break;
    if (t + 2 < T) {                       // tile t+1 in buffer 1; stage 0 holds t+2, stage 1 has t+3 in flight
      store_lds(0, S0{});
      if (t + 4 < T) load_global(t + 4, S0{});
    }
    compute(1, t + 1);
    __syncthreads();
  }
  float* Cb = p.C;
  const int64_t ldc = p.ldc;
  const bool n4 = bx_n4(p, Cb, ldc, p.bias);
  if (BXA_EPI && n4 && !p.gg_gates) {
    // every load of the epilogue in front of its first store (see bx_areg_body: a load may not pass a store that could alias
    // it, so the per-tile "exponents, addends, store" was a chain of load latencies)
    int colv[NJW]; bool okc[NJW];
    int4 ex[NJW]; float4 bs[NJW];
#pragma unroll
    for (int jj = 0; jj < NJW; ++jj) {
      const int j = wave + 4 * jj;
      const int col = n0 + 16 * j + 4 * g;
      okc[jj] = j < NT && col < p.N;
      colv[jj] = okc[jj] ? col : n0;                           // (n0 < N: a valid quad to read instead)
      ex[jj] = int4{0, 0, 0, 0}; bs[jj] = float4{0.f, 0.f, 0.f, 0.f};
      if constexpr (FMT == 1) ex[jj] = *reinterpret_cast<const int4*>(bexp1 + colv[jj]);
      if (p.bias) bs[jj] = *reinterpret_cast<const float4*>(p.bias + colv[jj]);
    }
    int rowv[2]; bool okr[2]; float rscale[2]; bool zero[2];
    float4 tc[2][NJW], ta[2][NJW], tr[2][NJW];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = m0 + 16 * i + r;
      okr[i] = row < Mlim;
      rowv[i] = okr[i] ? row : m0;
      rscale[i] = p.row_scale ? p.row_scale[(int64_t)rowv[i] * p.rs_ld] : 1.f;
      zero[i] = p.row_zero ? (p.row_zero[rowv[i]] != 0) : false;
      const float* addrow = p.add_src ? p.add_src + (int64_t)(p.add_idx ? p.add_idx[rowv[i]] : rowv[i]) * p.add_ld : nullptr;
#pragma unroll
      for (int jj = 0; jj < NJW; ++jj) {
        if (p.accumulate) tc[i][jj] = *reinterpret_cast<const float4*>(Cb + (int64_t)rowv[i] * ldc + colv[jj]);
        if (addrow) ta[i][jj] = *reinterpret_cast<const float4*>(addrow + colv[jj]);
        if (p.relu_src) tr[i][jj] = *reinterpret_cast<const float4*>(p.relu_src + (int64_t)rowv[i] * p.relu_ld + colv[jj]);
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int jj = 0; jj < NJW; ++jj) {
        f32x4 v = acc[i][jj];
        if constexpr (FMT == 1) v = hx_scale4(v, ex[jj], accE[i] - 2 * HX_TOP);
        if (p.accumulate) { v[0] += tc[i][jj].x; v[1] += tc[i][jj].y; v[2] += tc[i][jj].z; v[3] += tc[i][jj].w; }
        if (p.bias) { v[0] = fmaf(bs[jj].x, rscale[i], v[0]); v[1] = fmaf(bs[jj].y, rscale[i], v[1]); v[2] = fmaf(bs[jj].z, rscale[i], v[2]); v[3] = fmaf(bs[jj].w, rscale[i], v[3]); }
        if (p.add_src) { v[0] += ta[i][jj].x; v[1] += ta[i][jj].y; v[2] += ta[i][jj].z; v[3] += ta[i][jj].w; }
        if (zero[i]) v = f32x4{0.f, 0.f, 0.f, 0.f};
        if (p.relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
        if (p.relu_src) {
          v[0] = tr[i][jj].x > 0.f ? v[0] : 0.f; v[1] = tr[i][jj].y > 0.f ? v[1] : 0.f;
          v[2] = tr[i][jj].z > 0.f ? v[2] : 0.f; v[3] = tr[i][jj].w > 0.f ? v[3] : 0.f;
        }
        acc[i][jj] = v;
      }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int jj = 0; jj < NJW; ++jj)
        if (okr[i] && okc[jj]) *reinterpret_cast<f32x4*>(Cb + (int64_t)rowv[i] * ldc + colv[jj]) = acc[i][jj];
    return;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = m0 + 16 * i + r;
    if (row >= Mlim) continue;
    const float rscale = p.row_scale ? p.row_scale[(int64_t)row * p.rs_ld] : 1.f;
    const bool zero = p.row_zero ? (p.row_zero[row] != 0) : false;
    const float* addrow = p.add_src ? p.add_src + (int64_t)(p.add_idx ? p.add_idx[row] : row) * p.add_ld : nullptr;
#pragma unroll
    for (int jj = 0; jj < NJW; ++jj) {
      const int j = wave + 4 * jj;
      const int col = n0 + 16 * j + 4 * g;
      if (j < NT && col < p.N) {
        f32x4 v = acc[i][jj];
        if constexpr (FMT == 1) v = hx_scale4(v, *reinterpret_cast<const int4*>(bexp1 + col), accE[i] - 2 * HX_TOP);
        if (p.gg_gates) bx_gru_gates4(p, row, col, v);           // (wave-uniform: a kernel argument)
        else bx_store4(p, Cb, ldc, p.bias, rscale, zero, n4, row, col, v, addrow);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// bf16x3 weight gradient tile: C[m][n] = sum_k A[k][m] B[k][n], both operands k-major (k = instance row), so the
// MFMA fragments (8 consecutive k per lane) are TRANSPOSED reads of the staged tile: ds_read_b64_tr_b16 delivers a
// 4(k) x 16(m) block column-major, two of them make one 16x16x32 operand.  LDS image per piece: [k 0..31][row of
// 32-byte slots, one slot = 16 columns], A rows 256 B (8 slots), B rows 512 B (11 slots used); slot j of row k sits
// at slot (j + pi(k)) mod {8,16}, pi(k) = (k & 3) + 4 * ((k >> 3) & 1): the 8 rows a 32-lane half reads in one
// instruction ({0..3, 8..11} + 4h + 16 * half) land on 8 different 32-byte bank groups - conflict-free.
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
#define TX_A_ROW 256
#define TX_B_ROW 512
#define TX_A_PIECE (BK * TX_A_ROW)
#define TX_B_PIECE (BK * TX_B_ROW)
#define TX_LDS_BYTES (3 * TX_A_PIECE + 3 * TX_B_PIECE)

__device__ __forceinline__ int tx_pi(int k) { return (k & 3) + ((k >> 1) & 4); }

// float4 = 4 consecutive columns (m or n) of k-row `k`: 8 bytes of bf16 per piece
__device__ __forceinline__ void tx_split_store(char* base, int piece_bytes, int off, const float4 v) {
  uint2 o1, o2, o3;
  bx_split4(v, o1, o2, o3);
  *reinterpret_cast<uint2*>(base + off) = o1;
  *reinterpret_cast<uint2*>(base + piece_bytes + off) = o2;
  *reinterpret_cast<uint2*>(base + 2 * piece_bytes + off) = o3;
}
// byte offset of columns col..col+3 of k-row `k` in a transposed-read image
__device__ __forceinline__ int tx_off(int row_bytes, int slot_mask, int k, int col) {
  return k * row_bytes + ((((col >> 4) + tx_pi(k)) & slot_mask) << 5) + ((col & 15) << 1);
}

__device__ __forceinline__ bf16x8 tx_read(const char* piece, int row_bytes, int slot_mask, int slot, int g, int idx) {
  // lane (g, idx): block rows 8g + 4h + (idx >> 2), 8 bytes at column quad idx & 3
  typedef __attribute__((address_space(3))) bf16x4* lds_p;
  const int k0 = 8 * g + (idx >> 2), k1 = k0 + 4;
  const char* a0 = piece + k0 * row_bytes + (((slot + tx_pi(k0)) & slot_mask) << 5) + ((idx & 3) << 3);
  const char* a1 = piece + k1 * row_bytes + (((slot + tx_pi(k1)) & slot_mask) << 5) + ((idx & 3) << 3);
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p)(a0));
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p)(a1));
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// one 128 x 176 tile of a split-K weight gradient into its slab (the grouped launch below); VEC operands only
// FMT 1 (two fp16 pieces, three products): the contraction runs over the instance rows, so a per-row scale cannot be factored
// out; each OPERAND of the workgroup's K-slab takes ONE power-of-two scale instead - a running maximum over the k-tiles seen so
// far, with HX_GROW binades of headroom when it moves.  The maxima of the tile being staged are found while it waits in
// registers (per-wavefront maxima through eight LDS words, in front of the barrier that already separates the MFMAs of tile t
// from the staging of tile t+1); every thread derives the same two exponents from them, so the state lives in registers, the
// conversion takes a wave-uniform exponent, and when a scale moves the accumulators are multiplied by one exact power of two.
// Elements below 2^-16 of the slab's largest magnitude keep an absolute error of 2^-38 of it: norm-wise bound, as for the
// image kernels, now per slab and operand rather than per row.
__device__ __forceinline__ uint32_t tx_wave_max_u32(uint32_t u) {
  const auto a = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  u = max(a[0], a[1]);
  const auto b = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  u = max(b[0], b[1]);
  u = max(u, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)u, 0xB1, 0xF, 0xF, false));
  u = max(u, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)u, 0x4E, 0xF, 0xF, false));
  u = max(u, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)u, 0x124, 0xF, 0xF, false));
  return max(u, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)u, 0x128, 0xF, 0xF, false));
}
// W wavefronts per workgroup, 32 rows of the tile each: 4 (128 x 176, two workgroups per CU) or 8 (256 x 176, one per CU): the
// second form stages the B tile once per 256 rows instead of once per 128 - the launch's L2-level traffic (A once, B once per row
// tile) is what its time follows (profiles/r5_tn_stamps.txt).
template <int FMT, int W = 4>
__device__ __forceinline__ void gemm_tile_tn_bx(const GemmDev& p, int bx, int by, int split, char* lds) {
  constexpr int NP = BxFmt<FMT>::NP;
  constexpr int NT = 64 * W, TBM = 32 * W;                     // threads, rows of the tile
  constexpr int TXA_ROW = 2 * TBM, TXA_PIECE = BK * TXA_ROW, TXA_MASK = TBM / 16 - 1;
  constexpr int NA = BK * (TBM / 4) / NT, NB = (BK * 44 + NT - 1) / NT;      // float4 of the A / B tile per thread
  char* const As = lds;
  char* const Bs = lds + NP * TXA_PIECE;
  uint32_t* const wmax = reinterpret_cast<uint32_t*>(lds + NP * TXA_PIECE + NP * TX_B_PIECE);     // [W wavefronts: A | W: B]
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int m0 = bx * TBM, n0 = by * BN;
  int Kext = p.K[0];
  if (p.m_dev) Kext = min(Kext, *p.m_dev);
  const int chunk = p.dyn_chunk ? ((Kext + p.nsplit - 1) / p.nsplit + BK - 1) / BK * BK : p.split_chunk;
  const int kbeg = split * chunk;
  const int Ks = min(Kext, kbeg + chunk);
  const int wrow = 32 * wave;

  f32x4 acc[2][11];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 11; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int T = Ks > kbeg ? (Ks - kbeg + BK - 1) / BK : 0;
  const float* safe = p.A[0];
  const int64_t lda = p.lda[0], ldb = p.ldb[0];
  // per-thread invariants of the staging pass: which (k-row, column quad) of the tile each of the 4 + 6 float4 covers,
  // where it lands in LDS, and whether it holds the bias column
  float4 a_reg[NA], b_reg[NB];
  const float* a_ptr[NA];         // row kbeg + kr of A at this thread's columns; advanced by BK rows per tile
  int a_kr[NA], a_off[NA];
  bool a_col_ok[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int f = tid + NT * i;
    a_kr[i] = f / (TBM / 4);
    const int mc = 4 * (f % (TBM / 4));
    a_col_ok[i] = m0 + mc < p.M;                        // M % 4 == 0 (VEC): a quad is all in or all out
    a_ptr[i] = p.A[0] + (int64_t)(kbeg + a_kr[i]) * lda + m0 + mc;
    a_off[i] = tx_off(TXA_ROW, TXA_MASK, a_kr[i], mc);
  }
  int b_kr[NB], b_off[NB], b_n[NB], b_bias_e[NB];
  bool b_live[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int f = tid + NT * i;
    b_kr[i] = f / 44;
    const int nc = 4 * (f - b_kr[i] * 44);
    b_live[i] = f < BK * 44;
    b_n[i] = n0 + nc;
    b_off[i] = tx_off(TX_B_ROW, 15, b_live[i] ? b_kr[i] : 0, nc);
    const int e = p.n_real - b_n[i];                     // bias column inside this quad?
    b_bias_e[i] = (p.n_real < p.N && e >= 0 && e < 4) ? e : -1;
  }

  // Branch-free staging loads: every lane loads from a clamped, always-valid address and the out-of-range values are
  // selected to zero afterwards (per-lane conditions around loads compile to exec-mask branch sequences: the first
  // version of this loop spent as many scalar as vector instructions).
  const int k_last = max(Ks - 1, 0);
  // gathered B rows: the row indices of tile t+1 are fetched while tile t is staged, so the row loads of a tile never
  // wait for an index load issued just before them
  int b_row_next[NB];
  auto fetch_rows = [&](int t) {
    const int k0 = kbeg + t * BK;
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int kc = min(k0 + b_kr[i], k_last);
      b_row_next[i] = p.b_idx ? p.b_idx[kc] : kc;                       // wave-uniform test, unconditional load
    }
  };
  fetch_rows(0);
  auto load_tile = [&](int t) {
    const int k0 = kbeg + t * BK;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const bool ok = a_col_ok[i] && (k0 + a_kr[i] < Ks);
      a_reg[i] = ld4<true>(a_ptr[i] + (int64_t)t * BK * lda, ok ? 4 : 0, safe);
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int k = k0 + b_kr[i];
      const bool ok = b_live[i] && k < Ks && b_n[i] < p.N;
      b_reg[i] = ld4<true>(p.B[0] + (int64_t)b_row_next[i] * ldb + b_n[i], ok ? p.n_real - b_n[i] : 0, safe);
      // bias column: dW's extra column accumulates sum_k dY[k][m]
      const int e = ok ? b_bias_e[i] : -1;
      b_reg[i].x = e == 0 ? 1.f : b_reg[i].x; b_reg[i].y = e == 1 ? 1.f : b_reg[i].y;
      b_reg[i].z = e == 2 ? 1.f : b_reg[i].z; b_reg[i].w = e == 3 ? 1.f : b_reg[i].w;
    }
    if (t + 1 < T) fetch_rows(t + 1);
  };
  int eA = HX_EMIN, eB = HX_EMIN;           // FMT 1: exponents of the two operands' scales (the same in every thread)
  // maxima of the tile in the staging registers -> one word per wavefront and operand (called in front of a barrier)
  auto publish_max = [&]() {
    float ma = 0.f, mb = 0.f;
#pragma unroll
    for (int i = 0; i < NA; ++i) ma = fmaxf(fmaxf(ma, fmaxf(fabsf(a_reg[i].x), fabsf(a_reg[i].y))), fmaxf(fabsf(a_reg[i].z), fabsf(a_reg[i].w)));
#pragma unroll
    for (int i = 0; i < NB; ++i) mb = fmaxf(fmaxf(mb, fmaxf(fabsf(b_reg[i].x), fabsf(b_reg[i].y))), fmaxf(fabsf(b_reg[i].z), fabsf(b_reg[i].w)));
    const uint32_t ua = tx_wave_max_u32(__float_as_uint(ma)), ub = tx_wave_max_u32(__float_as_uint(mb));
    if (lane == 0) { wmax[wave] = ua; wmax[W + wave] = ub; }
  };
  auto hstore = [&](char* base, int piece_bytes, int off, const float4 v, const int se) {
    uint2 oh, ol;
    hx_split4(v, se, oh, ol);
    *reinterpret_cast<uint2*>(base + off) = oh;
    *reinterpret_cast<uint2*>(base + piece_bytes + off) = ol;
  };
  auto store_tile = [&](bool first) {
    if constexpr (FMT == 0) {
#pragma unroll
      for (int i = 0; i < NA; ++i) tx_split_store(As, TXA_PIECE, a_off[i], a_reg[i]);
#pragma unroll
      for (int i = 0; i < NB; ++i)
        if (b_live[i]) tx_split_store(Bs, TX_B_PIECE, b_off[i], b_reg[i]);
    } else {
      uint32_t mwa = 0, mwb = 0;
#pragma unroll
      for (int q4 = 0; q4 < W / 4; ++q4) {
        const uint4 wa = *reinterpret_cast<const uint4*>(wmax + 4 * q4), wb = *reinterpret_cast<const uint4*>(wmax + W + 4 * q4);
        mwa = max(mwa, max(max(wa.x, wa.y), max(wa.z, wa.w))); mwb = max(mwb, max(max(wb.x, wb.y), max(wb.z, wb.w)));
      }
      const int ta = hx_exp_of_bits(mwa), tb = hx_exp_of_bits(mwb);
      const int nA = ta > eA ? min(ta + HX_GROW, HX_EMAX) : eA, nB = tb > eB ? min(tb + HX_GROW, HX_EMAX) : eB;
      if (nA != eA || nB != eB) {                              // a scale moves (the same decision in every thread): the sums follow
        if (!first) {
          const int d = (eA - nA) + (eB - nB);
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 11; ++j)
#pragma unroll
              for (int q = 0; q < 4; ++q) acc[i][j][q] = __builtin_amdgcn_ldexpf(acc[i][j][q], d);
        }
        eA = nA; eB = nB;
      }
      const int sa = HX_TOP - eA, sb = HX_TOP - eB;
#pragma unroll
      for (int i = 0; i < NA; ++i) hstore(As, TXA_PIECE, a_off[i], a_reg[i], sa);
#pragma unroll
      for (int i = 0; i < NB; ++i)
        if (b_live[i]) hstore(Bs, TX_B_PIECE, b_off[i], b_reg[i], sb);
    }
  };
  const int strips = __builtin_amdgcn_readfirstlane((m0 + wrow + 16 < p.M) ? 2 : ((m0 + wrow < p.M) ? 1 : 0));
  auto compute_tile = [&]() {
    u32x4 a[2][NP];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int q = 0; q < NP; ++q) a[i][q] = __builtin_bit_cast(u32x4, tx_read(As + q * TXA_PIECE, TXA_ROW, TXA_MASK, 2 * wave + i, g, r));
    auto ldb = [&](u32x4 (&b)[NP], int j) {
#pragma unroll
      for (int q = 0; q < NP; ++q) b[q] = __builtin_bit_cast(u32x4, tx_read(Bs + q * TX_B_PIECE, TX_B_ROW, 15, j, g, r));
    };
    // a wavefront whose strips lie beyond M skips their MFMAs (172-row gradients): `strips` is wave-uniform (an SGPR
    // after readfirstlane), so these are scalar branches, not exec-mask sequences
    auto mma = [&](const u32x4 (&b)[NP], int j) {
      if (strips >= 1) acc[0][j] = bx_mma<FMT>(a[0], b, acc[0][j]);
      if (strips >= 2) acc[1][j] = bx_mma<FMT>(a[1], b, acc[1][j]);
    };
    u32x4 b0[NP], b1[NP];                     // double-buffered B fragments (see gemm_bf16x3_kernel)
    ldb(b0, 0);
#pragma unroll
    for (int j = 0; j < 11; j += 2) {
      if (j + 1 < 11) ldb(b1, j + 1);
      mma(b0, j);
      if (j + 2 < 11) ldb(b0, j + 2);
      if (j + 1 < 11) mma(b1, j + 1);
    }
  };
  if (T > 0) {
#if BXA_STAMPS
    uint32_t st_acc[5] = {0, 0, 0, 0, 0};      // shader cycles of wavefront 0 per phase of a k-tile step (tools/probes/tn_stamps.py)
    BXA_STAMP(0);
#endif
    load_tile(0);
    if constexpr (FMT == 1) { publish_max(); __syncthreads(); }
    store_tile(true);
    __syncthreads();
#if BXA_STAMPS
    BXA_STAMP(1);
    const uint64_t st_loop0 = __builtin_amdgcn_s_memtime();
#endif
    for (int t = 0; t < T; ++t) {
      const bool more = (BX_EXP == 1) ? false : (t + 1 < T);
#if BXA_STAMPS
      const uint64_t c0 = __builtin_amdgcn_s_memtime();
#endif
      if (more) load_tile(t + 1);
#if BXA_STAMPS
      asm volatile("s_nop 0" ::: "memory");
      const uint64_t c1 = __builtin_amdgcn_s_memtime();
#endif
      if (BX_EXP != 2) compute_tile();
#if BXA_STAMPS
      asm volatile("s_nop 0" ::: "memory");
      const uint64_t c2 = __builtin_amdgcn_s_memtime();
#endif
      if constexpr (FMT == 1) { if (more) publish_max(); }
#if BXA_STAMPS
      asm volatile("s_nop 0" ::: "memory");
      const uint64_t c3 = __builtin_amdgcn_s_memtime();
#endif
      __syncthreads();
#if BXA_STAMPS
      const uint64_t c4 = __builtin_amdgcn_s_memtime();
#endif
      if (more && BX_EXP != 3) store_tile(false);
      __syncthreads();
#if BXA_STAMPS
      const uint64_t c5 = __builtin_amdgcn_s_memtime();
      st_acc[0] += (uint32_t)(c1 - c0); st_acc[1] += (uint32_t)(c2 - c1); st_acc[2] += (uint32_t)(c3 - c2); st_acc[3] += (uint32_t)(c4 - c3); st_acc[4] += (uint32_t)(c5 - c4);
#endif
    }
#if BXA_STAMPS
    BXA_STAMP(2);
    if (threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.x < 4096) {
      g_bxa_stamps[blockIdx.x * 8 + 5] = ((uint64_t)st_acc[1] << 32) | st_acc[0];
      g_bxa_stamps[blockIdx.x * 8 + 6] = ((uint64_t)st_acc[3] << 32) | st_acc[2];
      g_bxa_stamps[blockIdx.x * 8 + 3] = ((uint64_t)st_acc[4] << 32) | (uint32_t)T;
      g_bxa_stamps[blockIdx.x * 8 + 7] = __builtin_amdgcn_s_memtime() - st_loop0;
    }
#endif
  }
  float* Cb = p.slab_base + (int64_t)split * (int64_t)p.M * p.N;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int row = m0 + wrow + 16 * i + 4 * g + reg;
      if (row >= p.M) continue;
#pragma unroll
      for (int j = 0; j < 11; ++j) {
        const int col = n0 + 16 * j + r;
        if (col < p.N) Cb[(int64_t)row * p.N + col] = FMT == 1 ? __builtin_amdgcn_ldexpf(acc[i][j][reg], eA + eB - 2 * HX_TOP) : acc[i][j][reg];
      }
    }
#if BXA_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  BXA_STAMP(4);
#endif
}

// Pre-split image of a weight operand for gemm_bf16x3_kernel<true>: [k-tile][piece][row n, padded to BN][64 B],
// the exact LDS image of the kernel (same swizzle), zero-padded in n and k.  W(n, k) = src[n*ld + k] or, with
// `trans`, src[k*ld + n] - so a k-major ("NN") operand becomes a row-major one for free.
struct BimgDev {
  const float* src[PFO_BIMG_MAX]; int64_t ld[PFO_BIMG_MAX]; int N[PFO_BIMG_MAX], K[PFO_BIMG_MAX], trans[PFO_BIMG_MAX];
  void* dst[PFO_BIMG_MAX]; int rows[PFO_BIMG_MAX];      // rows: image rows this problem fills (incl. zero padding)
  int row0[PFO_BIMG_MAX], rows_total[PFO_BIMG_MAX];     // first image row of the problem, padded rows of the whole image
  int gate[PFO_BIMG_MAX], gate_D[PFO_BIMG_MAX];         // GRU gate-ordered image (gemm.hpp PfoBimg::gate): 0 = plain
};
// Source row of image row n of a GRU gate-ordered image (gru_fused_kernel): blocks of GF_BN = 128 image rows = two groups of
// four 16-row tiles (r, z, n_i, n_h) for 16 hidden units each.  gate 1: the message weights W_ih [3D, M] (no n_h rows),
// gate 2: the hidden-state weights W_hh [3D, D] (no n_i rows).  -1: a zero row.
__device__ __forceinline__ int bimg_gate_row(int n, int which, int D) {
  const int b = n >> 7, t = (n & 127) >> 4, i = n & 15;
  const int u = 32 * b + 16 * (t >> 2) + i, gt = t & 3;
  if (u >= D) return -1;
  if (gt == 0) return u;
  if (gt == 1) return D + u;
  if (gt == 2) return which == 1 ? 2 * D + u : -1;
  return which == 2 ? 2 * D + u : -1;
}
__global__ __launch_bounds__(256) void bimg_kernel(const BimgDev g) {
  const int z = blockIdx.y;
  const int N = g.N[z], K = g.K[z], rows = g.rows[z];
  const int T = (K + 31) / 32;
  const int64_t total = (int64_t)T * rows * 4;                 // one thread per (tile, row, 16-byte chunk)
  const float* __restrict__ src = g.src[z];
  const int64_t ld = g.ld[z];
  char* dst = reinterpret_cast<char*>(g.dst[z]);
  for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    int t, n, c;
    if (g.trans[z]) { n = (int)(i % rows); c = (int)((i / rows) & 3); t = (int)(i / (4 * (int64_t)rows)); }
    else            { c = (int)(i & 3); n = (int)((i >> 2) % rows); t = (int)((i >> 2) / rows); }
    uint32_t w[3][4];
#pragma unroll
    for (int e2 = 0; e2 < 4; ++e2) {
      uint32_t pc[2][3];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int k = 32 * t + 8 * c + 2 * e2 + h;
        float x = 0.f;
        if (g.gate[z]) {
          const int sr = bimg_gate_row(n, g.gate[z], g.gate_D[z]);
          if (sr >= 0 && k < K) x = src[(int64_t)sr * ld + k];
        } else if (n < N && k < K) x = g.trans[z] ? src[(int64_t)k * ld + n] : src[(int64_t)n * ld + k];
        const uint32_t b1 = __float_as_uint(x) & 0xFFFF0000u;
        const float r1 = x - __uint_as_float(b1);
        const uint32_t b2 = __float_as_uint(r1) & 0xFFFF0000u;
        const float r2 = r1 - __uint_as_float(b2);
        pc[h][0] = b1; pc[h][1] = b2; pc[h][2] = __float_as_uint(r2);
      }
#pragma unroll
      for (int q = 0; q < 3; ++q) w[q][e2] = (pc[1][q] & 0xFFFF0000u) | (pc[0][q] >> 16);
    }
    const int ng = n + g.row0[z];                               // row of the (possibly stacked) image
    const int64_t off = (int64_t)ng * 64 + ((c ^ bx_swz(ng)) << 4);
#pragma unroll
    for (int q = 0; q < 3; ++q)
      *reinterpret_cast<uint4*>(dst + ((int64_t)t * 3 + q) * g.rows_total[z] * 64 + off) = uint4{w[q][0], w[q][1], w[q][2], w[q][3]};
  }
}


// The fp16x2 image (FMT 1): [k-tile][piece h | l][row][64 B] in the same swizzled layout, then one int32 per image row: the
// biased exponent E of the row's largest magnitude (clamped); the row is stored times 2^(HX_TOP - E).
__device__ __forceinline__ int bimg_src_row(const BimgDev& g, int z, int n) {
  if (g.gate[z]) return bimg_gate_row(n, g.gate[z], g.gate_D[z]);
  return n < g.N[z] ? n : -1;
}
__device__ __forceinline__ int32_t* bimg_exps(const BimgDev& g, int z) {
  const int T = (g.K[z] + 31) / 32;
  return reinterpret_cast<int32_t*>(reinterpret_cast<char*>(g.dst[z]) + (int64_t)T * 2 * g.rows_total[z] * 64);
}
// Row-major operands (and the GRU's gate-ordered ones): one wavefront per image row, lanes along k with 16-byte loads; the
// first pass finds the row maximum, the second (L1 / L2 hits) converts.  vec: every row start and K are multiples of 4 floats.
__global__ __launch_bounds__(256) void bimg_h_rows_kernel(const BimgDev g) {
  const int z = blockIdx.y;
  const int K = g.K[z], rows = g.rows[z];
  const int T = (K + 31) / 32;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= rows) return;
  const int lane = threadIdx.x & 63;
  const int sr = bimg_src_row(g, z, n);
  const float* __restrict__ row = g.src[z] + (int64_t)max(sr, 0) * g.ld[z];
  const bool vec = ((g.ld[z] | K) & 3) == 0 && (((uintptr_t)g.src[z]) & 15) == 0;
  auto ld4k = [&](int k) -> float4 {                            // k % 4 == 0
    if (sr < 0 || k >= K) return float4{0.f, 0.f, 0.f, 0.f};
    if (vec) return *reinterpret_cast<const float4*>(row + k);
    return float4{row[k], k + 1 < K ? row[k + 1] : 0.f, k + 2 < K ? row[k + 2] : 0.f, k + 3 < K ? row[k + 3] : 0.f};
  };
  float m = 0.f;
  for (int ci = lane; ci < 4 * T; ci += 64) m = fmaxf(m, hx_absmax8(ld4k(8 * ci), ld4k(8 * ci + 4)));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  const int E = hx_exp_of(m);
  const int se = HX_TOP - E;
  char* dst = reinterpret_cast<char*>(g.dst[z]);
  const int ng = n + g.row0[z];
  const int64_t piece = (int64_t)g.rows_total[z] * 64;
  for (int ci = lane; ci < 4 * T; ci += 64) {
    const int t = ci >> 2, c = ci & 3;
    uint2 h0, l0, h1, l1;
    hx_split4(ld4k(8 * ci), se, h0, l0);
    hx_split4(ld4k(8 * ci + 4), se, h1, l1);
    const int64_t off = (int64_t)ng * 64 + ((c ^ bx_swz(ng)) << 4);
    *reinterpret_cast<uint4*>(dst + ((int64_t)t * 2 + 0) * piece + off) = uint4{h0.x, h0.y, h1.x, h1.y};
    *reinterpret_cast<uint4*>(dst + ((int64_t)t * 2 + 1) * piece + off) = uint4{l0.x, l0.y, l1.x, l1.y};
  }
  if (lane == 0) bimg_exps(g, z)[ng] = E;
}
// Lists with k-major ("trans") operands take two launches: the row exponents (one wavefront per image row; a k-major row is
// a strided column of the source, its 64-byte lines shared with the neighbouring rows' wavefronts), then the conversion in
// bimg_kernel's thread order (coalesced along n for k-major sources).
__global__ __launch_bounds__(256) void bimg_exp_kernel(const BimgDev g) {
  const int z = blockIdx.y;
  const int K = g.K[z], rows = g.rows[z];
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= rows) return;
  const int lane = threadIdx.x & 63;
  const int sr = bimg_src_row(g, z, n);
  const float* __restrict__ src = g.src[z];
  const int64_t ld = g.ld[z];
  float m = 0.f;
  if (sr >= 0) {
    if (g.trans[z]) for (int k = lane; k < K; k += 64) m = fmaxf(m, fabsf(src[(int64_t)k * ld + sr]));
    else for (int k = lane; k < K; k += 64) m = fmaxf(m, fabsf(src[(int64_t)sr * ld + k]));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if (lane == 0) bimg_exps(g, z)[n + g.row0[z]] = hx_exp_of(m);
}
__global__ __launch_bounds__(256) void bimg_h_kernel(const BimgDev g) {
  const int z = blockIdx.y;
  const int N = g.N[z], K = g.K[z], rows = g.rows[z];
  const int T = (K + 31) / 32;
  const int64_t total = (int64_t)T * rows * 4;                 // one thread per (tile, row, 16-byte chunk)
  const float* __restrict__ src = g.src[z];
  const int64_t ld = g.ld[z];
  char* dst = reinterpret_cast<char*>(g.dst[z]);
  const int32_t* exps = bimg_exps(g, z);
  const int64_t piece = (int64_t)g.rows_total[z] * 64;
  for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    int t, n, c;
    if (g.trans[z]) { n = (int)(i % rows); c = (int)((i / rows) & 3); t = (int)(i / (4 * (int64_t)rows)); }
    else            { c = (int)(i & 3); n = (int)((i >> 2) % rows); t = (int)((i >> 2) / rows); }
    const int sr = bimg_src_row(g, z, n);
    float x[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = 32 * t + 8 * c + e;
      x[e] = (sr >= 0 && k < K) ? (g.trans[z] ? src[(int64_t)k * ld + sr] : src[(int64_t)sr * ld + k]) : 0.f;
    }
    const int ng = n + g.row0[z];
    const int se = HX_TOP - exps[ng];
    uint2 h0, l0, h1, l1;
    hx_split4(float4{x[0], x[1], x[2], x[3]}, se, h0, l0);
    hx_split4(float4{x[4], x[5], x[6], x[7]}, se, h1, l1);
    const int64_t off = (int64_t)ng * 64 + ((c ^ bx_swz(ng)) << 4);
    *reinterpret_cast<uint4*>(dst + ((int64_t)t * 2 + 0) * piece + off) = uint4{h0.x, h0.y, h1.x, h1.y};
    *reinterpret_cast<uint4*>(dst + ((int64_t)t * 2 + 1) * piece + off) = uint4{l0.x, l0.y, l1.x, l1.y};
  }
}

// ---------------------------------------------------------------------------------------------
// Several SMALL independent contractions in one launch (the composite-weight products of a layer and their
// gradient chain: each is far too small to fill the chip or to amortise a launch).  32 x 64 tiles; the operand
// layout is a per-problem switch (wavefront-uniform).
#define MULTI_MAX PFO_GEMM_MULTI_MAX
struct MultiDev {
  GemmDev p[MULTI_MAX];
  int layout[MULTI_MAX];      // a_kmajor * 2 + b_kmajor
  int tile_begin[MULTI_MAX];  // first flattened tile of each problem
  int tm[MULTI_MAX], tn[MULTI_MAX];
  int n;
};
template <bool VEC>
__global__ __launch_bounds__(GEMM_THREADS) void gemm_multi_kernel(const MultiDev g) {
  // the 32 x 64 tile's staging: A max(32 * 34, 32 * 48), B max(64 * 34, 32 * 80) floats - 16 KB, so these launches take
  // little room beside the large ones they run next to
  __shared__ __attribute__((aligned(16))) float lds_a[1536];
  __shared__ __attribute__((aligned(16))) float lds_b[2560];
  int q = 0;
#pragma unroll
  for (int i = 1; i < MULTI_MAX; ++i)
    if (i < g.n && (int)blockIdx.x >= g.tile_begin[i]) q = i;
  int t = blockIdx.x - g.tile_begin[q];
  const int per_batch = g.tm[q] * g.tn[q];
  const int zb = t / per_batch;
  t -= zb * per_batch;
  const int bx = t / g.tn[q], by = t % g.tn[q];
  const GemmDev p = g.p[q];                       // a copy: indexing the by-value argument through a reference costs scratch
  switch (g.layout[q]) {
    case 0: gemm_tile<false, false, 2, VEC>(p, bx, by, zb, 0, lds_a, lds_b); break;
    case 1: gemm_tile<false, true, 2, VEC>(p, bx, by, zb, 0, lds_a, lds_b); break;
    case 2: gemm_tile<true, false, 2, VEC>(p, bx, by, zb, 0, lds_a, lds_b); break;
    default: gemm_tile<true, true, 2, VEC>(p, bx, by, zb, 0, lds_a, lds_b); break;
  }
}

// DIRECT form of the same grouped launch (round 6).  These products (composite weights, their gradient chains: M, N, K <= 704,
// operands L2-resident weights) are chains of k-tiles, each a global -> LDS -> barrier -> fragment round trip: 10-57 us per
// launch for ~40 MFLOP, 0.32 ms per step of side-stream residency.  Here a wavefront owns ONE 16 x 16 output tile and takes
// its MFMA fragments straight from global memory - no LDS, no barrier: lane (r, g) holds A[row r][16 q + 4 g + j] and
// B[16 q + 4 g + j][col r], j = 0..3, as one float4 (or four dwords on a k-major operand) per 16 k; MFMA step j multiplies the
// j-th elements, so every k of the block is taken exactly once (a permutation of the k order inside a block of 16 - the sum
// does not care).  Loads run MULTI_PD blocks ahead; four times the workgroups (16 x 64 per workgroup instead of 32 x 64).
#ifndef MULTI_PD
#define MULTI_PD 4
#endif
template <bool A_KM, bool B_KM, bool VEC>
__device__ __forceinline__ void gemm_direct_tile(const GemmDev& p, const int bx, const int by, const int zb) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int m0 = bx * 16, n0 = by * 64 + 16 * wave;
  if (n0 >= p.N) return;                                    // (no barrier in this kernel: a wavefront may leave)
  const int K = p.K[0];
  const float* Ab = p.A[0] + zb * p.a_bs[0];
  const float* Bb = p.B[0] + zb * p.b_bs[0];
  const int64_t lda = p.lda[0], ldb = p.ldb[0];
  const int mr = min(m0 + r, p.M - 1), nr = min(n0 + r, p.N - 1);       // clamped: rows / columns beyond the edge only feed outputs that are not stored
  const int T = (K + 15) >> 4;
  float a[MULTI_PD][4], b[MULTI_PD][4];
  auto load = [&](int q, float (&av)[4], float (&bv)[4]) {
    const int k0 = 16 * q + 4 * g;
#pragma unroll
    for (int j = 0; j < 4; ++j) { av[j] = 0.f; bv[j] = 0.f; }
    if (!A_KM) {
      const float* src = Ab + (int64_t)mr * lda + min(k0, max(K - 4, 0));
      if (VEC) { const float4 v = *reinterpret_cast<const float4*>(src); av[0] = v.x; av[1] = v.y; av[2] = v.z; av[3] = v.w; }
      else {
#pragma unroll
        for (int j = 0; j < 4; ++j) av[j] = Ab[(int64_t)mr * lda + min(k0 + j, K - 1)];
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) av[j] = Ab[(int64_t)min(k0 + j, K - 1) * lda + mr];
    }
    if (!B_KM) {
      const float* src = Bb + (int64_t)nr * ldb + min(k0, max(K - 4, 0));
      if (VEC) { const float4 v = *reinterpret_cast<const float4*>(src); bv[0] = v.x; bv[1] = v.y; bv[2] = v.z; bv[3] = v.w; }
      else {
#pragma unroll
        for (int j = 0; j < 4; ++j) bv[j] = Bb[(int64_t)nr * ldb + min(k0 + j, K - 1)];
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) bv[j] = Bb[(int64_t)min(k0 + j, K - 1) * ldb + nr];
    }
    // beyond K both operands are zero (selects behind the unconditional loads)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool in = k0 + j < K;
      av[j] = in ? av[j] : 0.f;
      bv[j] = in ? bv[j] : 0.f;
    }
  };
  f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int d = 0; d < MULTI_PD; ++d)
    if (d < T) load(d, a[d], b[d]);
  for (int q0 = 0; q0 < T; q0 += MULTI_PD) {
#pragma unroll
    for (int d = 0; d < MULTI_PD; ++d) {
      if (q0 + d < T) {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[d][j], b[d][j], acc, 0, 0, 0);
        if (q0 + d + MULTI_PD < T) load(q0 + d + MULTI_PD, a[d], b[d]);
      }
    }
  }
  // C/D layout of 16x16x4: col = lane & 15, row = 4 * (lane >> 4) + reg
  float* Cb = p.C + zb * p.c_bs;
  const float* bias = p.bias ? p.bias + zb * p.bias_bs : nullptr;
  const int col = n0 + r;
  if (col < p.N) {
    const float bv = bias ? bias[col] : 0.f;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int row = m0 + 4 * g + reg;
      if (row < p.M) {
        float v = acc[reg];
        if (p.accumulate) v += Cb[(int64_t)row * p.ldc + col];
        Cb[(int64_t)row * p.ldc + col] = v + bv;
      }
    }
  }
}
template <bool VEC>
__global__ __launch_bounds__(GEMM_THREADS) void gemm_multi_direct_kernel(const MultiDev g) {
  int q = 0;
#pragma unroll
  for (int i = 1; i < MULTI_MAX; ++i)
    if (i < g.n && (int)blockIdx.x >= g.tile_begin[i]) q = i;
  int t = blockIdx.x - g.tile_begin[q];
  const int per_batch = g.tm[q] * g.tn[q];
  const int zb = t / per_batch;
  t -= zb * per_batch;
  const int bx = t / g.tn[q], by = t % g.tn[q];
  const GemmDev p = g.p[q];
  switch (g.layout[q]) {
    case 0: gemm_direct_tile<false, false, VEC>(p, bx, by, zb); break;
    case 1: gemm_direct_tile<false, true, VEC>(p, bx, by, zb); break;
    case 2: gemm_direct_tile<true, false, VEC>(p, bx, by, zb); break;
    default: gemm_direct_tile<true, true, VEC>(p, bx, by, zb); break;
  }
}

// out[m, n] += sum_r u_r[m] * v_r[n], several independent updates in one launch (blockIdx.y = the update)
struct Rank1Dev {
  const float* u[PFO_RANK1_MAX]; const float* v[PFO_RANK1_MAX]; float* out[PFO_RANK1_MAX];
  int64_t ldu[PFO_RANK1_MAX], ldv[PFO_RANK1_MAX], ldo[PFO_RANK1_MAX], u_rs[PFO_RANK1_MAX], v_rs[PFO_RANK1_MAX];
  int M[PFO_RANK1_MAX], N[PFO_RANK1_MAX], reps[PFO_RANK1_MAX];
};
__global__ void rank1_kernel(const Rank1Dev g) {
  const int q = blockIdx.y;
  const int N = g.N[q], reps = g.reps[q];
  const int64_t total = (int64_t)g.M[q] * N;
  const float* __restrict__ u = g.u[q];
  const float* __restrict__ v = g.v[q];
  float* __restrict__ out = g.out[q];
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int m = (int)(e / N), n = (int)(e - (int64_t)m * N);
    float acc = 0.f;
    for (int r = 0; r < reps; ++r) acc = fmaf(u[r * g.u_rs[q] + (int64_t)m * g.ldu[q]], v[r * g.v_rs[q] + (int64_t)n * g.ldv[q]], acc);
    out[(int64_t)m * g.ldo[q] + n] += acc;
  }
}
int pfo_rank1_multi_launch(const PfoRank1* list, int n, hipStream_t stream) {
  PFO_REQUIRE(list && n >= 1 && n <= PFO_RANK1_MAX, "bad rank-1 list");
  Rank1Dev g;
  int64_t most = 0;
  for (int i = 0; i < n; ++i) {
    const PfoRank1& r = list[i];
    PFO_REQUIRE(r.u && r.v && r.out && r.M > 0 && r.N > 0 && r.reps >= 1, "bad rank-1 update");
    g.u[i] = r.u; g.v[i] = r.v; g.out[i] = r.out; g.ldu[i] = r.ldu; g.ldv[i] = r.ldv; g.ldo[i] = r.ldo; g.M[i] = r.M; g.N[i] = r.N;
    g.reps[i] = r.reps; g.u_rs[i] = r.u_rs; g.v_rs[i] = r.v_rs;
    most = std::max(most, (int64_t)r.M * r.N);
  }
  const int nb = (int)std::min<int64_t>(512, pfo_ceil_div(most, 256));
  PFO_KLAUNCH(rank1_kernel, dim3(nb, n), dim3(256), 0, stream, g);
  PFO_LAUNCH_CHECK();
  return PFO_OK;
}
// dst[i] (+)= sum_s src[s * stride + i]  (fixed order: reproducible), several destinations in one launch
struct SumSlabsDev {
  float* dst[PFO_SUM_SLABS_MAX]; const float* src[PFO_SUM_SLABS_MAX];
  int64_t stride[PFO_SUM_SLABS_MAX], count[PFO_SUM_SLABS_MAX];
  int n_slabs[PFO_SUM_SLABS_MAX], accumulate[PFO_SUM_SLABS_MAX];
};
__global__ void sum_slabs_kernel(const SumSlabsDev g) {
  const int q = blockIdx.y;
  const float* __restrict__ src = g.src[q];
  float* __restrict__ dst = g.dst[q];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < g.count[q]; i += (int64_t)gridDim.x * blockDim.x) {
    float acc = g.accumulate[q] ? dst[i] : 0.f;
    for (int sl = 0; sl < g.n_slabs[q]; ++sl) acc += src[sl * g.stride[q] + i];
    dst[i] = acc;
  }
}
int pfo_sum_slabs_launch(const PfoSumSlabs* list, int n, hipStream_t stream) {
  PFO_REQUIRE(list && n >= 1 && n <= PFO_SUM_SLABS_MAX, "bad slab list");
  SumSlabsDev g;
  int64_t most = 0;
  for (int i = 0; i < n; ++i) {
    PFO_REQUIRE(list[i].dst && list[i].src && list[i].count > 0 && list[i].n_slabs >= 1, "bad slab sum");
    g.dst[i] = list[i].dst; g.src[i] = list[i].src; g.stride[i] = list[i].stride; g.count[i] = list[i].count;
    g.n_slabs[i] = list[i].n_slabs; g.accumulate[i] = list[i].accumulate;
    most = std::max(most, list[i].count);
  }
  const int nb = (int)std::min<int64_t>(256, pfo_ceil_div(most, 256));
  PFO_KLAUNCH(sum_slabs_kernel, dim3(nb, n), dim3(256), 0, stream, g);
  PFO_LAUNCH_CHECK();
  return PFO_OK;
}
int pfo_rank1_launch(const float* u, int64_t ldu, const float* v, int64_t ldv, int M, int N, float* out, int64_t ldo,
                     hipStream_t stream) {
  PfoRank1 r;
  r.u = u; r.ldu = ldu; r.v = v; r.ldv = ldv; r.M = M; r.N = N; r.out = out; r.ldo = ldo;
  return pfo_rank1_multi_launch(&r, 1, stream);
}

// ---------------------------------------------------------------------------------------------
// Grouped weight gradients: every dW (+ bias gradient as one extra column) of a layer in ONE launch.
// All problems share the K extent (the layer's instance rows); K is split into `nsplit` chunks, each
// workgroup writes a partial slab tile, one grouped reduce folds the slabs into the gradient buffers.
#define TN_MAX_PROBLEMS 16
struct TnProbDev {
  const float* A; int64_t lda; const float* B; int64_t ldb; const int32_t* b_idx;
  float* C; int64_t ldc; float* bias_out; int bias_accumulate, c_accumulate;
  int M, N_real, N;          // N = N_real + (bias_out ? 1 : 0)
  int tile_begin, tn;        // first flattened tile, column tiles
  int64_t slab_off;          // floats
};
struct TnGroupDev {
  TnProbDev p[TN_MAX_PROBLEMS];
  int n, K, nsplit, chunk, total_tiles;
  int xcd_map;               // bx kernel, 1-D grid: every tile of one K split on the same XCD (split s on XCD s & 7)
  const int32_t* k_dev;
  float* slabs;
};

template <bool VEC>
__global__ __launch_bounds__(GEMM_THREADS) void gemm_tn_group_kernel(const TnGroupDev g) {
  GEMM_LDS_DECL;
  int q = 0;
#pragma unroll
  for (int i = 1; i < TN_MAX_PROBLEMS; ++i)
    if (i < g.n && (int)blockIdx.x >= g.p[i].tile_begin) q = i;
  const TnProbDev& pr = g.p[q];
  const int t = blockIdx.x - pr.tile_begin;
  GemmDev d;
  d.A[0] = pr.A; d.A[1] = nullptr; d.lda[0] = pr.lda; d.lda[1] = 0; d.a_idx[0] = d.a_idx[1] = nullptr;
  d.B[0] = pr.B; d.B[1] = nullptr; d.ldb[0] = pr.ldb; d.ldb[1] = 0; d.b_idx = pr.b_idx;
  d.K[0] = g.K; d.K[1] = 0;
  d.C = nullptr; d.ldc = 0; d.bias = nullptr; d.row_scale = nullptr; d.rs_ld = 0; d.row_zero = nullptr;
  d.relu_src = nullptr; d.relu_ld = 0; d.M = pr.M; d.N = pr.N; d.m_dev = g.k_dev; d.relu = 0; d.accumulate = 0;
  d.add_src = nullptr; d.add_ld = 0; d.add_idx = nullptr;
  d.nsplit = 2;                      // any value > 1: selects the slab epilogue; the real split index comes from blockIdx.y
  d.split_chunk = g.chunk;
  d.a_bs[0] = d.a_bs[1] = d.b_bs[0] = d.b_bs[1] = d.c_bs = d.bias_bs = d.rs_bs = 0;
  d.n_real = pr.N_real;
  // slab region of this problem: [nsplit][M][N]; gemm_tile indexes it as (zb * nsplit + split) with zb = 0
  d.slab_base = g.slabs + pr.slab_off - (int64_t)0;
  d.nsplit = g.nsplit > 1 ? g.nsplit : 2;
  d.dyn_chunk = (g.k_dev != nullptr && g.nsplit > 1) ? 1 : 0;
  gemm_tile<true, true, 0, VEC>(d, t / pr.tn, t % pr.tn, 0, blockIdx.y, lds_a, lds_b);
}

// shader-clock pair of the grouped weight-gradient kernel (common.hpp pfo_clock_*)
__device__ unsigned long long g_gemm_clock[1][2];
int pfo_gemm_clock_read(double* out, int reset) {
  unsigned long long h[1][2];
  if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_gemm_clock), sizeof(h)) != hipSuccess) return PFO_ERR_HIP;
  out[0] = (double)h[0][0]; out[1] = (double)h[0][1];
  if (reset) { memset(h, 0, sizeof(h)); if (hipMemcpyToSymbol(HIP_SYMBOL(g_gemm_clock), h, sizeof(h)) != hipSuccess) return PFO_ERR_HIP; }
  return PFO_OK;
}
// the same grouped launch on the bf16 matrix cores (3-way split, transposed LDS reads)
template <int FMT, int W = 4>
__global__ __launch_bounds__(64 * W, 2) void gemm_tn_group_bx_kernel(const TnGroupDev g) {
  __shared__ __attribute__((aligned(16))) char lds[BxFmt<FMT>::NP * (BK * 64 * W + TX_B_PIECE) + 64];
  const bool clk_on = blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 64;
  PfoClockStamp clk;
  if (clk_on) clk = pfo_clock_begin();
  // Workgroups go to the XCDs round-robin by linear id.  All tiles of one K split read the same rows of A and B (a tile
  // takes 128 of A's columns and all of B's 172): with the (tile, split) grid the six tiles of dW1ovT's split sat on six
  // different L2s and B came from HBM six times (FETCH 188 MB per launch on average where the operands are 83 MB).  The
  // 1-D grid puts split s on XCD s & 7: id = (s / 8) * 8 * tiles + tile * 8 + (s & 7); a last group of fewer than eight
  // splits is laid out tile-major.
  int tile_id = blockIdx.x, split = blockIdx.y;
  if (g.xcd_map) {
    const int id = blockIdx.x, T = g.total_tiles;
    const int grp = id / (8 * T), loc = id - grp * 8 * T;
    const int in_grp = min(8, g.nsplit - 8 * grp);
    tile_id = loc / in_grp;
    split = 8 * grp + loc % in_grp;
  }
  int q = 0;
#pragma unroll
  for (int i = 1; i < TN_MAX_PROBLEMS; ++i)
    if (i < g.n && tile_id >= g.p[i].tile_begin) q = i;
  const TnProbDev& pr = g.p[q];
  const int t = tile_id - pr.tile_begin;
  GemmDev d;
  d.A[0] = pr.A; d.lda[0] = pr.lda; d.B[0] = pr.B; d.ldb[0] = pr.ldb; d.b_idx = pr.b_idx;
  d.K[0] = g.K; d.M = pr.M; d.N = pr.N; d.m_dev = g.k_dev;
  d.split_chunk = g.chunk;
  d.n_real = pr.N_real;
  d.slab_base = g.slabs + pr.slab_off;
  d.nsplit = g.nsplit;
  d.dyn_chunk = (g.k_dev != nullptr && g.nsplit > 1) ? 1 : 0;
  gemm_tile_tn_bx<FMT, W>(d, t / pr.tn, t % pr.tn, split, lds);
  if (clk_on && threadIdx.x == 0) pfo_clock_end(clk, g_gemm_clock[0]);
}
__global__ __launch_bounds__(256) void tn_group_reduce_kernel(const TnGroupDev g) {
  int K = g.K;
  if (g.k_dev) K = min(K, *g.k_dev);
  int chunk = g.chunk;
  if (g.k_dev && g.nsplit > 1) chunk = ((K + g.nsplit - 1) / g.nsplit + BK - 1) / BK * BK;   // same rule as gemm_tile
  int nz = chunk > 0 ? (K + chunk - 1) / chunk : 0;
  nz = min(nz, g.nsplit);
  for (int q = 0; q < g.n; ++q) {
    const TnProbDev& pr = g.p[q];
    const int64_t total = (int64_t)pr.M * pr.N;
    const float* slab = g.slabs + pr.slab_off;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
      // eight slab loads in flight per thread; the summation order (z ascending within four interleaved chains, then a
      // fixed tree) depends on nz only, so the result is reproducible
      float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
      int z = 0;
      for (; z + 8 <= nz; z += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = slab[(int64_t)(z + u) * total + e];
        s0 += v[0]; s1 += v[1]; s2 += v[2]; s3 += v[3];
        s0 += v[4]; s1 += v[5]; s2 += v[6]; s3 += v[7];
      }
      for (; z < nz; ++z) s0 += slab[(int64_t)z * total + e];
      const float s = (s0 + s1) + (s2 + s3);
      const int m = (int)(e / pr.N), n = (int)(e - (int64_t)m * pr.N);
      if (n < pr.N_real) {
        float* o = pr.C + (int64_t)m * pr.ldc + n;
        *o = pr.c_accumulate ? *o + s : s;
      } else {
        pr.bias_out[m] = pr.bias_accumulate ? pr.bias_out[m] + s : s;
      }
    }
  }
}

int pfo_gemm_tn_group_launch(const PfoTnProblem* probs, int n, int K, const int32_t* k_dev, float* slabs,
                             int64_t slab_floats, hipStream_t stream) {
  PFO_REQUIRE(n >= 1 && n <= TN_MAX_PROBLEMS && K > 0 && probs && slabs, "bad arguments");
  TnGroupDev g;
  memset(&g, 0, sizeof(g));
  bool vec = true;
  int tiles = 0;
  // 256-row tiles of eight wavefronts (one workgroup per CU) when every problem is at least that tall: 0 = never  (A/B switch)
  static const int tn8_env = getenv("PFO_TN8") ? atoi(getenv("PFO_TN8")) : PFO_DEFAULT_TN8;
  static const int tn_fmt0 = getenv("PFO_TN_FMT") ? atoi(getenv("PFO_TN_FMT")) : PFO_DEFAULT_TN_FMT;
  static const int bx0 = getenv("PFO_GEMM_BF16X3") ? atoi(getenv("PFO_GEMM_BF16X3")) : PFO_DEFAULT_BF16X3;
  bool tn8 = tn8_env != 0 && tn_fmt0 != 0 && bx0 >= 1;
  for (int i = 0; i < n; ++i) {
    const PfoTnProblem& q = probs[i];
    tn8 = tn8 && q.M >= tn8_env && aligned4(q.A) && aligned4(q.B) && (q.lda % 4) == 0 && (q.ldb % 4) == 0 && (q.M % 4) == 0 && (q.N % 4) == 0;
  }
  static const int slots4_env = getenv("PFO_TN_SLOTS") ? atoi(getenv("PFO_TN_SLOTS")) : 512;      // A/B: workgroups of the 128-row form the split count is sized for (2 per CU; 768 = 3: LDS 3 x 49.6 KB and 3 x 120 registers fit)
  const int tbm = tn8 ? 256 : BM, slots = tn8 ? 256 : slots4_env;
  int64_t per_split = 0;
  for (int i = 0; i < n; ++i) {
    const PfoTnProblem& s = probs[i];
    PFO_REQUIRE(s.A && s.B && s.C && s.M > 0 && s.N > 0, "bad problem");
    TnProbDev& d = g.p[i];
    d.A = s.A; d.lda = s.lda; d.B = s.B; d.ldb = s.ldb; d.b_idx = s.b_idx;
    d.C = s.C; d.ldc = s.ldc;
    d.bias_out = s.bias_out; d.bias_accumulate = s.bias_accumulate; d.c_accumulate = s.c_accumulate;
    d.M = s.M; d.N_real = s.N; d.N = s.N + (s.bias_out ? 1 : 0);
    d.tn = (int)pfo_ceil_div(d.N, BN);
    d.tile_begin = tiles;
    tiles += (int)pfo_ceil_div(d.M, tbm) * d.tn;
    d.slab_off = per_split;            // scaled by nsplit below
    per_split += (int64_t)d.M * d.N;
    vec = vec && aligned4(s.A) && aligned4(s.B) && (s.lda % 4) == 0 && (s.ldb % 4) == 0 && (s.M % 4) == 0 && (s.N % 4) == 0;
  }
  // one full round of resident workgroups (2 per CU x 256 CUs), never a nearly empty second one
  int nsplit = (int)std::max<int64_t>(1, std::min<int64_t>(slots / std::max(1, tiles), pfo_ceil_div(K, 4 * BK)));
  int chunk = (int)pfo_align_up(pfo_ceil_div(K, nsplit), BK);
  nsplit = (int)pfo_ceil_div(K, chunk);
  PFO_REQUIRE(slab_floats >= per_split * nsplit, "split-K workspace too small");
  for (int i = 0; i < n; ++i) g.p[i].slab_off *= nsplit;
  g.n = n; g.K = K; g.nsplit = nsplit; g.chunk = chunk; g.total_tiles = tiles; g.k_dev = k_dev; g.slabs = slabs;
  double flops = 0;
  for (int i = 0; i < n; ++i) flops += 2.0 * probs[i].M * probs[i].N * (double)K;
  static const int bx = getenv("PFO_GEMM_BF16X3") ? atoi(getenv("PFO_GEMM_BF16X3")) : PFO_DEFAULT_BF16X3;
  const bool use_bx = vec && bx >= 1;
  pfo_prof_begin(stream);
  static const int tn_fmt = getenv("PFO_TN_FMT") ? atoi(getenv("PFO_TN_FMT")) : PFO_DEFAULT_TN_FMT;      // A/B switch
  static const int xcd_map = getenv("PFO_TN_XCD") ? atoi(getenv("PFO_TN_XCD")) : 1;                      // A/B switch
  g.xcd_map = use_bx && xcd_map && nsplit > 1;
  const dim3 grid_bx = g.xcd_map ? dim3(tiles * nsplit, 1) : dim3(tiles, nsplit);
  if (use_bx && tn_fmt && tn8) PFO_KLAUNCH((gemm_tn_group_bx_kernel<1, 8>), grid_bx, dim3(512), 0, stream, g);
  else if (use_bx && tn_fmt) PFO_KLAUNCH(gemm_tn_group_bx_kernel<1>, grid_bx, dim3(GEMM_THREADS), 0, stream, g);
  else if (use_bx) PFO_KLAUNCH(gemm_tn_group_bx_kernel<0>, grid_bx, dim3(GEMM_THREADS), 0, stream, g);
  else if (vec) PFO_KLAUNCH(gemm_tn_group_kernel<true>, dim3(tiles, nsplit), dim3(GEMM_THREADS), 0, stream, g);
  else PFO_KLAUNCH(gemm_tn_group_kernel<false>, dim3(tiles, nsplit), dim3(GEMM_THREADS), 0, stream, g);
  PFO_LAUNCH_CHECK();
  // the GEMM kernel alone; with a device-side K bound the work is (flops per k-row) x the count read back at collect time
  if (use_bx) pfo_prof_end_dev((tn_fmt && tn8) ? PFO_PROF_GEMM_TN_BX8 : PFO_PROF_GEMM_TN_BX, flops / (double)K, k_dev, K, stream);
  if (use_bx) pfo_prof_begin(stream);                           // the slab fold is a family of its own: slabs in, matrices out
  PFO_KLAUNCH(tn_group_reduce_kernel, dim3((unsigned)std::min<int64_t>(1024, pfo_ceil_div(per_split, 256))), dim3(256), 0,
                     stream, g);
  PFO_LAUNCH_CHECK();
  if (!use_bx) pfo_prof_end_dev(PFO_PROF_GEMM_TN, flops / (double)K, k_dev, K, stream);
  else pfo_prof_end(PFO_PROF_TN_REDUCE, (double)per_split * (nsplit + 1) * sizeof(float), stream);
  return PFO_OK;
}


static void to_dev(const PfoGemm& g, GemmDev& d) {
  d.xcd_tm = d.xcd_tn = 0;
  for (int s = 0; s < 2; ++s) {
    d.A[s] = g.A[s]; d.lda[s] = g.lda[s]; d.a_idx[s] = g.a_idx[s];
    d.B[s] = g.B[s]; d.ldb[s] = g.ldb[s]; d.K[s] = g.K[s];
    d.a_bs[s] = g.a_bs[s]; d.b_bs[s] = g.b_bs[s];
  }
  d.b_idx = g.b_idx;
  d.C = g.C; d.ldc = g.ldc; d.bias = g.bias; d.row_scale = g.row_scale; d.rs_ld = g.rs_ld; d.row_zero = g.row_zero;
  d.relu_src = g.relu_src; d.relu_ld = g.relu_ld; d.M = g.M; d.N = g.N; d.m_dev = g.m_dev;
  d.add_src = g.add_src; d.add_ld = g.add_ld; d.add_idx = g.add_idx;
  d.relu = g.relu; d.accumulate = g.accumulate; d.nsplit = 1; d.split_chunk = 0;
  d.c_bs = g.c_bs; d.bias_bs = g.bias_bs; d.rs_bs = g.rs_bs;
  d.n_real = g.N; d.slab_base = g.slabs; d.dyn_chunk = 0;
  d.b_img = nullptr; d.b_img_rows = 0; d.b_img2 = nullptr;
  d.gg_gates = g.gg_gates; d.gg_h = g.gg_h; d.gg_hm = g.gg_hm; d.gg_dh0 = g.gg_dh0; d.gg_dgi = g.gg_dgi; d.gg_dgh = g.gg_dgh;
}


int64_t pfo_bimg_bytes(int N, int K) { return (int64_t)pfo_ceil_div(K, 32) * 3 * pfo_align_up(N, BN) * 64; }

int pfo_bimg_launch(const PfoBimg* list, int n, hipStream_t stream) {
  PFO_REQUIRE(n >= 1 && n <= PFO_BIMG_MAX, "bad image count");
  BimgDev d;
  int64_t most = 0;
  for (int i = 0; i < n; ++i) {
    PFO_REQUIRE(list[i].src && list[i].dst && list[i].N > 0 && list[i].K > 0, "bad image problem");
    PFO_REQUIRE((((uintptr_t)list[i].dst) & 15) == 0, "image must be 16-byte aligned");
    d.src[i] = list[i].src; d.ld[i] = list[i].ld; d.N[i] = list[i].N; d.K[i] = list[i].K; d.trans[i] = list[i].trans;
    d.dst[i] = list[i].dst;
    d.row0[i] = list[i].row0;
    d.gate[i] = list[i].gate; d.gate_D[i] = list[i].gate_D;
    if (list[i].gate) {
      // gate-ordered: the image has pfo_gru_img_rows(D) rows, all written by this problem (zero rows included); N = 3 D source rows
      PFO_REQUIRE(list[i].gate_D > 0 && list[i].N == 3 * list[i].gate_D && !list[i].trans && list[i].rows_total == 0, "bad gate-ordered image");
      d.rows_total[i] = d.rows[i] = pfo_gru_img_rows(list[i].gate_D);
      most = std::max<int64_t>(most, (int64_t)pfo_ceil_div(list[i].K, 32) * d.rows[i] * 4);
      continue;
    }
    d.rows_total[i] = list[i].rows_total > 0 ? (int)pfo_align_up(list[i].rows_total, BN) : (int)pfo_align_up(list[i].N, BN);
    // rows this problem writes: a plain image and the last operand of a stack include the zero padding up to the padded end
    d.rows[i] = list[i].rows_total > 0 ? (list[i].last ? d.rows_total[i] - d.row0[i] : list[i].N) : d.rows_total[i];
    PFO_REQUIRE(d.row0[i] >= 0 && d.rows[i] >= list[i].N && d.row0[i] + d.rows[i] <= d.rows_total[i], "bad stacked image rows");
    most = std::max<int64_t>(most, (int64_t)pfo_ceil_div(list[i].K, 32) * d.rows[i] * 4);
  }
  if (pfo_bx_fmt()) {
    int most_rows = 0;
    bool any_trans = false;
    for (int i = 0; i < n; ++i) { most_rows = std::max(most_rows, d.rows[i]); any_trans = any_trans || d.trans[i] != 0; }
    if (!any_trans) PFO_KLAUNCH(bimg_h_rows_kernel, dim3((unsigned)pfo_ceil_div(most_rows, 4), n), dim3(256), 0, stream, d);
    else {
      PFO_KLAUNCH(bimg_exp_kernel, dim3((unsigned)pfo_ceil_div(most_rows, 4), n), dim3(256), 0, stream, d);
      PFO_KLAUNCH(bimg_h_kernel, dim3((unsigned)pfo_ceil_div(most, 256), n), dim3(256), 0, stream, d);
    }
  } else
    PFO_KLAUNCH(bimg_kernel, dim3((unsigned)pfo_ceil_div(most, 256), n), dim3(256), 0, stream, d);
  PFO_LAUNCH_CHECK();
  return PFO_OK;
}

bool pfo_gemm_takes_skinny(int M, int N) {
  static const int bx = getenv("PFO_GEMM_BF16X3") ? atoi(getenv("PFO_GEMM_BF16X3")) : PFO_DEFAULT_BF16X3;
  static const int bx_min_tiles = getenv("PFO_BX_MIN_TILES") ? atoi(getenv("PFO_BX_MIN_TILES")) : PFO_BX_MIN_TILES;
  static const bool forced = getenv("PFO_GEMM_TILE") != nullptr;
  static const int sk = getenv("PFO_GEMM_SKINNY") ? atoi(getenv("PFO_GEMM_SKINNY")) : 1;
  if (bx < 1 || forced || !sk) return false;
  return (int64_t)pfo_ceil_div(M, BM) * pfo_ceil_div(N, BN) < bx_min_tiles;
}
bool pfo_gemm_takes_bx(int M, int N) {
  static const int bx = getenv("PFO_GEMM_BF16X3") ? atoi(getenv("PFO_GEMM_BF16X3")) : PFO_DEFAULT_BF16X3;
  static const int bx_min_tiles = getenv("PFO_BX_MIN_TILES") ? atoi(getenv("PFO_BX_MIN_TILES")) : PFO_BX_MIN_TILES;
  static const bool forced = getenv("PFO_GEMM_TILE") != nullptr;
  static const int sk = getenv("PFO_GEMM_SKINNY") ? atoi(getenv("PFO_GEMM_SKINNY")) : 1;
  if (bx < 1 || forced) return false;
  return sk != 0 || (int64_t)pfo_ceil_div(M, BM) * pfo_ceil_div(N, BN) >= bx_min_tiles;
}

int pfo_gemm_launch(const PfoGemm& g, hipStream_t stream) {
  PFO_REQUIRE(g.M > 0 && g.N > 0 && g.K[0] > 0 && g.batch >= 1, "bad sizes");
  PFO_REQUIRE(g.A[0] && g.B[0] && g.C, "null operand");
  bool a_vec = true, b_vec = true;
  for (int s = 0; s < 2; ++s) {
    if (s == 1 && g.K[1] == 0) continue;
    PFO_REQUIRE(g.A[s] && g.B[s], "null operand (source 2)");
    a_vec = a_vec && aligned4(g.A[s]) && (g.lda[s] % 4) == 0 && (g.a_bs[s] % 4) == 0 &&
            (g.a_kmajor ? (g.M % 4) == 0 : (g.K[s] % 4) == 0);
    b_vec = b_vec && aligned4(g.B[s]) && (g.ldb[s] % 4) == 0 && (g.b_bs[s] % 4) == 0 &&
            (g.b_kmajor ? (g.N % 4) == 0 : (g.K[s] % 4) == 0);
  }
  GemmDev d;
  to_dev(g, d);
  const bool vec = a_vec && b_vec;
  const double flops = 2.0 * g.M * g.N * ((double)g.K[0] + g.K[1]) * g.batch;
  int kind = g.m_dev ? PFO_PROF_GEMM_DEVM : (g.a_kmajor ? PFO_PROF_GEMM_TN : (g.b_kmajor ? PFO_PROF_GEMM_NN : PFO_PROF_GEMM_NT));
  pfo_prof_begin(stream);
  const int tn = (int)pfo_ceil_div(g.N, BN);
#define GEMM_GO(AK, BK_, SM, grid)                                                                                   \
  do {                                                                                                                \
    if (vec) PFO_KLAUNCH((gemm_f32_kernel<AK, BK_, SM, true>), grid, dim3(GEMM_THREADS), 0, stream, d);        \
    else PFO_KLAUNCH((gemm_f32_kernel<AK, BK_, SM, false>), grid, dim3(GEMM_THREADS), 0, stream, d);           \
  } while (0)
  if (g.a_kmajor) {
    const int tm = (int)pfo_ceil_div(g.M, BM);
    PFO_REQUIRE(g.K[1] == 0, "k-major A supports one source");
    // weight gradient: few output tiles, long K -> split K over workgroups, deterministic slab reduce
    const int K = g.K[0];
    int want = (int)pfo_ceil_div(512, (int64_t)tm * tn * g.batch);
    int nsplit = (int)std::max<int64_t>(1, std::min<int64_t>(want, pfo_ceil_div(K, 4 * BK)));
    int chunk = (int)pfo_align_up(pfo_ceil_div(K, nsplit), BK);
    nsplit = (int)pfo_ceil_div(K, chunk);
    if (nsplit > 1 && g.b_kmajor && g.slabs) {
      // the weight-gradient form goes through the grouped split-K launch (one problem)
      PFO_REQUIRE(g.batch == 1, "split-K with batch is not supported");
      PFO_REQUIRE(!g.bias && !g.relu && !g.row_zero && !g.relu_src, "split-K takes no epilogue");
      PfoTnProblem q;
      q.A = g.A[0]; q.lda = g.lda[0]; q.B = g.B[0]; q.ldb = g.ldb[0]; q.b_idx = g.b_idx; q.M = g.M; q.N = g.N;
      q.C = g.C; q.ldc = g.ldc; q.c_accumulate = g.accumulate;
      return pfo_gemm_tn_group_launch(&q, 1, K, g.m_dev, g.slabs, g.slab_floats, stream);
    }
    if (g.b_kmajor) GEMM_GO(true, true, 0, dim3(tm, tn, g.batch)); else GEMM_GO(true, false, 0, dim3(tm, tn, g.batch));
  } else {
    // tile rows per workgroup: 128 (BIG) or 32 (SMALL: launches where 128-row tiles would leave most of the 256 CUs
    // without work)
    static const int force = getenv("PFO_GEMM_TILE") ? (atoi(getenv("PFO_GEMM_TILE")) != 0 ? 1 : 0) : -1;   // A/B switch: 0 BIG, 1 SMALL
    const int64_t big_tiles = (int64_t)pfo_ceil_div(g.M, BM) * tn * g.batch;
    int tile = big_tiles < 400 ? 1 : PFO_DEFAULT_TILE;
    if (force >= 0) tile = force;
    const int rows = tile == 1 ? 32 : BM;
    const dim3 grid((unsigned)pfo_ceil_div(g.M, rows), tn, g.batch);
    // bf16x3 split contraction: 1 = when the caller supplies the pre-split image of B, 2 = also for plain row-major
    // B (split in the kernel), 0 = never (fp32 MFMA everywhere).  A/B switch.
    static const int bx = getenv("PFO_GEMM_BF16X3") ? atoi(getenv("PFO_GEMM_BF16X3")) : PFO_DEFAULT_BF16X3;
    static const int bx_min_tiles = getenv("PFO_BX_MIN_TILES") ? atoi(getenv("PFO_BX_MIN_TILES")) : PFO_BX_MIN_TILES;
    const bool a_rowvec = a_vec && g.batch == 1;
    static const int sk = getenv("PFO_GEMM_SKINNY") ? atoi(getenv("PFO_GEMM_SKINNY")) : 1;                  // A/B switch
    if (g.b_img && (g.K[1] == 0 || g.b_img2) && a_rowvec && (g.bx_force == 2 || (bx >= 1 && sk && !g.bx_force && force < 0 && big_tiles < bx_min_tiles))) {
      d.b_img = g.b_img; d.b_img_rows = (int)pfo_align_up(g.N, BN); d.b_img2 = g.b_img2;
      kind = PFO_PROF_GEMM_BX_SKINNY;
      // few row tiles (< one per CU even with 176-column workgroups): 64-column workgroups fill the chip three times better
      static const int narrow = getenv("PFO_SKINNY_NARROW") ? atoi(getenv("PFO_SKINNY_NARROW")) : 512;        // A/B switch: workgroup threshold, 0 = never
      const int64_t sk_wgs = (int64_t)pfo_ceil_div(g.M, SK_ROWS) * tn;
      const dim3 g4((unsigned)pfo_ceil_div(g.M, SK_ROWS), (unsigned)pfo_ceil_div(g.N, 64), 1), g11((unsigned)pfo_ceil_div(g.M, SK_ROWS), tn, 1);
      if (sk_wgs < narrow) {
        if (pfo_bx_fmt()) PFO_KLAUNCH((gemm_bx_skinny_kernel<4, 1>), g4, dim3(GEMM_THREADS), 0, stream, d);
        else PFO_KLAUNCH((gemm_bx_skinny_kernel<4, 0>), g4, dim3(GEMM_THREADS), 0, stream, d);
      } else {
        if (pfo_bx_fmt()) PFO_KLAUNCH((gemm_bx_skinny_kernel<11, 1>), g11, dim3(GEMM_THREADS), 0, stream, d);
        else PFO_KLAUNCH((gemm_bx_skinny_kernel<11, 0>), g11, dim3(GEMM_THREADS), 0, stream, d);
      }
    } else if (g.b_img && (g.K[1] == 0 || g.b_img2) && a_rowvec && (g.bx_force || (bx >= 1 && big_tiles >= bx_min_tiles && force < 0))) {
      d.b_img = g.b_img; d.b_img_rows = (int)pfo_align_up(g.N, BN); d.b_img2 = g.b_img2;
      kind = PFO_PROF_GEMM_BX;
      PFO_REQUIRE(!g.gg_gates, "the GRU gate epilogue exists in the 32-row image kernel only (pfo_gemm_takes_skinny)");
      static const int areg = getenv("PFO_GEMM_AREG") ? atoi(getenv("PFO_GEMM_AREG")) : PFO_DEFAULT_AREG;    // A/B switch
      if (areg && g.batch == 1)
      {
        static const int xcd = getenv("PFO_GEMM_XCD") ? atoi(getenv("PFO_GEMM_XCD")) : 1;                  // A/B switch
        const int tmr = (int)pfo_ceil_div(g.M, BM);
        dim3 grid((unsigned)tmr, tn, 1);
        if (xcd && tn > 1) { d.xcd_tm = tmr; d.xcd_tn = (int)tn; grid = dim3((unsigned)(pfo_ceil_div(tmr, 8) * 8 * tn), 1, 1); }
        // (eight single-strip wavefronts from PFO_AREG8 workgroups on: 0 = never; below one workgroup per slot the four-wavefront form's
        //  fewer barriers win - QX / GRU shapes at 470 workgroups: 32.0 against 35.7 us)
        // short contraction, many columns, plain stores: the A-stationary form (gemm_bx_astat_kernel)
        bool as_done = false;
        static const int astat = getenv("PFO_ASTAT") ? atoi(getenv("PFO_ASTAT")) : PFO_DEFAULT_ASTAT;           // A/B switch
        if (astat && pfo_bx_fmt() && g.K[1] == 0 && g.K[0] <= AS_TMAX * BK && (g.N % 32) == 0 && g.N >= 352 && (g.ldc % 4) == 0 &&
            aligned4(g.C) && !g.bias && !g.relu && !g.relu_src && !g.add_src && !g.accumulate && !g.row_scale && !g.row_zero &&
            !g.gg_gates && g.M >= 128 * 256) {
          if (g.N <= AS_NMAX) {
            PFO_KLAUNCH(gemm_bx_astat_kernel, dim3((unsigned)pfo_ceil_div(g.M, 128)), dim3(GEMM_THREADS), 0, stream, d);
            as_done = true;
          }
        }
        static const int areg8_min = getenv("PFO_AREG8") ? atoi(getenv("PFO_AREG8")) : PFO_DEFAULT_AREG8;      // A/B switch
        const bool areg8 = areg8_min > 0 && big_tiles >= areg8_min;
        if (as_done) { }
        else if (pfo_bx_fmt() && areg8) PFO_KLAUNCH(gemm_bx_areg8_kernel<1>, grid, dim3(512), 0, stream, d);
        else if (pfo_bx_fmt()) PFO_KLAUNCH(gemm_bx_areg_kernel<1>, grid, dim3(GEMM_THREADS), 0, stream, d);
        else PFO_KLAUNCH(gemm_bx_areg_kernel<0>, grid, dim3(GEMM_THREADS), 0, stream, d);
      }
      else {
        PFO_REQUIRE(!pfo_bx_fmt(), "the fp16x2 images are read by the row-major-A kernels only (batch 1)");
        PFO_KLAUNCH(gemm_bf16x3_kernel<true>, dim3((unsigned)pfo_ceil_div(g.M, BM), tn, 1), dim3(GEMM_THREADS), 0,
                           stream, d);
      }
    } else if (g.K[1] > 0 && g.b_img2) {
      // the caller fused two sources whose float B operands may differ in layout: only the image kernels can take that
      pfo_set_error("pfo_gemm_launch: a two-source launch with weight images needs 16-byte aligned row-major A operands");
      return PFO_ERR_INVALID;
    } else if (bx >= 2 && !g.b_kmajor && tile == 0 && vec) {
      kind = PFO_PROF_GEMM_BX;
      PFO_KLAUNCH(gemm_bf16x3_kernel<false>, grid, dim3(GEMM_THREADS), 0, stream, d);
    } else if (g.b_kmajor) {
      if (tile == 1) GEMM_GO(false, true, 1, grid); else GEMM_GO(false, true, 0, grid);
    } else {
      if (tile == 1) GEMM_GO(false, false, 1, grid); else GEMM_GO(false, false, 0, grid);
    }
  }
#undef GEMM_GO
  PFO_LAUNCH_CHECK();
  // a device-side extent (rows, or the K of a k-major A) scales the work: read back when the records are collected
  if (g.m_dev) pfo_prof_end_dev(kind, flops / (double)(g.a_kmajor ? g.K[0] : g.M), g.m_dev, g.a_kmajor ? g.K[0] : g.M, stream);
  else pfo_prof_end(kind, flops, stream);
  return PFO_OK;
}

static bool gemm_vec_ok(const PfoGemm& g) {
  bool a_vec = true, b_vec = true;
  for (int s = 0; s < 2; ++s) {
    if (s == 1 && g.K[1] == 0) continue;
    a_vec = a_vec && aligned4(g.A[s]) && (g.lda[s] % 4) == 0 && (g.a_bs[s] % 4) == 0 &&
            (g.a_kmajor ? (g.M % 4) == 0 : (g.K[s] % 4) == 0);
    b_vec = b_vec && aligned4(g.B[s]) && (g.ldb[s] % 4) == 0 && (g.b_bs[s] % 4) == 0 &&
            (g.b_kmajor ? (g.N % 4) == 0 : (g.K[s] % 4) == 0);
  }
  return a_vec && b_vec;
}

int pfo_gemm_multi_launch(const PfoGemm* list, int n, hipStream_t stream) {
  PFO_REQUIRE(list && n >= 1, "bad arguments");
  for (int base = 0; base < n; base += MULTI_MAX) {
    const int cnt = std::min(MULTI_MAX, n - base);
    MultiDev g;
    memset(&g, 0, sizeof(g));
    bool vec = true;
    int tiles = 0;
    // the direct form (gemm_multi_direct_kernel): plain problems of one source - everything these launches are used for
    static const int direct_env = getenv("PFO_MULTI_DIRECT") ? atoi(getenv("PFO_MULTI_DIRECT")) : 0;      // A/B switch (off: 1.223-1.239 against 1.216-1.220 ms per step, profiles/r6_experiments.txt 17)
    bool direct = direct_env != 0;
    for (int i = 0; i < cnt; ++i) {
      const PfoGemm& s = list[base + i];
      direct = direct && s.K[1] == 0 && !s.a_idx[0] && !s.b_idx && !s.relu && !s.relu_src && !s.row_scale && !s.row_zero && !s.add_src &&
               !s.b_img && !s.gg_gates;
    }
    for (int i = 0; i < cnt; ++i) {
      const PfoGemm& s = list[base + i];
      PFO_REQUIRE(s.M > 0 && s.N > 0 && s.K[0] > 0 && s.A[0] && s.B[0] && s.C, "bad problem");
      PFO_REQUIRE(!s.m_dev && !s.slabs, "multi launch takes plain problems only");
      to_dev(s, g.p[i]);
      g.layout[i] = (s.a_kmajor ? 2 : 0) + (s.b_kmajor ? 1 : 0);
      g.tm[i] = (int)pfo_ceil_div(s.M, direct ? 16 : 32);
      g.tn[i] = (int)pfo_ceil_div(s.N, 64);            // gemm_tile's TINY shape: 32 x 64; the direct form: 16 x 64 (four 16 x 16 tiles)
      g.tile_begin[i] = tiles;
      tiles += g.tm[i] * g.tn[i] * s.batch;
      vec = vec && gemm_vec_ok(s);
    }
    g.n = cnt;
    double mflops = 0;
    for (int i = 0; i < cnt; ++i) mflops += 2.0 * list[base + i].M * list[base + i].N * (double)list[base + i].K[0] * list[base + i].batch;
    pfo_prof_begin(stream);
    static const int abl_multi = getenv("PFO_ABL_MULTI") ? atoi(getenv("PFO_ABL_MULTI")) : 0;   // timing-only ablation (wrong results): 1 = a single tile per launch
    if (abl_multi) tiles = 1;
    if (direct && vec) PFO_KLAUNCH(gemm_multi_direct_kernel<true>, dim3(tiles), dim3(GEMM_THREADS), 0, stream, g);
    else if (direct) PFO_KLAUNCH(gemm_multi_direct_kernel<false>, dim3(tiles), dim3(GEMM_THREADS), 0, stream, g);
    else if (vec) PFO_KLAUNCH(gemm_multi_kernel<true>, dim3(tiles), dim3(GEMM_THREADS), 0, stream, g);
    else PFO_KLAUNCH(gemm_multi_kernel<false>, dim3(tiles), dim3(GEMM_THREADS), 0, stream, g);
    PFO_LAUNCH_CHECK();
    pfo_prof_end(PFO_PROF_GEMM_MULTI, mflops, stream);
  }
  return PFO_OK;
}


// ---------------------------------------------------------------------------------------------
extern "C" int pfo_gemm_f32(const float* A, int64_t lda, int32_t a_kmajor, const float* B, int64_t ldb,
                            int32_t b_kmajor, float* C, int64_t ldc, const float* bias, int32_t M, int32_t N, int32_t K,
                            int32_t relu, float* workspace, int64_t workspace_floats, void* stream) {
  PfoGemm g;
  g.A[0] = A; g.lda[0] = lda; g.B[0] = B; g.ldb[0] = ldb; g.K[0] = K;
  g.C = C; g.ldc = ldc; g.bias = bias; g.M = M; g.N = N; g.relu = relu;
  g.a_kmajor = a_kmajor; g.b_kmajor = b_kmajor;
  g.slabs = workspace; g.slab_floats = workspace_floats;
  if (a_kmajor && bias) { pfo_set_error("pfo_gemm_f32: k-major A takes no bias"); return PFO_ERR_INVALID; }
  return pfo_gemm_launch(g, (hipStream_t)stream);
}

extern "C" int64_t pfo_gemm_bf16x3_workspace_bytes(int32_t N, int32_t K) { return pfo_bimg_bytes(N, K); }

extern "C" int pfo_gemm_bf16x3(const float* A, int64_t lda, const float* B, int64_t ldb, int32_t b_kmajor, float* C,
                               int64_t ldc, const float* bias, int32_t M, int32_t N, int32_t K, int32_t relu,
                               void* workspace, int64_t workspace_bytes, void* stream) {
  PFO_REQUIRE(A && B && C && workspace, "null operand");
  PFO_REQUIRE(M > 0 && N > 0 && K > 0, "bad sizes");
  PFO_REQUIRE(workspace_bytes >= pfo_bimg_bytes(N, K), "workspace too small for the image of B");
  PFO_REQUIRE(aligned4(A) && (lda % 4) == 0 && (K % 4) == 0, "A must be 16-byte aligned with lda and K multiples of 4");
  PfoBimg im;
  im.src = B; im.ld = ldb; im.N = N; im.K = K; im.trans = b_kmajor ? 1 : 0; im.dst = workspace;
  if (int rc = pfo_bimg_launch(&im, 1, (hipStream_t)stream)) return rc;
  PfoGemm g;
  g.A[0] = A; g.lda[0] = lda; g.B[0] = B; g.ldb[0] = ldb; g.K[0] = K; g.C = C; g.ldc = ldc; g.bias = bias;
  g.M = M; g.N = N; g.relu = relu; g.b_kmajor = b_kmajor ? 1 : 0; g.b_img = workspace;
  g.bx_force = M <= 4096 ? 2 : 1;            // short operands take the 32-row kernel, as in the TGN step
  return pfo_gemm_launch(g, (hipStream_t)stream);
}
