// Internal interface of the fp32 MFMA contraction (gemm.hip).
#pragma once
#include "common.hpp"

// C[M,N] = epilogue( sum over up to two K-concatenated sources  A_s[M,K_s] * op(B_s) )
//   A_s row-major [M,K_s] (a_kmajor=0, optional row gather a_idx) or [K_s,M] (a_kmajor=1)
//   B_s [N,K_s] (b_kmajor=0, nn.Linear layout) or [K_s,N] (b_kmajor=1, optional k-row gather b_idx)
// epilogue order: (+= C)  ->  + bias[n]*row_scale[m]  ->  + add_src[add_idx[m]][n]  ->  row_zero  ->  relu  ->  * (relu_src>0)
struct PfoGemm {
  const float* A[2] = {nullptr, nullptr};
  int64_t lda[2] = {0, 0};
  const int32_t* a_idx[2] = {nullptr, nullptr};
  const float* B[2] = {nullptr, nullptr};
  int64_t ldb[2] = {0, 0};
  const int32_t* b_idx = nullptr;   // k-major B only: gathers the k rows of source 0
  int K[2] = {0, 0};
  float* C = nullptr;
  int64_t ldc = 0;
  const float* bias = nullptr;
  const float* row_scale = nullptr; int64_t rs_ld = 1;
  const uint8_t* row_zero = nullptr;
  const float* relu_src = nullptr; int64_t relu_ld = 0;
  // row-gathered addend (bf16x3 image kernels only): C[m][n] += add_src[add_idx ? add_idx[m] : m][n]
  const float* add_src = nullptr; int64_t add_ld = 0; const int32_t* add_idx = nullptr;
  int M = 0, N = 0;
  const int32_t* m_dev = nullptr;   // device-side row count (rows M for row-major A, extent K for k-major A)
  int relu = 0, accumulate = 0;
  int a_kmajor = 0, b_kmajor = 0;
  // batch (grid.z), element strides
  int batch = 1;
  int64_t a_bs[2] = {0, 0}, b_bs[2] = {0, 0}, c_bs = 0, bias_bs = 0, rs_bs = 0;
  // split-K for a_kmajor && b_kmajor (weight gradients): partial slabs in `slabs`, then reduced into C
  float* slabs = nullptr; int64_t slab_floats = 0;
  // optional pre-split bf16x3 image of op(B) for source 0 (pfo_bimg_launch); used by the large row-major launches
  const void* b_img = nullptr;
  const void* b_img2 = nullptr;     // image of the second source's B (two K-concatenated sources in one bf16x3 launch)
  int bx_force = 0;                 // tests: 1 = the 128-row bf16x3 kernel, 2 = the 32-row one, whatever the heuristic says
  // GRU gate backward as the EPILOGUE (32-row image kernel only, pfo_gemm_takes_skinny): the contraction's result is the
  // query-side gradient of the touched rows' level-0 features (dx_tab); instead of storing it, row m / hidden unit d gets
  //   dh = C[m][d] + gg_dh0[m][d];  the GRUCell gate derivatives of memory_updater.py:18-61 via autograd (memory.hip
  //   gru_gates_bwd_vec_kernel has the same arithmetic)  ->  gg_dgi[m][3N], gg_dgh[m][3N]     (N = the hidden size)
  // C is not written.  gg_gates [M, 4N] = r | z | n | gh_n kept by the forward, gg_h [M, N] the packed memory rows,
  // gg_hm u8[M] "the row had a pending message" (0: zero gradients).
  const float* gg_gates = nullptr; const float* gg_h = nullptr; const uint8_t* gg_hm = nullptr; const float* gg_dh0 = nullptr;
  float* gg_dgi = nullptr; float* gg_dgh = nullptr;
};
bool pfo_gemm_takes_skinny(int M, int N);   // a launch of this shape with weight images takes the 32-row kernel

// Pre-split ("bf16x3") image of a weight operand W(n, k) = src[n*ld + k] (trans = 0) or src[k*ld + n] (trans = 1):
// three bf16 pieces per element in the LDS layout of the split contraction kernel.  dst needs pfo_bimg_bytes(N, K).
#define PFO_BIMG_MAX 24
struct PfoBimg {
  const float* src = nullptr; int64_t ld = 0; int N = 0, K = 0, trans = 0; void* dst = nullptr;
  // an image whose rows come from several operands stacked along n (e.g. [Wqk ; W1[:, E:]]): this problem fills image rows
  // [row0, row0 + N) of an image of rows_total rows (padded to the tile width); the LAST operand of a stack sets `last`
  // and also writes the zero padding rows up to the padded end.  rows_total = 0: a plain single-operand image
  int row0 = 0, rows_total = 0, last = 0;
  // GRU gate-ordered image for pfo_gru_fused_launch: src = the stacked [3D, K] weight (N = 3 D), gate = 1 (W_ih) or 2 (W_hh),
  // gate_D = D; dst needs pfo_gru_img_bytes(D, K)
  int gate = 0, gate_D = 0;
};
int pfo_gru_img_rows(int D);
int64_t pfo_gru_img_bytes(int D, int K);

// The lazy GRU of cap_rows (device count n_rows) packed rows in one launch: [msg_rows | h_rows] against the gate-ordered images,
// gates in the epilogue.  upd_mem = h' (or h where hm == 0), h0_tab = h' + node_feat[touched], gates [rows, 4D] = r | z | n | gh_n.
struct PfoGruFused {
  const float* msg_rows = nullptr; int K_msg = 0;     // [rows, K_msg]
  const float* h_rows = nullptr;                      // [rows, D]
  const void* img_ih = nullptr; const void* img_hh = nullptr;
  const float* b_ih = nullptr; const float* b_hh = nullptr;
  const uint8_t* hm = nullptr; const int32_t* touched = nullptr; const float* node_feat = nullptr;
  float* upd_mem = nullptr; float* h0_tab = nullptr; float* gates = nullptr;
  int D = 0, cap_rows = 0; const int32_t* n_rows = nullptr;
  // gather = 1: msg_rows / h_rows / hm are the FULL per-node tables (message table, memory, has_msg) and row m of the launch is
  // node touched[m] of them - the packed copies the backward needs are then made off the critical path
  int gather = 0;
};
int pfo_gru_fused_launch(const PfoGruFused& f, hipStream_t stream);
int64_t pfo_bimg_bytes(int N, int K);
int pfo_bimg_launch(const PfoBimg* list, int n, hipStream_t stream);

int pfo_gemm_launch(const PfoGemm& g, hipStream_t stream);
// true when a row-major launch of this size (aligned operands, images supplied) takes a bf16x3 kernel: those accept two
// K-concatenated sources whose B operands differ in layout
bool pfo_gemm_takes_bx(int M, int N);
// several small plain problems (any operand layouts, no device-side counts, no split-K) in one launch
#define PFO_GEMM_MULTI_MAX 10
int pfo_gemm_multi_launch(const PfoGemm* list, int n, hipStream_t stream);
// out[m, n] += u[m * ldu] * v[n * ldv]
// several updates out += sum_{r < reps} u_r (x) v_r in one launch (u_r = u + r * u_rs, v_r = v + r * v_rs); the outputs must not overlap
#define PFO_RANK1_MAX 12
struct PfoRank1 {
  const float* u = nullptr; int64_t ldu = 1; const float* v = nullptr; int64_t ldv = 1; int M = 0, N = 0; float* out = nullptr; int64_t ldo = 0;
  int reps = 1; int64_t u_rs = 0, v_rs = 0;
};
int pfo_rank1_multi_launch(const PfoRank1* list, int n, hipStream_t stream);
// dst[i] (+)= sum_{s < n_slabs} src[s * stride + i], i < count; several in one launch
#define PFO_SUM_SLABS_MAX 4
struct PfoSumSlabs {
  float* dst = nullptr; const float* src = nullptr; int64_t stride = 0, count = 0; int n_slabs = 1, accumulate = 1;
};
int pfo_sum_slabs_launch(const PfoSumSlabs* list, int n, hipStream_t stream);
int pfo_rank1_launch(const float* u, int64_t ldu, const float* v, int64_t ldv, int M, int N, float* out, int64_t ldo,
                     hipStream_t stream);

// One weight gradient dW[M,N] += A[K,M]^T B[K,N] (A = dY, B = X, both row-major over the K instance rows) with an
// optional bias gradient bias_out[M] (+)= sum_k A[k][m] carried as column N.
struct PfoTnProblem {
  const float* A = nullptr; int64_t lda = 0;
  const float* B = nullptr; int64_t ldb = 0; const int32_t* b_idx = nullptr;
  int M = 0, N = 0;
  float* C = nullptr; int64_t ldc = 0; int c_accumulate = 1;
  float* bias_out = nullptr; int bias_accumulate = 1;
};
// all problems share the K extent (and its optional device-side bound); one GEMM launch + one reduce launch
int pfo_gemm_tn_group_launch(const PfoTnProblem* probs, int n, int K, const int32_t* k_dev, float* slabs,
                             int64_t slab_floats, hipStream_t stream);
