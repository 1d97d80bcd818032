// Internal interface of the neighbour-tile attention kernels (attn.hip).
#pragma once
#include "common.hpp"

// One attention layer's neighbour side for N instances with K neighbour slots each.
// key_j = [ nbr_tab[row_j] (D) | edge_feat[eidx_j] (Ef) | cos(fma(dt_j, w, b)) (D) ]   (temporal_attention.py:52)
// scores use the FOLDED query-key vector qk_h = Wk_h^T Q_h: score_jh = scale * qk_h . key_j
// Replicas of the level-0 gradient table the layer-1 backward adds into (replica = XCD id & (n - 1)): with one table the
// float atomics of all eight XCDs make every cache line migrate between the hardware-coherent L2s.  Power of two;
// measured at C2: 1 -> 0.537 ms, 2 -> 0.510, 4 -> 0.506, 8 -> 0.498 for the kernel, best step time at 4 (replicas are
// zeroed before and summed after).
#ifndef PFO_GRAD_REPLICAS
#define PFO_GRAD_REPLICAS 4
#endif

#define ATTN_TIME_BINS 64          // power of two

struct PfoAttn {
  int N = 0, K = 0, D = 0, Ef = 0, H = 0;
  int Cp = 0;                       // per-head row stride of QK / ctx / dctx / dQK: C = 2D+Ef feature columns, column C =
                                    // sum_j a'_jh, column C+1 = valid flag on head 0, zero padding up to a multiple of 4
  const float* QK = nullptr;        // [N, H*Cp], or rows of qk_ld floats picked by qk_row
  const int32_t* qk_row = nullptr;  // [N] row of QK per instance (layer 1: the touched-node table, several instances share a row); null: n
  int64_t qk_ld = 0;                // row stride of QK in floats (0: H*Cp)
  const float* nbr_tab = nullptr;   // rows of D floats
  int64_t nbr_ld = 0;
  const int32_t* nbr_row = nullptr; // [N*K] row of nbr_tab per slot, or null: row = nbr_row_base + n*K + j
  int64_t nbr_row_base = 0;
  int64_t nbr_rows = 0, edge_rows = 0;   // rows of nbr_tab / edge_feat (validated: the kernels use 32-bit element offsets)
  int nbr_relu = 0;                 // backward, plain-stored neighbour gradients: rows of nbr_tab are ReLU outputs, d row *= (row > 0)
  const int32_t* nbr_ids = nullptr; // [N*K] node ids; slot is padding iff id == 0 (embedding_module.py:154)
  const float* edge_feat = nullptr; // [E+1, Ef]
  const int32_t* eidx = nullptr;    // [N*K]
  const float* dt = nullptr;        // [N*K]
  const float* tw = nullptr;        // time-encoder weight [D]
  const float* tb = nullptr;        // time-encoder bias [D]
  float scale = 1.f;
  float dropout_p = 0.f;
  uint64_t seed = 0, offset = 0;
  const uint64_t* offset_dev = nullptr;   // optional device word added to offset (graph-captured steps)
  const uint8_t* keep_inject = nullptr;   // optional [N, K] injected dropout decisions (bit h = head h kept) instead of the Philox draws
  // forward outputs / backward inputs
  float* ctx = nullptr;             // [N, H*Cp] sum_j a'_jh key_j (+ the two extra columns)
  float* attw = nullptr;            // [N, H, K] softmax probabilities before dropout (0 on padding)
  uint8_t* inv = nullptr;           // [N]       1 = no valid neighbour (temporal_attention.py:60)
  // backward
  const float* dctx = nullptr;      // [N, H*Cp]
  float* dQK = nullptr;             // [N, H*Cp]
  float* d_nbr = nullptr;           // rows of D floats: direct rows (nbr_row == null) or atomically added rows
  int64_t d_nbr_ld = 0;
  int64_t d_nbr_rep = 0;            // atomically added rows: floats between the per-XCD replicas of the table (0 = a single table)
  int d_nbr_nrep = 1;               // replicas in use (power of two <= PFO_GRAD_REPLICAS): XCD x adds into replica x & (n - 1)
  double* dtime_part = nullptr;     // [ATTN_TIME_BINS, 2*D] fp64 accumulators (dw | db) of the time encoder: ADDED to (zero them per step)
  // deterministic mode (pfo_tgn_batch.deterministic): the time-encoder partials go to dtime_slab, one row of 2*D doubles per
  // workgroup (pfo_attn_bwd_det_parts(N) rows, every row written), and atomically added neighbour rows are int64 fixed point
  // (PFO_DET_SCALE) in a table of d_nbr_ld int64 per row at d_nbr
  int det = 0;
  double* dtime_slab = nullptr;
  // run-merged kernel, non-deterministic launches: the query-side gradient rows are ADDED to the per-table-row sums directly
  // (float atomics into dq_rows[qk_row[n] * dq_ld + column], cleared by the caller) instead of being stored per member for
  // pfo_segsum_launch - that pass and its 94 us on the serial tail of the step disappear.  dQK / dqk_live are then unused.
  float* dq_rows = nullptr;
  int64_t dq_ld = 0;
  uint8_t* dqk_live = nullptr;      // run-merged kernel (members given): [members] flags of the dQK rows that hold a sum - consecutive
                                    // members on one table row are summed on chip and stored once (pfo_segsum_launch src0_live)
  // optional (layer 1 over the touched-node table, atomically added rows, most-recent sampling): the instances ordered by
  // (table row, entries before the instance's time) as built by pfo_seg_build_launch with key_src = run_cnt.  Consecutive
  // members of a row have neighbour lists that are shifts of each other, so their key-side gradients are summed on chip by
  // row entry and added once per group.
  const int32_t* members = nullptr; // [seg_ptr[*n_rows]]
  const int32_t* seg_ptr = nullptr; // [rows + 1]
  const int32_t* n_rows = nullptr;  // device-side row count
  const int32_t* run_cnt = nullptr; // [N] entries of the instance's row before its time (sampler.hip out_cnt): members of one
                                    // row whose counts differ by d have neighbour lists shifted by d slots
};

int pfo_attn_fwd_launch(const PfoAttn& a, hipStream_t stream);
// *n_parts receives ATTN_TIME_BINS (rows of dtime_part that may hold contributions)
int pfo_attn_bwd_launch(const PfoAttn& a, int* n_parts, hipStream_t stream);
int pfo_attn_bwd_max_parts();
int64_t pfo_attn_bwd_det_parts(int64_t N);   // slab rows a deterministic backward launch over N instances writes (all of them)
#define PFO_DET_SCALE 1099511627776.0       // 2^40: level-0 gradient rows as fixed point, resolution 9e-13, range +-8e6
// true when pfo_attn_bwd_launch will take the run-merged kernel for `a`: dQK row m then belongs to the m-th MEMBER
// (a.members[m]), not to instance m
bool pfo_attn_bwd_uses_runs(const PfoAttn& a);
bool pfo_attn_bwd_runs_possible(int K, int D, int H);   // the switch and the shape limits alone (known before the launch is described)
