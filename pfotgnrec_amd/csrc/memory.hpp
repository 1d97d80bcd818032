// Internal interface of the memory-side kernels (memory.hip) and small ops (misc.hip).
#pragma once
#include "common.hpp"

// --- touched-node compaction: slot[v] = row of v in the per-step tables, -1 when the step does not touch v.  mark[] holds the
// flags (1 = referenced by the step's levels: set by the sampler's `mark` output when `marked`, else from nodes0 here; 2 = named
// only by `extra`), slot[] / touched_ids[] the result: class-1 nodes in id order first (n_counts[1] of them), class-2 nodes
// behind (n_counts[0] = all).  marks_are_zero: the caller has cleared mark[0, n_nodes) and scratch[0, 2 n_nodes / 1024) in
// stream order.
int64_t pfo_compact_scratch_ints(int n_nodes);
int pfo_touch_compact_launch(const int32_t* nodes0, int64_t n0, const int32_t* extra, int64_t n_extra, int n_nodes, int32_t* mark,
                             int32_t* slot, int32_t* touched_ids, int32_t* n_counts, int32_t* scratch, bool marks_are_zero,
                             bool marked, hipStream_t stream);
// packs the rows backward still needs after the state update overwrites them (pfo_pack_remap_launch):
//   msg_rows[s] = msg_table[id], h_rows[s] = memory[id], hm[s] = has_msg[id]   (id = touched_ids[s])
int pfo_remap_launch(const int32_t* nodes0, int64_t n0, const int32_t* slot, int32_t* idx0, hipStream_t stream);
// both in one launch (msg_table == null: only `memory` rows are copied, e.g. node features of a no-memory model)
int pfo_pack_remap_launch(const float* msg_table, int M, const float* memory, int D, const uint8_t* has_msg,
                          const int32_t* touched_ids, const int32_t* n_touched, int cap, float* msg_rows, float* h_rows,
                          uint8_t* hm, const int32_t* nodes0, int64_t n0, const int32_t* slot, int32_t* idx0, hipStream_t stream);

// --- GRU gates, backward (memory_updater.py:60; torch.nn.GRUCell gate order r, z, n).  The forward is gemm.hpp
// pfo_gru_fused_launch (contractions + gates in one launch; it keeps gates[s] = r | z | n | gh_n).  Writes d gi / d gh [rows, 3D]
// given d h' = sum over the n_rep replicas of d_h0[s] (+ d_extra: the rows' query-side gradient, layer 1); zeros where no
// message was applied.  det: d_h0 is ONE table of int64 fixed-point sums (attn.hpp PFO_DET_SCALE)
int pfo_gru_gates_bwd_launch(const float* gates, float* dgi, float* dgh, const float* h_rows, const uint8_t* hm,
                             const int32_t* n_touched, int cap, int D, const float* d_h0, int n_rep, int64_t rep_stride,
                             const float* d_extra, int det, hipStream_t stream);
// dst[s] = src[touched_ids[s]], s < *n_touched
int pfo_gather_rows_launch(const float* src, int D, const int32_t* touched_ids, const int32_t* n_touched, int cap, float* dst,
                           hipStream_t stream);
// exclusive scan of n ints (n < 2^31; scratch: n / 1024 + 16 ints); in == out is allowed
int pfo_iscan_launch(const int32_t* in, int64_t n, int32_t* out, int32_t* scratch, hipStream_t stream);
// --- instances grouped by the touched-table row they read: seg_ptr[cap_rows + 1], members[N] (ascending inside a group)
int64_t pfo_seg_scratch_ints(int cap_rows);
// inside a group the members are ordered by (key_src[instance], instance): key = the number of row entries before the
// instance's time (equal keys <=> identical most-recent neighbour lists), or the instance index when key_src is null
int pfo_seg_build_launch(const int32_t* idx, const int32_t* nodes, int N, int cap_rows, const int32_t* key_src,
                         int32_t* seg_ptr, int32_t* cursor, int32_t* tmp, int32_t* members, int32_t* seg_of, int32_t* scratch,
                         hipStream_t stream);
// (seg_of, optional: pfo_seg_of_ints(N) ints - seg_of[m] = the group of member position m, for pfo_segsum_launch)
int64_t pfo_seg_of_ints(int64_t n_members);
// out[s] = [ sum_{n in group s} src0[n] | sum_{n in group s} src1[n] ]  (row widths W0, W1; s < *n_rows).
// src0_by_position: src0's rows are stored in MEMBER order (row m belongs to members[m]) - contiguous per group
// seg_of (optional, with cap_members = the most members there can be): the work is cut by members instead of by groups
int pfo_segsum_launch(const float* src0, int W0, const float* src1, int W1, const int32_t* seg_ptr, const int32_t* members,
                      const int32_t* seg_of, int64_t cap_members, const int32_t* n_rows, int cap_rows, int src0_by_position,
                      const uint8_t* src0_live, float* out, hipStream_t stream);
int pfo_segsum_cols_launch(const float* src, int W, const int32_t* seg_ptr, const int32_t* members, const int32_t* seg_of,
                           int64_t cap_members, const int32_t* n_rows, float* out, int64_t out_ld, hipStream_t stream);
// (src0_live: optional byte flags per position; rows flagged 0 are not read)
int pfo_zero_rows_launch(float* dst, const int32_t* n_rows, int cap_rows, int D, int n_rep, int64_t rep_stride, hipStream_t stream);

// --- state update (tgn.py:290-317)
int pfo_persist_launch(const int32_t* src, const int32_t* dst, int B, const int32_t* slot, const float* upd_mem,
                       const uint8_t* has_msg, const float* msg_time, float* memory, float* last_update, int D, int32_t* winner,
                       hipStream_t stream);
// winner (i32[n_nodes]) is only used when pfo_msg_store_needs_winner(B): persist then resets it, the store takes an atomicMax pass
bool pfo_msg_store_needs_winner(int B);
int pfo_msg_store_launch(const int32_t* src, const int32_t* dst, const double* ts, const int32_t* eidx, int B,
                         const float* memory, const float* last_update, const float* edge_feat, const float* tw,
                         const float* tb, int D, int Ef, float* msg_table, float* msg_time, uint8_t* has_msg,
                         int32_t* winner, hipStream_t stream);

// --- small ops (misc.hip)
// out[0] = mean(src[0 .. n)), one wavefront, fixed order
int pfo_mean_launch(const float* src, int64_t n, float* out, hipStream_t stream);
// cq = Wq[:, D:2D] cos(b) + bq folded query bias: backward of that term
//   gq[E] = colsum(dQ);  d bq += gq;  d Wq[:, D:] += gq (x) cosb;  d tb += -sin(tb) * (Wq[:, D:]^T gq)
// tb_part != null: the time-bias term is STORED there instead of added to d_tb (a partial launch; a later launch passes it as
// tb_add).  dtime != null (final launch only): the step's fp64 time-encoder partial sums [n_bins][2D] are folded into
// d_tw / d_tb by the same launch (fixed order).
int pfo_cq_backward_launch(const float* const* gq, const float* const* Wq, int n_layers, const float* tb, int D, float* const* d_bq,
                           float* const* d_Wq, float* d_tb, float* tb_part, const float* tb_add, const double* dtime, int n_bins,
                           float* d_tw, hipStream_t stream);
// scratch: pfo_fold_parts_scratch_doubles(n) doubles; tickets: 64 ints, zero before the first use (self-resetting)
int64_t pfo_fold_parts_scratch_doubles(int n);
int pfo_fold_parts_launch(const double* parts, int n_parts, int n, float* out, int accumulate, double* scratch, int* tickets,
                          hipStream_t stream, double* out64 = nullptr);
// (out64 != null: the sums are stored there as doubles instead of being cast into `out`)
