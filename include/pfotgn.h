/*
 * pfotgn.h - C ABI of the MI355X-native TGN-recommender training path.
 *
 * Drop-in boundary for the hot path of youngandbin/PfoTGNRec (reference is pure
 * Python; citations are file:line under /root/reference).  Every entry point takes
 * plain DEVICE pointers (unless marked host), sizes and a hipStream_t passed as
 * void*; no torch / C++ types cross this boundary.  All kernels are gfx950 HIP.
 *
 * Conventions
 *   - return value: 0 = ok, <0 = error (pfo_last_error() gives the message, per thread)
 *   - node / edge ids on device are int32 (node 0 and edge 0 are padding)
 *   - all floating-point tensors are fp32 row-major unless stated (timestamps fp64)
 *   - nothing here allocates, frees or synchronises: callers own every buffer
 */
#ifndef PFOTGN_H
#define PFOTGN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PFO_OK 0
#define PFO_ERR_INVALID (-1)
#define PFO_ERR_HIP (-2)

#define PFO_MAX_NEIGHBORS 64 /* K: one wavefront lane per neighbour slot */
#define PFO_MAX_LAYERS 4
#define PFO_MAX_HEADS 8

int pfo_abi_version(void);
const char* pfo_last_error(void);

/* ------------------------------------------------------------------------------------------
 * K1  temporal neighbour lookup over a time-sorted CSR.
 * Replaces NeighborFinder.find_before + get_temporal_neighbor (utils/utils.py:150-219).
 *
 *   indptr  i64[n_nodes+1]; adj_nbr i32[nnz]; adj_eidx i32[nnz]; adj_ts f64[nnz] (per node ascending,
 *   ties in edge order - what utils/utils.py:139 `sorted(key=ts)` yields).
 *   Query i = (q_nodes[i], q_ts[i]); entries with ts STRICTLY < q_ts qualify (searchsorted side='left').
 *   mode 0: most-recent K, right-aligned, left-padded with (0,0,0.0)              (:206-218)
 *   mode 1: uniform with INJECTED draws  draws i64[n_q,K] in [0, #qualifying)     (:194)
 *   mode 2: uniform with counter-based Philox draws (seed, stream offset)
 *   uniform modes re-sort each row by f32 time, STABLE (tie policy SURVEY App. A-9)   (:201-204)
 *   K == 0 is treated as one all-padding column by the host mirror (:175); here K >= 1.
 * Outputs (any may be NULL): out_nbr/out_eidx i32[n_q,K], out_et f32[n_q,K] (edge time cast to f32),
 *   out_dt f32[n_q,K] = f32( q_ts - f64(out_et) )  (embedding_module.py:133-135).
 * Frontier expansion (optional, next_nodes != NULL): writes the next recursion level
 *   next_nodes[0:n_q] = q_nodes, next_nodes[n_q + i*K + j] = nbr[i][j]; next_ts (optional) likewise, q_ts[i] repeated
 *   (embedding_module.py:115,141-145: neighbours are evaluated at the ROOT's timestamp).
 */
int pfo_tnbr_sample(const int64_t* indptr, const int32_t* adj_nbr, const int32_t* adj_eidx, const double* adj_ts,
                    int64_t n_nodes, const int32_t* q_nodes, const double* q_ts, int64_t n_q, int32_t K,
                    int32_t mode, const int64_t* draws, uint64_t seed, uint64_t offset,
                    int32_t* out_nbr, int32_t* out_eidx, float* out_et, float* out_dt,
                    int32_t* next_nodes, double* next_ts, void* stream);

/* ------------------------------------------------------------------------------------------
 * Candidate-negative draw.  Replaces RandEdgeSampler.sample (utils/utils.py:86-114).
 *   item_avail u8[n_items]: 1 if the item occurs among the train destinations (np.unique(dst_list), :73)
 *   port_idx i32[B,port_stride] 0-based item indices, port_len i32[B] (the '' entry is already dropped, :76)
 *   out i32[B,size] = item NODE ids (index + upper_u + 1), drawn without replacement from
 *   avail \ portfolio when that set has >= size members, otherwise with replacement (:99-111).
 *   RNG: Philox(seed, offset + interaction) - semantics-level parity only.
 */
int pfo_neg_draw(const uint8_t* item_avail, int32_t n_items, const int32_t* port_idx, const int32_t* port_len,
                 int32_t port_stride, int64_t B, int32_t size, int32_t upper_u, uint64_t seed, uint64_t offset,
                 int32_t* out, void* stream);
/* the same with a device word added to `offset` (graph-captured steps, see pfo_tgn_batch.offset_dev) */
int pfo_neg_draw_dev(const uint8_t* item_avail, int32_t n_items, const int32_t* port_idx, const int32_t* port_len,
                     int32_t port_stride, int64_t B, int32_t size, int32_t upper_u, uint64_t seed, uint64_t offset,
                     const uint64_t* offset_dev, int32_t* out, void* stream);

/* ------------------------------------------------------------------------------------------
 * K2  mean-variance-efficient rank fusion.  Replaces the inline block main.py:209-304.
 *   returns f64[n_days,n_items,n_ret]: log-returns log(p[t+1]/p[t]) (main.py:218,226-227; n_ret = 29)
 *   day_idx i32[B]; cand i32[B,n_cand] item NODE ids, column 0 = true destination (main.py:207)
 *   y_mv = (mu/gamma - 0.5*mean_j cov_ij)/cov_ii, or (mu/gamma)/var_i for an empty portfolio (main.py:243-271)
 *   invest_rank = average-tie rank of y_mv; tgn_rank = n..1; new = lam*invest + (1-lam)*tgn (main.py:282-286)
 *   order = stable ascending argsort of new, reversed (tie policy SURVEY App. A-9; main.py:289)
 *   p_pos i32[B,n_pos] = first n_pos of order, p_neg i32[B,n_neg] = last n_neg (main.py:291-292), as node ids.
 *   y_out f64[B,n_cand] and rank_out f64[B,n_cand] (new_rank) are optional diagnostics.
 */
int pfo_mv_select(const double* returns, int32_t n_days, int32_t n_items, int32_t n_ret, const int32_t* day_idx,
                  const int32_t* cand, int32_t n_cand, const int32_t* port_idx, const int32_t* port_len,
                  int32_t port_stride, int64_t B, int32_t upper_u, double gamma, double lambda_mv,
                  int32_t n_pos, int32_t n_neg, int32_t* p_pos, int32_t* p_neg, double* y_out, double* rank_out,
                  void* stream);

/* ------------------------------------------------------------------------------------------
 * TimeEncode forward: out[i,d] = cos(fma(t[i], w[d], b[d]))   (model/time_encoding.py:17-25;
 * single fp32 FMA then a full-range cosine - SURVEY §7 hard part 1).
 */
int pfo_time_encode(const float* t, int64_t n, const float* w, const float* b, int32_t D, float* out, void* stream);

/* The train-mode dropout multipliers of one attention layer, exactly as the step's kernels draw them
 * (nn.MultiheadAttention(dropout=p) on the softmax weights, temporal_attention.py:28,70): out[n, h, j] = 1/(1-p) where the
 * weight of key slot j / head h of instance n is kept, 0 where it is dropped.  Philox4x32 keyed by `seed`, counter
 * (n * 64 + j, offset): the step uses offset = pfo_tgn_batch.offset + 0x51ED0000 + layer (1-based), n = the instance's index
 * in the layer's level list.  Test / parity infrastructure: lets an oracle replay a dropout-0.1 step with the SAME masks
 * (the RNG stream itself is not part of the reference's contract). */
int pfo_attn_dropout_mask(uint64_t seed, uint64_t offset, int64_t N, int32_t K, int32_t H, float p, float* out /* [N,H,K] */,
                          void* stream);

/* ------------------------------------------------------------------------------------------
 * Gradient rows of the instances that sit on one node, summed per node: the backward of the reference's reuse of one
 * level-0 row [memory + features] by every instance of that node (embedding_module.py:93-98, the index by node id there;
 * autograd's index backward is this sum).  Exposed for tests; the step calls it between layer 1's attention backward and
 * the touched-table contraction.
 *   out[s, :] = sum over m in [seg_ptr[s], seg_ptr[s+1]) of [ src0[p(m), 0:W0] | src1[members[m], 0:W1] ],  s < *n_rows
 *   p(m) = m (src0_by_position != 0: rows stored in member order) or members[m]
 *   src0_live (optional, with src0_by_position): byte per position, 0 = that src0 row holds nothing and is not read
 *   seg_of (optional): int32 per member position, the segment it belongs to, in an array of at least
 *   (n_members / 16 + 2) * 16 entries (the tail unused).  With it the launch is cut by members instead of by segments
 *   (balanced whatever the segment lengths); without it, four segments per workgroup.
 * Rows are added in member order: the result is the sequential sum, the same on every run.
 * seg_ptr / members / seg_of / n_rows are device arrays (n_rows: one int32, <= cap_rows; n_members >= seg_ptr[*n_rows]).
 */
int pfo_segment_sum(const float* src0, int32_t W0, const float* src1, int32_t W1, const int32_t* seg_ptr,
                    const int32_t* members, const int32_t* seg_of, int64_t n_members, const int32_t* n_rows,
                    int32_t cap_rows, int32_t src0_by_position, const uint8_t* src0_live,
                    float* out /* [cap_rows, W0+W1] */, void* stream);

/* ------------------------------------------------------------------------------------------
 * Dense fp32 contraction on the matrix cores (v_mfma_f32_16x16x4_f32), exposed for tests.
 *   C[M,N] = A[M,K] * op(B) + bias,  op(B) = B[N,K]^T (b_kmajor = 0, the nn.Linear weight layout)
 *                                    or      B[K,N]   (b_kmajor = 1)
 *   a_kmajor = 1 reads A as [K,M] (used for weight gradients dW = dY^T X).
 */
int pfo_gemm_f32(const float* A, int64_t lda, int32_t a_kmajor, const float* B, int64_t ldb, int32_t b_kmajor,
                 float* C, int64_t ldc, const float* bias, int32_t M, int32_t N, int32_t K, int32_t relu,
                 float* workspace, int64_t workspace_floats, void* stream);

/* ------------------------------------------------------------------------------------------
 * The same contraction for row-major A, computed on the 16-bit matrix cores by splitting every fp32 operand into pieces
 * with exact residuals and accumulating the piece products in fp32: fp32-level accuracy (error <= ~2^-22 sum_k |a||b|,
 * the bound of an fp32 accumulation).  Default: TWO fp16 pieces of the operand times a per-row power of two, x 2^s = h + l,
 * three v_mfma_f32_16x16x32_f16 per block (weight rows scaled from their maximum, activation rows from a running maximum with
 * exact accumulator rescaling; elements below 2^-16 of their row's maximum keep an absolute error <= 2^-38 of that maximum).
 * With PFO_BX_FMT=0 in the environment: THREE bf16 pieces, x = x1 + x2 + x3, six v_mfma_f32_16x16x32_bf16 per block (the
 * symbol's name).  This is the kernel the large launches of pfo_tgn_forward / pfo_tgn_backward use for torch.nn.Linear
 * (utils.py:7-17, torch.nn.MultiheadAttention in temporal_attention.py:27-31); it is exported so that it can be checked alone.
 *   workspace: pfo_gemm_bf16x3_workspace_bytes(N, K) bytes, 16-byte aligned (receives the split image of B).
 *   A 16-byte aligned, lda % 4 == 0, K % 4 == 0.
 */
int64_t pfo_gemm_bf16x3_workspace_bytes(int32_t N, int32_t K);
int pfo_gemm_bf16x3(const float* A, int64_t lda, const float* B, int64_t ldb, int32_t b_kmajor, float* C, int64_t ldc,
                    const float* bias, int32_t M, int32_t N, int32_t K, int32_t relu, void* workspace,
                    int64_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Root list of one batch shard, the node/timestamp arrays the reference concatenates before compute_embedding
 * (tgn.py:118-124 for the `ours` path, tgn.py:237-239 for the baseline):
 *   roots   = [ src[lo:hi] | dst[lo:hi] | groups[0][lo:hi, :] | groups[1][lo:hi, :] ... ]
 *   root_ts = ts of the interaction each root belongs to (edge_times repeated per group element)
 * groups[g] i32[B * reps[g]] (device pointers in a HOST array of n_groups <= PFO_MAX_ROOT_GROUPS entries), row-major
 * per interaction.  roots i32[R], root_ts f64[R], R = (hi - lo) * (2 + sum reps).
 */
#define PFO_MAX_ROOT_GROUPS 4
int pfo_roots_assemble(const int32_t* src, const int32_t* dst, const double* ts, int32_t lo, int32_t hi,
                       const int32_t* const* groups, const int32_t* reps, int32_t n_groups, int32_t* roots,
                       double* root_ts, void* stream);

/* ------------------------------------------------------------------------------------------
 * BPR loss, forward + gradient in one pass (main.py:321-337 / 364-381):
 *   loss = -mean_b log sigmoid( mean_k( s_b.p_b - s_b.n_bk ) )       (sigma of the MEAN difference)
 * emb f32[R,D] holds the roots in the reference's order: [src B | dst B | (p_pos B*n_pos) | neg B*n_neg];
 * pos_off = row offset of the positives block (B for the baseline path where the positive is dst,
 * 2B for the `ours` path), neg_off = row offset of the negatives block.  n_pos must be 1 (main.py:33).
 * loss_out f32[1]; d_emb f32[R,D] receives scale * dloss/demb (rows not involved are zeroed).
 */
int pfo_bpr_loss(const float* emb, int64_t B, int32_t D, int64_t pos_off, int64_t neg_off, int32_t n_neg,
                 int64_t R, float scale, float* loss_out, float* d_emb, float* workspace, void* stream);
/* The same in ONE launch: the workgroup that finishes last takes the mean (index order: reproducible).  ticket i32[1]: zero
 * before the first use, resets itself; workspace f32[B]. */
int pfo_bpr_loss_fused(const float* emb, int64_t B, int32_t D, int64_t pos_off, int64_t neg_off, int32_t n_neg,
                       int64_t R, float scale, float* loss_out, float* d_emb, float* workspace, int32_t* ticket, void* stream);
/* Only the per-interaction losses loss_parts f32[B] (-log sigmoid of the mean score difference) and the gradient rows; the
 * mean over the batch is left to the caller (pfo_tgn_backward_ev takes it on its side stream). */
int pfo_bpr_loss_parts(const float* emb, int64_t B, int32_t D, int64_t pos_off, int64_t neg_off, int32_t n_neg,
                       int64_t R, float scale, float* loss_parts, float* d_emb, void* stream);

/* ------------------------------------------------------------------------------------------
 * Ranking metrics of evaluation.py:114-145 for one positive per interaction.
 *   emb f32[R,D] = [src B | dst B | neg B*n_items]; score = src . item (evaluation.py:114-115).
 *   rank_b = #{ k : score(neg_bk) >= score(dst_b) }  - position of the positive in the descending order of
 *   concat(pos, neg) with the canonical tie policy (stable ascending argsort reversed, SURVEY App. A-9;
 *   evaluation.py:138 uses the platform's unstable argsort).
 *   rank_out i32[B]; hits_out f32[B,3] = recall@{1,3,5}; ndcg_out f32[B,3] = NDCG@{1,3,5} (evaluation.py:11-21).
 */
int pfo_rank_metrics(const float* emb, int64_t B, int32_t D, int32_t n_items, int32_t* rank_out, float* hits_out,
                     float* ndcg_out, void* stream);

/* ------------------------------------------------------------------------------------------
 * Time-sorted adjacency built on the device: replaces get_neighbor_finder / NeighborFinder.__init__ (utils/utils.py:117-148).
 * Every edge e contributes (dst, eidx, ts) to row src[e] and (src, eidx, ts) to row dst[e]; rows are sorted by timestamp,
 * ties in edge order (Python's stable sorted(key=ts), utils.py:139).  A stable LSD radix sort of the 2E entries; the eight
 * timestamp passes are skipped when the log is already chronological.  All pointers are device pointers; `workspace` holds
 * pfo_csr_build_workspace_bytes(E, n_nodes) bytes; indptr has n_nodes + 1 entries, the adjacency arrays 2E.
 */
int64_t pfo_csr_build_workspace_bytes(int64_t E, int64_t n_nodes);
int pfo_csr_build(const int32_t* src, const int32_t* dst, const int32_t* eidx, const double* ts, int64_t E, int64_t n_nodes,
                  int64_t* indptr, int32_t* adj_nbr, int32_t* adj_eidx, double* adj_ts, void* workspace,
                  int64_t workspace_bytes, void* stream);
/* Merge of a CSR of NEW edges (built by pfo_csr_build over n_nodes rows) into an existing one (n_old_nodes <= n_nodes rows):
 * the result equals a rebuild over [old edges ; new edges] - inside a row a new entry goes behind every old entry whose
 * timestamp is <= its own.  Output arrays hold old + new entries; new_indptr has n_nodes + 1 entries.
 */
int pfo_csr_append(const int64_t* old_indptr, const int32_t* old_nbr, const int32_t* old_eidx, const double* old_ts,
                   int64_t n_old_nodes, const int64_t* add_indptr, const int32_t* add_nbr, const int32_t* add_eidx,
                   const double* add_ts, int64_t n_nodes, int64_t* new_indptr, int32_t* new_nbr, int32_t* new_eidx,
                   double* new_ts, void* stream);

/* ------------------------------------------------------------------------------------------
 * Adam step over a flat parameter buffer (torch.optim.Adam defaults, main.py:123,389).
 */
int pfo_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                  float beta1, float beta2, float eps, int32_t step, void* stream);
/* The same update on up to PFO_ADAM_MAX_RANGES element ranges [lo[r], hi[r]) of the flat buffers, each with its own step
 * count, in ONE launch.  torch.optim.Adam keeps a step counter per parameter tensor and skips tensors whose .grad is
 * None (main.py:123 hands it every parameter; the GRU's are None on the first batch of an epoch, when no message is
 * pending: memory_updater.py:38-40), so tensors can be one step apart for a whole run; ranges that are left out are
 * not touched at all.  lo / hi / step are HOST arrays. */
/* ..._dev: the step count of range r is step[r] + *step_dev (a device word): a graph-captured step advances it on the device
 * and the bias corrections are computed there */
#define PFO_ADAM_MAX_RANGES 16
int pfo_adam_step_ranges_dev(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int32_t n_ranges,
                             const int64_t* lo, const int64_t* hi, const int32_t* step, const int32_t* step_dev, float lr,
                             float beta1, float beta2, float eps, void* stream);
int pfo_adam_step_ranges(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int32_t n_ranges,
                         const int64_t* lo, const int64_t* hi, const int32_t* step, float lr, float beta1, float beta2,
                         float eps, void* stream);

/* ------------------------------------------------------------------------------------------
 * The TGN step (model/tgn.py:102-378 + modules/).  One POD config, one flat parameter buffer,
 * caller-owned state tables and workspace.
 */
typedef struct pfo_tgn_config {
  int32_t n_nodes;      /* incl. padding node 0 */
  int32_t n_edges_p1;   /* rows of edge_feat (E+1) */
  int32_t D;            /* memory = node-feature = time dim (tgn.py:43-54); must be a multiple of 4 */
  int32_t Ef;           /* edge-feature dim (multiple of 4) */
  int32_t n_layers;     /* L */
  int32_t n_heads;      /* H, divides 2D */
  int32_t use_memory;   /* 0 = TGAT (main.py:70-74) */
  int32_t max_roots;    /* capacity R the workspace is sized for */
  int32_t max_neighbors;/* capacity K */
  int32_t max_batch;    /* capacity B (positives = 2B) */
} pfo_tgn_config;

/* flat parameter layout (element offsets into the fp32 parameter / gradient buffers) */
typedef struct pfo_tgn_layer_layout {
  int64_t wq, wk, wv, b_in, wo, bo, w1, b1, w2, b2; /* attention_models.{l}: MHA in/out proj + MergeLayer */
} pfo_tgn_layer_layout;
typedef struct pfo_tgn_layout {
  int64_t time_w, time_b;                   /* time_encoder.w.{weight,bias} */
  int64_t gru_w_ih, gru_w_hh, gru_b_ih, gru_b_hh; /* memory_updater.memory_updater.* (-1 without memory) */
  pfo_tgn_layer_layout layer[PFO_MAX_LAYERS];
  int64_t total;
} pfo_tgn_layout;

int pfo_tgn_param_layout(const pfo_tgn_config* cfg, pfo_tgn_layout* out);
/* bytes of scratch pfo_tgn_forward/backward need for (cfg.max_roots, cfg.max_neighbors) */
int64_t pfo_tgn_workspace_bytes(const pfo_tgn_config* cfg);

/* device-resident model state; all caller-owned */
typedef struct pfo_tgn_state {
  const int64_t* indptr; const int32_t* adj_nbr; const int32_t* adj_eidx; const double* adj_ts; /* CSR (K1) */
  const float* node_feat;   /* [n_nodes,D]  tgn.py:35 */
  const float* edge_feat;   /* [E+1,Ef] z-scored, tgn.py:38-41 */
  float* memory;            /* [n_nodes,D]  modules/memory.py:28 */
  float* last_update;       /* [n_nodes]    modules/memory.py:30 */
  float* msg_table;         /* [n_nodes,3D+Ef] last pending raw message per node */
  float* msg_time;          /* [n_nodes] */
  uint8_t* has_msg;         /* [n_nodes] */
  const float* params;      /* flat, pfo_tgn_layout */
  /* Parameter cache (optional, abi 3): a caller-owned device buffer of pfo_tgn_pcache_bytes(cfg) bytes that holds
     everything the step derives from the PARAMETERS alone - the composite weights of every layer (Wqk, cqk, W1ov^T, the
     fc2-folded forms), cos(b) and the fp16 weight images of all contractions incl. the GRU's.  The reference re-reads
     nn.Linear weights per batch and recomputes nothing (tgn.py:219-327); here that work is ~15 small dependent launches,
     so it is done once per parameter VERSION: with pcache_valid == 0 pfo_tgn_forward builds the cache (side stream, as it
     did per step before), with pcache_valid != 0 it launches none of it.  The caller owns the validity: it must pass 0
     after anything wrote `params` (optimizer step, load_state_dict ...) unless pfo_tgn_refresh ran since.  pfo_tgn_backward
     reads the composites from the same buffer: parameters must not change between a forward and its backward.
     The buffer must be ZERO-INITIALISED by the caller when it is allocated (padding rows / columns of the composites are never
     written).  NULL: the composites live in the workspace and are rebuilt by every forward. */
  void* pcache;
  int32_t pcache_valid;
} pfo_tgn_state;

typedef struct pfo_tgn_batch {
  const int32_t* roots;     /* [R] = [src | dst | (p_pos) | neg]   tgn.py:121 / :235 */
  const double* root_ts;    /* [R] timestamps, negatives repeat their interaction's time (tgn.py:123-124,238-239) */
  int32_t R, K;
  int32_t uniform;          /* 0 most-recent, 1 injected draws, 2 Philox */
  const int64_t* const* draws; /* HOST array of n_layers device pointers (level L..1), mode 1 only */
  uint64_t seed, offset;
  float dropout_p;          /* attention-weight dropout, 0 in eval / parity mode (temporal_attention.py:28-32) */
  int32_t training;         /* keep what backward needs */
  const int32_t* extra_nodes; /* [n_extra] nodes whose lazily-updated memory is needed although they are not
                                 embedded here: data-parallel ranks pass the GLOBAL batch's positives so that
                                 pfo_tgn_update_state can persist all of them; may be NULL */
  int32_t n_extra;
  const uint64_t* offset_dev; /* optional device word ADDED to `offset` by every kernel that draws random numbers (dropout
                                 masks, Philox neighbour draws).  A step captured into a HIP graph keeps its kernel
                                 arguments: the per-step stream position then lives here and is advanced on the device */
  int32_t deterministic;    /* != 0: pfo_tgn_backward is bitwise reproducible run to run.  The two order-dependent sums of the
                                 default path - float atomics into the level-0 gradient rows, fp64 atomics into the time-encoder
                                 partial bins - become order-free: the rows are added as 2^-40 fixed-point int64 (integer addition
                                 is associative), the partials go to one slab row per workgroup and are folded in row order.
                                 Costs ~4 % of a C2 step (8-byte atomics); every other sum of the step is ordered already */
  int32_t prepared;         /* != 0: pfo_tgn_prepare already ran for this batch on this workspace (and the caller's stream is
                                 ordered behind it): pfo_tgn_forward skips the frontier sampling, the compaction and the row pack */
  /* optional: the interactions whose state update (pfo_tgn_update_state's arguments: the GLOBAL batch's src / dst / time /
     edge index, upd_B of them) pfo_tgn_forward performs ITSELF - on its side stream, behind the lazy GRU, beside layer 1,
     joined by the event the call's layer 2 waits for anyway: two small launches leave the critical path.  Taken with
     use_memory and n_layers >= 2; otherwise (upd_src NULL, one layer, no memory) the caller calls pfo_tgn_update_state. */
  const int32_t* upd_src; const int32_t* upd_dst; const double* upd_ts; const int32_t* upd_eidx; int32_t upd_B;
  /* Optional INJECTED dropout decisions (parity tests against masks captured from the reference, like `draws` for the uniform
     sampler): dropout_keep[L - l] for layer l (same order as `draws`: the roots' level first) is a device array u8 [n_l, K],
     bit h of entry (n, j) = the attention weight of key slot j / head h of instance n is KEPT; null = the step's own Philox
     draws.  Only read when dropout_p > 0 and training != 0. */
  const uint8_t* const* dropout_keep;
  /* Optional hipEvent_t pfo_tgn_backward records on the caller's stream right behind layer 1's d ctx' contraction, i.e. when
     the layer-1 attention backward - the longest kernel of the step, ~1/3 of it - is about to start.  A caller that prepares
     the NEXT batch (pfo_tgn_prepare on another stream: sampling, compaction, row pack - small latency-bound launches) makes
     that stream wait for this event, so the preparation runs beside the one phase of the step that hides it. */
  void* mid_event;
  int32_t defer_join;       /* pfo_tgn_backward: != 0 leaves the END of the backward on the library's first side stream: the
                               caller's stream is NOT made to wait for the side streams' last launches (the chain back to the
                               layer-1 projection weights, the time-encoder fold: ~40 us after the caller's stream has run dry at
                               C2); instead that side stream waits for the caller's stream, so "everything of this backward" is
                               complete THERE.  The caller must then take the optimizer step with pfo_tgn_adam_side (same side
                               stream) or call pfo_tgn_join before it touches gradients or parameters on its own stream.  The
                               next pfo_tgn_forward joins by itself - behind its neighbour sampling, which reads neither. */
  int32_t seg_in_forward;   /* training calls whose state update runs inside the forward (upd_* given): != 0 lets the forward also
                               queue - on its side stream, beside layer 1 - what the backward's layer-1 kernels need from it and
                               that depends on the sampled levels alone (the instance groups per touched row, the cleared
                               level-0 gradient table); the backward (same batch struct) then forks nothing at its start and
                               waits for nothing in front of its attention kernel */
  int32_t mid_event_late;   /* != 0: record it BEHIND the layer-1 attention backward instead - the preparation then runs beside the
                               backward's serial tail (per-row sums, the touched-table contractions, the GRU's weight gradients:
                               small launches that leave most of the chip idle) */
} pfo_tgn_batch;

/* The part of pfo_tgn_forward that depends on neither parameters nor gradients - frontier sampling (utils.py:163-219 per level,
 * embedding_module.py:125), compaction of the touched nodes, the packed copies of their memory / message rows (tgn.py:251) - on
 * ANY stream.  A training loop issues it for batch n+1 on a second stream as soon as batch n's forward (whose state update
 * writes the tables the pack reads) is queued, into a second workspace: it then runs beside batch n's backward, and batch
 * n+1's forward (batch.prepared = 1, its stream made to wait for this one) starts at the GRU.  The reference does the same
 * work on the host between batches (main.py:190-207 + the neighbour finder inside the model call). */
int pfo_tgn_prepare(const pfo_tgn_config* cfg, const pfo_tgn_state* st, const pfo_tgn_batch* batch, void* workspace,
                    void* stream);

/* Lazy memory update for touched nodes (tgn.py:251, memory_updater.py:35-53) + L-layer temporal graph
 * attention (embedding_module.py:76-175, temporal_attention.py:34-90).  emb_out f32[R,D]. */
int pfo_tgn_forward(const pfo_tgn_config* cfg, const pfo_tgn_state* st, const pfo_tgn_batch* batch,
                    void* workspace, float* emb_out, void* stream);
/* Gradients of everything pfo_tgn_forward (training=1) computed, given d_emb f32[R,D]; ACCUMULATES into grad (flat). */
int pfo_tgn_backward(const pfo_tgn_config* cfg, const pfo_tgn_state* st, const pfo_tgn_batch* batch,
                     void* workspace, const float* d_emb, float* grad, void* stream);
/* The same with two options for the training loop:
 *   zero_grad_first  != 0: `grad` is cleared by this call (on its side stream, off the critical path) before anything is added -
 *                          replaces the caller's memset of the 6 MB flat buffer (optimizer.zero_grad(), main.py:386)
 *   top_ready_event  (hipEvent_t as void*, may be NULL): recorded, on an internal stream, at the point where every gradient of
 *                          the TOP layer's parameter block - elements [pfo_tgn_grad_split(cfg), layout.total) of `grad` - is
 *                          final, ~half a step before the call's last kernel: a data-parallel caller makes its communication
 *                          stream wait for it and all-reduces that block beside the rest of the backward.  Only
 *                          with n_layers >= 2 (pfo_tgn_grad_split returns layout.total otherwise and the event is not recorded).
 *   mean_src / mean_n / mean_out (may be NULL / 0 / NULL): out[0] = mean(mean_src[0 .. mean_n)) is taken on this call's side
 *                          stream, beside the backward, and is complete when the call's work is (pfo_bpr_loss_parts leaves the
 *                          loss's final reduction to it: the loss VALUE is not an input of the backward) */
int pfo_tgn_backward_ev(const pfo_tgn_config* cfg, const pfo_tgn_state* st, const pfo_tgn_batch* batch,
                        void* workspace, const float* d_emb, float* grad, int32_t zero_grad_first, void* top_ready_event,
                        const float* mean_src, int64_t mean_n, float* mean_out, void* stream);
int pfo_tgn_grad_split(const pfo_tgn_config* cfg, int64_t* split);
/* Persist memory for the positives, clear their pending messages, build and store the new raw messages with
 * last-wins semantics (tgn.py:290-317, memory_updater.py:18-33, memory.py:35-37,73-75, tgn.py:357-378).
 * src/dst i32[B], ts f64[B], eidx i32[B]; needs the workspace of the forward call that preceded it. */
int pfo_tgn_update_state(const pfo_tgn_config* cfg, const pfo_tgn_state* st, const int32_t* src, const int32_t* dst,
                         const double* ts, const int32_t* eidx, int32_t B, void* workspace, void* stream);

/* diagnostics for tests: copies of internals of the last forward (device pointers into the workspace) */
typedef struct pfo_tgn_debug {
  const int32_t* n_touched;  /* [1] */
  const int32_t* touched_ids;/* [n_touched] */
  const float* h0_table;     /* [n_touched,D] layer-0 features memory'+node_feat (embedding_module.py:98) */
  const int32_t* slot;       /* [n_nodes] */
  /* operands and results of the two largest weight-gradient contractions of the last backward (valid until the workspace's
     next forward): parity tests re-contract the SAME fp32 operands in fp64 (per-element evidence for the split contraction) */
  const int32_t* n_core;     /* [1] table rows the step's levels reference (the GRU backward's row count) */
  const float* l1_ctx;       /* [n_1, H*Cp]   layer-1 context rows ctx' (a operand of dW1ov)                     */
  const float* l1_dh1;       /* [n_1, D]      d loss / d (layer-1 fc1 pre-activation) (b operand)               */
  const float* l1_dW1ovT;    /* [H*Cp, D]     = ctx'^T dh1                                                      */
  const float* gru_dgi;      /* [n_core, 3D]  d loss / d (GRU input-side pre-activations) (a operand of dW_ih)  */
  const float* gru_msg_rows; /* [n_core, 3D+Ef] packed message rows (b operand); dW_ih lands in the gradient buffer */
  int32_t Cp;                /* per-head row stride of ctx' */
} pfo_tgn_debug;
int pfo_tgn_debug_views(const pfo_tgn_config* cfg, void* workspace, pfo_tgn_debug* out);

/* Parameter cache (pfo_tgn_state.pcache): its size, and the call that (re)builds it from state->params on the library's
 * side stream, forked from `stream` and NOT joined back - the next pfo_tgn_forward with pcache_valid = 1 queues its own
 * side-stream work behind it, so the caller's stream never waits for the build itself.  Meant to be called right behind the
 * optimizer's kernel: the build then runs beside the next batch's sampling phase.  Not capturable into a HIP graph
 * (unjoined fork): a captured step passes pcache_valid = 0 instead. */
/* Optimizer step on the library's first side stream, behind a backward that ran with pfo_tgn_batch.defer_join: same
 * arguments and arithmetic as pfo_adam_step_ranges.  Gradients, moments and parameters are then in flight on that stream;
 * pfo_tgn_forward / pfo_tgn_prepare / pfo_tgn_refresh / pfo_tgn_update_state wait for it where they first need them,
 * pfo_tgn_join(stream) makes any other stream wait (no-op when nothing is pending). */
int pfo_tgn_adam_side(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int32_t n_ranges,
                      const int64_t* lo, const int64_t* hi, const int32_t* step, float lr, float beta1, float beta2, float eps);
/* The same step in BUCKETS ordered by first use in the next forward (data-parallel ranks: reduce + step per bucket).
 * bucket 1: the ranges the next pfo_tgn_forward reads on the CALLER's stream (time encoder, GRU, layer 1: everything below
 * pfo_tgn_grad_split) - an event behind this kernel is all that forward's caller's stream waits for; bucket 2: a later bucket of
 * the same step (the top layer's block, whose all-reduce ran beside the backward) - the forward meets it through the side
 * stream's own order (composite weights, fc2 fold); bucket 0: pfo_tgn_adam_side.  Buckets 1 / 2: models with n_layers >= 2. */
int pfo_tgn_adam_side_bucket(float* param, float* grad, float* exp_avg, float* exp_avg_sq, int32_t n_ranges,
                             const int64_t* lo, const int64_t* hi, const int32_t* step, float lr, float beta1, float beta2, float eps,
                             int32_t flags /* bits 0-1: the bucket; bit 2 (4): the kernel also CLEARS the gradient ranges it read -
                                              optimizer.zero_grad() folded in, so that the next pfo_tgn_backward_ev may run with
                                              zero_grad_first = 0 (no clear on its critical path) when the ranges cover the buffer */);
int pfo_tgn_join(void* stream);
/* The library's first side stream (hipStream_t; null on failure): the stream a deferred backward end is left on and
 * pfo_tgn_adam_side runs on.  A data-parallel caller queues its gradient all-reduce THERE, between pfo_tgn_backward
 * (defer_join) and pfo_tgn_adam_side - ordered behind the backward's last launch, in front of the optimizer's kernel, and off the
 * caller's stream, which goes on to the next batch's sampling.  One stream per device, valid for the life of the process. */
void* pfo_tgn_side_stream(void);
int64_t pfo_tgn_pcache_bytes(const pfo_tgn_config* cfg);
int pfo_tgn_refresh(const pfo_tgn_config* cfg, const pfo_tgn_state* state, void* stream);

/* ------------------------------------------------------------------------------------------
 * Live per-kernel timing with HIP events on the launch stream (bench.py roofline numbers).
 * While enabled every launch of the kinds below is bracketed by an event pair; pfo_prof_collect
 * waits for them and returns, per kind, the summed device time [ms], the summed ALGORITHMIC work
 * (FLOP for the GEMM kinds, bytes otherwise - DESIGN.md states the per-unit figures) and the launch count.
 */
#define PFO_PROF_GEMM_NT 0   /* C = A B^T (+bias...)  forward projections                */
#define PFO_PROF_GEMM_NN 1   /* C = A B             backward-data, folded key projection */
#define PFO_PROF_GEMM_TN 2   /* dW = A^T B          weight gradients (split-K + reduce)  */
#define PFO_PROF_GEMM_DEVM 3 /* fp32-MFMA contractions whose extent is a device-side count (work = per-row flops x the count read back) */
#define PFO_PROF_ATTN_FWD 4
#define PFO_PROF_ATTN_BWD 5
#define PFO_PROF_SAMPLER 6
#define PFO_PROF_GEMM_BX 7   /* gemm_bx_areg_kernel and nothing else: every row-major split-contraction launch, device-side row counts included (flops = 2MNK, M read back) */
#define PFO_PROF_GEMM_TN_BX 8 /* grouped weight gradients on the bf16x3 kernel (GEMM kernel only)  */
#define PFO_PROF_GEMM_BX_SKINNY 9 /* the 32-row bf16x3 kernel of the short (layer-2) launches           */
#define PFO_PROF_ATTN_BWD_RUNS 10 /* layer-1 attention backward, run-merged kernel (attn_bwd_runs_kernel)  */
#define PFO_PROF_GRU_FUSED 11 /* gru_fused_kernel: both GRUCell contractions + gates (flops = per-row FLOPs x touched rows read back) */
#define PFO_PROF_GEMM_MULTI 12 /* gemm_multi_kernel: the grouped small fp32 products of the composite-weight builds and their chain-back (FLOP) */
#define PFO_PROF_SEGSUM 13     /* segsum_*_kernel: per-table-row sums of the layer-1 gradient rows (bytes: every member row once + the sums) */
#define PFO_PROF_TN_REDUCE 14  /* tn_group_reduce_kernel: the split-K slabs of a grouped weight-gradient launch folded (bytes: slabs in, matrices out) */
#define PFO_PROF_GRU_GATES_BWD 15 /* gru_gates_bwd_*_kernel: GRU gate backward (bytes: gates, h, d h in; dgi, dgh out) */
#define PFO_PROF_GEMM_TN_BX8 16   /* the same grouped weight gradients on 256-row tiles of eight wavefronts (gemm_tn_group_bx_kernel<1, 8>): a kernel of its own in the traces */
#define PFO_PROF_KINDS 17
/* Milestones: while enabled (pfo_marks_enable(1)) the step's native calls record a timing event on the CALLER's stream at
 * named points of the critical path (sampling done, lazy GRU done, every large launch of every layer ...; callers may add
 * their own with pfo_mark).  pfo_marks_dump waits for them and writes one line per consecutive pair "from -> to  mean_us  n"
 * in first-seen order: the segments of the caller's stream as the GPU ran them, WITHOUT a tracer slowing the host down.
 * ~1 us per mark; off by default. */
int pfo_marks_enable(int32_t on);
int pfo_mark(const char* name /* static string */, void* stream);
int64_t pfo_marks_dump(char* out, int64_t cap); /* HOST buffer; returns the bytes written (0-terminated), clears the records */
int pfo_prof_enable(int32_t on);
int pfo_prof_collect(double* ms, double* work, int64_t* count); /* HOST arrays of PFO_PROF_KINDS entries */
/* Shader clock under the product kernels [GHz]: the first wavefront of three large kernels stamps the shader cycle counter
 * against the constant 100 MHz counter on every launch (two scalar reads, two atomics per launch).  out[0] attention forward
 * (ring form), out[1] run-merged attention backward, out[2] grouped weight-gradient kernel; 0 where the kernel has not run since
 * the last reset.  The peaks of MI355X_MICROARCH.md are priced at 2.4 GHz: bench.py prints both. */
#define PFO_CLOCK_KERNELS 3
int pfo_shader_clock(double* ghz_out /* HOST, PFO_CLOCK_KERNELS entries */, int32_t reset);

#ifdef __cplusplus
}
#endif
#endif /* PFOTGN_H */
