// The TGN step as one native call per phase: pfo_tgn_forward / pfo_tgn_backward / pfo_tgn_update_state.
// Host code only: carves the caller's workspace, walks the recursion levels of
// embedding_module.py:76-175 iteratively (frontier lists instead of recursion), and queues the kernels
// on the caller's stream.  No allocation, no synchronisation, no host<->device copies.
//
// Level structure (L layers, R roots, K neighbour slots):
//   S_L = roots;  S_{l-1} = [ S_l ; neighbours(S_l) flattened ]  (|S_{l-1}| = |S_l| (1+K))
//   layer l maps features of S_{l-1} to embeddings of S_l:  x_i = H_{l-1}[i],  neighbour j of i = H_{l-1}[|S_l| + i K + j]
//   level 0 is never materialised: its rows are looked up in a per-step table of touched nodes
//   h0_tab[slot[v]] = memory'[v] + node_feat[v]   (embedding_module.py:93-98)
#include "gemm.hpp"
#include "attn.hpp"
#include "memory.hpp"
#include <algorithm>
#include <functional>
#include <math.h>
#include <string.h>
#include <stdlib.h>

int pfo_tnbr_sample_dev(const int64_t*, const int32_t*, const int32_t*, const double*, int64_t, const int32_t*, const double*,
                        int64_t, int32_t, int32_t, const int64_t*, uint64_t, uint64_t, const uint64_t*, int32_t*, int32_t*,
                        float*, float*, int32_t*, double*, int32_t*, int32_t*, void*, int32_t*, int64_t);


namespace {

struct LayerWs {
  float *QK, *attw, *ctx, *h1;             // activations kept for backward
  float* dQK;                              // layers >= 2: their own d qk' rows - their weight-gradient launch reads them on a side
                                           // stream while the caller's stream is already writing layer 1's
  float *cq, *Wqk, *cqk, *W1oT, *W1ovT;   // per-step composite weights (see the layer comment in pfo_tgn_forward)
  float *dWqk, *gqk, *dW1ovT, *dW1oT, *gq; // their gradients (per layer: the chain-back runs on the side stream)
  // layers >= 2 take the PREVIOUS layer's h1 rows as input, with that layer's fc2 (out = W2 h1 + b2) folded into their own
  // composites (see "fc2 fold" in pfo_tgn_forward): the folded weights, the two intermediates, and the gradients of all
  float *T1, *tq, *Wqk_f, *cqk_f, *W1ovT_f, *W1b_f, *b1_f;
  float *dT1, *dWqk_f, *gqk_f, *dW1ovT_f, *dW1b_f, *db1_f, *fold_slabs, *fold_vslabs;
  uint8_t* inv;
  // pre-split bf16x3 images of the weight operands of the large contractions (gemm.hpp PfoBimg): per step, side stream
  void *iWqk, *iWqkT, *iW1ov, *iW1ovT, *iW1b, *iW1bT, *iW2, *iW2T;
};
struct Ws {
  int32_t* nodes[PFO_MAX_LAYERS + 1];
  double* ts[PFO_MAX_LAYERS + 1];
  int32_t* eidx[PFO_MAX_LAYERS + 1];
  float* dt[PFO_MAX_LAYERS + 1];
  int32_t *mark, *slot, *touched, *n_touched, *n_core, *scan, *idx0, *winner;   // n_touched = counts[0] (all rows), n_core = counts[1]
  float *gi, *gh, *gates, *upd_mem, *h0_tab, *d_h0, *msg_rows, *h_rows;   // gi / gh: backward only (d gi / d gh)
  uint8_t* hm;
  float *cosb, *zero, *tb_part;
  void *iWih, *iWhh;
  // layer 1 works on the touched-node table: QX[s] = h0_tab[s] [Wqk ; W1[:, E:]]^T + [cqk | 0] for every touched row s
  // (the query-side projections of all instances that sit on node s), Dq = per-row sums of the instances' gradients
  float *QX, *Dq, *dx_tab, *l1_bias;
  void* iQX;
  int32_t *seg_ptr, *seg_cur, *seg_tmp, *seg_mem, *seg_of, *seg_scratch;
  uint8_t* dqk_live;           // per layer-1 member position: does dQK row m hold a sum (attn.hip, run-merged backward)
  int32_t* cnt1;               // per layer-1 instance: entries of its node's row before its time (the run key, sampler.hip)
  LayerWs layer[PFO_MAX_LAYERS + 1];
  float *dh1, *dctx, *dQK;
  float* dH[PFO_MAX_LAYERS + 1];
  float *slabs, *slabs2, *slabs3;   // split-K slabs of the weight-gradient launches: main stream / side stream / second side stream
  double *dtime, *fold_scratch, *dtime_slab;
  int32_t* tickets;
  int64_t slab_floats;
  // cleared per step: [zero | tickets | dtime] (zero_bytes); cleared per composite build: pz = [64 zero floats | Wqk, W1ovT, cqk
  // of every layer] (pz_bytes) - in the workspace right behind the per-step region, or in the parameter cache
  size_t zero_bytes, mark_bytes;
  float* pz; size_t pz_bytes;
  bool pz_in_cache;            // pz lives in the caller's parameter cache, which the caller zeroed once (no per-build memset)
  int64_t bytes;
};

struct Dims {
  int L, D, Ef, H, E, C, Cp, dh, M;
  int64_t ncap[PFO_MAX_LAYERS + 1];
  int64_t capP;
};

Dims dims_of(const pfo_tgn_config* c) {
  Dims d;
  d.L = c->n_layers; d.D = c->D; d.Ef = c->Ef; d.H = c->n_heads;
  d.E = 2 * d.D; d.C = 2 * d.D + d.Ef; d.dh = d.E / d.H; d.M = 3 * d.D + d.Ef;
  d.Cp = (int)pfo_align_up(d.C + 2, 4);   // key columns + (sum of weights) + (valid flag), padded for 16-byte rows
  d.ncap[d.L] = c->max_roots;
  for (int l = d.L; l >= 1; --l) d.ncap[l - 1] = d.ncap[l] * (1 + (int64_t)c->max_neighbors);
  d.capP = std::min<int64_t>(c->n_nodes, d.ncap[0] + 2 * (int64_t)c->max_batch);
  return d;
}

const int64_t SLAB_FLOATS = (int64_t)(768 + 32) * 128 * 176;

template <typename T>
T* take(char*& p, int64_t count) {
  T* r = reinterpret_cast<T*>(p);
  p += pfo_align_up(count * (int64_t)sizeof(T), 256);
  return r;
}

Ws carve(const pfo_tgn_config* c, void* base) {
  const Dims d = dims_of(c);
  Ws w;
  memset(&w, 0, sizeof(w));
  char* p = reinterpret_cast<char*>(base);
  const int64_t Km = c->max_neighbors;
  for (int l = 0; l <= d.L; ++l) {
    w.nodes[l] = take<int32_t>(p, d.ncap[l]);
    if (l >= 1) {
      w.ts[l] = take<double>(p, d.ncap[l]);
      w.eidx[l] = take<int32_t>(p, d.ncap[l] * Km);
      w.dt[l] = take<float>(p, d.ncap[l] * Km);
    }
  }
  // one memset per step clears [zero | tickets | dtime]; a composite build clears pz = [64 floats | Wqk, W1ovT, cqk of every
  // layer] (padding rows / columns of the composites must be zero; with a parameter cache pz lives there: pcache_bind)
  w.zero = take<float>(p, 64);
  w.tickets = take<int32_t>(p, 64);
  w.dtime = take<double>(p, (int64_t)pfo_attn_bwd_max_parts() * 2 * d.D);      // time-encoder gradient bins, also cleared per step
  w.zero_bytes = (size_t)(p - reinterpret_cast<char*>(w.zero));
  w.pz = take<float>(p, 64);
  for (int l = 1; l <= d.L; ++l) {
    LayerWs& lw = w.layer[l];
    lw.Wqk = take<float>(p, (int64_t)d.H * d.Cp * d.D);
    lw.W1ovT = take<float>(p, (int64_t)d.H * d.Cp * d.D);
    // layer 1: cqk is the head of the bias of the stacked [Wqk ; W1[:, E:]] projection; its tail (D floats) stays zero
    lw.cqk = take<float>(p, (int64_t)d.H * d.Cp + (l == 1 ? d.D : 0));
  }
  w.l1_bias = w.layer[1].cqk;
  w.pz_bytes = (size_t)(p - reinterpret_cast<char*>(w.pz));
  w.cosb = take<float>(p, d.D);
  w.tb_part = take<float>(p, d.D);
  {
    const int WQ = d.H * d.Cp + d.D;
    // one memset per step clears [touched-node flags | block flags of the one-pass compaction]
    w.mark = take<int32_t>(p, c->n_nodes);
    w.scan = take<int32_t>(p, pfo_compact_scratch_ints(c->n_nodes));
    w.mark_bytes = (size_t)(p - reinterpret_cast<char*>(w.mark));
    w.slot = take<int32_t>(p, c->n_nodes);
    w.touched = take<int32_t>(p, d.capP);
    w.n_touched = take<int32_t>(p, 64);
    w.n_core = w.n_touched + 1;
    w.idx0 = take<int32_t>(p, d.ncap[0]);
    w.h0_tab = take<float>(p, d.capP * d.D);
    w.QX = take<float>(p, d.capP * WQ);
    w.Dq = take<float>(p, d.capP * WQ);
    w.dx_tab = take<float>(p, d.capP * d.D);
    w.iQX = take<char>(p, pfo_bimg_bytes(WQ, d.D));
    w.seg_ptr = take<int32_t>(p, d.capP + 1);
    w.seg_cur = take<int32_t>(p, d.capP + 1);
    w.seg_tmp = take<int32_t>(p, d.ncap[1]);
    w.seg_mem = take<int32_t>(p, d.ncap[1]);
    w.seg_of = take<int32_t>(p, pfo_seg_of_ints(d.ncap[1]));
    w.seg_scratch = take<int32_t>(p, pfo_seg_scratch_ints((int)d.capP));
    w.cnt1 = take<int32_t>(p, d.ncap[1]);
    w.dqk_live = take<uint8_t>(p, d.ncap[1]);
  }
  if (c->use_memory) {
    w.winner = take<int32_t>(p, c->n_nodes);
    w.gi = take<float>(p, d.capP * 3 * d.D);
    w.gh = take<float>(p, d.capP * 3 * d.D);
    w.gates = take<float>(p, d.capP * 4 * d.D);
    w.upd_mem = take<float>(p, d.capP * d.D);
    w.d_h0 = take<float>(p, PFO_GRAD_REPLICAS * d.capP * d.D);     // one replica per XCD (attn.hip, DMODE 1)
    w.msg_rows = take<float>(p, d.capP * d.M);
    w.h_rows = take<float>(p, d.capP * d.D);
    w.hm = take<uint8_t>(p, d.capP);
    w.iWih = take<char>(p, pfo_gru_img_bytes(d.D, d.M));
    w.iWhh = take<char>(p, pfo_gru_img_bytes(d.D, d.D));
  }
  for (int l = 1; l <= d.L; ++l) {
    const int64_t N = d.ncap[l];
    LayerWs& lw = w.layer[l];
    lw.cq = take<float>(p, d.E);
    lw.W1oT = take<float>(p, (int64_t)d.E * d.D);
    lw.dWqk = take<float>(p, (int64_t)d.H * d.Cp * d.D);
    lw.gqk = take<float>(p, (int64_t)d.H * d.Cp);
    lw.dW1ovT = take<float>(p, (int64_t)d.H * d.Cp * d.D);
    lw.dW1oT = take<float>(p, (int64_t)d.E * d.D);
    lw.gq = take<float>(p, d.E);
    {
      const int HCp = d.H * d.Cp;
      lw.iWqk = take<char>(p, pfo_bimg_bytes(HCp, d.D));   lw.iWqkT = take<char>(p, pfo_bimg_bytes(d.D, HCp));
      lw.iW1ovT = take<char>(p, pfo_bimg_bytes(HCp, d.D)); lw.iW1ov = take<char>(p, pfo_bimg_bytes(d.D, HCp));
      lw.iW1b = take<char>(p, pfo_bimg_bytes(d.D, d.D));   lw.iW1bT = take<char>(p, pfo_bimg_bytes(d.D, d.D));
      lw.iW2 = take<char>(p, pfo_bimg_bytes(d.D, d.D));    lw.iW2T = take<char>(p, pfo_bimg_bytes(d.D, d.D));
    }
    lw.QK = l > 1 ? take<float>(p, N * d.H * d.Cp) : nullptr;      // layer 1 reads its rows from the touched-node table QX
    lw.attw = take<float>(p, N * d.H * Km);
    lw.inv = take<uint8_t>(p, N);
    lw.ctx = take<float>(p, N * d.H * d.Cp);
    lw.h1 = take<float>(p, N * d.D);
    if (l < d.L) w.dH[l] = take<float>(p, N * d.D);
    if (l >= 2) {
      lw.dQK = take<float>(p, N * d.H * d.Cp);
      const int64_t HCpD = (int64_t)d.H * d.Cp * d.D, HCp = (int64_t)d.H * d.Cp, DD = (int64_t)d.D * d.D;
      lw.T1 = take<float>(p, HCpD);      lw.tq = take<float>(p, HCp);
      lw.Wqk_f = take<float>(p, HCpD);   lw.cqk_f = take<float>(p, HCp);
      lw.W1ovT_f = take<float>(p, HCpD); lw.W1b_f = take<float>(p, DD);   lw.b1_f = take<float>(p, d.D);
      lw.dT1 = take<float>(p, HCpD);
      lw.dWqk_f = take<float>(p, HCpD);  lw.gqk_f = take<float>(p, HCp);
      lw.dW1ovT_f = take<float>(p, HCpD); lw.dW1b_f = take<float>(p, DD); lw.db1_f = take<float>(p, d.D);
      lw.fold_slabs = take<float>(p, (2 * d.H + 2) * DD);
      lw.fold_vslabs = take<float>(p, (int64_t)(d.H + 2) * d.D);
    }
  }
  const int64_t N1 = d.ncap[1];
  w.dh1 = take<float>(p, N1 * d.D);
  w.dctx = take<float>(p, N1 * d.H * d.Cp);
  w.dQK = take<float>(p, N1 * d.H * d.Cp);
  w.slab_floats = SLAB_FLOATS;
  w.slabs = take<float>(p, w.slab_floats);
  w.slabs2 = take<float>(p, w.slab_floats);
  w.slabs3 = take<float>(p, w.slab_floats);
  w.fold_scratch = take<double>(p, pfo_fold_parts_scratch_doubles(2 * d.D));
  {
    // deterministic mode: one slab row of time-encoder partials per attention-backward workgroup, all layers
    int64_t rows = 0;
    for (int l = 1; l <= d.L; ++l) rows += pfo_attn_bwd_det_parts(d.ncap[l]);
    w.dtime_slab = take<double>(p, rows * 2 * d.D);
  }
  w.bytes = p - reinterpret_cast<char*>(base);
  return w;
}

// The parameter cache (pfo_tgn_state.pcache): every buffer of `w` that depends on the parameters alone is re-pointed into
// `base` (same sizes as in the workspace, which keeps its own - then unused - copies).  Returns the cache's size.
int64_t pcache_bind(const pfo_tgn_config* c, void* base, Ws& w) {
  const Dims d = dims_of(c);
  char* p = reinterpret_cast<char*>(base);
  const int HCp = d.H * d.Cp, WQ = HCp + d.D;
  const int64_t HCpD = (int64_t)HCp * d.D, DD = (int64_t)d.D * d.D;
  w.pz = take<float>(p, 64);
  for (int l = 1; l <= d.L; ++l) {
    LayerWs& lw = w.layer[l];
    lw.Wqk = take<float>(p, HCpD);
    lw.W1ovT = take<float>(p, HCpD);
    lw.cqk = take<float>(p, (int64_t)HCp + (l == 1 ? d.D : 0));
  }
  w.l1_bias = w.layer[1].cqk;
  w.pz_bytes = (size_t)(p - reinterpret_cast<char*>(w.pz));
  w.pz_in_cache = true;
  w.cosb = take<float>(p, d.D);
  w.iQX = take<char>(p, pfo_bimg_bytes(WQ, d.D));
  if (c->use_memory) {
    w.iWih = take<char>(p, pfo_gru_img_bytes(d.D, d.M));
    w.iWhh = take<char>(p, pfo_gru_img_bytes(d.D, d.D));
  }
  for (int l = 1; l <= d.L; ++l) {
    LayerWs& lw = w.layer[l];
    lw.cq = take<float>(p, d.E);
    lw.W1oT = take<float>(p, (int64_t)d.E * d.D);
    lw.iWqk = take<char>(p, pfo_bimg_bytes(HCp, d.D));   lw.iWqkT = take<char>(p, pfo_bimg_bytes(d.D, HCp));
    lw.iW1ovT = take<char>(p, pfo_bimg_bytes(HCp, d.D)); lw.iW1ov = take<char>(p, pfo_bimg_bytes(d.D, HCp));
    lw.iW1b = take<char>(p, pfo_bimg_bytes(d.D, d.D));   lw.iW1bT = take<char>(p, pfo_bimg_bytes(d.D, d.D));
    lw.iW2 = take<char>(p, pfo_bimg_bytes(d.D, d.D));    lw.iW2T = take<char>(p, pfo_bimg_bytes(d.D, d.D));
    if (l >= 2) {
      lw.T1 = take<float>(p, HCpD);      lw.tq = take<float>(p, HCp);
      lw.Wqk_f = take<float>(p, HCpD);   lw.cqk_f = take<float>(p, HCp);
      lw.W1ovT_f = take<float>(p, HCpD); lw.W1b_f = take<float>(p, DD);   lw.b1_f = take<float>(p, d.D);
    }
  }
  return (int64_t)(p - reinterpret_cast<char*>(base));
}

int check_cfg(const pfo_tgn_config* c) {
  PFO_REQUIRE(c != nullptr, "null config");
  PFO_REQUIRE(c->n_layers >= 1 && c->n_layers <= PFO_MAX_LAYERS, "n_layers must be in [1, 4]");
  PFO_REQUIRE(c->D >= 4 && c->D <= 256 && (c->D % 4) == 0, "D must be a multiple of 4 in [4, 256]");
  PFO_REQUIRE(c->Ef >= 0 && c->Ef <= 64 && (c->Ef % 4) == 0, "Ef must be a multiple of 4 in [0, 64]");
  PFO_REQUIRE(c->n_heads == 1 || c->n_heads == 2 || c->n_heads == 4, "n_heads must be 1, 2 or 4");
  PFO_REQUIRE(((2 * c->D) % c->n_heads) == 0, "n_heads must divide 2D");
  PFO_REQUIRE(c->max_batch >= 0, "bad max_batch");
  PFO_REQUIRE(c->n_nodes >= 2 && c->n_edges_p1 >= 1, "bad graph sizes");
  PFO_REQUIRE(c->max_roots >= 1 && c->max_neighbors >= 1 && c->max_neighbors <= PFO_MAX_NEIGHBORS, "bad capacities");
  return PFO_OK;
}

struct Params {
  const float *tw, *tb, *w_ih, *w_hh, *b_ih, *b_hh;
  struct { const float *wq, *wk, *wv, *b_in, *wo, *bo, *w1, *b1, *w2, *b2; } l[PFO_MAX_LAYERS + 1];
};
struct Grads {
  float *tw, *tb, *w_ih, *w_hh, *b_ih, *b_hh;
  struct { float *wq, *wk, *wv, *b_in, *wo, *bo, *w1, *b1, *w2, *b2; } l[PFO_MAX_LAYERS + 1];
};

template <typename PT, typename FT>
void bind(const pfo_tgn_layout& lay, FT* base, PT& p, int L, bool mem) {
  p.tw = base + lay.time_w; p.tb = base + lay.time_b;
  if (mem) {
    p.w_ih = base + lay.gru_w_ih; p.w_hh = base + lay.gru_w_hh; p.b_ih = base + lay.gru_b_ih; p.b_hh = base + lay.gru_b_hh;
  } else {
    p.w_ih = p.w_hh = p.b_ih = p.b_hh = nullptr;
  }
  for (int l = 1; l <= L; ++l) {
    const pfo_tgn_layer_layout& q = lay.layer[l - 1];
    p.l[l].wq = base + q.wq; p.l[l].wk = base + q.wk; p.l[l].wv = base + q.wv; p.l[l].b_in = base + q.b_in;
    p.l[l].wo = base + q.wo; p.l[l].bo = base + q.bo; p.l[l].w1 = base + q.w1; p.l[l].b1 = base + q.b1;
    p.l[l].w2 = base + q.w2; p.l[l].b2 = base + q.b2;
  }
}

// plain C = A[M,K] * B[N,K]^T (+bias)
PfoGemm g_nt(const float* A, int64_t lda, const int32_t* a_idx, const float* B, int64_t ldb, float* C, int64_t ldc, int M,
             int N, int K, const float* bias) {
  PfoGemm g;
  g.A[0] = A; g.lda[0] = lda; g.a_idx[0] = a_idx; g.B[0] = B; g.ldb[0] = ldb; g.K[0] = K;
  g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.bias = bias;
  return g;
}
// C = A[M,K] * B[K,N]
PfoGemm g_nn(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int M, int N, int K) {
  PfoGemm g = g_nt(A, lda, nullptr, B, ldb, C, ldc, M, N, K, nullptr);
  g.b_kmajor = 1;
  return g;
}
// dW[M,N] += A[K,M]^T * B[K,N]   (rows = K)
PfoGemm g_tn(const float* A, int64_t lda, const float* B, int64_t ldb, const int32_t* b_idx, float* C, int64_t ldc, int M,
             int N, int K, const Ws& w) {
  PfoGemm g = g_nt(A, lda, nullptr, B, ldb, C, ldc, M, N, K, nullptr);
  g.a_kmajor = 1; g.b_kmajor = 1; g.b_idx = b_idx; g.accumulate = 1;
  g.slabs = w.slabs; g.slab_floats = w.slab_floats;
  return g;
}

#define PFO_MAX_DEVICES 16
// Internal side stream: the composite-weight products (forward) and their gradient chain (backward) are tiny
// dependent launches; they run beside the main stream's work and are joined by events where their results are needed.
struct Side {
  hipStream_t s = nullptr, s2 = nullptr;
  hipEvent_t tn_a_done = nullptr, done2 = nullptr, gru_done = nullptr, comp_done = nullptr, pc_a = nullptr, pc_b = nullptr;
  hipEvent_t main_done = nullptr, side_done = nullptr;
  // Deferred work on `s` (a backward end, the optimizer step behind it: pfo_tgn_batch.defer_join) is counted in GENERATIONS:
  // side_gen grows when work is queued there, every joining stream remembers the generation it last waited for.  (One
  // "pending" flag that the first joiner cleared let a prepare call on the prefetch stream consume the join: tgn.join(),
  // state_dict() and the next backward on the caller's stream then read parameters the side stream was still writing.)
  uint64_t side_gen = 0, recorded_gen = 0;
  bool last_backward_deferred = false;                           // pfo_tgn_adam_side may only follow such a backward
  hipStream_t deferred_from = nullptr;                           // the caller's stream of that backward (the only stream the side stream is already ordered behind)
  // Buckets of a side-stream optimizer step in order of FIRST USE (pfo_tgn_adam_side_bucket): early_done fires behind the
  // kernel that finishes the parameters the next forward reads on the caller's stream (time encoder, GRU, layer 1's biases);
  // it stands for the whole side stream as long as nothing but later buckets of the same step was queued behind it.
  hipEvent_t early_done = nullptr, late_done = nullptr;
  uint64_t early_gen = 0;
  bool early_ok = false;
  struct Joined { hipStream_t s; uint64_t gen; } joined[8] = {};
  int n_joined = 0;
  bool pending_for(hipStream_t s) const {
    if (side_gen == 0) return false;
    for (int i = 0; i < n_joined; ++i) if (joined[i].s == s) return joined[i].gen != side_gen;
    return true;
  }
  void mark_joined(hipStream_t s) {
    for (int i = 0; i < n_joined; ++i) if (joined[i].s == s) { joined[i].gen = side_gen; return; }
    if (n_joined < 8) { joined[n_joined].s = s; joined[n_joined].gen = side_gen; ++n_joined; return; }
    for (int i = 1; i < 8; ++i) joined[i - 1] = joined[i];       // more streams than slots: the oldest entry goes (it will wait again)
    joined[7].s = s; joined[7].gen = side_gen;
  }
  hipEvent_t fork = nullptr, done = nullptr, seg_done = nullptr, tn_a = nullptr, tn_b = nullptr, fold_done = nullptr, dh1_sum = nullptr;
  hipEvent_t layer[PFO_MAX_LAYERS + 1] = {};
  bool ok = false;
};
Side& side() {
  // one set per device (streams and events belong to the device that was current when they were made)
  static Side sds[PFO_MAX_DEVICES];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= PFO_MAX_DEVICES) dev = 0;
  Side& sd = sds[dev];
  if (!sd.ok) {
    // The side streams live in the HIGH and the LOW priority class: the runtime multiplexes all streams of a process onto
    // GPU_MAX_HW_QUEUES hardware queues (default 4) PER PRIORITY CLASS, and once a process holds more streams than that (a
    // process group: RCCL, c10d) a normal-priority side stream shares a hardware queue with the caller's stream - every
    // "beside" of this file silently becomes "behind" (rank path at world 1 on RCCL: 1.65 ms per step against 1.39).  In
    // classes of their own they cannot.  (Dispatch priority itself changes nothing measurable: round 3.)  PFO_SIDE_PRIO=0: A/B.
    static const int prio = getenv("PFO_SIDE_PRIO") ? atoi(getenv("PFO_SIDE_PRIO")) : 1;
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);           // lo = least (numerically largest), hi = greatest
    bool good = (prio ? hipStreamCreateWithPriority(&sd.s, hipStreamNonBlocking, hi) : hipStreamCreateWithFlags(&sd.s, hipStreamNonBlocking)) == hipSuccess;
    good = good && (prio ? hipStreamCreateWithPriority(&sd.s2, hipStreamNonBlocking, lo) : hipStreamCreateWithFlags(&sd.s2, hipStreamNonBlocking)) == hipSuccess;
    good = good && hipEventCreateWithFlags(&sd.tn_a_done, hipEventDisableTiming) == hipSuccess;
    good = good && hipEventCreateWithFlags(&sd.done2, hipEventDisableTiming) == hipSuccess;
    good = good && hipEventCreateWithFlags(&sd.fork, hipEventDisableTiming) == hipSuccess;
    good = good && hipEventCreateWithFlags(&sd.done, hipEventDisableTiming) == hipSuccess;
    good = good && hipEventCreateWithFlags(&sd.seg_done, hipEventDisableTiming) == hipSuccess;
    good = good && hipEventCreateWithFlags(&sd.tn_a, hipEventDisableTiming) == hipSuccess;
    good = good && hipEventCreateWithFlags(&sd.tn_b, hipEventDisableTiming) == hipSuccess;
    good = good && hipEventCreateWithFlags(&sd.fold_done, hipEventDisableTiming) == hipSuccess;
    good = good && hipEventCreateWithFlags(&sd.dh1_sum, hipEventDisableTiming) == hipSuccess;
    good = good && hipEventCreateWithFlags(&sd.gru_done, hipEventDisableTiming) == hipSuccess;
    good = good && hipEventCreateWithFlags(&sd.comp_done, hipEventDisableTiming) == hipSuccess;
    good = good && hipEventCreateWithFlags(&sd.pc_a, hipEventDisableTiming) == hipSuccess;
    good = good && hipEventCreateWithFlags(&sd.pc_b, hipEventDisableTiming) == hipSuccess;
    good = good && hipEventCreateWithFlags(&sd.main_done, hipEventDisableTiming) == hipSuccess;
    good = good && hipEventCreateWithFlags(&sd.side_done, hipEventDisableTiming) == hipSuccess;
    good = good && hipEventCreateWithFlags(&sd.early_done, hipEventDisableTiming) == hipSuccess;
    good = good && hipEventCreateWithFlags(&sd.late_done, hipEventDisableTiming) == hipSuccess;
    for (int l = 0; l <= PFO_MAX_LAYERS; ++l) good = good && hipEventCreateWithFlags(&sd.layer[l], hipEventDisableTiming) == hipSuccess;
    sd.ok = good;
  }
  return sd;
}
#define HIPOK(expr, msg) PFO_REQUIRE((expr) == hipSuccess, msg)
// `s` waits for whatever a deferred backward end / side-stream optimizer step left in flight (pfo_tgn_batch.defer_join)
// allow_early (pfo_tgn_forward only): when the side stream's tail is a bucketed optimizer step, the caller's stream waits for
// the first-use bucket alone and is NOT marked joined - everything else the forward takes from the side streams arrives
// through events recorded there behind the later buckets (composite weights, weight images, the fc2 fold), and the next full
// join (the backward's) still waits for all of it.
int side_join(Side& sd, hipStream_t s, bool allow_early = false) {
  if (!sd.pending_for(s)) return PFO_OK;
  if (allow_early && sd.early_ok && sd.early_gen == sd.side_gen) {
    hipStreamCaptureStatus cap0 = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cap0) == hipSuccess && cap0 == hipStreamCaptureStatusNone) {
      HIPOK(hipStreamWaitEvent(s, sd.early_done, 0), "event wait failed");
      return PFO_OK;
    }
  }
  // A capturing stream does not join: a capture starts from a synchronised device (torch.cuda.graph, GraphedTrainStep.capture),
  // so nothing deferred is still in flight, and the calls below would be captured - by the time a forward joins, the side stream
  // belongs to the capture (it waited for the fork event), a record on it would turn side_done into a capture-only event and
  // every later plain wait on it would fail.  (A backward under capture never defers its end: pfo_tgn_backward_ev.)
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) { sd.mark_joined(s); return PFO_OK; }
  if (sd.recorded_gen != sd.side_gen) {                          // one record per generation, shared by every joiner
    HIPOK(hipEventRecord(sd.side_done, sd.s), "event record failed");
    sd.recorded_gen = sd.side_gen;
  }
  HIPOK(hipStreamWaitEvent(s, sd.side_done, 0), "event wait failed");
  sd.mark_joined(s);
  return PFO_OK;
}

#define RUN(expr)                  \
  do {                             \
    int rc__ = (expr);             \
    if (rc__ != PFO_OK) return rc__; \
  } while (0)

}  // namespace

// =============================================================================================
extern "C" int pfo_tgn_param_layout(const pfo_tgn_config* c, pfo_tgn_layout* out) {
  if (int rc = check_cfg(c)) return rc;
  PFO_REQUIRE(out != nullptr, "null output");
  const Dims d = dims_of(c);
  int64_t o = 0;
  auto put = [&](int64_t n) { int64_t r = o; o += n; return r; };
  memset(out, 0, sizeof(*out));
  out->time_w = put(d.D);
  out->time_b = put(d.D);
  if (c->use_memory) {
    out->gru_w_ih = put((int64_t)3 * d.D * d.M);
    out->gru_w_hh = put((int64_t)3 * d.D * d.D);
    out->gru_b_ih = put(3 * d.D);
    out->gru_b_hh = put(3 * d.D);
  } else {
    out->gru_w_ih = out->gru_w_hh = out->gru_b_ih = out->gru_b_hh = -1;
  }
  for (int l = 0; l < d.L; ++l) {
    pfo_tgn_layer_layout& q = out->layer[l];
    q.wq = put((int64_t)d.E * d.E);
    q.wk = put((int64_t)d.E * d.C);
    q.wv = put((int64_t)d.E * d.C);
    q.b_in = put(3 * d.E);
    q.wo = put((int64_t)d.E * d.E);
    q.bo = put(d.E);
    q.w1 = put((int64_t)d.D * (d.E + d.D));
    q.b1 = put(d.D);
    q.w2 = put((int64_t)d.D * d.D);
    q.b2 = put(d.D);
  }
  out->total = o;
  return PFO_OK;
}

extern "C" int64_t pfo_tgn_workspace_bytes(const pfo_tgn_config* c) {
  if (check_cfg(c) != PFO_OK) return -1;
  return carve(c, nullptr).bytes + 256;
}

extern "C" int pfo_tgn_debug_views(const pfo_tgn_config* c, void* workspace, pfo_tgn_debug* out) {
  if (int rc = check_cfg(c)) return rc;
  PFO_REQUIRE(workspace && out, "null argument");
  const Ws w = carve(c, workspace);
  out->n_touched = w.n_touched; out->touched_ids = w.touched; out->h0_table = w.h0_tab; out->slot = w.slot;
  const Dims d = dims_of(c);
  out->n_core = w.n_core;
  out->l1_ctx = w.layer[1].ctx;
  out->l1_dh1 = d.L == 1 ? w.dh1 : w.dH[1];       // the top layer writes dh1 itself, below it the layer above does
  out->l1_dW1ovT = w.layer[1].dW1ovT;
  out->gru_dgi = c->use_memory ? w.gi : nullptr;
  out->gru_msg_rows = c->use_memory ? w.msg_rows : nullptr;
  out->Cp = d.Cp;
  return PFO_OK;
}

static int level_sizes(const pfo_tgn_config* c, const pfo_tgn_batch* b, int64_t* n) {
  PFO_REQUIRE(b != nullptr, "null batch");
  PFO_REQUIRE(b->R >= 1 && b->R <= c->max_roots, "R exceeds the workspace capacity (max_roots)");
  PFO_REQUIRE(b->K >= 1 && b->K <= c->max_neighbors, "K exceeds the workspace capacity (max_neighbors)");
  n[c->n_layers] = b->R;
  for (int l = c->n_layers; l >= 1; --l) n[l - 1] = n[l] * (1 + (int64_t)b->K);
  PFO_REQUIRE(n[0] < (int64_t)1 << 31, "too many level-0 references");
  return PFO_OK;
}

// Replicas of the level-0 gradient table the layer-1 attention backward adds into: per-XCD replicas pay off only for the
// per-instance atomics (uniform sampling: 4 replicas 0.537 -> 0.506 ms); the run-merged kernel issues 2-3x fewer and measures
// best on ONE table (1.656 vs 1.665 ms/step); the deterministic int64 table is always one
static int grad_replicas(const pfo_tgn_config* c, const pfo_tgn_batch* b) {
  const int det = b->deterministic ? 1 : 0;
  return (det || (c->use_memory && !b->uniform && pfo_attn_bwd_runs_possible(b->K, c->D, c->n_heads))) ? 1 : PFO_GRAD_REPLICAS;
}
// Layer-1 backward, run-merged attention kernel, non-deterministic calls (PFO_DQ_ATOMIC=1): the attention kernel adds the
// query-side gradient rows straight into the per-table-row sums Dq[:, :H Cp] (float atomics, attn.hpp dq_rows) and the d h1
// half Dq[:, H Cp:] is summed on the side stream beside it - no segment-sum pass on the serial tail.  OFF by default: measured
// at C2 (round 5, one box, interleaved) 1.382 / 1.392 ms per step against 1.302 / 1.303 - the ~20 k extra rows of atomics
// cost the attention kernel 36 us (270 -> 306 alone), and the table-row contraction that now follows it directly runs
// beside the tail of the instance weight-gradient launch (114 us instead of 49) where the HBM-bound segment sum used to
// overlap with it for free: this phase is bound by the chip's total work, not by the order of its launches.
static bool dq_atomic_mode(const pfo_tgn_config* c, const pfo_tgn_batch* b) {
  static const int on = getenv("PFO_DQ_ATOMIC") ? atoi(getenv("PFO_DQ_ATOMIC")) : 0;          // A/B switch
  return on && c->use_memory && !b->uniform && !b->deterministic && pfo_attn_bwd_runs_possible(b->K, c->D, c->n_heads);
}
// What the backward's layer-1 kernels need and that depends on the sampled levels alone: the cleared level-0 gradient rows and
// the layer-1 instances grouped by the touched-table row they sit on.  Queued by the backward on its side stream - or, for
// calls with pfo_tgn_batch.seg_in_forward, by the forward on ITS side stream (beside layer 1, joined by the event layer 2 waits
// for anyway), which takes one event record and one wait off the backward's critical path.
static int seg_prologue(const pfo_tgn_config* c, const pfo_tgn_batch* b, const Ws& w, const Dims& d, const int64_t* n, hipStream_t ss) {
  const int capP = (int)std::min<int64_t>(c->n_nodes, n[0] + b->n_extra);
  const int det = b->deterministic ? 1 : 0;
  const int64_t rep_stride = (int64_t)d.capP * d.D;
  // (deterministic: the table holds int64 fixed-point sums - rows of 2 D floats' worth)
  if (c->use_memory) RUN(pfo_zero_rows_launch(w.d_h0, w.n_core, capP, det ? 2 * d.D : d.D, grad_replicas(c, b), rep_stride, ss));
  if (dq_atomic_mode(c, b)) RUN(pfo_zero_rows_launch(w.Dq, w.n_core, capP, d.H * d.Cp + d.D, 1, 0, ss));   // the row sums the attention backward adds into
  RUN(pfo_seg_build_launch(w.idx0, w.nodes[0], (int)n[1], capP, b->uniform ? nullptr : w.cnt1, w.seg_ptr, w.seg_cur,
                           w.seg_tmp, w.seg_mem, w.seg_of, w.seg_scratch, ss));
  return PFO_OK;
}

// persist the lazily updated memory of the batch's positives + store their raw messages (tgn.py:290-317, 357-378)
static int state_update(const pfo_tgn_config* c, const pfo_tgn_state* st, const Ws& w, const int32_t* src, const int32_t* dst,
                        const double* ts, const int32_t* eidx, int32_t B, hipStream_t s) {
  pfo_tgn_layout lay;
  RUN(pfo_tgn_param_layout(c, &lay));
  const float* tw = st->params + lay.time_w;
  const float* tb = st->params + lay.time_b;
  // (the last-message-wins test needs the per-node table only for very large batches: memory.hip MSG_INLINE_MAX)
  int32_t* winner = pfo_msg_store_needs_winner(B) ? w.winner : nullptr;
  RUN(pfo_persist_launch(src, dst, B, w.slot, w.upd_mem, st->has_msg, st->msg_time, st->memory, st->last_update, c->D, winner, s));
  RUN(pfo_msg_store_launch(src, dst, ts, eidx, B, st->memory, st->last_update, st->edge_feat, tw, tb, c->D, c->Ef,
                           st->msg_table, st->msg_time, st->has_msg, winner, s));
  return PFO_OK;
}

// ---- the part of a forward call that depends on neither parameters nor gradients, in two halves (pfo_tgn_forward runs the
// GRU's weight-image launch between them; pfo_tgn_prepare runs them back to back on any stream, ahead of time)
static int prepare_sample(const pfo_tgn_config* c, const pfo_tgn_state* st, const pfo_tgn_batch* b, const Ws& w, const int64_t* n,
                          hipStream_t s) {
  const int L = c->n_layers, K = b->K;
  void* stream = (void*)s;
  // The compaction's flags are cleared on THIS stream (4 us): a wait for a side-stream memset costs the waiting stream 5-17 us
  // on this part (r3 timeline), more than the memset itself.  Same reasoning for the GRU's two weight images below.
  // With two or more levels the UPPER level's sampler launch clears the flags (it does not mark) and the memset is gone
  // (fwd.begin -> fwd.sampled 40.4 -> 35.6 us).  Both levels of most-recent sampling in ONE launch (a workgroup per root: its
  // own query, then the 1 + K queries that depend on it) was built and measured too: it marks, so the memset comes back, and
  // the step does not move (1.302 / 1.308 against 1.297 / 1.305 ms, round 5) - removed again.
  // (only while the clear stays a small share of that launch: it runs on ceil(n * 16 / 256) workgroups, a memset on the
  //  whole chip - beyond ~4 K ints per workgroup (large node tables under small batches) the memset is the faster head)
  const int64_t clear_blocks = std::max<int64_t>(1, pfo_ceil_div(n[L] * 16, 256));
  const bool clear_in_sampler = L >= 2 && (w.mark_bytes % 4) == 0 && (int64_t)(w.mark_bytes / 4) <= 4096 * clear_blocks;
  if (!clear_in_sampler) HIPOK(hipMemsetAsync(w.mark, 0, w.mark_bytes, s), "memset failed");
  // ---- frontier expansion: K1 per level (utils.py:163-219 called from embedding_module.py:125).  Enqueued before the
  // side-stream work below: it depends on nothing else, and the GPU samples while the host is still enqueueing
  for (int l = L; l >= 1; --l) {
    const int64_t* dr = (b->uniform == 1) ? b->draws[L - l] : nullptr;
    PFO_REQUIRE(b->uniform != 1 || dr, "missing draws for a level");
    // level L reads the caller's roots directly; every level writes [its own nodes | their neighbours] as the next one
    const int32_t* lvl_nodes = (l == L) ? b->roots : w.nodes[l];
    const double* lvl_ts = (l == L) ? b->root_ts : w.ts[l];
    // the last launch writes the whole level-0 list [S_1 ; neighbours(S_1)]: it also sets the touched-node flags, so the
    // compaction needs no marking pass
    RUN(pfo_tnbr_sample_dev(st->indptr, st->adj_nbr, st->adj_eidx, st->adj_ts, c->n_nodes, lvl_nodes, lvl_ts, n[l], K,
                            b->uniform, dr, b->seed, b->offset + (uint64_t)l * 0x100000000ull, b->offset_dev, nullptr,
                            w.eidx[l], nullptr, w.dt[l], w.nodes[l - 1], l > 1 ? w.ts[l - 1] : nullptr, l == 1 ? w.mark : nullptr,
                            l == 1 ? w.cnt1 : nullptr, stream, (clear_in_sampler && l == L) ? w.mark : nullptr,
                            (clear_in_sampler && l == L) ? (int64_t)(w.mark_bytes / 4) : 0));
  }

  return PFO_OK;
}
static int prepare_compact(const pfo_tgn_config* c, const pfo_tgn_batch* b, const Ws& w, hipStream_t s) {
  // slot[v] = row of v in the per-step tables (roots and every sampled neighbour, all levels; + the caller's extra nodes)
  RUN(pfo_touch_compact_launch(nullptr, 0, b->extra_nodes, b->n_extra, c->n_nodes, w.mark, w.slot, w.touched, w.n_touched,
                               w.scan, true, true, s));
  return PFO_OK;
}
static int prepare_pack(const pfo_tgn_config* c, const pfo_tgn_state* st, const pfo_tgn_batch* b, const Ws& w, const Dims& d,
                        const int64_t* n, hipStream_t s) {
  const int D = d.D;
  const int capP = (int)std::min<int64_t>(c->n_nodes, n[0] + b->n_extra);
  // ---- level-0 rows of the touched nodes: memory' + node features (embedding_module.py:93-98), memory' = the lazily
  // updated memory (tgn.py:251; memory_updater.py:35-53).  One launch copies the rows the GRU reads (and the backward reads
  // again after the state update has overwritten the tables) and translates the level-0 list into table rows
  // (idx0[i] = row of the i-th level-0 reference).
  if (c->use_memory)
    RUN(pfo_pack_remap_launch(st->msg_table, d.M, st->memory, D, st->has_msg, w.touched, w.n_touched, capP, w.msg_rows,
                              w.h_rows, w.hm, w.nodes[0], n[0], w.slot, w.idx0, s));
  else
    RUN(pfo_pack_remap_launch(nullptr, d.M, st->node_feat, D, nullptr, w.touched, w.n_touched, capP, nullptr, w.h0_tab, nullptr,
                              w.nodes[0], n[0], w.slot, w.idx0, s));
  return PFO_OK;
}
static int prepare_compact_pack(const pfo_tgn_config* c, const pfo_tgn_state* st, const pfo_tgn_batch* b, const Ws& w, const Dims& d,
                                const int64_t* n, hipStream_t s) {
  RUN(prepare_compact(c, b, w, s));
  return prepare_pack(c, st, b, w, d, n, s);
}

// ---- Everything the step derives from the PARAMETERS alone, in two stages of small dependent launches on one stream
// (pfo_tgn_forward runs them on its side stream when the parameter cache is absent or stale; pfo_tgn_refresh right behind the
// optimizer).  Stage A: cos(b), the composite weights of every layer and the weight images layer 1 and the top layer's fc2
// need.  Stage B ("fc2 fold"): the composites of the layers >= 2 with the fc2 of the layer below folded in, and their images.
static int build_gru_images(const pfo_tgn_config* c, const Dims& d, const Ws& w, const Params& P, hipStream_t s) {
  // gate-ordered rows: r | z | n_i | n_h per 16 hidden units (gemm.hip gru_fused_kernel)
  const int D = d.D;
  PfoBimg im[2];
  im[0].src = P.w_ih; im[0].ld = d.M; im[0].N = 3 * D; im[0].K = d.M; im[0].trans = 0; im[0].dst = w.iWih;
  im[1].src = P.w_hh; im[1].ld = D; im[1].N = 3 * D; im[1].K = D; im[1].trans = 0; im[1].dst = w.iWhh;
  im[0].gate = 1; im[0].gate_D = D; im[1].gate = 2; im[1].gate_D = D;
  (void)c;
  return pfo_bimg_launch(im, 2, s);
}
static int build_stage_a(const Dims& d, const Ws& w, const Params& P, hipStream_t ss) {
  const int L = d.L, D = d.D, H = d.H, E = d.E, C = d.C, dh = d.dh, Cp = d.Cp, HCp = d.H * d.Cp;
  // 64 zero floats | Wqk / W1ovT / cqk of every layer: the builds below OVERWRITE the live rows and never touch the padding
  // rows / columns, which must be zero - in the workspace (recycled memory) that takes a memset per build, in the parameter
  // cache the caller cleared the buffer once when it allocated it (pfo_tgn_state.pcache: "zero-initialised")
  if (!w.pz_in_cache) HIPOK(hipMemsetAsync(w.pz, 0, w.pz_bytes, ss), "memset failed");
  RUN(pfo_time_encode(w.pz, 1, P.tw, P.tb, D, w.cosb, ss));                     // cos(fma(0, w, b)) (embedding_module.py:92)
  PfoGemm st1[3 * PFO_MAX_LAYERS], st2[4 * PFO_MAX_LAYERS];
  PfoBimg im[8 * PFO_MAX_LAYERS + 2];
  int n1 = 0, n2 = 0, ni = 0;
  auto img = [&](const float* src, int64_t ld, int N_, int K_, int trans, void* dst) {
    im[ni].src = src; im[ni].ld = ld; im[ni].N = N_; im[ni].K = K_; im[ni].trans = trans; im[ni].dst = dst; ++ni;
  };
  for (int l = 1; l <= L; ++l) {
    const LayerWs& lw = w.layer[l];
    const auto& p = P.l[l];
    PfoGemm* a1 = st1 + n1;
    a1[0] = g_nt(w.cosb, D, nullptr, p.wq + D, E, lw.cq, E, 1, E, D, p.b_in);            // cq = Wq[:, D:] cos(b) + bq
    a1[1] = g_nt(p.wo, E, nullptr, p.w1, E + D, lw.W1oT, D, E, D, E, nullptr);           // W1oT = Wo^T W1[:, :E]^T
    a1[1].a_kmajor = 1;
    a1[2] = g_nt(p.wk, C, nullptr, p.wq, E, lw.Wqk, D, C, D, dh, nullptr);               // Wqk_h = Wk_h^T Wq_h[:, :D]
    a1[2].a_kmajor = 1; a1[2].b_kmajor = 1; a1[2].batch = H;
    a1[2].a_bs[0] = (int64_t)dh * C; a1[2].b_bs[0] = (int64_t)dh * E; a1[2].c_bs = (int64_t)Cp * D;
    n1 += 3;
    PfoGemm* a2 = st2 + n2;
    a2[0] = g_nn(lw.cq, E, p.wk, C, lw.cqk, HCp, 1, C, dh);                               // cqk_h = Wk_h^T cq_h
    a2[0].batch = H; a2[0].a_bs[0] = dh; a2[0].b_bs[0] = (int64_t)dh * C; a2[0].c_bs = Cp;
    a2[1] = g_nt(p.wv, C, nullptr, lw.W1oT, D, lw.W1ovT, D, C, D, dh, nullptr);           // W1ovT_h = Wv_h^T W1oT_h
    a2[1].a_kmajor = 1; a2[1].b_kmajor = 1; a2[1].batch = H;
    a2[1].a_bs[0] = (int64_t)dh * C; a2[1].b_bs[0] = (int64_t)dh * D; a2[1].c_bs = (int64_t)Cp * D;
    a2[2] = g_nn(p.b_in + 2 * E, dh, lw.W1oT, D, lw.W1ovT + (int64_t)C * D, D, 1, D, dh); // row C: (W1 Wo_h bv_h)^T
    a2[2].batch = H; a2[2].a_bs[0] = dh; a2[2].b_bs[0] = (int64_t)dh * D; a2[2].c_bs = (int64_t)Cp * D;
    a2[3] = g_nt(p.bo, E, nullptr, p.w1, E + D, lw.W1ovT + (int64_t)(C + 1) * D, D, 1, D, E, nullptr);   // row C+1 of head 0: (W1 bo)^T
    n2 += 4;
    // fp16 images of this layer's weight operands, in both orientations (forward and data-gradient launches).
    // Layers >= 2 take theirs from the fc2-folded composites (stage B).
    if (l == 1) {
      // layer 1 projects the touched-node table once: [Wqk ; W1[:, E:]] stacked along the output dimension
      img(lw.Wqk, D, HCp, D, 0, w.iQX);        im[ni - 1].row0 = 0;   im[ni - 1].rows_total = HCp + D;
      img(p.w1 + E, E + D, D, D, 0, w.iQX);    im[ni - 1].row0 = HCp; im[ni - 1].rows_total = HCp + D; im[ni - 1].last = 1;
      img(lw.Wqk, D, D, HCp, 1, lw.iWqkT);
      img(lw.W1ovT, D, HCp, D, 0, lw.iW1ovT);  img(lw.W1ovT, D, D, HCp, 1, lw.iW1ov);
      img(p.w1 + E, E + D, D, D, 1, lw.iW1bT);
    }
    if (l == L) { img(p.w2, D, D, D, 0, lw.iW2); img(p.w2, D, D, D, 1, lw.iW2T); }   // only the top layer applies its fc2
  }
  for (int i = 0; i < n1; i += PFO_GEMM_MULTI_MAX) RUN(pfo_gemm_multi_launch(st1 + i, std::min(PFO_GEMM_MULTI_MAX, n1 - i), ss));
  for (int i = 0; i < n2; i += PFO_GEMM_MULTI_MAX) RUN(pfo_gemm_multi_launch(st2 + i, std::min(PFO_GEMM_MULTI_MAX, n2 - i), ss));
  for (int i = 0; i < ni; i += PFO_BIMG_MAX) RUN(pfo_bimg_launch(im + i, std::min(PFO_BIMG_MAX, ni - i), ss));
  return PFO_OK;
}
// fc2 fold.  A layer l >= 2 reads rows of the previous layer, out = A h + b (A = W2, b = b2 of layer l-1, h = that
// layer's h1 row).  Every use of such a row is linear, so A and b move into THIS layer's composites and the previous
// layer's fc2 contraction over all its instances (and, backward, d h1 = d out W2 and dW2 = d out^T h1) disappears:
//   query side   qk'_h = Q_h x + cqk_h,  x = A s + b        ->  T1_h = Q_h A,  t_h = Q_h b + cqk_h
//   scores       qk'_h,node . (A k + b) = (A^T qk'_h,node) . k + const(j)   (the constant drops out of the softmax)
//                                                         ->  Q_f,h = [A^T T1_h,node ; T1_h,edge|time],  c_f,h likewise from t_h
//   context      sum_j a'_j (A k_j + b) = A c + b sum_j a'_j  ->  V_f,h,node = A^T V_h,node,  V_f,h[C] = V_h[C] + b^T V_h,node
//   x term       W1b (A s + b)                              ->  W1b_f = W1b A,  b1_f = b1 + W1b b
// (Q_h = Wqk_h, V_h = W1ovT_h; tiny products, two more dependent stages.)
static int build_stage_b(const Dims& d, const Ws& w, const Params& P, hipStream_t ss) {
  const int L = d.L, D = d.D, H = d.H, E = d.E, C = d.C, Cp = d.Cp, HCp = d.H * d.Cp;
  if (L < 2) return PFO_OK;
  PfoGemm f1[4 * PFO_MAX_LAYERS], f2[4 * PFO_MAX_LAYERS];
  PfoBimg fim[6 * PFO_MAX_LAYERS];
  int m1 = 0, m2 = 0, mi = 0;
  const int64_t HCpD = (int64_t)HCp * D;
  for (int l = 2; l <= L; ++l) {
    const LayerWs& lw = w.layer[l];
    const auto& p = P.l[l];
    const float* A = P.l[l - 1].w2;
    const float* bv = P.l[l - 1].b2;
    PfoGemm* a = f1 + m1;
    a[0] = g_nn(lw.Wqk, D, A, D, lw.T1, D, HCp, D, D);                                    // T1 = Q A (all heads' rows)
    a[1] = g_nt(bv, D, nullptr, lw.Wqk, D, lw.tq, HCp, 1, HCp, D, lw.cqk);               // t = Q b + cqk
    a[2] = g_nn(p.w1 + E, E + D, A, D, lw.W1b_f, D, D, D, D);                            // W1b_f = W1b A
    a[3] = g_nt(bv, D, nullptr, p.w1 + E, E + D, lw.b1_f, D, 1, D, D, p.b1);             // b1_f = b1 + W1b b
    m1 += 4;
  }
  for (int i = 0; i < m1; i += PFO_GEMM_MULTI_MAX) RUN(pfo_gemm_multi_launch(f1 + i, std::min(PFO_GEMM_MULTI_MAX, m1 - i), ss));
  for (int l = 2; l <= L; ++l) {
    const LayerWs& lw = w.layer[l];
    const float* A = P.l[l - 1].w2;
    const float* bv = P.l[l - 1].b2;
    static const int grouped_copy = getenv("PFO_GROUPED_COPY") ? atoi(getenv("PFO_GROUPED_COPY")) : 1;      // A/B switch
    if (!grouped_copy) {
      HIPOK(hipMemcpyAsync(lw.W1ovT_f, lw.W1ovT, HCpD * sizeof(float), hipMemcpyDeviceToDevice, ss), "copy failed");
      HIPOK(hipMemcpyAsync(lw.Wqk_f, lw.T1, HCpD * sizeof(float), hipMemcpyDeviceToDevice, ss), "copy failed");   // edge | time rows stay
      HIPOK(hipMemcpyAsync(lw.cqk_f, lw.tq, (size_t)HCp * sizeof(float), hipMemcpyDeviceToDevice, ss), "copy failed");
    } else {
      // the rows the fold does not touch (edge | time, the two bias rows) come over as they are - ONE grouped launch instead of
      // three runtime blits (each a launch of its own in this chain of small dependent launches)
      PfoSumSlabs cp[3];
      cp[0].dst = lw.W1ovT_f; cp[0].src = lw.W1ovT; cp[0].count = HCpD;
      cp[1].dst = lw.Wqk_f;   cp[1].src = lw.T1;    cp[1].count = HCpD;
      cp[2].dst = lw.cqk_f;   cp[2].src = lw.tq;    cp[2].count = HCp;
      for (int q = 0; q < 3; ++q) { cp[q].n_slabs = 1; cp[q].accumulate = 0; cp[q].stride = 0; }
      RUN(pfo_sum_slabs_launch(cp, 3, ss));
    }
    PfoGemm* a = f2 + m2;
    a[0] = g_nn(A, D, lw.T1, D, lw.Wqk_f, D, D, D, D);        a[0].a_kmajor = 1;          // Q_f,node = A^T T1_node
    a[0].batch = H; a[0].b_bs[0] = (int64_t)Cp * D; a[0].c_bs = (int64_t)Cp * D;
    a[1] = g_nn(lw.tq, D, A, D, lw.cqk_f, D, 1, D, D);                                    // c_f,node = A^T t_node
    a[1].batch = H; a[1].a_bs[0] = Cp; a[1].c_bs = Cp;
    a[2] = g_nn(A, D, lw.W1ovT, D, lw.W1ovT_f, D, D, D, D);   a[2].a_kmajor = 1;          // V_f,node = A^T V_node
    a[2].batch = H; a[2].b_bs[0] = (int64_t)Cp * D; a[2].c_bs = (int64_t)Cp * D;
    a[3] = g_nn(bv, D, lw.W1ovT, D, lw.W1ovT_f + (int64_t)C * D, D, 1, D, D);             // V_f[C] = V[C] + b^T V_node
    a[3].batch = H; a[3].b_bs[0] = (int64_t)Cp * D; a[3].c_bs = (int64_t)Cp * D; a[3].accumulate = 1;
    m2 += 4;
    auto fimg = [&](const float* src, int64_t ld, int N_, int K_, int trans, void* dst) {
      fim[mi].src = src; fim[mi].ld = ld; fim[mi].N = N_; fim[mi].K = K_; fim[mi].trans = trans; fim[mi].dst = dst; ++mi;
    };
    fimg(lw.Wqk_f, D, HCp, D, 0, lw.iWqk);      fimg(lw.W1b_f, D, D, D, 0, lw.iW1b);
    fimg(lw.Wqk_f, D, D, HCp, 1, lw.iWqkT);     fimg(lw.W1b_f, D, D, D, 1, lw.iW1bT);
    fimg(lw.W1ovT_f, D, HCp, D, 0, lw.iW1ovT);  fimg(lw.W1ovT_f, D, D, HCp, 1, lw.iW1ov);
  }
  for (int i = 0; i < m2; i += PFO_GEMM_MULTI_MAX) RUN(pfo_gemm_multi_launch(f2 + i, std::min(PFO_GEMM_MULTI_MAX, m2 - i), ss));
  for (int i = 0; i < mi; i += PFO_BIMG_MAX) RUN(pfo_bimg_launch(fim + i, std::min(PFO_BIMG_MAX, mi - i), ss));
  return PFO_OK;
}

// =============================================================================================
extern "C" int64_t pfo_tgn_pcache_bytes(const pfo_tgn_config* c) {
  if (check_cfg(c) != PFO_OK) return -1;
  Ws w;
  memset(&w, 0, sizeof(w));
  return pcache_bind(c, nullptr, w) + 256;
}

extern "C" int pfo_tgn_refresh(const pfo_tgn_config* c, const pfo_tgn_state* st, void* stream) {
  if (int rc = check_cfg(c)) return rc;
  PFO_REQUIRE(st && st->params && st->pcache, "pfo_tgn_refresh needs parameters and a parameter cache");
  const Dims d = dims_of(c);
  Ws w;
  memset(&w, 0, sizeof(w));
  pcache_bind(c, st->pcache, w);
  pfo_tgn_layout lay;
  RUN(pfo_tgn_param_layout(c, &lay));
  Params P;
  bind(lay, st->params, P, d.L, c->use_memory != 0);
  Side& sd = side();
  PFO_REQUIRE(sd.ok, "could not create the side stream");
  hipStream_t s = (hipStream_t)stream, sr = sd.s2;
  PfoRange range("pfo_tgn_refresh");
  RUN(side_join(sd, s));
  // Forked from the caller's stream (behind the kernel that wrote the parameters, and behind every reader of the old
  // composites: the previous backward joined its side streams into that stream) onto the SECOND side stream, not joined.
  // The next forward's first side stream - which packs the touched rows as soon as the compaction is done - waits for pc_a
  // only in front of the event layer 1 waits for anyway, and for pc_b in front of the one layer 2 waits for: the caller's
  // stream itself never waits for this chain, and the row pack does not queue behind it (on one stream the ~15 dependent
  // launches of both stages sat in front of the pack: +45 us per step, measured).
  // (the GRU's two images stay on the caller's stream - one 4 us launch, as in the forward: the fused GRU launch reads them
  //  on that stream BEFORE the forward's wait for the side stream)
  if (c->use_memory) RUN(build_gru_images(c, d, w, P, s));
  HIPOK(hipEventRecord(sd.fork, s), "event record failed");
  HIPOK(hipStreamWaitEvent(sr, sd.fork, 0), "event wait failed");
  RUN(build_stage_a(d, w, P, sr));
  HIPOK(hipEventRecord(sd.pc_a, sr), "event record failed");
  RUN(build_stage_b(d, w, P, sr));
  HIPOK(hipEventRecord(sd.pc_b, sr), "event record failed");
  return PFO_OK;
}

// =============================================================================================
extern "C" int pfo_tgn_prepare(const pfo_tgn_config* c, const pfo_tgn_state* st, const pfo_tgn_batch* b, void* workspace,
                               void* stream) {
  if (int rc = check_cfg(c)) return rc;
  PFO_REQUIRE(st && workspace && b, "null argument");
  PFO_REQUIRE(st->indptr && st->adj_nbr && st->adj_eidx && st->adj_ts && st->node_feat, "null state");
  PFO_REQUIRE(!c->use_memory || (st->memory && st->msg_table && st->has_msg), "null memory state");
  int64_t n[PFO_MAX_LAYERS + 1];
  RUN(level_sizes(c, b, n));
  PFO_REQUIRE(b->roots && b->root_ts, "null batch arrays");
  PFO_REQUIRE(b->uniform >= 0 && b->uniform <= 2, "bad sampling mode");
  PFO_REQUIRE(b->uniform != 1 || b->draws, "mode 1 needs draws");
  PFO_REQUIRE(b->n_extra >= 0 && b->n_extra <= 2 * c->max_batch, "n_extra exceeds 2 * max_batch");
  PFO_REQUIRE(b->n_extra == 0 || b->extra_nodes, "null extra_nodes");
  const Dims d = dims_of(c);
  const Ws w = carve(c, workspace);
  PfoRange range("pfo_tgn_prepare");
  RUN(side_join(side(), (hipStream_t)stream));
  RUN(prepare_sample(c, st, b, w, n, (hipStream_t)stream));
  RUN(prepare_compact_pack(c, st, b, w, d, n, (hipStream_t)stream));
  return PFO_OK;
}

// =============================================================================================
extern "C" int pfo_tgn_forward(const pfo_tgn_config* c, const pfo_tgn_state* st, const pfo_tgn_batch* b, void* workspace,
                               float* emb_out, void* stream) {
  if (int rc = check_cfg(c)) return rc;
  PFO_REQUIRE(st && workspace && emb_out, "null argument");
  PFO_REQUIRE(st->indptr && st->adj_nbr && st->adj_eidx && st->adj_ts && st->node_feat && st->params, "null state");
  PFO_REQUIRE(c->Ef == 0 || st->edge_feat, "null edge features");
  PFO_REQUIRE(!c->use_memory || (st->memory && st->last_update && st->msg_table && st->msg_time && st->has_msg),
              "null memory state");
  int64_t n[PFO_MAX_LAYERS + 1];
  RUN(level_sizes(c, b, n));
  PFO_REQUIRE(b->roots && b->root_ts, "null batch arrays");
  PFO_REQUIRE(b->uniform >= 0 && b->uniform <= 2, "bad sampling mode");
  PFO_REQUIRE(b->uniform != 1 || b->draws, "mode 1 needs draws");
  const Dims d = dims_of(c);
  Ws w_ = carve(c, workspace);
  if (st->pcache) pcache_bind(c, st->pcache, w_);       // parameter-only buffers live in the caller's cache
  const Ws& w = w_;
  const bool build = !st->pcache || !st->pcache_valid;  // composites / images (re)built by this call
  hipStream_t s = (hipStream_t)stream;
  pfo_tgn_layout lay;
  RUN(pfo_tgn_param_layout(c, &lay));
  Params P;
  bind(lay, st->params, P, d.L, c->use_memory != 0);
  const int L = d.L, D = d.D, Ef = d.Ef, H = d.H, E = d.E, C = d.C, dh = d.dh, K = b->K;
  (void)C; (void)dh; (void)E;

  PfoRange range_call(b->training ? "pfo_tgn_forward (training)" : "pfo_tgn_forward");
  // ---- composite weights of every layer: on the side stream, beside the sampling / memory phase
  Side& sd = side();
  PFO_REQUIRE(sd.ok, "could not create the side stream");
  hipStream_t ss = sd.s;
  const int Cp = d.Cp, HCp = H * d.Cp;
  // Everything on the side stream depends on the parameters only, so it may start at once (after whatever the main
  // stream did before this call); all layers share each launch.
  hipStreamCaptureStatus cap_early = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &cap_early) != hipSuccess) cap_early = hipStreamCaptureStatusActive;
  const bool bind_events = cap_early == hipStreamCaptureStatusNone;      // events bound to a launch (common.hpp): not under capture
  PFO_MARK("fwd.begin", s);
  // (with a deferred backward end in flight the side stream already waits for the caller's stream's last launch of that
  //  backward, and everything this call gives it either reads parameters only or sits behind an event of this call: no fork)
  static const int skip_fork = getenv("PFO_SKIP_FORK") ? atoi(getenv("PFO_SKIP_FORK")) : 1;      // A/B switch
  // (only for the stream that queued the deferred backward: the side stream waits for THAT stream's last launch; a writer of
  //  the parameters on any other stream - a native optimizer step there, a graph replay - is ordered by the fork below)
  if (!(skip_fork && sd.pending_for(s) && sd.deferred_from == s && bind_events)) {
    HIPOK(hipEventRecord(sd.fork, s), "event record failed");
    HIPOK(hipStreamWaitEvent(ss, sd.fork, 0), "event wait failed");
  }
  // The GRU's two weight images depend on the parameters alone.  With a deferred backward end + optimizer step in flight on
  // the side stream they are made THERE, right behind that step's kernel (this call is queued while the device still runs the
  // backward) - the join below covers them - instead of on the caller's stream between the join and the lazy GRU (6 us + a
  // launch gap of the step's serial head).  PFO_GRU_IMG_SIDE=1: A/B (off: +6 us, as round 4 found for the refresh path).
  static const int early_join = getenv("PFO_EARLY_JOIN") ? atoi(getenv("PFO_EARLY_JOIN")) : 1;      // A/B switch
  static const int gru_img_side = getenv("PFO_GRU_IMG_SIDE") ? atoi(getenv("PFO_GRU_IMG_SIDE")) : 0;   // (measured: 1.2238 with, 1.2173 without - off)
  const bool early_path = early_join != 0 && c->n_layers >= 2 && sd.early_ok && sd.early_gen == sd.side_gen;
  bool gru_img_done = false;
  if (gru_img_side && c->use_memory && build && bind_events && sd.pending_for(s) && sd.deferred_from == s && !early_path) {
    RUN(build_gru_images(c, d, w, P, ss));
    gru_img_done = true;
  }
  if (!b->prepared) RUN(prepare_sample(c, st, b, w, n, s));
  PFO_MARK("fwd.sampled", s);
  // a deferred backward end + optimizer step of the previous step may still be running on the side stream: the sampling above
  // reads neither its buffers nor the parameters; everything below does (compaction counts, the GRU's weights ...)
  // (early: needs the fused state update's ordering - persist / message store run on the side stream behind every bucket -
  //  and layers >= 2 wait for fold_done, recorded there too; the top layer's raw b2 is the only late-bucket value this stream reads)
  RUN(side_join(sd, s, early_join != 0 && c->n_layers >= 2));
  PFO_MARK("fwd.side_joined", s);
  const bool fused_state = b->upd_src != nullptr && c->use_memory && L >= 2;
  static const int side_late_env = getenv("PFO_FWD_SIDE_LATE") ? atoi(getenv("PFO_FWD_SIDE_LATE")) : 0;   // EXPERIMENT switch
  const bool side_late = side_late_env != 0 && bind_events;
  PFO_REQUIRE(!fused_state || (b->upd_dst && b->upd_ts && b->upd_eidx && b->upd_B >= 1), "bad state-update arguments");

  // the GRU contractions come first on the main stream: their two weight images are made there too (one 4 us launch)
  if (c->use_memory && build && !gru_img_done) RUN(build_gru_images(c, d, w, P, s));
  bool composites_awaited = false;

  // ---- the nodes this step reads, compacted, and their level-0 rows packed (prepare_compact_pack; a prepared batch has them)
  PFO_REQUIRE(b->n_extra >= 0 && b->n_extra <= 2 * c->max_batch, "n_extra exceeds 2 * max_batch");
  PFO_REQUIRE(b->n_extra == 0 || b->extra_nodes, "null extra_nodes");
  // With memory the row pack leaves the caller's stream: the fused GRU gathers its rows straight from the per-node tables, and
  // the packed copies (for the backward, which runs after the state update has overwritten the tables) + the level-0 remap
  // (first used by layer 1's attention) are made on the side stream, in front of the event layer 1 waits for anyway.
  static const int pack_side_env = getenv("PFO_PACK_SIDE") ? atoi(getenv("PFO_PACK_SIDE")) : 1;      // A/B switch
  const bool pack_side = pack_side_env && c->use_memory && !b->prepared;
  if (!b->prepared) {
    if (pack_side) {
      // (comp_done rides on the completion of the compaction's LAST launch - its only one, or the third when the caller's extra
      //  list is marked and compacted too (data-parallel ranks: memory.hip pfo_touch_compact_launch): no marker packet)
      PFO_RUN_BOUND(bind_events, sd.comp_done, b->n_extra > 0 ? 2 : 0, s, prepare_compact(c, b, w, s));
      if (!bind_events) HIPOK(hipEventRecord(sd.comp_done, s), "event record failed");
    } else {
      RUN(prepare_compact_pack(c, st, b, w, d, n, s));
    }
  }
  PFO_MARK("fwd.compacted", s);
  const int capP = (int)std::min<int64_t>(c->n_nodes, n[0] + b->n_extra);
  if (c->use_memory) {
    PfoRange range_gru("forward lazy GRU");
    // both GRU contractions and the gate math in ONE launch (gemm.hip gru_fused_kernel): gi / gh never exist in HBM
    PfoGruFused f;
    f.msg_rows = w.msg_rows; f.K_msg = d.M; f.h_rows = w.h_rows; f.img_ih = w.iWih; f.img_hh = w.iWhh;
    f.b_ih = P.b_ih; f.b_hh = P.b_hh; f.hm = w.hm; f.touched = w.touched; f.node_feat = st->node_feat;
    f.upd_mem = w.upd_mem; f.h0_tab = w.h0_tab; f.gates = w.gates; f.D = D; f.cap_rows = capP; f.n_rows = w.n_touched;
    if (pack_side) { f.gather = 1; f.msg_rows = st->msg_table; f.h_rows = st->memory; f.hm = st->has_msg; }
    PFO_RUN_BOUND(fused_state && bind_events, sd.gru_done, 0, s, pfo_gru_fused_launch(f, s));
    if (fused_state && !bind_events) HIPOK(hipEventRecord(sd.gru_done, s), "event record failed");
    PFO_MARK("fwd.gru", s);
  }
  // ---- composite weights of every layer and their fp16 images (build_stage_a / _b): side stream.  Enqueued HERE, after the
  // sampling / compaction / GRU launches of the caller's stream (which need none of it): when the host is the slower side
  // (small batches, profilers) the critical chain is already queued while these ~14 launches are being issued; layer 1 waits
  // for them below.  With a valid parameter cache (pfo_tgn_state.pcache_valid) NONE of it is launched: the side stream only
  // clears the step's accumulators, packs the rows and runs the state update.
  PFO_REQUIRE(hipMemsetAsync(w.zero, 0, w.zero_bytes, ss) == hipSuccess, "memset failed");   // w.zero, w.tickets, time-gradient bins
  {
    // (a pfo_tgn_refresh chain may still be running on the second side stream: a rebuild waits for all of it, a call that uses
    //  the cache waits for its two stages where their results are first needed - no-ops when no refresh is outstanding)
    hipStreamCaptureStatus cap_st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cap_st) != hipSuccess) cap_st = hipStreamCaptureStatusActive;
    const bool capturing = cap_st != hipStreamCaptureStatusNone;         // (events recorded outside a capture are not waited for inside one)
    if (build && st->pcache && !capturing) HIPOK(hipStreamWaitEvent(ss, sd.pc_b, 0), "event wait failed");
    if (build) RUN(build_stage_a(d, w, P, ss));
    // (in front of the pack, not behind it: a cross-stream wait costs the waiting stream 5-17 us even when the event fired long
    //  ago, and the record below is what layer 1 waits for - stage A of a refresh is done ~80 us after the optimizer's kernel)
    if (!build && !capturing) HIPOK(hipStreamWaitEvent(ss, sd.pc_a, 0), "event wait failed");
    if (pack_side) {
      HIPOK(hipStreamWaitEvent(ss, sd.comp_done, 0), "event wait failed");
      RUN(prepare_pack(c, st, b, w, d, n, ss));
    }
    HIPOK(hipEventRecord(sd.layer[0], ss), "event record failed");      // layer 1 (and the top layer's fc2) can go
    if (L >= 2) {
      if (build) RUN(build_stage_b(d, w, P, ss));
      else if (!capturing) HIPOK(hipStreamWaitEvent(ss, sd.pc_b, 0), "event wait failed");
      if (fused_state && !side_late) {
        // the batch's state update, here: behind the lazy GRU (whose rows it persists), beside layer 1, in front of the event
        // layer 2 waits for - persist + message store leave the critical path and are joined at no extra wait
        HIPOK(hipStreamWaitEvent(ss, sd.gru_done, 0), "event wait failed");
        RUN(state_update(c, st, w, b->upd_src, b->upd_dst, b->upd_ts, b->upd_eidx, b->upd_B, ss));
        if (b->seg_in_forward) RUN(seg_prologue(c, b, w, d, n, ss));       // (behind gru_done: the compaction and the row pack are done)
      }
      HIPOK(hipEventRecord(sd.fold_done, ss), "event record failed");
    }
  }
  const float* tab0 = w.h0_tab;
  const int32_t* idx0 = w.idx0;
  const int WQ = HCp + D;

  // One attention layer = THREE large contractions around the neighbour-tile attention kernel (taken
  // to its end).  With one query per instance every projection that touches only that instance folds into a
  // per-step composite weight:
  //   qk'_h  = Wqk_h x + cqk_h,            Wqk_h  = Wk_h^T Wq_h[:, :D],   cqk_h = Wk_h^T (Wq_h[:, D:] cos(b) + bq_h)
  //   h1     = relu( sum_h W1ov_h ctx'_h + W1[:, E:] x + b1 ),
  //            W1ov_h = W1[:, :E] Wo[:, h] Wv_h   (fc1 . out_proj . value projection),
  //            ctx'_h = [ sum_j a'_jh key_j | sum_j a'_jh | valid ]  - the two extra columns carry the folded value
  //            bias (W1 Wo_h bv_h, scaled by the post-dropout weight sum) and the folded out_proj bias (W1 bo, only
  //            on rows that have a valid neighbour: temporal_attention.py:84 zero-fills the others)
  //   out    = W2 h1 + b2
  // 0.60 MFLOP per instance instead of 1.13 (and 10.3 un-folded); Q, O and attn_out are never formed.
  const float scale = 1.0f / sqrtf((float)dh);
  static const char* const fwd_names[PFO_MAX_LAYERS + 1] = {"", "forward layer 1", "forward layer 2", "forward layer 3", "forward layer 4"};
  static const char* const mk_qk[PFO_MAX_LAYERS + 1] = {"", "fwd.L1.qk", "fwd.L2.qk", "fwd.L3.qk", "fwd.L4.qk"};
  static const char* const mk_at[PFO_MAX_LAYERS + 1] = {"", "fwd.L1.attn", "fwd.L2.attn", "fwd.L3.attn", "fwd.L4.attn"};
  static const char* const mk_h1[PFO_MAX_LAYERS + 1] = {"", "fwd.L1.h1", "fwd.L2.h1", "fwd.L3.h1", "fwd.L4.h1"};
  for (int l = 1; l <= L; ++l) {
    PfoRange range_layer(fwd_names[l]);
    const int N = (int)n[l];
    const LayerWs& lw = w.layer[l];
    const auto& p = P.l[l];
    // layers >= 2 read the previous layer's h1 rows: its fc2 lives in this layer's folded composites (fc2 fold above)
    const float* xA = (l == 1) ? tab0 : w.layer[l - 1].h1;
    const int32_t* x_idx = (l == 1) ? idx0 : nullptr;
    const float* Wqk_l = (l == 1) ? lw.Wqk : lw.Wqk_f;
    const float* cqk_l = (l == 1) ? lw.cqk : lw.cqk_f;
    const float* W1ovT_l = (l == 1) ? lw.W1ovT : lw.W1ovT_f;
    const float* W1b_l = (l == 1) ? p.w1 + E : lw.W1b_f;
    const int64_t W1b_ld = (l == 1) ? E + D : D;
    const float* b1_l = (l == 1) ? p.b1 : lw.b1_f;

    if (!composites_awaited) {
      HIPOK(hipStreamWaitEvent(s, sd.layer[0], 0), "event wait failed");   // composite weights and layer 1's images are ready
      composites_awaited = true;
    }
    // ... and the fc2-folded ones of the layers >= 2.  (Round 3 tried ONE wait for both, on the later event: the side stream's
    // chain - 14 small dependent launches, ~150 us - then ends about when this stream reaches layer 1, and the touched-table
    // projection started 25 us late; layer 1 needs only the first ~90 us of that chain.)
    if (l == 2) HIPOK(hipStreamWaitEvent(s, sd.fold_done, 0), "event wait failed");
    // ---- qk' = x Wqk^T + cqk.  Layer 1: x is a row of the touched-node table, shared by every instance that sits on
    // that node (~54 k instances on ~11 k nodes at C2): ONE projection of the table, [qk' | x W1[:, E:]^T] per row
    if (l == 1) {
      PfoGemm g = g_nt(tab0, D, nullptr, lw.Wqk, D, w.QX, WQ, capP, WQ, D, w.l1_bias);
      g.m_dev = w.n_core; g.b_img = w.iQX;                 // (rows only the caller's extra list names are never queried)
      RUN(pfo_gemm_launch(g, s));
    } else {
      PfoGemm g = g_nt(xA, D, x_idx, Wqk_l, D, lw.QK, HCp, N, HCp, D, cqk_l);
      g.b_img = lw.iWqk;
      RUN(pfo_gemm_launch(g, s));
    }
    PfoAttn a;
    a.N = N; a.K = K; a.D = D; a.Ef = Ef; a.H = H; a.Cp = Cp;
    a.QK = (l == 1) ? w.QX : lw.QK; a.qk_row = (l == 1) ? idx0 : nullptr; a.qk_ld = (l == 1) ? WQ : HCp;
    a.nbr_tab = xA; a.nbr_ld = D;
    a.nbr_rows = (l == 1) ? capP : n[l - 1]; a.edge_rows = c->n_edges_p1;
    a.nbr_row = (l == 1) ? idx0 + N : nullptr;
    a.nbr_row_base = N;
    a.nbr_ids = w.nodes[l - 1] + N;
    a.edge_feat = st->edge_feat; a.eidx = w.eidx[l]; a.dt = w.dt[l]; a.tw = P.tw; a.tb = P.tb;
    a.scale = scale; a.dropout_p = b->dropout_p; a.seed = b->seed; a.offset = b->offset + 0x51ED0000ull + (uint64_t)l; a.offset_dev = b->offset_dev;
    a.keep_inject = (b->dropout_keep && b->training) ? b->dropout_keep[L - l] : nullptr;
    a.ctx = lw.ctx; a.attw = lw.attw; a.inv = lw.inv;
    PFO_MARK(mk_qk[l], s);
    RUN(pfo_attn_fwd_launch(a, s));
    PFO_MARK(mk_at[l], s);
    if (l == 1 && fused_state && side_late) {
      // EXPERIMENT (PFO_FWD_SIDE_LATE=1): the state update and the instance groups behind layer 1's attention kernel instead of
      // beside it (they are ~12 small launches that share the chip with the step's second-longest kernel)
      HIPOK(hipEventRecord(sd.layer[1], s), "event record failed");
      HIPOK(hipStreamWaitEvent(ss, sd.layer[1], 0), "event wait failed");
      RUN(state_update(c, st, w, b->upd_src, b->upd_dst, b->upd_ts, b->upd_eidx, b->upd_B, ss));
      if (b->seg_in_forward) RUN(seg_prologue(c, b, w, d, n, ss));
      HIPOK(hipEventRecord(sd.late_done, ss), "event record failed");
    }
    // ---- h1 = relu(ctx' W1ovT + x W1[:, E:]^T + b1)   (MergeLayer fc1 with out_proj and the value projection folded in)
    if (l == 1) {
      // the x term was projected with the table: it arrives as a row-gathered addend of the epilogue
      PFO_REQUIRE(pfo_gemm_takes_bx(N, D), "the touched-table layer needs the bf16x3 image kernels");
      PfoGemm g = g_nn(lw.ctx, HCp, lw.W1ovT, D, lw.h1, D, N, D, HCp);
      g.bias = p.b1; g.relu = 1; g.b_img = lw.iW1ov;
      g.add_src = w.QX + HCp; g.add_ld = WQ; g.add_idx = idx0;
      RUN(pfo_gemm_launch(g, s));
    } else if (pfo_gemm_takes_bx(N, D)) {
      // both K-concatenated sources ([ctx' | x] against [W1ov | W1[:, E:]]) in one launch: h1 is written once
      PfoGemm g = g_nn(lw.ctx, HCp, W1ovT_l, D, lw.h1, D, N, D, HCp);
      g.A[1] = xA; g.lda[1] = D; g.a_idx[1] = x_idx; g.B[1] = W1b_l; g.ldb[1] = W1b_ld; g.K[1] = D;
      g.bias = b1_l; g.relu = 1; g.b_img = lw.iW1ov; g.b_img2 = lw.iW1b;
      RUN(pfo_gemm_launch(g, s));
    } else {
      PfoGemm g = g_nn(lw.ctx, HCp, W1ovT_l, D, lw.h1, D, N, D, HCp);
      g.b_img = lw.iW1ov;
      RUN(pfo_gemm_launch(g, s));
      g = g_nt(xA, D, x_idx, W1b_l, W1b_ld, lw.h1, D, N, D, D, b1_l);
      g.accumulate = 1; g.relu = 1; g.b_img = lw.iW1b;
      RUN(pfo_gemm_launch(g, s));
    }
    PFO_MARK(mk_h1[l], s);
    if (l == L) {
      // out = W2 h1 + b2: only the top layer's rows ARE embeddings (N == R; written in place, no copy) - below it the
      // contraction is folded into the next layer's composites
      PfoGemm g = g_nt(lw.h1, D, nullptr, p.w2, D, emb_out, D, N, D, D, p.b2);
      g.b_img = lw.iW2;
      RUN(pfo_gemm_launch(g, s));
    }
  }
  if (fused_state && side_late) HIPOK(hipStreamWaitEvent(s, sd.late_done, 0), "event wait failed");
  PFO_MARK("fwd.end", s);
  return PFO_OK;
}

// =============================================================================================
extern "C" int pfo_tgn_grad_split(const pfo_tgn_config* c, int64_t* split) {
  if (int rc = check_cfg(c)) return rc;
  PFO_REQUIRE(split != nullptr, "null output");
  pfo_tgn_layout lay;
  RUN(pfo_tgn_param_layout(c, &lay));
  *split = c->n_layers >= 2 ? lay.layer[c->n_layers - 1].wq : lay.total;
  return PFO_OK;
}

extern "C" int pfo_tgn_backward(const pfo_tgn_config* c, const pfo_tgn_state* st, const pfo_tgn_batch* b, void* workspace,
                                const float* d_emb, float* grad, void* stream) {
  return pfo_tgn_backward_ev(c, st, b, workspace, d_emb, grad, 0, nullptr, nullptr, 0, nullptr, stream);
}

extern "C" int pfo_tgn_backward_ev(const pfo_tgn_config* c, const pfo_tgn_state* st, const pfo_tgn_batch* b, void* workspace,
                                   const float* d_emb, float* grad, int32_t zero_grad_first, void* top_ready_event,
                                   const float* mean_src, int64_t mean_n, float* mean_out, void* stream) {
  if (int rc = check_cfg(c)) return rc;
  PFO_REQUIRE(st && workspace && d_emb && grad && st->params, "null argument");
  PFO_REQUIRE(!mean_src || (mean_n > 0 && mean_out), "bad deferred mean");
  int64_t n[PFO_MAX_LAYERS + 1];
  RUN(level_sizes(c, b, n));
  const Dims d = dims_of(c);
  Ws w_ = carve(c, workspace);
  if (st->pcache) pcache_bind(c, st->pcache, w_);       // the composites and images the forward used
  const Ws& w = w_;
  hipStream_t s = (hipStream_t)stream;
  pfo_tgn_layout lay;
  RUN(pfo_tgn_param_layout(c, &lay));
  Params P;
  Grads G;
  bind(lay, st->params, P, d.L, c->use_memory != 0);
  bind(lay, grad, G, d.L, c->use_memory != 0);
  const int L = d.L, D = d.D, Ef = d.Ef, H = d.H, E = d.E, C = d.C, dh = d.dh, K = b->K;
  const float scale = 1.0f / sqrtf((float)dh);
  const float* tab0 = w.h0_tab;
  const int32_t* idx0 = w.idx0;
  const int capP = (int)std::min<int64_t>(c->n_nodes, n[0] + b->n_extra);
  const int64_t rep_stride = (int64_t)d.capP * D;             // floats between the per-XCD replicas of d_h0
  // Per-XCD replicas of the level-0 gradient table pay off only for the per-instance atomics (uniform sampling: 4 replicas
  // 0.537 -> 0.506 ms); the run-merged kernel issues 2-3x fewer and measures best on ONE table (1.656 vs 1.665 ms/step)
  const int det = b->deterministic ? 1 : 0;
  const int n_rep = grad_replicas(c, b);
  static_assert(PFO_GRAD_REPLICAS >= 2, "the deterministic int64 gradient table needs the room of two float replicas");
  int64_t det_rows = 0;                                        // slab rows written so far (deterministic mode)

  const int Cp = d.Cp, HCp = H * d.Cp, WQ = HCp + D;
  PfoRange range_call("pfo_tgn_backward");
  Side& sd = side();
  PFO_REQUIRE(sd.ok, "could not create the side stream");
  hipStream_t ss = sd.s;
  RUN(side_join(sd, s));
  // layer-1 instances grouped by the touched-table row they sit on (needed only when the layer-1 gradients are summed
  // per row, late in this call): built on the side stream, beside the layer-L .. 2 work.  The same stream first clears what
  // this call accumulates into: the level-0 gradient rows (layer 1's attention backward waits for seg_done)
  // (the gradient buffer is cleared on the caller's stream, before the fork: every writer on any stream comes after it)
  hipStreamCaptureStatus cap_early = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &cap_early) != hipSuccess) cap_early = hipStreamCaptureStatusActive;
  const bool bind_events = cap_early == hipStreamCaptureStatusNone;      // events bound to a launch (common.hpp): not under capture
  PFO_MARK("bwd.begin", s);
  if (zero_grad_first) HIPOK(hipMemsetAsync(grad, 0, (size_t)lay.total * sizeof(float), s), "memset failed");
  // The side stream's opening work: the loss mean the caller left to this call, the cleared level-0 gradient rows and the
  // layer-1 instance groups (seg_prologue).  A call whose forward already queued the last two (pfo_tgn_batch.seg_in_forward)
  // forks nothing here - an event record costs the caller's stream a ~6 us bubble in front of its next kernel - and takes
  // the mean behind the first event the side stream waits for anyway.
  const bool seg_fwd = b->seg_in_forward && b->upd_src != nullptr && c->use_memory && L >= 2;
  bool mean_pending = mean_src != nullptr;
  auto side_mean_once = [&]() -> int {
    if (mean_pending) { mean_pending = false; RUN(pfo_mean_launch(mean_src, mean_n, mean_out, ss)); }
    return PFO_OK;
  };
  if (!seg_fwd) {
    HIPOK(hipEventRecord(sd.fork, s), "event record failed");
    HIPOK(hipStreamWaitEvent(ss, sd.fork, 0), "event wait failed");
    RUN(side_mean_once());
    RUN(seg_prologue(c, b, w, d, n, ss));
    HIPOK(hipEventRecord(sd.seg_done, ss), "event record failed");
  }
  std::function<int()> deferred_chain;                       // a layer's chain-back launches, issued one layer later (below)
  bool gates_fused = false;                                  // the GRU gate backward ran as the epilogue of layer 1's dx_tab launch
  static const char* const bwd_names[PFO_MAX_LAYERS + 1] = {"", "backward layer 1", "backward layer 2", "backward layer 3", "backward layer 4"};
  static const char* const mk_dh1[PFO_MAX_LAYERS + 1] = {"", "bwd.L1.dh1", "bwd.L2.dh1", "bwd.L3.dh1", "bwd.L4.dh1"};
  static const char* const mk_dctx[PFO_MAX_LAYERS + 1] = {"", "bwd.L1.dctx", "bwd.L2.dctx", "bwd.L3.dctx", "bwd.L4.dctx"};
  static const char* const mk_battn[PFO_MAX_LAYERS + 1] = {"", "bwd.L1.attn", "bwd.L2.attn", "bwd.L3.attn", "bwd.L4.attn"};
  static const char* const mk_dx[PFO_MAX_LAYERS + 1] = {"", "bwd.L1.dx_tab", "bwd.L2.dx", "bwd.L3.dx", "bwd.L4.dx"};
  for (int l = L; l >= 1; --l) {
    PfoRange range_layer(bwd_names[l]);
    const int N = (int)n[l];
    const LayerWs& lw = w.layer[l];
    const auto& p = P.l[l];
    const auto& g = G.l[l];
    // layers >= 2 work on the previous layer's h1 rows through their fc2-folded composites (pfo_tgn_forward, "fc2 fold"):
    // their weight gradients land in the folded composites' gradient buffers and are unfolded on the side stream below
    const bool folded = l >= 2;
    const float* xA = (l == 1) ? tab0 : w.layer[l - 1].h1;
    const int32_t* x_idx = (l == 1) ? idx0 : nullptr;
    float* dx = (l == 1) ? nullptr : w.dH[l - 1];            // rows [0, N) of the previous level's gradient
    const float* Wqk_l = folded ? lw.Wqk_f : lw.Wqk;
    const float* W1ovT_l = folded ? lw.W1ovT_f : lw.W1ovT;
    const float* W1b_l = folded ? lw.W1b_f : p.w1 + E;
    const int64_t W1b_ld = folded ? D : E + D;
    float* dWqk_l = folded ? lw.dWqk_f : lw.dWqk;
    float* gqk_l = folded ? lw.gqk_f : lw.gqk;
    float* dW1ovT_l = folded ? lw.dW1ovT_f : lw.dW1ovT;

    // Data gradients first (a chain of GEMMs + the attention core); every weight / bias gradient of the layer is
    // then taken in ONE grouped split-K launch (bias gradients ride along as an extra column).
    PfoTnProblem tn[4];
    int ntn = 0;
    auto set_tn = [&](PfoTnProblem& q, const float* A, int64_t lda, const float* B, int64_t ldb, const int32_t* b_idx, int M_,
                      int N_, float* C_, int64_t ldc, float* bias_out) {
      q = PfoTnProblem();
      q.A = A; q.lda = lda; q.B = B; q.ldb = ldb; q.b_idx = b_idx; q.M = M_; q.N = N_; q.C = C_; q.ldc = ldc; q.bias_out = bias_out;
    };
    const float* dh1;                                          // d loss / d (fc1 pre-activation)
    if (l == L) {
      // the top layer applies its fc2 (utils.py:17): dW2 / db2 and d h1 = relu'(.) (d out W2)
      set_tn(tn[ntn++], d_emb, D, lw.h1, D, nullptr, D, D, g.w2, D, g.b2);
      PfoGemm q = g_nn(d_emb, D, p.w2, D, w.dh1, D, N, D, D);
      q.relu_src = lw.h1; q.relu_ld = D;                     // ReLU backward
      q.b_img = lw.iW2T;
      RUN(pfo_gemm_launch(q, s));
      PFO_MARK(mk_dh1[l], s);
      dh1 = w.dh1;
    } else {
      // below the top the layer above wrote d h1 directly, ReLU mask applied by its producers (attention backward's key
      // rows, the x-side contraction's epilogue): fc2 is part of that layer's composites
      dh1 = w.dH[l];
    }
    set_tn(tn[ntn], lw.ctx, HCp, dh1, D, nullptr, HCp, D, dW1ovT_l, D, nullptr);       // dW1ovT = ctx'^T dh1
    tn[ntn].c_accumulate = 0;
    ++ntn;
    const int n_tn_a = ntn;
    // The weight gradients over the INSTANCES (dW2 / db2 at the top, dW1ovT) need only d out, h1, ctx' and dh1: they go to the side
    // stream.  With the bf16x3 tile they were best started as soon as dh1 existed, beside the d ctx' contraction (15 us/step
    // better than next to the attention backward alone, whose single-wavefront workgroups starved a 74 KB-LDS kernel of slots);
    // the fp16 tile (49 KB, half the matrix work) does better behind d ctx', beside the attention backward: 1.4495 against
    // 1.4532 ms over four interleaved pairs.  (Behind the attention backward, beside the serial tail: +2 % per step.)
    static const int tna_mode = getenv("PFO_TNA_MODE") ? atoi(getenv("PFO_TNA_MODE")) : 0;
    const int tna_mode_g = tna_mode;   // A/B: 0 fork behind the d ctx' contraction (beside the attention backward), 2 beside d ctx', 1 main stream
    bool tn_a_bound = false;                                   // tn_a already rides on the d ctx' launch
    const bool dq_atomic = l == 1 && dq_atomic_mode(c, b);
    bool dh1_summed = false;
    auto tn_a_side = [&]() -> int {
      if (!tn_a_bound) HIPOK(hipEventRecord(sd.tn_a, s), "event record failed");
      HIPOK(hipStreamWaitEvent(ss, sd.tn_a, 0), "event wait failed");
      RUN(side_mean_once());
      if (dq_atomic && l == 1) {
        // the d h1 half of the per-row sums, here beside the attention backward (whose atomics fill the other half)
        RUN(pfo_segsum_cols_launch(dh1, D, w.seg_ptr, w.seg_mem, w.seg_of, n[1], w.n_core, w.Dq + HCp, WQ, ss));
        HIPOK(hipEventRecord(sd.dh1_sum, ss), "event record failed");
        dh1_summed = true;
      }
      PFO_MARK("@side1.tn_a.begin", ss);
      RUN(pfo_gemm_tn_group_launch(tn, n_tn_a, N, nullptr, w.slabs2, w.slab_floats, ss));
      PFO_MARK("@side1.tn_a.end", ss);
      HIPOK(hipEventRecord(sd.tn_a_done, ss), "event record failed");
      return PFO_OK;
    };
    if (l == 1 && tna_mode == 2 && !pfo_prof_on()) RUN(tn_a_side());
    // merged fc1: d ctx' = dh1 W1ovT^T (dx = dh1 W1[:, E:] is taken together with the query/key part below)
    {
      PfoGemm q = g_nt(dh1, D, nullptr, W1ovT_l, D, w.dctx, HCp, N, HCp, D, nullptr);
      q.b_img = lw.iW1ovT;
      const bool bind_tn_a = bind_events && l == 1 && tna_mode == 0 && !pfo_prof_on();
      PFO_RUN_BOUND(bind_tn_a, sd.tn_a, 0, s, pfo_gemm_launch(q, s));
      if (bind_tn_a) tn_a_bound = true;
      PFO_MARK(mk_dctx[l], s);
      if (l == 1 && b->mid_event && !b->mid_event_late) HIPOK(hipEventRecord((hipEvent_t)b->mid_event, s), "event record failed");
    }
    if (l > 1) {
      set_tn(tn[ntn], dh1, D, xA, D, x_idx, D, D, lw.dW1b_f, D, lw.db1_f);              // d (W1[:, E:] A), d (b1 + W1[:, E:] b)
      tn[ntn].c_accumulate = 0; tn[ntn].bias_accumulate = 0;
      ++ntn;
    }
    if (l == 1) {
      if (pfo_prof_on() || tna_mode == 1) {        // event-bracketed step (bench.py's roofline sample): serial, so the bracket times the kernel alone
        RUN(pfo_gemm_tn_group_launch(tn, n_tn_a, N, nullptr, w.slabs, w.slab_floats, s));
        HIPOK(hipEventRecord(sd.tn_a_done, s), "event record failed");
      } else if (tna_mode == 0) {
        RUN(tn_a_side());
      }
    }
    // attention core
    PfoAttn a;
    a.N = N; a.K = K; a.D = D; a.Ef = Ef; a.H = H; a.Cp = Cp;
    a.QK = (l == 1) ? w.QX : lw.QK; a.qk_row = (l == 1) ? idx0 : nullptr; a.qk_ld = (l == 1) ? WQ : HCp;
    a.nbr_tab = xA; a.nbr_ld = D;
    a.nbr_rows = (l == 1) ? capP : n[l - 1]; a.edge_rows = c->n_edges_p1;
    a.nbr_row = (l == 1) ? idx0 + N : nullptr;
    a.nbr_row_base = N;
    a.nbr_relu = folded ? 1 : 0;                               // the keys are h1 rows of the layer below: d row *= (row > 0)
    a.nbr_ids = w.nodes[l - 1] + N;
    a.edge_feat = st->edge_feat; a.eidx = w.eidx[l]; a.dt = w.dt[l]; a.tw = P.tw; a.tb = P.tb;
    a.scale = scale; a.dropout_p = b->dropout_p; a.seed = b->seed; a.offset = b->offset + 0x51ED0000ull + (uint64_t)l; a.offset_dev = b->offset_dev;
    a.keep_inject = (b->dropout_keep && b->training) ? b->dropout_keep[L - l] : nullptr;
    a.ctx = lw.ctx; a.attw = lw.attw; a.inv = lw.inv;
    float* const dqk_l = (l == 1) ? w.dQK : lw.dQK;
    a.dctx = w.dctx; a.dQK = dqk_l;
    if (l == 1) { a.d_nbr = c->use_memory ? w.d_h0 : nullptr; a.d_nbr_ld = D; a.d_nbr_rep = rep_stride; a.d_nbr_nrep = n_rep; }
    else        { a.d_nbr = w.dH[l - 1]; a.d_nbr_ld = D; }
    a.dtime_part = w.dtime;
    a.det = det; a.dtime_slab = w.dtime_slab + det_rows * 2 * D;
    int n_parts = 0;
    if (l == 1 && c->use_memory && !seg_fwd) HIPOK(hipStreamWaitEvent(s, sd.seg_done, 0), "event wait failed");   // d_h0 is clear, the groups exist
    if (l == 1 && c->use_memory && !b->uniform) {
      // key-side gradients of instances with identical neighbour lists leave as one set of atomics (attn.hip)
      a.members = w.seg_mem; a.seg_ptr = w.seg_ptr; a.n_rows = w.n_core; a.run_cnt = w.cnt1; a.dqk_live = w.dqk_live;
    }
    if (dq_atomic) { a.dq_rows = w.Dq; a.dq_ld = WQ; }
    const int dqk_by_member = pfo_attn_bwd_uses_runs(a) ? 1 : 0;
    const bool dq_added = dq_atomic && dqk_by_member;         // (the alignment test of pfo_attn_bwd_uses_runs may still say no)
    if (dq_added && !dh1_summed) {
      // serial forms (bracketed steps, PFO_TNA_MODE): the d h1 half on this stream, in front of the attention backward
      // (the groups exist: seg_done was awaited above, or the forward built them)
      RUN(pfo_segsum_cols_launch(dh1, D, w.seg_ptr, w.seg_mem, w.seg_of, n[1], w.n_core, w.Dq + HCp, WQ, s));
    }
    PFO_RUN_BOUND(dq_added && bind_events, sd.tn_b, 0, s, pfo_attn_bwd_launch(a, &n_parts, s));   // (atomic row sums: tn_b rides on the attention launch, no segment sum behind it)
    if (l == 1 && tna_mode == 3 && !pfo_prof_on()) RUN(tn_a_side());     // A/B: the instance weight gradients BEHIND the attention backward, beside the serial tail
    PFO_MARK(mk_battn[l], s);
    if (l == 1 && b->mid_event && b->mid_event_late) HIPOK(hipEventRecord((hipEvent_t)b->mid_event, s), "event record failed");
    if (det) det_rows += n_parts;
    if (deferred_chain) { RUN(deferred_chain()); deferred_chain = nullptr; }     // the layer above's chain-back (side streams)
    bool layer_event_bound = false;
    // merged query/key projection: dx += dqk' Wqk, dWqk = dqk'^T x, gqk = colsum(dqk')
    if (l == 1) {
      // Layer 1: x is a row of the touched-node table shared by all instances on that node, so everything that is linear
      // in the per-instance gradients (dqk', dh1) and otherwise depends on x only - the data gradient of x and the two
      // weight gradients against x - is taken AFTER summing those gradients per table row: contractions over the
      // ~11 k touched rows instead of the ~54 k instances.
      if (!c->use_memory) HIPOK(hipStreamWaitEvent(s, sd.seg_done, 0), "event wait failed");   // (with memory: awaited before the attention backward)
      if (dq_added) {
        // Dq is complete when the attention launch and the side stream's d h1 sum are (the latter fired long ago)
        if (!bind_events) HIPOK(hipEventRecord(sd.tn_b, s), "event record failed");
        if (dh1_summed) HIPOK(hipStreamWaitEvent(s, sd.dh1_sum, 0), "event wait failed");
      } else {
      PFO_RUN_BOUND(bind_events, sd.tn_b, 0, s,
                    pfo_segsum_launch(dqk_l, HCp, dh1, D, w.seg_ptr, w.seg_mem, w.seg_of, n[1], w.n_core, capP, dqk_by_member,
                                      dqk_by_member ? w.dqk_live : nullptr, w.Dq, s));   // Dq = [sum dqk' | sum dh1]
      // the weight gradients over the table rows go to the side stream too (beside d h0 / the GRU backward on this one)
      if (!bind_events) HIPOK(hipEventRecord(sd.tn_b, s), "event record failed");
      }
      PFO_MARK("bwd.L1.segsum", s);
      HIPOK(hipStreamWaitEvent(ss, sd.tn_b, 0), "event wait failed");
      RUN(side_mean_once());
      PfoTnProblem tb[2];
      set_tn(tb[0], w.Dq, WQ, tab0, D, nullptr, HCp, D, lw.dWqk, D, lw.gqk);               // dWqk = (sum dqk')^T h0, gqk
      tb[0].c_accumulate = 0; tb[0].bias_accumulate = 0;
      set_tn(tb[1], w.Dq + HCp, WQ, tab0, D, nullptr, D, D, g.w1 + E, E + D, g.b1);        // dW1[:, E:], db1
      // (the layer's chain-back on the side stream needs dWqk / gqk: this launch stays there also while profiling)
      PFO_MARK("@side1.tn_b.begin", ss);
      RUN(pfo_gemm_tn_group_launch(tb, 2, capP, w.n_core, w.slabs2, w.slab_floats, ss));
      PFO_MARK("@side1.tn_b.end", ss);
      if (c->use_memory) {
        // d h0_tab (query side) = Dq [Wqk ; W1[:, E:]]; the GRU backward adds it to the key-side rows the attention scattered
        PfoGemm q = g_nn(w.Dq, WQ, lw.Wqk, D, w.dx_tab, D, capP, D, HCp);
        q.A[1] = w.Dq + HCp; q.lda[1] = WQ; q.B[1] = p.w1 + E; q.ldb[1] = E + D; q.K[1] = D;
        q.b_img = lw.iWqkT; q.b_img2 = lw.iW1bT; q.m_dev = w.n_core;
        // The GRU's gate backward rides in this launch's epilogue when it takes the 32-row kernel (one table, float rows):
        // d h0 = key side (scattered by the attention backward) + this contraction's query side never goes to HBM as dx_tab,
        // and the separate gate launch (26 us + a launch gap on the serial tail at C2) disappears
        const int fuse_env = getenv("PFO_FUSE_GATES") ? atoi(getenv("PFO_FUSE_GATES")) : 0;   // (read per call: a test flips it)         // A/B switch (measured: 1.458-1.461 ms with, 1.453 without - the epilogue runs on the ~375 live workgroups of a 32-row launch, the separate kernel on the whole chip; off)
        if (fuse_env && n_rep == 1 && !det && pfo_gemm_takes_skinny(capP, D) && (D % 4) == 0) {
          q.gg_gates = w.gates; q.gg_h = w.h_rows; q.gg_hm = w.hm; q.gg_dh0 = w.d_h0; q.gg_dgi = w.gi; q.gg_dgh = w.gh;
          gates_fused = true;
        }
        RUN(pfo_gemm_launch(q, s));
        PFO_MARK(mk_dx[l], s);
      }
    } else {
      // d h1 of the layer below, self rows [0, N): [dqk' | dh1] [Q_f ; W1b_f], masked by that layer's ReLU
      if (pfo_gemm_takes_bx(N, D)) {
        // both sources in one launch, dx written once
        PfoGemm q = g_nn(dqk_l, HCp, Wqk_l, D, dx, D, N, D, HCp);
        q.A[1] = dh1; q.lda[1] = D; q.B[1] = W1b_l; q.ldb[1] = W1b_ld; q.K[1] = D;
        q.b_img = lw.iWqkT; q.b_img2 = lw.iW1bT;
        q.relu_src = xA; q.relu_ld = D;
        PFO_RUN_BOUND(bind_events, sd.layer[l], 0, s, pfo_gemm_launch(q, s));     // (layer[l], recorded below, rides on this launch)
        if (bind_events) layer_event_bound = true;
      } else {
        PfoGemm q = g_nn(dh1, D, W1b_l, W1b_ld, dx, D, N, D, D);
        q.b_img = lw.iW1bT;
        RUN(pfo_gemm_launch(q, s));
        q = g_nn(dqk_l, HCp, Wqk_l, D, dx, D, N, D, HCp);
        q.accumulate = 1; q.b_img = lw.iWqkT;
        q.relu_src = xA; q.relu_ld = D;
        RUN(pfo_gemm_launch(q, s));
      }
      PFO_MARK(mk_dx[l], s);
      set_tn(tn[ntn], dqk_l, HCp, xA, D, x_idx, HCp, D, dWqk_l, D, gqk_l);
      tn[ntn].c_accumulate = 0; tn[ntn].bias_accumulate = 0;
      ++ntn;
      // (launched below, on the second side stream in front of the layer's chain: nothing on this stream needs the weight
      //  gradients of a layer >= 2 - 36 us of launch-latency-bound work off the critical path at C2)
    }

    // ---- chain the composite-weight gradients back to the parameters (tiny products, side streams).  ~15 launches that no
    // launch of the caller's stream waits for: for a layer >= 2 the HOST issues them only after the next layer's data-gradient
    // launches are queued (r3 timeline: the caller's stream sat idle for ~60 us behind them while the host was the slower side);
    // the event that releases them on the device stays where it was.
    if (folded && !layer_event_bound) HIPOK(hipEventRecord(sd.layer[l], s), "event record failed");
    auto chain_back = [=]() -> int {
    if (folded) {
      // First undo the fc2 fold (notation of pfo_tgn_forward: A, b = W2, b2 of layer l-1; Q = Wqk, V = W1ovT, per head):
      //   dT1_node = A dQ_f,node      dT1_edge|time = dQ_f,edge|time         dt likewise from gqk_f
      //   dV_node  = A dV_f,node + b (x) dV_f[C]         dV elsewhere = dV_f
      //   dW1b     = dW1b_f A^T + db1_f (x) b            db1 += db1_f
      //   dQ       = dT1 A^T + dt (x) b                  d cqk = dt
      //   dA       = sum_h ( T1_node dQ_f,node^T + t_node (x) gqk_f,node + V_node dV_f,node^T ) + W1b^T dW1b_f + Q^T dT1
      //   db       = sum_h V_node dV_f[C]^T + W1b^T db1_f + Q^T dt
      // The dA / db terms are written to slabs (several heads add to one matrix) and summed once.
      const auto& gp = G.l[l - 1];
      const float* A = P.l[l - 1].w2;
      const float* bv = P.l[l - 1].b2;
      const int64_t HCpD = (int64_t)HCp * D, CpD = (int64_t)Cp * D, DD = (int64_t)D * D;
      // (on the second side stream: the first one must stay free for layer 1's weight gradients over the instances)
      hipStream_t sf = sd.s2;
      HIPOK(hipStreamWaitEvent(sf, sd.layer[l], 0), "event wait failed");
      // (A/B, PFO_L2_CHAIN_LATE=1: layer 2's weight gradients and chain wait for layer 1's d ctx' contraction too - they then
      //  run beside the attention backward instead of taking workgroup slots from the contraction the caller's stream waits for)
      static const int chain_late = getenv("PFO_L2_CHAIN_LATE") ? atoi(getenv("PFO_L2_CHAIN_LATE")) : 0;
      if (chain_late && l == 2 && !pfo_prof_on() && tna_mode_g == 0) HIPOK(hipStreamWaitEvent(sf, sd.tn_a, 0), "event wait failed");
      if (pfo_prof_on()) {
        // event-bracketed step (bench.py's roofline sample): on the caller's stream, so that the bracket times the kernel alone
        RUN(pfo_gemm_tn_group_launch(tn, ntn, N, nullptr, w.slabs, w.slab_floats, s));
        HIPOK(hipEventRecord(sd.layer[l], s), "event record failed");
        HIPOK(hipStreamWaitEvent(sf, sd.layer[l], 0), "event wait failed");
      } else {
        RUN(pfo_gemm_tn_group_launch(tn, ntn, N, nullptr, w.slabs3, w.slab_floats, sf));
      }
      static const int grouped_copy_b = getenv("PFO_GROUPED_COPY") ? atoi(getenv("PFO_GROUPED_COPY")) : 1;      // A/B switch
      if (!grouped_copy_b) {
        HIPOK(hipMemcpyAsync(lw.dT1, lw.dWqk_f, HCpD * sizeof(float), hipMemcpyDeviceToDevice, sf), "copy failed");
        HIPOK(hipMemcpyAsync(lw.gqk, lw.gqk_f, (size_t)HCp * sizeof(float), hipMemcpyDeviceToDevice, sf), "copy failed");
        HIPOK(hipMemcpyAsync(lw.dW1ovT, lw.dW1ovT_f, HCpD * sizeof(float), hipMemcpyDeviceToDevice, sf), "copy failed");
      } else {
        PfoSumSlabs cp[3];                        // (one grouped launch instead of three runtime blits, as in build_stage_b)
        cp[0].dst = lw.dT1;    cp[0].src = lw.dWqk_f;   cp[0].count = HCpD;
        cp[1].dst = lw.gqk;    cp[1].src = lw.gqk_f;    cp[1].count = HCp;
        cp[2].dst = lw.dW1ovT; cp[2].src = lw.dW1ovT_f; cp[2].count = HCpD;
        for (int q = 0; q < 3; ++q) { cp[q].n_slabs = 1; cp[q].accumulate = 0; cp[q].stride = 0; }
        RUN(pfo_sum_slabs_launch(cp, 3, sf));
      }
      float* sl = lw.fold_slabs;
      float* vs = lw.fold_vslabs;
      {
        PfoGemm u[9];
        u[0] = g_nn(A, D, lw.dWqk_f, D, lw.dT1, D, D, D, D);                                  // dT1_node = A dQ_f,node
        u[0].batch = H; u[0].b_bs[0] = CpD; u[0].c_bs = CpD;
        u[1] = g_nt(lw.gqk_f, D, nullptr, A, D, lw.gqk, D, 1, D, D, nullptr);                 // dt_node = A gqk_f,node
        u[1].batch = H; u[1].a_bs[0] = Cp; u[1].c_bs = Cp;
        u[2] = g_nt(lw.T1, D, nullptr, lw.dWqk_f, D, sl, D, D, D, D, nullptr);                // slab h: T1_node dQ_f,node^T
        u[2].batch = H; u[2].a_bs[0] = CpD; u[2].b_bs[0] = CpD; u[2].c_bs = DD;
        u[3] = g_nt(lw.W1ovT, D, nullptr, lw.dW1ovT_f, D, sl + H * DD, D, D, D, D, nullptr);  // slab H+h: V_node dV_f,node^T
        u[3].batch = H; u[3].a_bs[0] = CpD; u[3].b_bs[0] = CpD; u[3].c_bs = DD;
        u[4] = g_nn(p.w1 + E, E + D, lw.dW1b_f, D, sl + 2 * H * DD, D, D, D, D);              // slab 2H: W1b^T dW1b_f
        u[4].a_kmajor = 1;
        u[5] = g_nn(A, D, lw.dW1ovT_f, D, lw.dW1ovT, D, D, D, D);                             // dV_node = A dV_f,node
        u[5].batch = H; u[5].b_bs[0] = CpD; u[5].c_bs = CpD;
        u[6] = g_nt(lw.dW1ovT_f + (int64_t)C * D, D, nullptr, lw.W1ovT, D, vs, D, 1, D, D, nullptr);   // vslab h: V_node dV_f[C]^T
        u[6].batch = H; u[6].a_bs[0] = CpD; u[6].b_bs[0] = CpD; u[6].c_bs = D;
        u[7] = g_nn(lw.db1_f, D, p.w1 + E, E + D, vs + H * D, D, 1, D, D);                    // vslab H: W1b^T db1_f
        u[8] = g_nt(lw.dW1b_f, D, nullptr, A, D, g.w1 + E, E + D, D, D, D, nullptr);          // dW1[:, E:] += dW1b_f A^T
        u[8].accumulate = 1;
        RUN(pfo_gemm_multi_launch(u, 9, sf));
      }
      {
        PfoGemm u[3];
        u[0] = g_nt(lw.dT1, D, nullptr, A, D, lw.dWqk, D, HCp, D, D, nullptr);                // dQ = dT1 A^T
        u[1] = g_nn(lw.Wqk, D, lw.dT1, D, sl + (2 * H + 1) * DD, D, D, D, HCp);               // slab 2H+1: Q^T dT1 (all heads' rows)
        u[1].a_kmajor = 1;
        u[2] = g_nn(lw.gqk, HCp, lw.Wqk, D, vs + (H + 1) * D, D, 1, D, HCp);                  // vslab H+1: Q^T dt
        RUN(pfo_gemm_multi_launch(u, 3, sf));
      }
      {
        PFO_REQUIRE(H + 3 <= PFO_RANK1_MAX, "too many heads");
        PfoRank1 r[PFO_RANK1_MAX];
        int nr = 0;
        r[nr].u = lw.gqk; r[nr].v = bv; r[nr].M = HCp; r[nr].N = D; r[nr].out = lw.dWqk; r[nr].ldo = D; ++nr;                    // dQ += dt (x) b
        r[nr].u = lw.db1_f; r[nr].v = bv; r[nr].M = D; r[nr].N = D; r[nr].out = g.w1 + E; r[nr].ldo = E + D; ++nr;               // dW1[:, E:] += db1_f (x) b
        for (int h = 0; h < H; ++h) {                                                                                             // dV_node += b (x) dV_f[C]
          r[nr].u = bv; r[nr].v = lw.dW1ovT_f + ((int64_t)h * Cp + C) * D; r[nr].M = D; r[nr].N = D;
          r[nr].out = lw.dW1ovT + (int64_t)h * CpD; r[nr].ldo = D; ++nr;
        }
        RUN(pfo_rank1_multi_launch(r, nr, sf));
        PfoSumSlabs q[3];
        q[0].dst = gp.w2; q[0].src = sl; q[0].stride = DD; q[0].count = DD; q[0].n_slabs = 2 * H + 2;                             // dA
        q[1].dst = gp.b2; q[1].src = vs; q[1].stride = D; q[1].count = D; q[1].n_slabs = H + 2;                                   // db
        q[2].dst = g.b1; q[2].src = lw.db1_f; q[2].stride = 0; q[2].count = D; q[2].n_slabs = 1;                                  // db1 += db1_f
        RUN(pfo_sum_slabs_launch(q, 3, sf));
        PfoRank1 ta;                                                                                                              // dA += sum_h t_node (x) gqk_f,node
        ta.u = lw.tq; ta.v = lw.gqk_f; ta.M = D; ta.N = D; ta.out = gp.w2; ta.ldo = D; ta.reps = H; ta.u_rs = Cp; ta.v_rs = Cp;
        RUN(pfo_rank1_multi_launch(&ta, 1, sf));
      }
    }
    // Half A hangs off dW1ovT (the weight gradients over the instances), half B off dWqk / gqk (those over the table rows).
    // At layer 1 the two sources finish ~100 us apart on the side stream, so half A gets a stream of its own and is done
    // before half B starts.
    {
      PfoGemm ca[3], cb[3], c2[3];
      ca[0] = g_nn(p.wv, C, lw.dW1ovT, D, lw.dW1oT, D, dh, D, C);                          // dW1oT_h = Wv_h dW1ovT_h
      ca[0].batch = H; ca[0].a_bs[0] = (int64_t)dh * C; ca[0].b_bs[0] = (int64_t)Cp * D; ca[0].c_bs = (int64_t)dh * D;
      ca[1] = g_nt(lw.W1oT, D, nullptr, lw.dW1ovT, D, g.wv, C, dh, C, D, nullptr);          // dWv_h += W1oT_h dW1ovT_h^T
      ca[1].batch = H; ca[1].a_bs[0] = (int64_t)dh * D; ca[1].b_bs[0] = (int64_t)Cp * D; ca[1].c_bs = (int64_t)dh * C;
      ca[1].accumulate = 1;
      ca[2] = g_nt(lw.W1oT, D, nullptr, lw.dW1ovT + (int64_t)C * D, D, g.b_in + 2 * E, 1, dh, 1, D, nullptr);   // dbv_h += W1oT_h du_h
      ca[2].batch = H; ca[2].a_bs[0] = (int64_t)dh * D; ca[2].b_bs[0] = (int64_t)Cp * D; ca[2].c_bs = dh;
      ca[2].accumulate = 1;
      cb[0] = g_nt(p.wq, E, nullptr, lw.dWqk, D, g.wk, C, dh, C, D, nullptr);               // dWk_h += Wq_h[:, :D] dWqk_h^T
      cb[0].batch = H; cb[0].a_bs[0] = (int64_t)dh * E; cb[0].b_bs[0] = (int64_t)Cp * D; cb[0].c_bs = (int64_t)dh * C;
      cb[0].accumulate = 1;
      cb[1] = g_nn(p.wk, C, lw.dWqk, D, g.wq, E, dh, D, C);                                 // dWq_h[:, :D] += Wk_h dWqk_h
      cb[1].batch = H; cb[1].a_bs[0] = (int64_t)dh * C; cb[1].b_bs[0] = (int64_t)Cp * D; cb[1].c_bs = (int64_t)dh * E;
      cb[1].accumulate = 1;
      cb[2] = g_nt(p.wk, C, nullptr, lw.gqk, C, lw.gq, 1, dh, 1, C, nullptr);                // d cq_h = Wk_h gqk_h
      cb[2].batch = H; cb[2].a_bs[0] = (int64_t)dh * C; cb[2].b_bs[0] = Cp; cb[2].c_bs = dh;
      const float* dc = lw.dW1ovT + (int64_t)(C + 1) * D;                                   // gradient of (W1 bo)^T
      c2[0] = g_nt(lw.dW1oT, D, nullptr, p.wo, E, g.w1, E + D, D, E, E, nullptr);           // dW1[:, :E] += dW1o Wo^T
      c2[0].a_kmajor = 1; c2[0].accumulate = 1;
      c2[1] = g_nt(p.w1, E + D, nullptr, lw.dW1oT, D, g.wo, E, E, E, D, nullptr);           // dWo += W1[:, :E]^T dW1o
      c2[1].a_kmajor = 1; c2[1].accumulate = 1;
      c2[2] = g_nt(p.w1, E + D, nullptr, dc, D, g.bo, 1, E, 1, D, nullptr);                // dbo += W1[:, :E]^T dc
      c2[2].a_kmajor = 1; c2[2].accumulate = 1;
      // the outer-product halves of the composite gradients, one launch per half (disjoint outputs):
      //   A: W1ovT row C = bv_h^T W1oT_h (into dW1oT, which c2 reads),  dW1[:, :E] += dc (x) bo     B: cqk_h = Wk_h^T cq_h
      PFO_REQUIRE(H + 1 <= PFO_RANK1_MAX, "too many heads");
      PfoRank1 ra[PFO_RANK1_MAX], rb[PFO_RANK1_MAX];
      for (int h = 0; h < H; ++h) {
        ra[h].u = p.b_in + 2 * E + h * dh; ra[h].ldu = 1; ra[h].v = lw.dW1ovT + ((int64_t)h * Cp + C) * D; ra[h].ldv = 1;
        ra[h].M = dh; ra[h].N = D; ra[h].out = lw.dW1oT + (int64_t)h * dh * D; ra[h].ldo = D;
        rb[h].u = lw.cq + h * dh; rb[h].ldu = 1; rb[h].v = lw.gqk + (int64_t)h * Cp; rb[h].ldv = 1;
        rb[h].M = dh; rb[h].N = C; rb[h].out = g.wk + (int64_t)h * dh * C; rb[h].ldo = C;
      }
      ra[H].u = dc; ra[H].ldu = 1; ra[H].v = p.bo; ra[H].ldv = 1; ra[H].M = D; ra[H].N = E; ra[H].out = g.w1; ra[H].ldo = E + D;
      hipStream_t sa = sd.s2, sb = sd.s2;
      if (l == 1) {
        // both sources were launched on side streams (or, bracketed for profiling, the first on the main stream): no need to
        // hold the chain behind the main stream's d h0 contraction
        sb = ss;
        HIPOK(hipStreamWaitEvent(sa, sd.tn_a_done, 0), "event wait failed");
      }
      if (l == 1) PFO_MARK("@side2.L1.chainA.begin", sa);
      RUN(pfo_gemm_multi_launch(ca, 3, sa));
      RUN(pfo_rank1_multi_launch(ra, H + 1, sa));
      RUN(pfo_gemm_multi_launch(c2, 3, sa));
      if (l == 1) PFO_MARK("@side2.L1.chainA.end", sa);
      else PFO_MARK("@side2.L2.chainA.end", sa);
      RUN(pfo_gemm_multi_launch(cb, 3, sb));
      RUN(pfo_rank1_multi_launch(rb, H, sb));
      if (l == 1) PFO_MARK("@side1.L1.chainB.end", sb);
      else PFO_MARK("@side2.L2.chainB.end", sb);
      if (l == L && L >= 2) {
        // The top layer's folded query-bias backward runs here, on the stream of its chain, with its time-bias term parked in
        // tb_part (the final launch adds it): from this point every gradient of the top layer's parameter block
        // [layer[L-1].wq, total) is FINAL - a data-parallel caller starts all-reducing that block while layers L-1 .. 1 are
        // still being differentiated (pfo_tgn_grad_split, distributed.py)
        const float* gq1[1] = {lw.gq}; const float* wq1[1] = {p.wq};
        float* dbq1[1] = {g.b_in}; float* dwq1[1] = {g.wq};
        RUN(pfo_cq_backward_launch(gq1, wq1, 1, P.tb, D, dbq1, dwq1, G.tb, w.tb_part, nullptr, nullptr, 0, nullptr, sb));
        if (top_ready_event) HIPOK(hipEventRecord((hipEvent_t)top_ready_event, sb), "event record failed");
      }
    }
      return PFO_OK;
    };
    static const int defer = getenv("PFO_DEFER_CHAIN") ? atoi(getenv("PFO_DEFER_CHAIN")) : 1;      // A/B switch
    if (l > 1 && defer && !pfo_prof_on()) deferred_chain = chain_back;
    else RUN(chain_back());
  }
  if (deferred_chain) { RUN(deferred_chain()); deferred_chain = nullptr; }

  // ---- GRU parameters (messages and stored memory are constants: SURVEY App. A-6)
  bool main_done_bound = false;
  if (c->use_memory) {
    PfoRange range_gru("backward GRU");
    // (the GRU's backward covers the rows the layers read: rows only the extra list names carry no gradient)
    if (!gates_fused)
      RUN(pfo_gru_gates_bwd_launch(w.gates, w.gi, w.gh, w.h_rows, w.hm, w.n_core, capP, D, w.d_h0, n_rep, rep_stride,
                                   w.dx_tab, det, s));
    PFO_MARK("bwd.gru.gates", s);
    {
      PfoTnProblem gp[2];
      gp[0].A = w.gi; gp[0].lda = 3 * D; gp[0].B = w.msg_rows; gp[0].ldb = d.M; gp[0].M = 3 * D; gp[0].N = d.M;
      gp[0].C = G.w_ih; gp[0].ldc = d.M; gp[0].bias_out = G.b_ih;
      gp[1].A = w.gh; gp[1].lda = 3 * D; gp[1].B = w.h_rows; gp[1].ldb = D; gp[1].M = 3 * D; gp[1].N = D;
      gp[1].C = G.w_hh; gp[1].ldc = D; gp[1].bias_out = G.b_hh;
      // (a deferred join's main_done event rides on the slab reduce, the caller's stream's last launch of this call)
      main_done_bound = b->defer_join && bind_events;
      PFO_RUN_BOUND(main_done_bound, sd.main_done, 1, s, pfo_gemm_tn_group_launch(gp, 2, capP, w.n_core, w.slabs, w.slab_floats, s));
      PFO_MARK("bwd.gru.tn", s);
    }
  }
  // ---- the last small launches go to the first side stream, beside the GRU's weight gradients on the caller's stream:
  // [deterministic mode: the attention backwards' slab rows fold, in row order, into bin 0 of the (otherwise empty) fp64 bins]
  // then ONE launch finishes the time-encoder gradients: the folded query-bias backward of layers 1 .. L-1 (the top layer's ran
  // with its chain), + its parked time-bias term, + the fold of the fp64 partial sums into time_w / time_b (fixed order).
  // That stream has seen every attention backward (it waited for tn_b, recorded behind layer 1's) and runs layer 1's chain.
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &cap) != hipSuccess) cap = hipStreamCaptureStatusActive;
  const bool chained = cap == hipStreamCaptureStatusNone;
  // The side streams are joined in a chain - the first waits for the second, the caller's stream for the first: one wait on
  // the critical path instead of two - except while the caller's stream is being captured into a HIP graph, where that
  // topology makes hipStreamEndCapture of ROCm 7.0 segfault: there the caller's stream waits for both itself.
  HIPOK(hipEventRecord(sd.done2, sd.s2), "event record failed");
  HIPOK(hipStreamWaitEvent(chained ? ss : s, sd.done2, 0), "event wait failed");
  if (!chained) {                                             // (the launches below read the second side stream's results)
    HIPOK(hipEventRecord(sd.tn_a, s), "event record failed");
    HIPOK(hipStreamWaitEvent(ss, sd.tn_a, 0), "event wait failed");
  }
  if (det) RUN(pfo_fold_parts_launch(w.dtime_slab, (int)det_rows, 2 * D, nullptr, 0, w.fold_scratch, w.tickets, ss, w.dtime));
  {
    const float *gq[PFO_MAX_LAYERS], *wq[PFO_MAX_LAYERS];
    float *dbq[PFO_MAX_LAYERS], *dwq[PFO_MAX_LAYERS];
    const int nl = L >= 2 ? L - 1 : L;
    for (int l = 1; l <= nl; ++l) { gq[l - 1] = w.layer[l].gq; wq[l - 1] = P.l[l].wq; dbq[l - 1] = G.l[l].b_in; dwq[l - 1] = G.l[l].wq; }
    RUN(pfo_cq_backward_launch(gq, wq, nl, P.tb, D, dbq, dwq, G.tb, nullptr, L >= 2 ? w.tb_part : nullptr, w.dtime,
                               pfo_attn_bwd_max_parts(), G.tw, ss));                  // cq = Wq[:, D:] cos(b) + bq
  }
  PFO_MARK("@side1.cq.end", ss);
  if (b->defer_join && chained && bind_events) {               // (bind_events: not under capture - side_join)
    // the end of the backward stays on the side stream (pfo_tgn_batch.defer_join): it waits for the caller's stream's last
    // launch instead of the other way round - the caller's stream is free for the next batch's neighbour sampling
    if (!main_done_bound) HIPOK(hipEventRecord(sd.main_done, s), "event record failed");
    HIPOK(hipStreamWaitEvent(ss, sd.main_done, 0), "event wait failed");
    sd.side_gen += 1;
    sd.last_backward_deferred = true;
    sd.deferred_from = s;
  } else {
    sd.last_backward_deferred = false;
    sd.deferred_from = nullptr;
    HIPOK(hipEventRecord(sd.done, ss), "event record failed");
    HIPOK(hipStreamWaitEvent(s, sd.done, 0), "event wait failed");
  }
  PFO_MARK("bwd.end", s);
  return PFO_OK;
}

int pfo_adam_step_ranges_impl(float*, float*, float*, float*, int32_t, const int64_t*, const int64_t*, const int32_t*, float, float, float,
                              float, bool, void*);
static int adam_side_impl(float* param, float* grad, float* exp_avg, float* exp_avg_sq, int32_t n_ranges,
                          const int64_t* lo, const int64_t* hi, const int32_t* step, float lr, float beta1, float beta2,
                          float eps, bool zero_grad) {
  Side& sd = side();
  PFO_REQUIRE(sd.ok, "could not create the side stream");
  PFO_REQUIRE(sd.last_backward_deferred, "pfo_tgn_adam_side follows a backward that ran with pfo_tgn_batch.defer_join");
  const int rc = pfo_adam_step_ranges_impl(param, grad, exp_avg, exp_avg_sq, n_ranges, lo, hi, step, lr, beta1, beta2, eps, zero_grad, (void*)sd.s);
  sd.side_gen += 1;                                              // (a stream that joined between the backward and this call joins again)
  PFO_MARK("@side1.adam.end", sd.s);
  return rc;
}
extern "C" int pfo_tgn_adam_side(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int32_t n_ranges,
                                 const int64_t* lo, const int64_t* hi, const int32_t* step, float lr, float beta1, float beta2,
                                 float eps) {
  return adam_side_impl(param, const_cast<float*>(grad), exp_avg, exp_avg_sq, n_ranges, lo, hi, step, lr, beta1, beta2, eps, false);
}

extern "C" int pfo_tgn_adam_side_bucket(float* param, float* grad, float* exp_avg, float* exp_avg_sq, int32_t n_ranges,
                                        const int64_t* lo, const int64_t* hi, const int32_t* step, float lr, float beta1, float beta2,
                                        float eps, int32_t flags) {
  Side& sd = side();
  PFO_REQUIRE(sd.ok, "could not create the side stream");
  const int bucket = flags & 3;
  PFO_REQUIRE(flags >= 0 && flags < 8 && bucket <= 2, "flags: bucket 0 (plain), 1 (first use) or 2 (later bucket of the same step), + 4 = clear the gradient ranges");
  PFO_REQUIRE(bucket != 2 || sd.early_ok, "a later bucket follows a first-use bucket of the same step");
  const int rc = adam_side_impl(param, grad, exp_avg, exp_avg_sq, n_ranges, lo, hi, step, lr, beta1, beta2, eps, (flags & 4) != 0);
  if (rc != PFO_OK) { sd.early_ok = false; return rc; }
  if (bucket == 1) {
    HIPOK(hipEventRecord(sd.early_done, sd.s), "event record failed");
    sd.early_ok = true;
  }
  if (bucket == 0) sd.early_ok = false;
  sd.early_gen = sd.side_gen;                                    // (bucket 2: the event of bucket 1 still stands for the stream's tail)
  return PFO_OK;
}

extern "C" void* pfo_tgn_side_stream(void) {
  Side& sd = side();
  if (!sd.ok) { pfo_set_error("pfo_tgn_side_stream: could not create the side stream"); return nullptr; }
  return (void*)sd.s;
}

extern "C" int pfo_tgn_join(void* stream) {
  Side& sd = side();
  PFO_REQUIRE(sd.ok, "could not create the side stream");
  return side_join(sd, (hipStream_t)stream);
}

// =============================================================================================
extern "C" int pfo_tgn_update_state(const pfo_tgn_config* c, const pfo_tgn_state* st, const int32_t* src,
                                    const int32_t* dst, const double* ts, const int32_t* eidx, int32_t B, void* workspace,
                                    void* stream) {
  if (int rc = check_cfg(c)) return rc;
  if (!c->use_memory) return PFO_OK;
  PFO_REQUIRE(st && workspace && src && dst && ts && eidx && B >= 1, "bad arguments");
  const Ws w = carve(c, workspace);
  PfoRange range("pfo_tgn_update_state");
  RUN(side_join(side(), (hipStream_t)stream));
  return state_update(c, st, w, src, dst, ts, eidx, B, (hipStream_t)stream);
}
