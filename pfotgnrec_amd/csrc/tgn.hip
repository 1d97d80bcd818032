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
#include <math.h>
#include <string.h>

extern "C" int pfo_tnbr_sample(const int64_t*, const int32_t*, const int32_t*, const double*, int64_t, const int32_t*,
                               const double*, int64_t, int32_t, int32_t, const int64_t*, uint64_t, uint64_t, int32_t*,
                               int32_t*, float*, float*, int32_t*, double*, void*);

namespace {

struct LayerWs {
  float *Q, *QK, *attw, *ssum, *ctx, *O, *attn_out, *h1, *Hout, *cq;
  uint8_t* inv;
};
struct Ws {
  int32_t* nodes[PFO_MAX_LAYERS + 1];
  double* ts[PFO_MAX_LAYERS + 1];
  int32_t* eidx[PFO_MAX_LAYERS + 1];
  float* dt[PFO_MAX_LAYERS + 1];
  int32_t *slot, *touched, *n_touched, *scan, *idx0, *winner;
  float *gi, *gh, *upd_mem, *h0_tab, *d_h0, *msg_rows, *h_rows;
  uint8_t* hm;
  float *cosb, *zero, *gq;
  LayerWs layer[PFO_MAX_LAYERS + 1];
  float *dh1, *dattn, *dO, *dctx, *dQK, *dQ, *dx1;
  float* dH[PFO_MAX_LAYERS + 1];
  float *slabs, *colsum;
  double* dtime;
  int64_t slab_floats;
  int64_t bytes;
};

struct Dims {
  int L, D, Ef, H, E, C, dh, M;
  int64_t ncap[PFO_MAX_LAYERS + 1];
  int64_t capP;
};

Dims dims_of(const pfo_tgn_config* c) {
  Dims d;
  d.L = c->n_layers; d.D = c->D; d.Ef = c->Ef; d.H = c->n_heads;
  d.E = 2 * d.D; d.C = 2 * d.D + d.Ef; d.dh = d.E / d.H; d.M = 3 * d.D + d.Ef;
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
  w.zero = take<float>(p, 64);
  w.cosb = take<float>(p, d.D);
  w.gq = take<float>(p, d.E);
  if (c->use_memory) {
    w.slot = take<int32_t>(p, c->n_nodes);
    w.winner = take<int32_t>(p, c->n_nodes);
    w.touched = take<int32_t>(p, d.capP);
    w.n_touched = take<int32_t>(p, 64);
    w.scan = take<int32_t>(p, pfo_compact_scratch_ints(c->n_nodes));
    w.idx0 = take<int32_t>(p, d.ncap[0]);
    w.gi = take<float>(p, d.capP * 3 * d.D);
    w.gh = take<float>(p, d.capP * 3 * d.D);
    w.upd_mem = take<float>(p, d.capP * d.D);
    w.h0_tab = take<float>(p, d.capP * d.D);
    w.d_h0 = take<float>(p, d.capP * d.D);
    w.msg_rows = take<float>(p, d.capP * d.M);
    w.h_rows = take<float>(p, d.capP * d.D);
    w.hm = take<uint8_t>(p, d.capP);
  }
  for (int l = 1; l <= d.L; ++l) {
    const int64_t N = d.ncap[l];
    LayerWs& lw = w.layer[l];
    lw.cq = take<float>(p, d.E);
    lw.Q = take<float>(p, N * d.E);
    lw.QK = take<float>(p, N * d.H * d.C);
    lw.attw = take<float>(p, N * d.H * Km);
    lw.ssum = take<float>(p, N * d.H);
    lw.inv = take<uint8_t>(p, N);
    lw.ctx = take<float>(p, N * d.H * d.C);
    lw.O = take<float>(p, N * d.E);
    lw.attn_out = take<float>(p, N * d.E);
    lw.h1 = take<float>(p, N * d.D);
    lw.Hout = take<float>(p, N * d.D);
    if (l < d.L) w.dH[l] = take<float>(p, N * d.D);
  }
  const int64_t N1 = d.ncap[1];
  w.dh1 = take<float>(p, N1 * d.D);
  w.dattn = take<float>(p, N1 * d.E);
  w.dO = take<float>(p, N1 * d.E);
  w.dctx = take<float>(p, N1 * d.H * d.C);
  w.dQK = take<float>(p, N1 * d.H * d.C);
  w.dQ = take<float>(p, N1 * d.E);
  w.dx1 = take<float>(p, N1 * d.D);
  w.slab_floats = SLAB_FLOATS;
  w.slabs = take<float>(p, w.slab_floats);
  w.colsum = take<float>(p, pfo_colsum_scratch_floats(3 * d.D + 2 * d.E + d.M));
  w.dtime = take<double>(p, (int64_t)pfo_attn_bwd_max_parts() * 2 * d.D);
  w.bytes = p - reinterpret_cast<char*>(base);
  return w;
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
  const Ws w = carve(c, workspace);
  hipStream_t s = (hipStream_t)stream;
  pfo_tgn_layout lay;
  RUN(pfo_tgn_param_layout(c, &lay));
  Params P;
  bind(lay, st->params, P, d.L, c->use_memory != 0);
  const int L = d.L, D = d.D, Ef = d.Ef, H = d.H, E = d.E, C = d.C, dh = d.dh, K = b->K;

  // ---- frontier expansion: K1 per level (utils.py:163-219 called from embedding_module.py:125)
  PFO_REQUIRE(hipMemcpyAsync(w.nodes[L], b->roots, (size_t)b->R * sizeof(int32_t), hipMemcpyDeviceToDevice, s) == hipSuccess,
              "copy failed");
  PFO_REQUIRE(hipMemcpyAsync(w.ts[L], b->root_ts, (size_t)b->R * sizeof(double), hipMemcpyDeviceToDevice, s) == hipSuccess,
              "copy failed");
  for (int l = L; l >= 1; --l) {
    const int64_t* dr = (b->uniform == 1) ? b->draws[L - l] : nullptr;
    PFO_REQUIRE(b->uniform != 1 || dr, "missing draws for a level");
    RUN(pfo_tnbr_sample(st->indptr, st->adj_nbr, st->adj_eidx, st->adj_ts, c->n_nodes, w.nodes[l], w.ts[l], n[l], K,
                        b->uniform, dr, b->seed, b->offset + (uint64_t)l * 0x100000000ull, nullptr, w.eidx[l], nullptr,
                        w.dt[l], w.nodes[l - 1], l > 1 ? w.ts[l - 1] : nullptr, stream));
  }

  // ---- lazy memory update for the touched nodes (tgn.py:251; memory_updater.py:35-53)
  const float* tab0;
  const int32_t* idx0;
  if (c->use_memory) {
    PFO_REQUIRE(b->n_extra >= 0 && b->n_extra <= 2 * c->max_batch, "n_extra exceeds 2 * max_batch");
    PFO_REQUIRE(b->n_extra == 0 || b->extra_nodes, "null extra_nodes");
    RUN(pfo_touch_compact_launch(w.nodes[0], n[0], b->extra_nodes, b->n_extra, c->n_nodes, w.slot, w.touched, w.n_touched,
                                 w.scan, s));
    RUN(pfo_remap_launch(w.nodes[0], n[0], w.slot, w.idx0, s));
    const int capP = (int)std::min<int64_t>(c->n_nodes, n[0] + b->n_extra);
    RUN(pfo_pack_rows_launch(st->msg_table, d.M, st->memory, D, st->has_msg, w.touched, w.n_touched, capP, w.msg_rows,
                             w.h_rows, w.hm, s));
    PfoGemm gi = g_nt(w.msg_rows, d.M, nullptr, P.w_ih, d.M, w.gi, 3 * D, capP, 3 * D, d.M, P.b_ih);
    gi.m_dev = w.n_touched;
    RUN(pfo_gemm_launch(gi, s));
    PfoGemm gh = g_nt(w.h_rows, D, nullptr, P.w_hh, D, w.gh, 3 * D, capP, 3 * D, D, P.b_hh);
    gh.m_dev = w.n_touched;
    RUN(pfo_gemm_launch(gh, s));
    RUN(pfo_gru_gates_fwd_launch(w.gi, w.gh, w.h_rows, st->node_feat, w.hm, w.touched, w.n_touched, capP, D, w.upd_mem,
                                 w.h0_tab, s));
    tab0 = w.h0_tab;
    idx0 = w.idx0;
  } else {
    tab0 = st->node_feat;
    idx0 = w.nodes[0];
  }

  // ---- query time feature cos(fma(0, w, b)) (embedding_module.py:92)
  PFO_REQUIRE(hipMemsetAsync(w.zero, 0, 64 * sizeof(float), s) == hipSuccess, "memset failed");
  RUN(pfo_time_encode(w.zero, 1, P.tw, P.tb, D, w.cosb, stream));

  const float scale = 1.0f / sqrtf((float)dh);
  for (int l = 1; l <= L; ++l) {
    const int N = (int)n[l];
    const LayerWs& lw = w.layer[l];
    const auto& p = P.l[l];
    const float* xA = (l == 1) ? tab0 : w.layer[l - 1].Hout;
    const int32_t* x_idx = (l == 1) ? idx0 : nullptr;

    // folded query bias cq = Wq[:, D:] cos(b) + bq, then Q = x Wq[:, :D]^T + cq
    RUN(pfo_gemm_launch(g_nt(w.cosb, D, nullptr, p.wq + D, E, lw.cq, E, 1, E, D, p.b_in), s));
    RUN(pfo_gemm_launch(g_nt(xA, D, x_idx, p.wq, E, lw.Q, E, N, E, D, lw.cq), s));
    // folded key projection: qk_h = Wk_h^T Q_h   (Wk_h = rows [h dh, (h+1) dh) of k_proj_weight, a [dh, C] k-major operand)
    {
      PfoGemm g = g_nn(lw.Q, E, p.wk, C, lw.QK, (int64_t)H * C, N, C, dh);
      g.batch = H; g.a_bs[0] = dh; g.b_bs[0] = (int64_t)dh * C; g.c_bs = C;
      RUN(pfo_gemm_launch(g, s));
    }
    PfoAttn a;
    a.N = N; a.K = K; a.D = D; a.Ef = Ef; a.H = H; a.dh = dh;
    a.QK = lw.QK; a.nbr_tab = xA; a.nbr_ld = D;
    a.nbr_row = (l == 1) ? idx0 + N : nullptr;
    a.nbr_row_base = N;
    a.nbr_ids = w.nodes[l - 1] + N;
    a.edge_feat = st->edge_feat; a.eidx = w.eidx[l]; a.dt = w.dt[l]; a.tw = P.tw; a.tb = P.tb;
    a.scale = scale; a.dropout_p = b->dropout_p; a.seed = b->seed; a.offset = b->offset + 0x51ED0000ull + (uint64_t)l;
    a.ctx = lw.ctx; a.attw = lw.attw; a.ssum = lw.ssum; a.inv = lw.inv;
    RUN(pfo_attn_fwd_launch(a, s));
    // folded value projection O_h = Wv_h ctx_h + bv_h * sum_j a'_jh
    {
      PfoGemm g = g_nt(lw.ctx, (int64_t)H * C, nullptr, p.wv, C, lw.O, E, N, dh, C, p.b_in + 2 * E);
      g.batch = H; g.a_bs[0] = C; g.b_bs[0] = (int64_t)dh * C; g.c_bs = dh; g.bias_bs = dh;
      if (b->dropout_p > 0.f) { g.row_scale = lw.ssum; g.rs_ld = H; g.rs_bs = 1; }
      RUN(pfo_gemm_launch(g, s));
    }
    {
      PfoGemm g = g_nt(lw.O, E, nullptr, p.wo, E, lw.attn_out, E, N, E, E, p.bo);
      g.row_zero = lw.inv;                                   // temporal_attention.py:84
      RUN(pfo_gemm_launch(g, s));
    }
    {
      // MergeLayer fc1 on [attn_out | x] as two K-concatenated sources, ReLU fused (utils.py:14-16)
      PfoGemm g = g_nt(lw.attn_out, E, nullptr, p.w1, E + D, lw.h1, D, N, D, E, p.b1);
      g.A[1] = xA; g.lda[1] = D; g.a_idx[1] = x_idx; g.B[1] = p.w1 + E; g.ldb[1] = E + D; g.K[1] = D;
      g.relu = 1;
      RUN(pfo_gemm_launch(g, s));
    }
    RUN(pfo_gemm_launch(g_nt(lw.h1, D, nullptr, p.w2, D, lw.Hout, D, N, D, D, p.b2), s));
  }
  PFO_REQUIRE(hipMemcpyAsync(emb_out, w.layer[L].Hout, (size_t)b->R * D * sizeof(float), hipMemcpyDeviceToDevice, s) ==
                  hipSuccess,
              "copy failed");
  return PFO_OK;
}

// =============================================================================================
extern "C" int pfo_tgn_backward(const pfo_tgn_config* c, const pfo_tgn_state* st, const pfo_tgn_batch* b, void* workspace,
                                const float* d_emb, float* grad, void* stream) {
  if (int rc = check_cfg(c)) return rc;
  PFO_REQUIRE(st && workspace && d_emb && grad && st->params, "null argument");
  int64_t n[PFO_MAX_LAYERS + 1];
  RUN(level_sizes(c, b, n));
  const Dims d = dims_of(c);
  const Ws w = carve(c, workspace);
  hipStream_t s = (hipStream_t)stream;
  pfo_tgn_layout lay;
  RUN(pfo_tgn_param_layout(c, &lay));
  Params P;
  Grads G;
  bind(lay, st->params, P, d.L, c->use_memory != 0);
  bind(lay, grad, G, d.L, c->use_memory != 0);
  const int L = d.L, D = d.D, Ef = d.Ef, H = d.H, E = d.E, C = d.C, dh = d.dh, K = b->K;
  const float scale = 1.0f / sqrtf((float)dh);
  const float* tab0 = c->use_memory ? w.h0_tab : st->node_feat;
  const int32_t* idx0 = c->use_memory ? w.idx0 : w.nodes[0];
  const int capP = (int)std::min<int64_t>(c->n_nodes, n[0] + b->n_extra);
  if (c->use_memory)
    PFO_REQUIRE(hipMemsetAsync(w.d_h0, 0, (size_t)capP * D * sizeof(float), s) == hipSuccess, "memset failed");

  for (int l = L; l >= 1; --l) {
    const int N = (int)n[l];
    const LayerWs& lw = w.layer[l];
    const auto& p = P.l[l];
    const auto& g = G.l[l];
    const float* dOut = (l == L) ? d_emb : w.dH[l];
    const float* xA = (l == 1) ? tab0 : w.layer[l - 1].Hout;
    const int32_t* x_idx = (l == 1) ? idx0 : nullptr;
    float* dx = (l == 1) ? w.dx1 : w.dH[l - 1];              // rows [0, N) of the previous level's gradient

    // Data gradients first (a chain of GEMMs + the attention core); every weight / bias gradient of the layer is
    // then taken in ONE grouped split-K launch (bias gradients ride along as an extra column).
    PfoTnProblem tn[16];
    int ntn = 0;
    auto add_tn = [&](const float* A, int64_t lda, const float* B, int64_t ldb, const int32_t* b_idx, int M_, int N_, float* C,
                      int64_t ldc, float* bias_out) -> PfoTnProblem& {
      PfoTnProblem& q = tn[ntn++];
      q = PfoTnProblem();
      q.A = A; q.lda = lda; q.B = B; q.ldb = ldb; q.b_idx = b_idx; q.M = M_; q.N = N_; q.C = C; q.ldc = ldc; q.bias_out = bias_out;
      return q;
    };
    // fc2 (utils.py:17)
    add_tn(dOut, D, lw.h1, D, nullptr, D, D, g.w2, D, g.b2);
    {
      PfoGemm q = g_nn(dOut, D, p.w2, D, w.dh1, D, N, D, D);
      q.relu_src = lw.h1; q.relu_ld = D;                     // ReLU backward
      RUN(pfo_gemm_launch(q, s));
    }
    // fc1 on [attn_out | x]
    add_tn(w.dh1, D, lw.attn_out, E, nullptr, D, E, g.w1, E + D, g.b1);
    add_tn(w.dh1, D, xA, D, x_idx, D, D, g.w1 + E, E + D, nullptr);
    {
      PfoGemm q = g_nn(w.dh1, D, p.w1, E + D, w.dattn, E, N, E, D);
      q.row_zero = lw.inv;                                   // zero-filled rows pass no gradient (temporal_attention.py:84)
      RUN(pfo_gemm_launch(q, s));
    }
    RUN(pfo_gemm_launch(g_nn(w.dh1, D, p.w1 + E, E + D, dx, D, N, D, D), s));
    // out_proj
    add_tn(w.dattn, E, lw.O, E, nullptr, E, E, g.wo, E, g.bo);
    RUN(pfo_gemm_launch(g_nn(w.dattn, E, p.wo, E, w.dO, E, N, E, E), s));
    // folded value projection: d bv_h = sum_n ssum_h[n] dO_h[n] (ssum == 1 without dropout)
    for (int h = 0; h < H; ++h) {
      PfoTnProblem& q = add_tn(w.dO + h * dh, E, lw.ctx + h * C, (int64_t)H * C, nullptr, dh, C, g.wv + (int64_t)h * dh * C, C,
                               g.b_in + 2 * E + h * dh);
      if (b->dropout_p > 0.f) { q.ones_scale = lw.ssum + h; q.os_ld = H; }
    }
    {
      PfoGemm q = g_nn(w.dO, E, p.wv, C, w.dctx, (int64_t)H * C, N, C, dh);
      q.batch = H; q.a_bs[0] = dh; q.b_bs[0] = (int64_t)dh * C; q.c_bs = C;
      RUN(pfo_gemm_launch(q, s));
    }
    // attention core
    PfoAttn a;
    a.N = N; a.K = K; a.D = D; a.Ef = Ef; a.H = H; a.dh = dh;
    a.QK = lw.QK; a.nbr_tab = xA; a.nbr_ld = D;
    a.nbr_row = (l == 1) ? idx0 + N : nullptr;
    a.nbr_row_base = N;
    a.nbr_ids = w.nodes[l - 1] + N;
    a.edge_feat = st->edge_feat; a.eidx = w.eidx[l]; a.dt = w.dt[l]; a.tw = P.tw; a.tb = P.tb;
    a.scale = scale; a.dropout_p = b->dropout_p; a.seed = b->seed; a.offset = b->offset + 0x51ED0000ull + (uint64_t)l;
    a.ctx = lw.ctx; a.attw = lw.attw; a.ssum = lw.ssum; a.inv = lw.inv;
    a.dctx = w.dctx; a.dO = w.dO; a.bv = p.b_in + 2 * E; a.dQK = w.dQK;
    if (l == 1) { a.d_nbr = c->use_memory ? w.d_h0 : nullptr; a.d_nbr_ld = D; }
    else        { a.d_nbr = w.dH[l - 1]; a.d_nbr_ld = D; }
    a.dtime_part = w.dtime;
    int n_parts = 0;
    RUN(pfo_attn_bwd_launch(a, &n_parts, s));
    RUN(pfo_fold_parts_launch(w.dtime, n_parts, 2 * D, G.tw, 1, s));      // time_w and time_b are adjacent in the layout
    // folded key projection (the key bias has an exactly zero gradient: it cancels in the softmax)
    for (int h = 0; h < H; ++h)
      add_tn(lw.Q + h * dh, E, w.dQK + h * C, (int64_t)H * C, nullptr, dh, C, g.wk + (int64_t)h * dh * C, C, nullptr);
    {
      PfoGemm q = g_nt(w.dQK, (int64_t)H * C, nullptr, p.wk, C, w.dQ, E, N, dh, C, nullptr);
      q.batch = H; q.a_bs[0] = C; q.b_bs[0] = (int64_t)dh * C; q.c_bs = dh;
      RUN(pfo_gemm_launch(q, s));
    }
    // query projection (x part); its bias column is gq = colsum(dQ), consumed by the folded-bias backward
    add_tn(w.dQ, E, xA, D, x_idx, E, D, g.wq, E, w.gq).bias_accumulate = 0;
    {
      PfoGemm q = g_nn(w.dQ, E, p.wq, E, dx, D, N, D, E);
      q.accumulate = 1;
      RUN(pfo_gemm_launch(q, s));
    }
    RUN(pfo_gemm_tn_group_launch(tn, ntn, N, nullptr, w.slabs, w.slab_floats, s));
    RUN(pfo_cq_backward_launch(w.gq, p.wq, P.tb, D, g.b_in, g.wq, G.tb, s));
    if (l == 1 && c->use_memory) RUN(pfo_scatter_add_rows_launch(w.dx1, D, idx0, N, D, w.d_h0, D, s));
  }

  // ---- GRU parameters (messages and stored memory are constants: SURVEY App. A-6)
  if (c->use_memory) {
    RUN(pfo_gru_gates_bwd_launch(w.gi, w.gh, w.h_rows, w.hm, w.n_touched, capP, D, w.d_h0, s));
    {
      PfoTnProblem gp[2];
      gp[0].A = w.gi; gp[0].lda = 3 * D; gp[0].B = w.msg_rows; gp[0].ldb = d.M; gp[0].M = 3 * D; gp[0].N = d.M;
      gp[0].C = G.w_ih; gp[0].ldc = d.M; gp[0].bias_out = G.b_ih;
      gp[1].A = w.gh; gp[1].lda = 3 * D; gp[1].B = w.h_rows; gp[1].ldb = D; gp[1].M = 3 * D; gp[1].N = D;
      gp[1].C = G.w_hh; gp[1].ldc = D; gp[1].bias_out = G.b_hh;
      RUN(pfo_gemm_tn_group_launch(gp, 2, capP, w.n_touched, w.slabs, w.slab_floats, s));
    }
  }
  return PFO_OK;
}

// =============================================================================================
extern "C" int pfo_tgn_update_state(const pfo_tgn_config* c, const pfo_tgn_state* st, const int32_t* src,
                                    const int32_t* dst, const double* ts, const int32_t* eidx, int32_t B, void* workspace,
                                    void* stream) {
  if (int rc = check_cfg(c)) return rc;
  if (!c->use_memory) return PFO_OK;
  PFO_REQUIRE(st && workspace && src && dst && ts && eidx && B >= 1, "bad arguments");
  const Ws w = carve(c, workspace);
  hipStream_t s = (hipStream_t)stream;
  pfo_tgn_layout lay;
  RUN(pfo_tgn_param_layout(c, &lay));
  const float* tw = st->params + lay.time_w;
  const float* tb = st->params + lay.time_b;
  RUN(pfo_persist_launch(src, dst, B, w.slot, w.upd_mem, st->has_msg, st->msg_time, st->memory, st->last_update, c->D, s));
  RUN(pfo_msg_store_launch(src, dst, ts, eidx, B, st->memory, st->last_update, st->edge_feat, tw, tb, c->D, c->Ef,
                           st->msg_table, st->msg_time, st->has_msg, w.winner, s));
  return PFO_OK;
}
