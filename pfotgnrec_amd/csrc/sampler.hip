// Integer / indexing kernels of the path: temporal neighbour lookup (K1), candidate-negative
// draw, mean-variance rank fusion (K2).  All results are index-exact against the reference.
//
// These are latency/HBM-bound lookups: no LDS tiling tricks, just enough lanes in flight and
// coalesced row-tail copies.  One query per 16-lane group so a wave runs four binary searches
// in lock-step, each search 16-ary (4 dependent loads for a 20k-entry row instead of 15).
#include "common.hpp"

#pragma clang fp contract(off)   // fp64 rank arithmetic must round exactly like numpy/python

// ---------------------------------------------------------------------------------------------
// K1: utils/utils.py:150-219
template <int MODE>
__global__ __launch_bounds__(256) void tnbr_sample_kernel(
    const int64_t* __restrict__ indptr, const int32_t* __restrict__ adj_nbr, const int32_t* __restrict__ adj_eidx,
    const double* __restrict__ adj_ts, int64_t n_nodes, const int32_t* __restrict__ q_nodes,
    const double* __restrict__ q_ts, int64_t n_q, int K, const int64_t* __restrict__ draws, uint64_t seed,
    uint64_t offset, const uint64_t* __restrict__ offset_dev, int32_t* __restrict__ out_nbr, int32_t* __restrict__ out_eidx,
    float* __restrict__ out_et, float* __restrict__ out_dt, int32_t* __restrict__ next_nodes, double* __restrict__ next_ts,
    int32_t* __restrict__ mark, int32_t* __restrict__ out_cnt, int32_t* __restrict__ clear_ptr, int64_t clear_ints) {
  // clear_ptr (optional; never the launch that sets `mark`): clear_ints words zeroed by this launch - the step's touched-node
  // flags and compaction scratch are cleared by the UPPER level's sampler launch instead of by a memset of their own in front
  // of it (one launch boundary off the head of the step)
  if (clear_ptr) {
    const int64_t gtid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, gsz = (int64_t)gridDim.x * blockDim.x;
    int4* c4 = reinterpret_cast<int4*>(clear_ptr);
    for (int64_t i = gtid; i < (clear_ints >> 2); i += gsz) c4[i] = int4{0, 0, 0, 0};
    for (int64_t i = (clear_ints & ~3ll) + gtid; i < clear_ints; i += gsz) clear_ptr[i] = 0;
  }
  // mark (optional, with next_nodes): mark[v] = 1 for every node written to the next level - the touched-node flags of the
  // step's compaction, set here instead of by a pass of their own over the level-0 list
  // out_cnt (optional): entries of the node's row strictly before the query time.  Under most-recent sampling slot j of the
  // query holds row entry cnt - K + j, so two queries on one node have neighbour lists that are SHIFTS of each other by the
  // difference of their counts (the layer-1 attention backward merges their key-side gradients by row entry)
  if (MODE == 2 && offset_dev) offset += *offset_dev;
  __shared__ float s_time[16][PFO_MAX_NEIGHBORS];   // uniform modes: per-group sort scratch
  const int lane = threadIdx.x & 63;
  const int sub = lane & 15;
  const int grp = lane >> 4;
  const int grp_in_block = threadIdx.x >> 4;
  const int64_t q = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
  const bool active = q < n_q;
  const int node = active ? q_nodes[q] : 0;
  const double t = active ? q_ts[q] : 0.0;
  int64_t row_lo = 0, row_hi = 0;
  if (active && node >= 0 && (int64_t)node < n_nodes) {
    row_lo = indptr[node];
    row_hi = indptr[node + 1];
  }
  // 16-ary lower_bound: first index with ts >= t lies in [lo, hi]  (np.searchsorted side='left', utils.py:158)
  int64_t lo = row_lo, hi = row_hi;
  while (true) {
    const int64_t n = hi - lo;
    const bool need = n > 0;
    if (!__any(need)) break;
    bool pred = false;
    if (need) {
      if (n <= 16) {
        if (sub < n) pred = adj_ts[lo + sub] < t;
      } else {
        pred = adj_ts[lo + ((int64_t)(sub + 1) * n) / 17] < t;
      }
    }
    const unsigned long long bal = __ballot(pred);
    const int c = __popc((unsigned)((bal >> (16 * grp)) & 0xFFFFull));
    if (need) {
      if (n <= 16) {
        lo += c;
        hi = lo;
      } else {
        const int64_t nlo = (c > 0) ? lo + ((int64_t)c * n) / 17 + 1 : lo;
        const int64_t nhi = (c < 16) ? lo + ((int64_t)(c + 1) * n) / 17 : hi;
        lo = nlo;
        hi = nhi;
      }
    }
  }
  const int64_t cnt = lo - row_lo;   // entries strictly before t
  if (out_cnt && active && sub == 0) out_cnt[q] = (int32_t)(cnt < 0x7fffffff ? cnt : 0x7fffffff);

  if (active && sub == 0 && next_nodes) {
    next_nodes[q] = node;
    if (next_ts) next_ts[q] = t;
    if (mark && node >= 0 && (int64_t)node < n_nodes) mark[node] = 1;
  }

  if (MODE == 0) {
    if (!active) return;
    for (int j = sub; j < K; j += 16) {
      const int64_t src = cnt - K + j;       // right-aligned tail (utils.py:208-218)
      int32_t v = 0, e = 0;
      float et = 0.f;
      if (src >= 0) {
        const int64_t p = row_lo + src;
        v = adj_nbr[p];
        e = adj_eidx[p];
        et = (float)adj_ts[p];               // utils.py:179-180 (f32 edge time)
      }
      const int64_t o = q * K + j;
      if (out_nbr) out_nbr[o] = v;
      if (out_eidx) out_eidx[o] = e;
      if (out_et) out_et[o] = et;
      if (out_dt) out_dt[o] = (float)(t - (double)et);   // embedding_module.py:133-135
      if (next_nodes) {
        next_nodes[n_q + o] = v;
        if (next_ts) next_ts[n_q + o] = t;
        if (mark && (uint64_t)(uint32_t)v < (uint64_t)n_nodes) mark[v] = 1;
      }
    }
  } else {
    // uniform with replacement (utils.py:194-204): gather, then stable sort by f32 time
    int32_t v[4], e[4];
    float et[4];
    float* sc = s_time[grp_in_block];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int j = sub + 16 * u;
      v[u] = 0; e[u] = 0; et[u] = 0.f;
      if (active && j < K && cnt > 0) {
        int64_t idx;
        if (MODE == 1) {
          idx = draws[q * K + j];
        } else {
          const pfo_u4 r = pfo_philox(seed, (uint64_t)q, offset + (uint64_t)(j >> 2));
          idx = (int64_t)(((uint64_t)pfo_u4_get(r, j & 3) * (uint64_t)cnt) >> 32);
        }
        idx = idx < 0 ? 0 : (idx >= cnt ? cnt - 1 : idx);
        const int64_t p = row_lo + idx;
        v[u] = adj_nbr[p];
        e[u] = adj_eidx[p];
        et[u] = (float)adj_ts[p];
      }
      if (j < K) sc[j] = et[u];
    }
    __syncthreads();
    if (active) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int j = sub + 16 * u;
        if (j < K) {
          int rank = 0;
          for (int i = 0; i < K; ++i) {
            const float ti = sc[i];
            rank += (ti < et[u]) || (ti == et[u] && i < j);
          }
          const int64_t o = q * K + rank;
          if (out_nbr) out_nbr[o] = v[u];
          if (out_eidx) out_eidx[o] = e[u];
          if (out_et) out_et[o] = et[u];
          if (out_dt) out_dt[o] = (float)(t - (double)et[u]);
          if (next_nodes) {
            next_nodes[n_q + o] = v[u];
            if (next_ts) next_ts[n_q + o] = t;
            if (mark && (uint64_t)(uint32_t)v[u] < (uint64_t)n_nodes) mark[v[u]] = 1;
          }
        }
      }
    }
  }
}

int pfo_tnbr_sample_dev(const int64_t* indptr, const int32_t* adj_nbr, const int32_t* adj_eidx, const double* adj_ts,
                        int64_t n_nodes, const int32_t* q_nodes, const double* q_ts, int64_t n_q, int32_t K, int32_t mode,
                        const int64_t* draws, uint64_t seed, uint64_t offset, const uint64_t* offset_dev, int32_t* out_nbr,
                        int32_t* out_eidx, float* out_et, float* out_dt, int32_t* next_nodes, double* next_ts, int32_t* mark,
                        int32_t* out_cnt, void* stream, int32_t* clear_ptr, int64_t clear_ints) {
  PFO_REQUIRE(mark == nullptr || next_nodes != nullptr, "mark needs next_nodes");
  PFO_REQUIRE(clear_ptr == nullptr || (mark == nullptr && (((uintptr_t)clear_ptr) & 15) == 0 && n_q > 0), "clear_ptr: 16-byte aligned, not with mark");
  PFO_REQUIRE(K >= 1 && K <= PFO_MAX_NEIGHBORS, "K must be in [1, 64]");
  PFO_REQUIRE(mode >= 0 && mode <= 2, "mode must be 0, 1 or 2");
  PFO_REQUIRE(mode != 1 || draws != nullptr, "mode 1 needs injected draws");
  PFO_REQUIRE(next_ts == nullptr || next_nodes != nullptr, "next_ts needs next_nodes");
  PFO_REQUIRE(n_q >= 0 && n_nodes > 0, "bad sizes");
  if (n_q == 0) return PFO_OK;
  PFO_REQUIRE(indptr && adj_nbr && adj_eidx && adj_ts && q_nodes && q_ts, "null input");
  const int threads = 256;
  const int64_t blocks = pfo_ceil_div(n_q * 16, threads);
  hipStream_t s = (hipStream_t)stream;
#define LAUNCH(M)                                                                                                   \
  PFO_KLAUNCH(tnbr_sample_kernel<M>, dim3((unsigned)blocks), dim3(threads), 0, s, indptr, adj_nbr, adj_eidx, \
                     adj_ts, n_nodes, q_nodes, q_ts, n_q, (int)K, draws, seed, offset, offset_dev, out_nbr, out_eidx,  \
                     out_et, out_dt, next_nodes, next_ts, mark, out_cnt, clear_ptr, clear_ints)
  pfo_prof_begin(s);
  if (mode == 0) LAUNCH(0);
  else if (mode == 1) LAUNCH(1);
  else LAUNCH(2);
#undef LAUNCH
  PFO_LAUNCH_CHECK();
  // per query K*(4+4+8) adjacency bytes read + K*12 written (+8 per frontier slot) + ~11 probes + row bounds
  pfo_prof_end(PFO_PROF_SAMPLER, (double)n_q * (K * 36.0 + 112.0) + (clear_ptr ? 4.0 * (double)clear_ints : 0.0), s);   // (+ the flags this launch clears)
  return PFO_OK;
}

extern "C" int pfo_tnbr_sample(const int64_t* indptr, const int32_t* adj_nbr, const int32_t* adj_eidx,
                               const double* adj_ts, int64_t n_nodes, const int32_t* q_nodes, const double* q_ts,
                               int64_t n_q, int32_t K, int32_t mode, const int64_t* draws, uint64_t seed,
                               uint64_t offset, int32_t* out_nbr, int32_t* out_eidx, float* out_et, float* out_dt,
                               int32_t* next_nodes, double* next_ts, void* stream) {
  return pfo_tnbr_sample_dev(indptr, adj_nbr, adj_eidx, adj_ts, n_nodes, q_nodes, q_ts, n_q, K, mode, draws, seed, offset,
                             nullptr, out_nbr, out_eidx, out_et, out_dt, next_nodes, next_ts, nullptr, nullptr, stream, nullptr, 0);
}

// ---------------------------------------------------------------------------------------------
// candidate draw: utils/utils.py:86-114.  One wavefront per interaction, available items in LDS.
__global__ __launch_bounds__(64) void neg_draw_kernel(const uint8_t* __restrict__ item_avail, int n_items,
                                                      const int32_t* __restrict__ port_idx,
                                                      const int32_t* __restrict__ port_len, int port_stride,
                                                      int size, int upper_u, uint64_t seed, uint64_t offset,
                                                      const uint64_t* __restrict__ offset_dev, int32_t* __restrict__ out) {
  extern __shared__ int32_t s_list[];
  if (offset_dev) offset += *offset_dev;
  const int lane = threadIdx.x;
  const int64_t b = blockIdx.x;
  const int plen = min(port_len[b], port_stride);
  const int32_t* port = port_idx + b * port_stride;
  int base = 0;
  for (int i0 = 0; i0 < n_items; i0 += 64) {
    const int i = i0 + lane;
    bool ok = i < n_items && item_avail[i] != 0;
    if (ok)
      for (int p = 0; p < plen; ++p) ok = ok && (port[p] != i);          // np.setdiff1d, utils.py:96
    const unsigned long long bal = __ballot(ok);
    const int pre = __popcll(bal & ((1ull << lane) - 1ull));
    if (ok) s_list[base + pre] = i;
    base += __popcll(bal);
  }
  __syncthreads();
  const int n_avail = base;
  int32_t* o = out + b * size;
  if (n_avail <= 0) {
    for (int k = lane; k < size; k += 64) o[k] = 0;
    return;
  }
  if (n_avail >= size) {
    // without replacement (utils.py:109-111): partial Fisher-Yates, serial on lane 0
    if (lane == 0) {
      for (int k = 0; k < size; ++k) {
        const pfo_u4 r = pfo_philox(seed, (uint64_t)b + offset, (uint64_t)(k >> 2));
        const uint32_t span = (uint32_t)(n_avail - k);
        const int j = k + (int)(((uint64_t)pfo_u4_get(r, k & 3) * span) >> 32);
        const int32_t a = s_list[k], c = s_list[j];
        s_list[k] = c;
        s_list[j] = a;
        o[k] = c + upper_u + 1;
      }
    }
  } else {
    // fewer available than requested: with replacement (utils.py:99-105)
    for (int k = lane; k < size; k += 64) {
      const pfo_u4 r = pfo_philox(seed, (uint64_t)b + offset, (uint64_t)(k >> 2));
      const int j = (int)(((uint64_t)pfo_u4_get(r, k & 3) * (uint32_t)n_avail) >> 32);
      o[k] = s_list[j] + upper_u + 1;
    }
  }
}

extern "C" int pfo_neg_draw(const uint8_t* item_avail, int32_t n_items, const int32_t* port_idx,
                            const int32_t* port_len, int32_t port_stride, int64_t B, int32_t size, int32_t upper_u,
                            uint64_t seed, uint64_t offset, int32_t* out, void* stream) {
  return pfo_neg_draw_dev(item_avail, n_items, port_idx, port_len, port_stride, B, size, upper_u, seed, offset, nullptr, out, stream);
}
extern "C" int pfo_neg_draw_dev(const uint8_t* item_avail, int32_t n_items, const int32_t* port_idx,
                                const int32_t* port_len, int32_t port_stride, int64_t B, int32_t size, int32_t upper_u,
                                uint64_t seed, uint64_t offset, const uint64_t* offset_dev, int32_t* out, void* stream) {
  PFO_REQUIRE(n_items > 0 && n_items <= 16384, "n_items must be in [1, 16384]");
  PFO_REQUIRE(size >= 1 && port_stride >= 0 && B >= 0, "bad sizes");
  if (B == 0) return PFO_OK;
  PFO_REQUIRE(item_avail && port_len && out && (port_idx || port_stride == 0), "null input");
  PFO_KLAUNCH(neg_draw_kernel, dim3((unsigned)B), dim3(64), (size_t)n_items * sizeof(int32_t),
                     (hipStream_t)stream, item_avail, (int)n_items, port_idx, port_len, (int)port_stride, (int)size,
                     (int)upper_u, seed, offset, offset_dev, out);
  PFO_LAUNCH_CHECK();
  return PFO_OK;
}

// ---------------------------------------------------------------------------------------------
// K2: main.py:209-304.  One wavefront per interaction, one lane per candidate, fp64 throughout.
// numpy's pairwise summation order for 8 <= n <= 128 (used by np.mean, main.py:243): eight
// interleaved partial sums, combined as a balanced tree, then the tail.
__device__ __forceinline__ double np_sum(const double* a, int n) {
  if (n < 8) {
    double r = 0.0;
    for (int i = 0; i < n; ++i) r += a[i];
    return r;
  }
  double r0 = a[0], r1 = a[1], r2 = a[2], r3 = a[3], r4 = a[4], r5 = a[5], r6 = a[6], r7 = a[7];
  int i = 8;
  for (; i < n - (n % 8); i += 8) {
    r0 += a[i]; r1 += a[i + 1]; r2 += a[i + 2]; r3 += a[i + 3];
    r4 += a[i + 4]; r5 += a[i + 5]; r6 += a[i + 6]; r7 += a[i + 7];
  }
  double res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
  for (; i < n; ++i) res += a[i];
  return res;
}

#define PFO_MV_MAX_RET 128

__global__ __launch_bounds__(64) void mv_select_kernel(const double* __restrict__ returns, int n_items, int n_ret,
                                                       const int32_t* __restrict__ day_idx,
                                                       const int32_t* __restrict__ cand, int n_cand,
                                                       const int32_t* __restrict__ port_idx,
                                                       const int32_t* __restrict__ port_len, int port_stride,
                                                       int upper_u, double gamma, double lam, int n_pos, int n_neg,
                                                       int32_t* __restrict__ p_pos, int32_t* __restrict__ p_neg,
                                                       double* __restrict__ y_out, double* __restrict__ rank_out) {
  const int lane = threadIdx.x;
  const int64_t b = blockIdx.x;
  const bool have = lane < n_cand;
  const int node = have ? cand[b * n_cand + lane] : 0;
  const int item = node - upper_u - 1;
  const double* day = returns + (int64_t)day_idx[b] * n_items * n_ret;
  const int plen = min(port_len[b], port_stride);
  double y = 0.0;
  if (have) {
    const double* ri = day + (int64_t)item * n_ret;
    const double mu = np_sum(ri, n_ret) / (double)n_ret;                  // main.py:243
    const double inv = 1.0 / (double)(n_ret - 1);                         // np.cov: c *= 1/(N - ddof)
    double var = 0.0;
    for (int t = 0; t < n_ret; ++t) var += (ri[t] - mu) * (ri[t] - mu);
    var *= inv;
    if (plen == 0) {
      y = (mu / gamma) / var;                                             // main.py:254
    } else {
      double ssum = 0.0;
      for (int p = 0; p < plen; ++p) {
        const double* rp = day + (int64_t)port_idx[b * port_stride + p] * n_ret;
        const double mp = np_sum(rp, n_ret) / (double)n_ret;
        double cv = 0.0;
        for (int t = 0; t < n_ret; ++t) cv += (ri[t] - mu) * (rp[t] - mp);
        ssum += cv * inv;                                                 // np.sum(sigma_ij), main.py:268
      }
      const double sum_sigma = (1.0 / (double)plen) * ssum;               // y_uj/n_holding * sum
      y = (mu / gamma - 0.5 * sum_sigma) / var;                           // main.py:271
    }
  }
  // invest_rank: scipy.stats.rankdata average ties (main.py:282); tgn_rank = n..1 (main.py:283)
  int less = 0, eq = 0;
  for (int j = 0; j < n_cand; ++j) {
    const double yj = __shfl(y, j, 64);
    less += (yj < y);
    eq += (yj == y);
  }
  const double invest = (double)less + ((double)eq + 1.0) * 0.5;
  const double tgn = (double)(n_cand - lane);
  const double nr = invest * lam + tgn * (1.0 - lam);                     // main.py:286
  // order = stable ascending argsort, reversed (main.py:289 + tie policy)
  int pos = 0;
  for (int j = 0; j < n_cand; ++j) {
    const double nj = __shfl(nr, j, 64);
    pos += (nj < nr) || (nj == nr && j < lane);
  }
  if (have) {
    const int rev = n_cand - 1 - pos;
    if (rev < n_pos) p_pos[b * n_pos + rev] = node;                       // main.py:291
    if (rev >= n_cand - n_neg) p_neg[b * n_neg + (rev - (n_cand - n_neg))] = node;   // main.py:292
    if (y_out) y_out[b * n_cand + lane] = y;
    if (rank_out) rank_out[b * n_cand + lane] = nr;
  }
}

extern "C" int pfo_mv_select(const double* returns, int32_t n_days, int32_t n_items, int32_t n_ret,
                             const int32_t* day_idx, const int32_t* cand, int32_t n_cand, const int32_t* port_idx,
                             const int32_t* port_len, int32_t port_stride, int64_t B, int32_t upper_u, double gamma,
                             double lambda_mv, int32_t n_pos, int32_t n_neg, int32_t* p_pos, int32_t* p_neg,
                             double* y_out, double* rank_out, void* stream) {
  PFO_REQUIRE(n_cand >= 1 && n_cand <= 64, "n_cand must be in [1, 64]");
  PFO_REQUIRE(n_ret >= 2 && n_ret <= PFO_MV_MAX_RET, "n_ret must be in [2, 128]");
  PFO_REQUIRE(n_pos >= 0 && n_neg >= 0 && n_pos <= n_cand && n_neg <= n_cand, "bad n_pos / n_neg");
  PFO_REQUIRE(n_days > 0 && n_items > 0 && B >= 0, "bad sizes");
  if (B == 0) return PFO_OK;
  PFO_REQUIRE(returns && day_idx && cand && port_len && p_pos && p_neg, "null input");
  PFO_KLAUNCH(mv_select_kernel, dim3((unsigned)B), dim3(64), 0, (hipStream_t)stream, returns, (int)n_items,
                     (int)n_ret, day_idx, cand, (int)n_cand, port_idx, port_len, (int)port_stride, (int)upper_u, gamma,
                     lambda_mv, (int)n_pos, (int)n_neg, p_pos, p_neg, y_out, rank_out);
  PFO_LAUNCH_CHECK();
  return PFO_OK;
}
