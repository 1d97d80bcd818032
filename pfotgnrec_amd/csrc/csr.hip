// Time-sorted adjacency (CSR) built on the device - the one-off work of get_neighbor_finder / NeighborFinder.__init__
// (utils/utils.py:117-148): every edge contributes (dst, eidx, ts) to its source's row and (src, eidx, ts) to its
// destination's row, rows sorted by timestamp with ties in edge order (Python's stable sorted()).
//
// = a STABLE sort of the 2E entries by (owner, ts).  Hand-written LSD radix sort, 8 bits per pass, keys never move: a pass
// permutes the entry indices and recomputes its digit from the edge arrays.  One wavefront per 1024-entry tile: the
// tile's 16 steps are walked in order and same-digit lanes are matched by ballots, so ranks inside a tile follow the
// input order without any cross-wavefront hand-off (that is what makes each pass stable).  Timestamps of an interaction
// log normally arrive sorted already: a check kernel then skips the eight timestamp passes and only the owner passes run.
//
// pfo_csr_append merges a batch of new edges into an existing CSR with the same result as a rebuild over [old ; new]:
// inside a row a new entry goes behind every old entry with ts <= its own (new edges are later in edge order).
#include "memory.hpp"
#include <algorithm>

#define RS_TILE 1024        // entries per wavefront tile (16 steps of 64)
#define RS_STEPS (RS_TILE / 64)

namespace {

struct Edges {
  const int32_t* src; const int32_t* dst; const int32_t* eidx; const double* ts; int64_t E;
};

__device__ __forceinline__ uint64_t sortable_f64(double x) {
  const uint64_t u = (uint64_t)__double_as_longlong(x);
  return (u >> 63) ? ~u : (u | 0x8000000000000000ull);        // ascending order of doubles == ascending order of keys
}
// digit of entry `idx` (edge idx >> 1, side idx & 1) for pass `pass`: passes 0..7 = timestamp bytes, 8.. = owner bytes
__device__ __forceinline__ int digit_of(const Edges& g, int32_t idx, int which, int shift) {
  const int64_t e = idx >> 1;
  if (which == 0) return (int)((sortable_f64(g.ts[e]) >> shift) & 255ull);
  const uint32_t owner = (uint32_t)((idx & 1) ? g.dst[e] : g.src[e]);
  return (int)((owner >> shift) & 255u);
}

__global__ void ts_sorted_kernel(const double* __restrict__ ts, int64_t E, int32_t* __restrict__ flag) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i + 1 < E; i += (int64_t)gridDim.x * blockDim.x)
    if (ts[i] > ts[i + 1]) *flag = 1;              // benign race: every writer stores 1
}
__global__ void iota_kernel(int32_t* __restrict__ p, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = (int32_t)i;
}

// per-tile digit histogram, digit-major: hist[d * n_tiles + tile]
__global__ __launch_bounds__(64) void rs_hist_kernel(const Edges g, const int32_t* __restrict__ perm, int64_t n, int which, int shift,
                                                     const int32_t* __restrict__ skip, int32_t* __restrict__ hist, int n_tiles) {
  if (skip && *skip == 0) return;
  __shared__ int s_cnt[256];
  const int lane = threadIdx.x;
  for (int d = lane; d < 256; d += 64) s_cnt[d] = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * RS_TILE;
  for (int st = 0; st < RS_STEPS; ++st) {
    const int64_t i = base + st * 64 + lane;
    if (i < n) atomicAdd(&s_cnt[digit_of(g, perm[i], which, shift)], 1);
  }
  __syncthreads();
  for (int d = lane; d < 256; d += 64) hist[(int64_t)d * n_tiles + blockIdx.x] = s_cnt[d];
}

// stable scatter of one tile: out[base[d][tile] + rank of the entry among the tile's earlier entries with digit d]
__global__ __launch_bounds__(64) void rs_scatter_kernel(const Edges g, const int32_t* __restrict__ perm, int64_t n, int which, int shift,
                                                        const int32_t* __restrict__ skip, const int32_t* __restrict__ base,
                                                        int n_tiles, int32_t* __restrict__ out) {
  if (skip && *skip == 0) return;
  __shared__ int s_off[256];
  const int lane = threadIdx.x;
  for (int d = lane; d < 256; d += 64) s_off[d] = base[(int64_t)d * n_tiles + blockIdx.x];
  __syncthreads();
  const int64_t t0 = (int64_t)blockIdx.x * RS_TILE;
  for (int st = 0; st < RS_STEPS; ++st) {
    const int64_t i = t0 + st * 64 + lane;
    const bool live = i < n;
    const int32_t idx = live ? perm[i] : 0;
    const int d = live ? digit_of(g, idx, which, shift) : 0;
    // lanes holding the same digit (match-any by eight ballots); rank = earlier lanes among them
    unsigned long long same = __ballot(live);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const unsigned long long bal = __ballot(live && ((d >> b) & 1));
      same &= ((d >> b) & 1) ? bal : ~bal;
    }
    const int rank = __popcll(same & ((1ull << lane) - 1ull));
    int off = 0;
    if (live) off = s_off[d];
    if (live) out[off + rank] = idx;
    __syncthreads();
    if (live && rank == 0) s_off[d] = off + __popcll(same);      // one lane per digit advances the running offset
    __syncthreads();
  }
}

__global__ void owner_count_kernel(const Edges g, int64_t n, int64_t n_nodes, int32_t* __restrict__ cnt) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t e = i >> 1;
    const int owner = (i & 1) ? g.dst[e] : g.src[e];
    if (owner >= 0 && owner < n_nodes) atomicAdd(&cnt[owner], 1);
  }
}
__global__ void csr_emit_kernel(const Edges g, const int32_t* __restrict__ perm, int64_t n, const int32_t* __restrict__ ptr32,
                                int64_t n_nodes, int64_t* __restrict__ indptr, int32_t* __restrict__ nbr, int32_t* __restrict__ eid,
                                double* __restrict__ ts) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x, gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (int64_t i = gid; i <= n_nodes; i += stride) indptr[i] = ptr32[i];
  for (int64_t i = gid; i < n; i += stride) {
    const int32_t idx = perm[i];
    const int64_t e = idx >> 1;
    nbr[i] = (idx & 1) ? g.src[e] : g.dst[e];
    eid[i] = g.eidx[e];
    ts[i] = g.ts[e];
  }
}

// merge of an old row set with a new one: every old entry moves to new_ptr[v] + p + #{new entries of row v with ts < its ts},
// every new entry to new_ptr[v] + q + #{old entries of row v with ts <= its ts}
__global__ void csr_merge_kernel(const int64_t* __restrict__ optr, const int32_t* __restrict__ onbr, const int32_t* __restrict__ oeid,
                                 const double* __restrict__ ots, const int64_t* __restrict__ aptr, const int32_t* __restrict__ anbr,
                                 const int32_t* __restrict__ aeid, const double* __restrict__ ats, int64_t n_old_nodes,
                                 int64_t n_nodes, int64_t* __restrict__ nptr, int32_t* __restrict__ nnbr, int32_t* __restrict__ neid,
                                 double* __restrict__ nts) {
  const int64_t v = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);      // one wavefront per row
  const int lane = threadIdx.x & 63;
  if (v > n_nodes) return;
  const int64_t olo = v < n_old_nodes ? optr[v] : optr[n_old_nodes], ohi = v < n_old_nodes ? optr[v + 1] : olo;
  const int64_t alo = v < n_nodes ? aptr[v] : aptr[n_nodes], ahi = v < n_nodes ? aptr[v + 1] : alo;
  const int64_t nlo = olo + alo;
  if (lane == 0) nptr[v] = nlo;
  if (v == n_nodes) return;
  for (int64_t p = lane; p < ohi - olo; p += 64) {
    const double t = ots[olo + p];
    int64_t lo = alo, hi = ahi;                       // first new entry with ts >= t
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (ats[mid] < t) lo = mid + 1; else hi = mid; }
    const int64_t o = nlo + p + (lo - alo);
    nnbr[o] = onbr[olo + p]; neid[o] = oeid[olo + p]; nts[o] = t;
  }
  for (int64_t q = lane; q < ahi - alo; q += 64) {
    const double t = ats[alo + q];
    int64_t lo = olo, hi = ohi;                       // first old entry with ts > t
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (ots[mid] <= t) lo = mid + 1; else hi = mid; }
    const int64_t o = nlo + q + (lo - olo);
    nnbr[o] = anbr[alo + q]; neid[o] = aeid[alo + q]; nts[o] = t;
  }
}

int64_t tiles_of(int64_t n) { return pfo_ceil_div(n, RS_TILE); }

}  // namespace

extern "C" int64_t pfo_csr_build_workspace_bytes(int64_t E, int64_t n_nodes) {
  if (E < 0 || n_nodes <= 0) return -1;
  const int64_t n = 2 * E, nt = std::max<int64_t>(1, tiles_of(n));
  const int64_t hist = 256 * nt;
  int64_t b = 0;
  b += 2 * pfo_align_up(n * 4, 256);                                   // two permutation buffers
  b += 2 * pfo_align_up(hist * 4, 256);                                // tile histograms and their scan
  b += pfo_align_up((pfo_ceil_div(std::max(hist, n_nodes + 1), 1024) + 16) * 4, 256);   // scan scratch
  b += 2 * pfo_align_up((n_nodes + 1) * 4, 256);                       // per-node counts, 32-bit row offsets
  b += 256;                                                            // flags
  return b;
}

extern "C" int pfo_csr_build(const int32_t* src, const int32_t* dst, const int32_t* eidx, const double* ts, int64_t E,
                             int64_t n_nodes, int64_t* indptr, int32_t* adj_nbr, int32_t* adj_eidx, double* adj_ts,
                             void* workspace, int64_t workspace_bytes, void* stream) {
  PFO_REQUIRE(E >= 0 && n_nodes > 0 && 2 * E < ((int64_t)1 << 31), "bad sizes (2E must fit in int32)");
  PFO_REQUIRE(indptr && workspace && workspace_bytes >= pfo_csr_build_workspace_bytes(E, n_nodes), "null / short workspace");
  hipStream_t s = (hipStream_t)stream;
  const int64_t n = 2 * E, nt = std::max<int64_t>(1, tiles_of(n)), hist_n = 256 * nt;
  char* p = reinterpret_cast<char*>(workspace);
  auto take = [&](int64_t bytes) { char* r = p; p += pfo_align_up(bytes, 256); return r; };
  int32_t* perm[2] = {(int32_t*)take(n * 4), (int32_t*)take(n * 4)};
  int32_t* hist = (int32_t*)take(hist_n * 4);
  int32_t* base = (int32_t*)take(hist_n * 4);
  int32_t* scratch = (int32_t*)take((pfo_ceil_div(std::max(hist_n, n_nodes + 1), 1024) + 16) * 4);
  int32_t* cnt = (int32_t*)take((n_nodes + 1) * 4);
  int32_t* ptr32 = (int32_t*)take((n_nodes + 1) * 4);
  int32_t* unsorted = (int32_t*)take(256);
  PFO_REQUIRE(hipMemsetAsync(cnt, 0, (size_t)(n_nodes + 1) * 4, s) == hipSuccess, "memset failed");
  PFO_REQUIRE(hipMemsetAsync(unsorted, 0, 4, s) == hipSuccess, "memset failed");
  if (E == 0) {
    PFO_REQUIRE(hipMemsetAsync(indptr, 0, (size_t)(n_nodes + 1) * 8, s) == hipSuccess, "memset failed");
    return PFO_OK;
  }
  PFO_REQUIRE(src && dst && eidx && ts && adj_nbr && adj_eidx && adj_ts, "null input");
  Edges g{src, dst, eidx, ts, E};
  const int blk = (int)std::min<int64_t>(4096, pfo_ceil_div(n, 256));
  PFO_KLAUNCH(ts_sorted_kernel, dim3(blk), dim3(256), 0, s, ts, E, unsorted);
  PFO_KLAUNCH(iota_kernel, dim3(blk), dim3(256), 0, s, perm[0], n);
  int cur = 0;
  auto pass = [&](int which, int shift, const int32_t* skip) -> int {
    PFO_KLAUNCH(rs_hist_kernel, dim3((unsigned)nt), dim3(64), 0, s, g, perm[cur], n, which, shift, skip, hist, (int)nt);
    if (int rc = pfo_iscan_launch(hist, hist_n, base, scratch, s)) return rc;
    PFO_KLAUNCH(rs_scatter_kernel, dim3((unsigned)nt), dim3(64), 0, s, g, perm[cur], n, which, shift, skip, base, (int)nt, perm[cur ^ 1]);
    cur ^= 1;
    return PFO_OK;
  };
  // timestamp passes: skipped on the device when the log is already chronological.  A skipped pass must leave the
  // permutation where it is, so the eight passes are an even number of buffer swaps and copy nothing when skipped
  // (perm[cur] stays valid because both kernels return before touching anything).
  for (int b = 0; b < 8; ++b)
    if (int rc = pass(0, 8 * b, unsorted)) return rc;
  // when skipped, the data still sits in perm[0] and cur == 0 again after eight swaps: consistent either way
  int owner_bits = 1;
  while (((int64_t)1 << owner_bits) < n_nodes) ++owner_bits;
  for (int b = 0; b * 8 < owner_bits; ++b)
    if (int rc = pass(1, 8 * b, nullptr)) return rc;
  PFO_KLAUNCH(owner_count_kernel, dim3(blk), dim3(256), 0, s, g, n, n_nodes, cnt);
  if (int rc = pfo_iscan_launch(cnt, n_nodes + 1, ptr32, scratch, s)) return rc;
  PFO_KLAUNCH(csr_emit_kernel, dim3(blk), dim3(256), 0, s, g, perm[cur], n, ptr32, n_nodes, indptr, adj_nbr, adj_eidx, adj_ts);
  PFO_LAUNCH_CHECK();
  return PFO_OK;
}

extern "C" int pfo_csr_append(const int64_t* old_indptr, const int32_t* old_nbr, const int32_t* old_eidx, const double* old_ts,
                              int64_t n_old_nodes, const int64_t* add_indptr, const int32_t* add_nbr, const int32_t* add_eidx,
                              const double* add_ts, int64_t n_nodes, int64_t* new_indptr, int32_t* new_nbr, int32_t* new_eidx,
                              double* new_ts, void* stream) {
  PFO_REQUIRE(old_indptr && add_indptr && new_indptr && n_nodes >= n_old_nodes && n_old_nodes > 0, "bad arguments");
  PFO_KLAUNCH(csr_merge_kernel, dim3((unsigned)pfo_ceil_div(n_nodes + 1, 4)), dim3(256), 0, (hipStream_t)stream, old_indptr,
                     old_nbr, old_eidx, old_ts, add_indptr, add_nbr, add_eidx, add_ts, n_old_nodes, n_nodes, new_indptr, new_nbr,
                     new_eidx, new_ts);
  PFO_LAUNCH_CHECK();
  return PFO_OK;
}
