// Memory-side kernels: which nodes does this step read (touched-set compaction), the GRU gate
// math around the two MFMA contractions, and the persist / raw-message store that closes a step.
//
// Layout decision: the reference's defaultdict of per-node message lists becomes
// a dense table msg_table[n_nodes, 3D+Ef] + msg_time[n_nodes] + has_msg[n_nodes]; "last" aggregation
// (message_aggregator.py:38-55) is a last-index-wins scatter.  The lazy update of tgn.py:251 runs the
// GRU over EVERY node with a pending message (P -> n_nodes); only rows that this step actually reads
// can influence its outputs, so the GRU is applied to the compacted set of touched nodes instead.
#include "memory.hpp"
#include <algorithm>

#define SCAN_BLOCK 1024

// mark[v] = value for every listed node; keep_set: nodes already marked (by the sampler: the step's own references) stay as they are
__global__ void touch_mark_kernel(const int32_t* __restrict__ nodes0, int64_t n0, int n_nodes, int32_t* __restrict__ mark, int value,
                                  int keep_set) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n0; i += (int64_t)gridDim.x * blockDim.x) {
    const int v = nodes0[i];
    if (v >= 0 && v < n_nodes && !(keep_set && mark[v] != 0)) mark[v] = value;
  }
}

__global__ __launch_bounds__(1024) void scan_blocks_kernel(int32_t* __restrict__ block_counts, int n_blocks,
                                                           int32_t* __restrict__ n_touched) {
  // exclusive scan of block_counts in place, by one workgroup, 1024 entries per sweep
  __shared__ int s_w[16];
  __shared__ int s_carry;
  if (threadIdx.x == 0) s_carry = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int base = 0; base < n_blocks; base += 1024) {
    const int i = base + threadIdx.x;
    const int v = i < n_blocks ? block_counts[i] : 0;
    int incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int u = __shfl_up(incl, o, 64);
      if (lane >= o) incl += u;
    }
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < wave; ++w) woff += s_w[w];
    const int carry = s_carry;
    if (i < n_blocks) block_counts[i] = carry + woff + incl - v;
    __syncthreads();
    if (threadIdx.x == 1023) s_carry = carry + woff + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) *n_touched = s_carry;
}

// ---------------------------------------------------------------------------------------------
// The compaction in ONE launch (round 2: count / scan of the counts / scatter = three).  Block b counts its flags, publishes the count (+1: zero = not yet there) in flags[b], then sums the published counts of all blocks
// before it - no chain: every block only waits for earlier blocks to PUBLISH, which depends on nothing.  Earlier blocks are
// dispatched first, so they make progress whatever the occupancy (the forward-progress assumption of every look-back scan).
// flags[] must be zero when the kernel starts: they sit behind slot[] in the workspace and are cleared by the same memset.
// (Block b reads b flags: quadratic in n_nodes / 1024, 0.1 M reads at C4's 500 k nodes, ~50 M at 10 M nodes.)
__global__ __launch_bounds__(SCAN_BLOCK) void compact_onepass_kernel(const int32_t* __restrict__ mark, int match, int32_t* __restrict__ slot,
                                                                     int n_nodes, int32_t* flags, int32_t* __restrict__ touched_ids,
                                                                     const int32_t* __restrict__ base_dev, int32_t* __restrict__ n_out,
                                                                     int32_t* __restrict__ n_out2, int write_unmatched) {
  __shared__ int s_cnt[SCAN_BLOCK / 64];
  __shared__ int s_off[SCAN_BLOCK / 64];
  const int b = blockIdx.x;
  const int v = b * SCAN_BLOCK + threadIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int mk = v < n_nodes ? mark[v] : 0;
  const bool f = mk == match;
  const unsigned long long bal = __ballot(f);
  if (lane == 0) s_cnt[wave] = __popcll(bal);
  __syncthreads();
  int cnt = 0, woff = 0;
  for (int w = 0; w < SCAN_BLOCK / 64; ++w) { if (w == wave) woff = cnt; cnt += s_cnt[w]; }
  if (threadIdx.x == 0) __hip_atomic_store(&flags[b], cnt + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  // counts of the blocks before this one
  int part = 0;
  for (int t = threadIdx.x; t < b; t += SCAN_BLOCK) {
    int c;
    do { c = __hip_atomic_load(&flags[t], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT); } while (c == 0);
    part += c - 1;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
  if (lane == 0) s_off[wave] = part;
  __syncthreads();
  int off = base_dev ? *base_dev : 0;                        // second class: slots continue behind the first one's
  for (int w = 0; w < SCAN_BLOCK / 64; ++w) off += s_off[w];
  const int pos = off + woff + __popcll(bal & ((1ull << lane) - 1ull));
  if (v < n_nodes) {
    if (f) { slot[v] = pos; touched_ids[pos] = v; }
    else if (write_unmatched) slot[v] = -1;
  }
  if (b == (int)gridDim.x - 1 && threadIdx.x == 0) {
    *n_out = off + cnt;
    if (n_out2) *n_out2 = off + cnt;
  }
}

int64_t pfo_compact_scratch_ints(int n_nodes) { return 2 * pfo_ceil_div(n_nodes, SCAN_BLOCK) + 8; }

// Two CLASSES of touched nodes: those the step's levels reference (flag 1, set by the sampler or the marking pass here) get
// slots [0, n_core); nodes only the caller's `extra` list names (flag 2: a data-parallel rank's global positives, whose lazily
// updated memory the state update needs) get [n_core, n_touched).  Everything that is computed FOR the layers - the
// touched-table projection, the per-row gradient sums, the GRU backward - then stops at n_core (n_counts[1]); the GRU forward
// covers all n_touched (n_counts[0]) rows.  Without extras one pass, n_core == n_touched.
int pfo_touch_compact_launch(const int32_t* nodes0, int64_t n0, const int32_t* extra, int64_t n_extra, int n_nodes, int32_t* mark,
                             int32_t* slot, int32_t* touched_ids, int32_t* n_counts, int32_t* scratch, bool marks_are_zero,
                             bool marked, hipStream_t stream) {
  PFO_REQUIRE(mark && slot && touched_ids && n_counts && scratch && n_nodes > 0, "bad arguments");
  PFO_REQUIRE(marked || (nodes0 && n0 > 0), "no node list to mark");
  const int nb = (int)pfo_ceil_div(n_nodes, SCAN_BLOCK);
  if (!marks_are_zero) {
    PFO_REQUIRE(!marked, "marked flags need a cleared table");
    hipError_t e = hipMemsetAsync(mark, 0, (size_t)n_nodes * sizeof(int32_t), stream);
    PFO_REQUIRE(e == hipSuccess, "memset failed");
    e = hipMemsetAsync(scratch, 0, (size_t)2 * nb * sizeof(int32_t), stream);
    PFO_REQUIRE(e == hipSuccess, "memset failed");
  }
  if (!marked) {
    const int mb = (int)std::min<int64_t>(2048, pfo_ceil_div(n0, 256));
    PFO_KLAUNCH(touch_mark_kernel, dim3(mb), dim3(256), 0, stream, nodes0, n0, n_nodes, mark, 1, 0);
  }
  const bool two = extra && n_extra > 0;
  if (two) {
    const int eb = (int)std::min<int64_t>(2048, pfo_ceil_div(n_extra, 256));
    PFO_KLAUNCH(touch_mark_kernel, dim3(eb), dim3(256), 0, stream, extra, n_extra, n_nodes, mark, 2, 1);
  }
  // class 1 -> n_counts[1] (and n_counts[0] when it is the only class); class 2 continues behind it -> n_counts[0]
  PFO_KLAUNCH(compact_onepass_kernel, dim3(nb), dim3(SCAN_BLOCK), 0, stream, mark, 1, slot, n_nodes, scratch, touched_ids,
                     (const int32_t*)nullptr, n_counts + 1, two ? (int32_t*)nullptr : n_counts, 1);
  if (two)
    PFO_KLAUNCH(compact_onepass_kernel, dim3(nb), dim3(SCAN_BLOCK), 0, stream, mark, 2, slot, n_nodes, scratch + nb, touched_ids,
                       (const int32_t*)(n_counts + 1), n_counts, (int32_t*)nullptr, 0);
  PFO_LAUNCH_CHECK();
  return PFO_OK;
}

__global__ void remap_kernel(const int32_t* __restrict__ nodes0, int64_t n0, const int32_t* __restrict__ slot,
                             int32_t* __restrict__ idx0) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n0; i += (int64_t)gridDim.x * blockDim.x)
    idx0[i] = slot[nodes0[i]];
}

int pfo_remap_launch(const int32_t* nodes0, int64_t n0, const int32_t* slot, int32_t* idx0, hipStream_t stream) {
  const int mb = (int)std::min<int64_t>(2048, pfo_ceil_div(n0, 256));
  PFO_KLAUNCH(remap_kernel, dim3(mb), dim3(256), 0, stream, nodes0, n0, slot, idx0);
  PFO_LAUNCH_CHECK();
  return PFO_OK;
}

// rows of the touched nodes from a full table: dst[s] = src[touched_ids[s]]   (no-memory models: level 0 = node features)
__global__ void gather_rows_kernel(const float* __restrict__ src, int D, const int32_t* __restrict__ touched_ids,
                                   const int32_t* __restrict__ n_touched, float* __restrict__ dst) {
  const int lane = threadIdx.x & 63;
  const int nt = *n_touched;
  for (int s = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; s < nt; s += (gridDim.x * blockDim.x) >> 6) {
    const float4* a = reinterpret_cast<const float4*>(src + (int64_t)touched_ids[s] * D);
    float4* d = reinterpret_cast<float4*>(dst + (int64_t)s * D);
    for (int c = lane; c < D / 4; c += 64) d[c] = a[c];
  }
}
int pfo_gather_rows_launch(const float* src, int D, const int32_t* touched_ids, const int32_t* n_touched, int cap, float* dst,
                           hipStream_t stream) {
  PFO_REQUIRE((D % 4) == 0, "row length must be a multiple of 4");
  const int nb = (int)std::min<int64_t>(4096, std::max<int64_t>(1, pfo_ceil_div(cap, 4)));
  PFO_KLAUNCH(gather_rows_kernel, dim3(nb), dim3(256), 0, stream, src, D, touched_ids, n_touched, dst);
  PFO_LAUNCH_CHECK();
  return PFO_OK;
}

// ---------------------------------------------------------------------------------------------
// Instances grouped by the touched-table row they read (layer 1: instance n reads row idx[n]).  The backward sums
// per-instance gradients over each group BEFORE the contractions that only depend on the row, so those contractions run
// over the ~10 k touched rows instead of the ~54 k instances.  Counting sort; the order inside a group is fixed
// (ascending instance index) by a rank sort, so every sum over a group is reproducible.
__global__ void seg_count_kernel(const int32_t* __restrict__ idx, const int32_t* __restrict__ nodes, int N,
                                 int32_t* __restrict__ cnt) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n < N && nodes[n] != 0) atomicAdd(&cnt[idx[n]], 1);       // padding instances (node 0) carry no gradient
}
// exclusive scan, level 1: per block of SCAN_BLOCK entries; block totals to `block_sum`
__global__ __launch_bounds__(SCAN_BLOCK) void iscan_local_kernel(const int32_t* __restrict__ in, int n, int32_t* __restrict__ out,
                                                                 int32_t* __restrict__ block_sum) {
  __shared__ int s_w[SCAN_BLOCK / 64];
  const int i = blockIdx.x * SCAN_BLOCK + threadIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int v = i < n ? in[i] : 0;
  int incl = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int u = __shfl_up(incl, o, 64);
    if (lane >= o) incl += u;
  }
  if (lane == 63) s_w[wave] = incl;
  __syncthreads();
  int woff = 0;
  for (int w = 0; w < wave; ++w) woff += s_w[w];
  if (i < n) out[i] = woff + incl - v;
  if (threadIdx.x == SCAN_BLOCK - 1) block_sum[blockIdx.x] = woff + incl;
}
__global__ __launch_bounds__(SCAN_BLOCK) void iscan_add_kernel(int32_t* __restrict__ out, int n, const int32_t* __restrict__ block_off,
                                                               int32_t* __restrict__ copy) {
  const int i = blockIdx.x * SCAN_BLOCK + threadIdx.x;
  if (i < n) {
    const int v = out[i] + block_off[blockIdx.x];
    out[i] = v;
    copy[i] = v;
  }
}
__global__ void seg_place_kernel(const int32_t* __restrict__ idx, const int32_t* __restrict__ nodes, int N,
                                 int32_t* __restrict__ cursor, int32_t* __restrict__ members) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n < N && nodes[n] != 0) members[atomicAdd(&cursor[idx[n]], 1)] = n;
}
// one wavefront per group: members ordered by (key, instance index).  key = key_src[instance]: under most-recent sampling the
// number of row entries before the instance's time (sampler.hip out_cnt) - equal keys <=> identical neighbour lists, and a
// group in this order walks its node's history forwards.  key_src == null (uniform sampling: lists are random): the instance
// index itself.  Instance indices are distinct, so ranks are too; the order is reproducible.
__global__ void seg_sort_kernel(const int32_t* __restrict__ seg_ptr, int n_seg, const int32_t* __restrict__ in,
                                const int32_t* __restrict__ key_src, int32_t* __restrict__ out, int32_t* __restrict__ seg_of) {
  const int lane = threadIdx.x & 63;
  for (int s = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; s < n_seg; s += (gridDim.x * blockDim.x) >> 6) {
    const int lo = seg_ptr[s], cnt = seg_ptr[s + 1] - lo;
    if (cnt <= 0) continue;
    for (int i = lane; i < cnt; i += 64) {
      const int x = in[lo + i];
      const int kx = key_src ? key_src[x] : x;
      int rank = 0;
      for (int j = 0; j < cnt; ++j) {
        const int y = in[lo + j];
        const int ky = key_src ? key_src[y] : y;
        rank += (ky < kx) || (ky == kx && y < x);
      }
      out[lo + rank] = x;
      if (seg_of) seg_of[lo + i] = s;                       // (optional: the group of every member position)
    }
  }
}
int pfo_iscan_launch(const int32_t* in, int64_t n, int32_t* out, int32_t* scratch, hipStream_t stream) {
  PFO_REQUIRE(in && out && scratch && n > 0 && n < ((int64_t)1 << 31), "bad arguments");
  const int nb = (int)pfo_ceil_div(n, SCAN_BLOCK);
  PFO_KLAUNCH(iscan_local_kernel, dim3(nb), dim3(SCAN_BLOCK), 0, stream, in, (int)n, out, scratch);
  PFO_KLAUNCH(scan_blocks_kernel, dim3(1), dim3(1024), 0, stream, scratch, nb, scratch + nb);
  PFO_KLAUNCH(iscan_add_kernel, dim3(nb), dim3(SCAN_BLOCK), 0, stream, out, (int)n, scratch, out);
  PFO_LAUNCH_CHECK();
  return PFO_OK;
}

int64_t pfo_seg_scratch_ints(int cap_rows) { return pfo_ceil_div(cap_rows + 1, SCAN_BLOCK) + 8; }
int pfo_seg_build_launch(const int32_t* idx, const int32_t* nodes, int N, int cap_rows, const int32_t* key_src,
                         int32_t* seg_ptr, int32_t* cursor, int32_t* tmp, int32_t* members, int32_t* seg_of, int32_t* scratch,
                         hipStream_t stream) {
  PFO_REQUIRE(idx && nodes && seg_ptr && cursor && tmp && members && scratch && N > 0 && cap_rows > 0, "bad arguments");
  const int n = cap_rows + 1;
  PFO_REQUIRE(hipMemsetAsync(cursor, 0, (size_t)n * sizeof(int32_t), stream) == hipSuccess, "memset failed");
  PFO_KLAUNCH(seg_count_kernel, dim3((unsigned)pfo_ceil_div(N, 256)), dim3(256), 0, stream, idx, nodes, N, cursor);
  const int nb = (int)pfo_ceil_div(n, SCAN_BLOCK);
  PFO_KLAUNCH(iscan_local_kernel, dim3(nb), dim3(SCAN_BLOCK), 0, stream, cursor, n, seg_ptr, scratch);
  PFO_KLAUNCH(scan_blocks_kernel, dim3(1), dim3(1024), 0, stream, scratch, nb, scratch + nb);
  PFO_KLAUNCH(iscan_add_kernel, dim3(nb), dim3(SCAN_BLOCK), 0, stream, seg_ptr, n, scratch, cursor);
  PFO_KLAUNCH(seg_place_kernel, dim3((unsigned)pfo_ceil_div(N, 256)), dim3(256), 0, stream, idx, nodes, N, cursor, tmp);
  PFO_KLAUNCH(seg_sort_kernel, dim3((unsigned)std::min<int64_t>(4096, pfo_ceil_div(cap_rows, 4))), dim3(256), 0, stream,
                     seg_ptr, cap_rows, tmp, key_src, members, seg_of);
  PFO_LAUNCH_CHECK();
  return PFO_OK;
}

// out[s][0:W0) = sum over the group's members n of src0[n][0:W0), out[s][W0:W0+W1) = ... of src1[n][0:W1), s < *n_rows;
// groups without members get zeros.  One wavefront per group, fixed member order -> reproducible.
#define SEGSUM_R 16
// src0_live (optional, with src0_by_position): byte flags per position - a row whose flag is 0 holds nothing (its producer
// folded it into a later row of the same group, attn.hip) and is not read
__global__ __launch_bounds__(256) void segsum_kernel(const float* __restrict__ src0, int W0, const float* __restrict__ src1, int W1,
                                                     const int32_t* __restrict__ seg_ptr, const int32_t* __restrict__ members,
                                                     const int32_t* __restrict__ n_rows, int src0_by_position,
                                                     const uint8_t* __restrict__ src0_live, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int W = W0 + W1;
  const int nr = *n_rows;
  for (int s = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; s < nr; s += (gridDim.x * blockDim.x) >> 6) {
    const int lo = seg_ptr[s], hi = seg_ptr[s + 1];
    for (int c0 = 0; c0 < W; c0 += 64 * SEGSUM_R) {
      float acc[SEGSUM_R];
#pragma unroll
      for (int r = 0; r < SEGSUM_R; ++r) acc[r] = 0.f;
      int m = lo;
      for (; m + 1 < hi; m += 2) {                       // two member rows in flight
        const int64_t na = members[m], nb = members[m + 1];
        const int64_t pa = src0_by_position ? m : na, pb = src0_by_position ? m + 1 : nb;
        const bool la = !src0_live || src0_live[m] != 0, lb = !src0_live || src0_live[m + 1] != 0;   // wave-uniform
        float va[SEGSUM_R], vb[SEGSUM_R];
#pragma unroll
        for (int r = 0; r < SEGSUM_R; ++r) {
          const int c = c0 + lane + 64 * r;
          va[r] = c < W0 ? (la ? src0[pa * W0 + c] : 0.f) : (c < W ? src1[na * W1 + (c - W0)] : 0.f);
          vb[r] = c < W0 ? (lb ? src0[pb * W0 + c] : 0.f) : (c < W ? src1[nb * W1 + (c - W0)] : 0.f);
        }
#pragma unroll
        for (int r = 0; r < SEGSUM_R; ++r) acc[r] = (acc[r] + va[r]) + vb[r];
      }
      if (m < hi) {
        const int64_t na = members[m];
        const int64_t pa = src0_by_position ? m : na;
        const bool la = !src0_live || src0_live[m] != 0;
#pragma unroll
        for (int r = 0; r < SEGSUM_R; ++r) {
          const int c = c0 + lane + 64 * r;
          acc[r] += c < W0 ? (la ? src0[pa * W0 + c] : 0.f) : (c < W ? src1[na * W1 + (c - W0)] : 0.f);
        }
      }
#pragma unroll
      for (int r = 0; r < SEGSUM_R; ++r) {
        const int c = c0 + lane + 64 * r;
        if (c < W) out[(int64_t)s * W + c] = acc[r];
      }
    }
  }
}
// The same sums with 16-byte loads, organised around the MEMBER LIST instead of the segment (every row start a multiple of
// 4 floats).  A workgroup owns SEGSUM_NSEG consecutive segments = one contiguous range of the member list; each of its
// waves owns 64 float4 columns of the row (4 waves = 256 columns = the whole 876-float row of C2 in one pass) and walks the
// range in member order, SEGSUM_ROWS row loads in flight, storing its slice of an output row whenever a segment ends.
//   - the ids and flags of up to 64 members arrive in ONE vector load (lane i holds member i) and are broadcast with
//     v_readlane: a segment costs three dependent memory latencies (pointers, ids, rows) whatever its length up to
//     SEGSUM_ROWS, where a wave that owned a whole segment paid three per four members - the kernel was latency-bound
//     (C2, alone: 82-95 us for ~50 MB read + 47 MB written; tools/probes/segsum_bench.py)
//   - every load is a full 1 KB wave access, no partial rows; no LDS, no barrier
//   - rows are added strictly in member order: the sum is the sequential one, the same on every run.
#define SEGSUM_NSEG 4
#define SEGSUM_ROWS 8
__global__ __launch_bounds__(256) void segsum_vec_kernel(const float* __restrict__ src0, int W0, const float* __restrict__ src1, int W1,
                                                         const int32_t* __restrict__ seg_ptr, const int32_t* __restrict__ members,
                                                         const int32_t* __restrict__ n_rows, int src0_by_position,
                                                         const uint8_t* __restrict__ src0_live, float* __restrict__ out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int V0 = W0 >> 2, V = (W0 + W1) >> 2;
  const int nr = *n_rows;
  const float4 zero = {0.f, 0.f, 0.f, 0.f};
  for (int sb = blockIdx.x * SEGSUM_NSEG; sb < nr; sb += gridDim.x * SEGSUM_NSEG) {            // (workgroup-uniform)
    const int ns = min(SEGSUM_NSEG, nr - sb);
    const int my_ptr = lane <= ns ? seg_ptr[sb + lane] : 0;
    const int m_lo = __builtin_amdgcn_readlane(my_ptr, 0), m_hi = __builtin_amdgcn_readlane(my_ptr, ns);
    for (int v0 = 0; v0 < V; v0 += 256) {
      const int c = v0 + wave * 64 + lane;
      const bool in0 = c < V0, in1 = c >= V0 && c < V;
      float4* const orow = reinterpret_cast<float4*>(out + (int64_t)sb * (W0 + W1)) + c;
      const int64_t ostep = (W0 + W1) >> 2;
      int seg = 0;
      int seg_end = __builtin_amdgcn_readlane(my_ptr, 1);
      float4 acc = zero;
      for (int mb = m_lo; mb < m_hi; mb += 64) {
        const int nb = min(64, m_hi - mb);
        const int mid = lane < nb ? members[mb + lane] : 0;
        const int lv = (lane < nb && src0_live) ? (int)src0_live[mb + lane] : 1;
        for (int j0 = 0; j0 < nb; j0 += SEGSUM_ROWS) {
          // (ids and flags to scalars FIRST: a v_readlane between the row loads makes the compiler drain the load queue
          //  in front of it, one row in flight)
          int rn[SEGSUM_ROWS], rl[SEGSUM_ROWS];
#pragma unroll
          for (int j = 0; j < SEGSUM_ROWS; ++j) {
            const int jj = min(j0 + j, nb - 1);                              // (wave-uniform; rows past the end: the last one again, not added)
            rn[j] = __builtin_amdgcn_readlane(mid, jj);
            rl[j] = __builtin_amdgcn_readlane(lv, jj);
          }
          float4 v[SEGSUM_ROWS];
#pragma unroll
          for (int j = 0; j < SEGSUM_ROWS; ++j) {
            const int64_t pos = src0_by_position ? (int64_t)(mb + min(j0 + j, nb - 1)) : (int64_t)rn[j];
            const float4* p = in0 ? reinterpret_cast<const float4*>(src0 + pos * W0) + c
                                  : reinterpret_cast<const float4*>(src1 + (int64_t)rn[j] * W1) + (in1 ? c - V0 : 0);
            v[j] = (in0 ? rl[j] != 0 : in1) ? *p : zero;
          }
#pragma unroll
          for (int j = 0; j < SEGSUM_ROWS; ++j) {
            const int m = mb + j0 + j;
            if (m < m_hi) {
              while (m >= seg_end) {                                         // a segment ended in front of this member
                if (c < V) orow[seg * ostep] = acc;
                acc = zero;
                ++seg;
                seg_end = __builtin_amdgcn_readlane(my_ptr, seg + 1);
              }
              acc.x += v[j].x; acc.y += v[j].y; acc.z += v[j].z; acc.w += v[j].w;
            }
          }
        }
      }
      for (; seg < ns; ++seg) {                                              // the last segment (and empty ones behind it)
        if (c < V) orow[seg * ostep] = acc;
        acc = zero;
      }
    }
  }
}
// The same walk with the work cut by MEMBERS, not by segments (seg_of given: the segment of every member position, written
// by pfo_seg_build_launch).  A workgroup takes SEGSUM_CHUNK consecutive member positions and OWNS the segments that start
// inside them - it sums those to their ends, wherever that is - so every workgroup has about the same number of rows to
// read whatever the shape of the segments, no row is summed by two workgroups, and there is no atomic and no second pass.
// (with the segment-cut kernel above the 500 item segments of C2, 15-41 members each and all at the end of the table, were
//  the last 125 workgroups of the launch and ran after everything else: 64 us alone, 30 of them that tail.)
// Empty segments start nowhere: their rows are cleared by a sweep over the pointer array in front of the walk.
#define SEGSUM_CHUNK 16
__global__ __launch_bounds__(256) void segsum_chunk_kernel(const float* __restrict__ src0, int W0, const float* __restrict__ src1, int W1,
                                                           const int32_t* __restrict__ seg_ptr, const int32_t* __restrict__ members,
                                                           const int32_t* __restrict__ seg_of, const int32_t* __restrict__ n_rows,
                                                           int src0_by_position, const uint8_t* __restrict__ src0_live,
                                                           float* __restrict__ out, int64_t out_ld) {
  // (out_ld: row stride of `out` in floats - the sums may be a column block of wider rows; W1 = 0: one source only)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int V0 = W0 >> 2, V = (W0 + W1) >> 2;
  const int nr = *n_rows;
  const float4 zero = {0.f, 0.f, 0.f, 0.f};
  const int m0 = blockIdx.x * SEGSUM_CHUNK;
  // (speculative: seg_of holds gridDim.x * SEGSUM_CHUNK entries, the ones behind the last member are not used)
  int t0 = seg_of[m0], t1 = seg_of[m0 + SEGSUM_CHUNK - 1];
  const int M = seg_ptr[nr];
  for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < nr; s += gridDim.x * blockDim.x) {
    if (seg_ptr[s] == seg_ptr[s + 1]) {
      float4* o = reinterpret_cast<float4*>(out + (int64_t)s * out_ld);
      for (int c = 0; c < V; ++c) o[c] = zero;
    }
  }
  if (m0 >= M) return;
  if (m0 + SEGSUM_CHUNK > M) t1 = seg_of[M - 1];
  const int s_first = t0 + (seg_ptr[t0] != m0 ? 1 : 0);    // a segment that started in front of the chunk belongs to an earlier one
  const int ns = t1 - s_first + 1;                         // (empty segments between them included: they cost a store)
  if (ns <= 0) return;
  const int my_ptr = lane <= min(ns, 63) ? seg_ptr[s_first + lane] : 0;
  const int m_lo = __builtin_amdgcn_readlane(my_ptr, 0);
  const int m_hi = ns <= 63 ? __builtin_amdgcn_readlane(my_ptr, ns) : seg_ptr[s_first + ns];
  for (int v0 = 0; v0 < V; v0 += 256) {
    const int c = v0 + wave * 64 + lane;
    const bool in0 = c < V0, in1 = c >= V0 && c < V;
    float4* const orow = reinterpret_cast<float4*>(out + (int64_t)s_first * out_ld) + c;
    const int64_t ostep = out_ld >> 2;
    int seg = 0;
    int seg_end = __builtin_amdgcn_readlane(my_ptr, 1);
    float4 acc = zero;
    for (int mb = m_lo; mb < m_hi; mb += 64) {
      const int nb = min(64, m_hi - mb);
      const int mid = lane < nb ? members[mb + lane] : 0;
      const int lv = (lane < nb && src0_live) ? (int)src0_live[mb + lane] : 1;
      for (int j0 = 0; j0 < nb; j0 += SEGSUM_ROWS) {
        int rn[SEGSUM_ROWS], rl[SEGSUM_ROWS];
#pragma unroll
        for (int j = 0; j < SEGSUM_ROWS; ++j) {
          const int jj = min(j0 + j, nb - 1);
          rn[j] = __builtin_amdgcn_readlane(mid, jj);
          rl[j] = __builtin_amdgcn_readlane(lv, jj);
        }
        float4 v[SEGSUM_ROWS];
#pragma unroll
        for (int j = 0; j < SEGSUM_ROWS; ++j) {
          const int64_t pos = src0_by_position ? (int64_t)(mb + min(j0 + j, nb - 1)) : (int64_t)rn[j];
          const float4* p = in0 ? reinterpret_cast<const float4*>(src0 + pos * W0) + c
                                : reinterpret_cast<const float4*>(src1 + (int64_t)rn[j] * W1) + (in1 ? c - V0 : 0);
          v[j] = (in0 ? rl[j] != 0 : in1) ? *p : zero;
        }
#pragma unroll
        for (int j = 0; j < SEGSUM_ROWS; ++j) {
          const int m = mb + j0 + j;
          if (m < m_hi) {
            while (m >= seg_end) {
              if (c < V) orow[seg * ostep] = acc;
              acc = zero;
              ++seg;
              seg_end = seg + 1 <= 63 ? __builtin_amdgcn_readlane(my_ptr, seg + 1) : seg_ptr[s_first + seg + 1];
            }
            acc.x += v[j].x; acc.y += v[j].y; acc.z += v[j].z; acc.w += v[j].w;
          }
        }
      }
    }
    for (; seg < ns; ++seg) {
      if (c < V) orow[seg * ostep] = acc;
      acc = zero;
    }
  }
}

int64_t pfo_seg_of_ints(int64_t n_members) { return (pfo_ceil_div(n_members, SEGSUM_CHUNK) + 1) * SEGSUM_CHUNK; }
int pfo_segsum_launch(const float* src0, int W0, const float* src1, int W1, const int32_t* seg_ptr, const int32_t* members,
                      const int32_t* seg_of, int64_t cap_members, const int32_t* n_rows, int cap_rows, int src0_by_position,
                      const uint8_t* src0_live, float* out, hipStream_t stream) {
  PFO_REQUIRE(src0 && src1 && seg_ptr && members && n_rows && out && W0 > 0 && W1 > 0, "bad arguments");
  PFO_REQUIRE(!src0_live || src0_by_position, "row flags go with rows stored by position");
  const bool vec = ((W0 | W1) & 3) == 0 && ((((uintptr_t)src0) | ((uintptr_t)src1) | ((uintptr_t)out)) & 15) == 0;
  static const int chunked = getenv("PFO_SEGSUM_CHUNKED") ? atoi(getenv("PFO_SEGSUM_CHUNKED")) : 1;   // A/B switch
  if (vec && seg_of && cap_members > 0 && chunked) {
    const int nbm = (int)pfo_ceil_div(cap_members, SEGSUM_CHUNK);
    pfo_prof_begin(stream);
    PFO_KLAUNCH(segsum_chunk_kernel, dim3(nbm), dim3(256), 0, stream, src0, W0, src1, W1, seg_ptr, members, seg_of, n_rows,
                src0_by_position, src0_live, out, (int64_t)(W0 + W1));
    PFO_LAUNCH_CHECK();
    // every member row read once (the launch's capacity: a ~10 % over-count of the members; row flags skip part of src0) + the sums
    pfo_prof_end(PFO_PROF_SEGSUM, ((double)cap_members + (double)cap_rows) * (W0 + W1) * sizeof(float), stream);
    return PFO_OK;
  }
  const int nb = (int)std::min<int64_t>(8192, std::max<int64_t>(1, pfo_ceil_div(cap_rows, vec ? SEGSUM_NSEG : 4)));
  if (vec) PFO_KLAUNCH(segsum_vec_kernel, dim3(nb), dim3(256), 0, stream, src0, W0, src1, W1, seg_ptr, members, n_rows,
                              src0_by_position, src0_live, out);
  else PFO_KLAUNCH(segsum_kernel, dim3(nb), dim3(256), 0, stream, src0, W0, src1, W1, seg_ptr, members, n_rows,
                          src0_by_position, src0_live, out);
  PFO_LAUNCH_CHECK();
  return PFO_OK;
}

// One source, the sums written as a column block of wider rows: out[s * out_ld + 0 .. W) = sum of src[members of s] (layer-1
// backward: the d h1 half of Dq, taken beside the attention backward whose float atomics fill the d qk' half - attn.hpp dq_rows)
int pfo_segsum_cols_launch(const float* src, int W, const int32_t* seg_ptr, const int32_t* members, const int32_t* seg_of,
                           int64_t cap_members, const int32_t* n_rows, float* out, int64_t out_ld, hipStream_t stream) {
  PFO_REQUIRE(src && seg_ptr && members && seg_of && n_rows && out && W > 0 && cap_members > 0, "bad arguments");
  PFO_REQUIRE((W & 3) == 0 && (out_ld & 3) == 0 && ((((uintptr_t)src) | ((uintptr_t)out)) & 15) == 0, "rows must be float4-aligned");
  const int nbm = (int)pfo_ceil_div(cap_members, SEGSUM_CHUNK);
  PFO_KLAUNCH(segsum_chunk_kernel, dim3(nbm), dim3(256), 0, stream, src, W, (const float*)nullptr, 0, seg_ptr, members, seg_of, n_rows,
              0, (const uint8_t*)nullptr, out, out_ld);
  PFO_LAUNCH_CHECK();
  return PFO_OK;
}

extern "C" int pfo_segment_sum(const float* src0, int32_t W0, const float* src1, int32_t W1, const int32_t* seg_ptr,
                               const int32_t* members, const int32_t* seg_of, int64_t n_members, const int32_t* n_rows,
                               int32_t cap_rows, int32_t src0_by_position, const uint8_t* src0_live, float* out,
                               void* stream) {
  PFO_REQUIRE(cap_rows > 0 && n_members >= 0, "bad arguments");
  return pfo_segsum_launch(src0, W0, src1, W1, seg_ptr, members, seg_of, n_members, n_rows, cap_rows, src0_by_position,
                           src0_live, out, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------
// Two independent jobs that both wait for the compaction, in ONE launch: blocks [0, row_blocks) copy the touched nodes' rows
// out of the full tables (msg_table may be null: no-memory models copy node features only), the blocks behind them translate
// the level-0 node list into table rows (idx0[i] = slot[nodes0[i]]).
__global__ void pack_remap_kernel(const float* __restrict__ msg_table, int M, const float* __restrict__ memory, int D,
                                  const uint8_t* __restrict__ has_msg, const int32_t* __restrict__ touched_ids,
                                  const int32_t* __restrict__ n_touched, float* __restrict__ msg_rows,
                                  float* __restrict__ h_rows, uint8_t* __restrict__ hm, int row_blocks,
                                  const int32_t* __restrict__ nodes0, int64_t n0, const int32_t* __restrict__ slot,
                                  int32_t* __restrict__ idx0) {
  if ((int)blockIdx.x >= row_blocks) {
    const int64_t nb = gridDim.x - row_blocks;
    for (int64_t i = (int64_t)(blockIdx.x - row_blocks) * blockDim.x + threadIdx.x; i < n0; i += nb * blockDim.x)
      idx0[i] = slot[nodes0[i]];
    return;
  }
  const int lane = threadIdx.x & 63;
  const int nt = *n_touched;
  for (int s = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; s < nt; s += (row_blocks * blockDim.x) >> 6) {
    const int id = touched_ids[s];
    if (msg_table) {
      const float4* src = reinterpret_cast<const float4*>(msg_table + (int64_t)id * M);
      float4* dst = reinterpret_cast<float4*>(msg_rows + (int64_t)s * M);
      for (int c = lane; c < M / 4; c += 64) dst[c] = src[c];
    }
    const float4* hs = reinterpret_cast<const float4*>(memory + (int64_t)id * D);
    float4* hd = reinterpret_cast<float4*>(h_rows + (int64_t)s * D);
    for (int c = lane; c < D / 4; c += 64) hd[c] = hs[c];
    if (hm && lane == 0) hm[s] = has_msg[id];
  }
}

int pfo_pack_remap_launch(const float* msg_table, int M, const float* memory, int D, const uint8_t* has_msg,
                          const int32_t* touched_ids, const int32_t* n_touched, int cap, float* msg_rows, float* h_rows,
                          uint8_t* hm, const int32_t* nodes0, int64_t n0, const int32_t* slot, int32_t* idx0, hipStream_t stream) {
  PFO_REQUIRE((M % 4) == 0 && (D % 4) == 0, "row lengths must be multiples of 4");
  PFO_REQUIRE(memory && h_rows && touched_ids && n_touched && nodes0 && slot && idx0 && n0 > 0, "bad arguments");
  PFO_REQUIRE(!msg_table || (msg_rows && has_msg && hm), "null message buffers");
  const int rb = (int)std::min<int64_t>(4096, std::max<int64_t>(1, pfo_ceil_div(cap, 4)));
  const int mb = (int)std::min<int64_t>(2048, pfo_ceil_div(n0, 256));
  PFO_KLAUNCH(pack_remap_kernel, dim3(rb + mb), dim3(256), 0, stream, msg_table, M, memory, D, has_msg, touched_ids,
                     n_touched, msg_rows, h_rows, hm, rb, nodes0, n0, slot, idx0);
  PFO_LAUNCH_CHECK();
  return PFO_OK;
}

// GRU backward, gate part (torch.nn.GRUCell): from d h' and the gates the fused forward kept (gemm.hip gru_fused_kernel:
// gates[s] = r | z | n | gh_n) to the gradients of the two pre-activation blocks, d gi = (dr', dz', dn'), d gh = (dr', dz', dn' r)
// - the A operands of the GRU's weight-gradient launch.  d h' = the key-side rows the attention scattered (n_rep float replicas,
// or ONE int64 fixed-point table in deterministic mode) + d_extra (the rows' query-side gradient, already summed per row).
__global__ void gru_gates_bwd_kernel(const float* __restrict__ gates, float* __restrict__ dgi, float* __restrict__ dgh,
                                     const float* __restrict__ h_rows, const uint8_t* __restrict__ hm,
                                     const int32_t* __restrict__ n_touched, int D, const float* __restrict__ d_h0, int n_rep,
                                     int64_t rep_stride, const float* __restrict__ d_extra, int det) {
  const int64_t total = (int64_t)(*n_touched) * D;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int s = (int)(e / D), d = (int)(e - (int64_t)s * D);
    float dpr = 0.f, dpz = 0.f, dpn = 0.f, dpnr = 0.f;
    if (hm[s]) {
      const float* gs = gates + (int64_t)s * 4 * D;
      const float h = h_rows[e];
      const float r = gs[d], z = gs[D + d], nn = gs[2 * D + d], ghn = gs[3 * D + d];
      float dh = 0.f;                                  // the level-0 gradient is kept in one replica per XCD
      if (det) dh = (float)((double)reinterpret_cast<const long long*>(d_h0)[e] * (1.0 / 1099511627776.0));   // 2^-40 fixed point (attn.hpp)
      else
      for (int q = 0; q < n_rep; ++q) dh += d_h0[(int64_t)q * rep_stride + e];
      if (d_extra) dh += d_extra[e];                   // the rows' own (query-side) gradient, already summed per row
      const float dn = dh * (1.f - z);
      const float dz = dh * (h - nn);
      dpn = dn * (1.f - nn * nn);
      const float dr = dpn * ghn;
      dpr = dr * r * (1.f - r);
      dpz = dz * z * (1.f - z);
      dpnr = dpn * r;
    }
    float* gis = dgi + (int64_t)s * 3 * D;
    float* ghs = dgh + (int64_t)s * 3 * D;
    gis[d] = dpr; gis[D + d] = dpz; gis[2 * D + d] = dpn;
    ghs[d] = dpr; ghs[D + d] = dpz; ghs[2 * D + d] = dpnr;
  }
}

// the same, four hidden units per thread on 16-byte accesses (D % 4 == 0, all rows 16-byte aligned); same arithmetic per element
__global__ void gru_gates_bwd_vec_kernel(const float* __restrict__ gates, float* __restrict__ dgi, float* __restrict__ dgh,
                                         const float* __restrict__ h_rows, const uint8_t* __restrict__ hm,
                                         const int32_t* __restrict__ n_touched, int D, const float* __restrict__ d_h0, int n_rep,
                                         int64_t rep_stride, const float* __restrict__ d_extra, int det) {
  const int D4 = D >> 2;
  const int64_t total = (int64_t)(*n_touched) * D4;
  for (int64_t q4 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q4 < total; q4 += (int64_t)gridDim.x * blockDim.x) {
    const int s = (int)(q4 / D4), d = 4 * (int)(q4 - (int64_t)s * D4);
    const int64_t e = (int64_t)s * D + d;
    float o[4][4] = {};                                  // [dpr | dpz | dpn | dpn r][unit]
    if (hm[s]) {
      const float* gs = gates + (int64_t)s * 4 * D + d;
      const float4 h4 = *reinterpret_cast<const float4*>(h_rows + e);
      const float4 r4 = *reinterpret_cast<const float4*>(gs), z4 = *reinterpret_cast<const float4*>(gs + D);
      const float4 n4 = *reinterpret_cast<const float4*>(gs + 2 * D), g4 = *reinterpret_cast<const float4*>(gs + 3 * D);
      float dh[4] = {0.f, 0.f, 0.f, 0.f};
      if (det) {
        const long long* t = reinterpret_cast<const long long*>(d_h0) + e;
#pragma unroll
        for (int u = 0; u < 4; ++u) dh[u] = (float)((double)t[u] * (1.0 / 1099511627776.0));       // 2^-40 fixed point (attn.hpp)
      } else {
        for (int q = 0; q < n_rep; ++q) {
          const float4 v = *reinterpret_cast<const float4*>(d_h0 + (int64_t)q * rep_stride + e);
          dh[0] += v.x; dh[1] += v.y; dh[2] += v.z; dh[3] += v.w;
        }
      }
      if (d_extra) {
        const float4 v = *reinterpret_cast<const float4*>(d_extra + e);
        dh[0] += v.x; dh[1] += v.y; dh[2] += v.z; dh[3] += v.w;
      }
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
    float* gis = dgi + (int64_t)s * 3 * D + d;
    float* ghs = dgh + (int64_t)s * 3 * D + d;
    const float4 a = {o[0][0], o[0][1], o[0][2], o[0][3]}, b = {o[1][0], o[1][1], o[1][2], o[1][3]};
    const float4 c = {o[2][0], o[2][1], o[2][2], o[2][3]}, c2 = {o[3][0], o[3][1], o[3][2], o[3][3]};
    *reinterpret_cast<float4*>(gis) = a; *reinterpret_cast<float4*>(gis + D) = b; *reinterpret_cast<float4*>(gis + 2 * D) = c;
    *reinterpret_cast<float4*>(ghs) = a; *reinterpret_cast<float4*>(ghs + D) = b; *reinterpret_cast<float4*>(ghs + 2 * D) = c2;
  }
}

int pfo_gru_gates_bwd_launch(const float* gates, float* dgi, float* dgh, const float* h_rows, const uint8_t* hm,
                             const int32_t* n_touched, int cap, int D, const float* d_h0, int n_rep, int64_t rep_stride,
                             const float* d_extra, int det, hipStream_t stream) {
  const bool vec = (D & 3) == 0 && (rep_stride & 3) == 0 &&
                   ((((uintptr_t)gates) | ((uintptr_t)dgi) | ((uintptr_t)dgh) | ((uintptr_t)h_rows) | ((uintptr_t)d_h0) | ((uintptr_t)d_extra)) & 15) == 0;
  pfo_prof_begin(stream);
  if (vec) {
    const int nb = (int)std::min<int64_t>(4096, pfo_ceil_div((int64_t)cap * (D / 4), 256));
    PFO_KLAUNCH(gru_gates_bwd_vec_kernel, dim3(nb), dim3(256), 0, stream, gates, dgi, dgh, h_rows, hm, n_touched, D, d_h0, n_rep,
                       rep_stride, d_extra, det);
  } else {
    const int nb = (int)std::min<int64_t>(4096, pfo_ceil_div((int64_t)cap * D, 256));
    PFO_KLAUNCH(gru_gates_bwd_kernel, dim3(nb), dim3(256), 0, stream, gates, dgi, dgh, h_rows, hm, n_touched, D, d_h0, n_rep,
                       rep_stride, d_extra, det);
  }
  PFO_LAUNCH_CHECK();
  // per touched row: 4 D gate values + D memory + (n_rep + 1) D gradient floats in, 6 D floats out
  pfo_prof_end_dev(PFO_PROF_GRU_GATES_BWD, (double)D * (4 + 1 + n_rep + (d_extra ? 1 : 0) + 6) * sizeof(float), n_touched, cap, stream);
  return PFO_OK;
}

// ---------------------------------------------------------------------------------------------
// persist (tgn.py:295 -> memory_updater.py:18-33): the positives' lazily updated rows become the
// stored memory; last_update takes the consumed message's time.  Duplicates write equal values.
__global__ void persist_kernel(const int32_t* __restrict__ src, const int32_t* __restrict__ dst, int B,
                               const int32_t* __restrict__ slot, const float* __restrict__ upd_mem,
                               const uint8_t* __restrict__ has_msg, const float* __restrict__ msg_time,
                               float* __restrict__ memory, float* __restrict__ last_update, int D,
                               int32_t* __restrict__ winner) {
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (wave >= 2 * B) return;
  const int id = wave < B ? src[wave] : dst[wave - B];
  if (winner && lane == 0) winner[id] = -1;    // large batches: reset for the atomicMax pass that follows (msg_winner_max_kernel)
  if (!has_msg[id]) return;
  const int s = slot[id];
  if (s < 0) return;
  for (int d = lane; d < D; d += 64) memory[(int64_t)id * D + d] = upd_mem[(int64_t)s * D + d];
  if (lane == 0) last_update[id] = msg_time[id];
}

int pfo_persist_launch(const int32_t* src, const int32_t* dst, int B, const int32_t* slot, const float* upd_mem,
                       const uint8_t* has_msg, const float* msg_time, float* memory, float* last_update, int D,
                       int32_t* winner, hipStream_t stream) {
  PFO_KLAUNCH(persist_kernel, dim3((unsigned)pfo_ceil_div(2 * B, 4)), dim3(256), 0, stream, src, dst, B, slot,
                     upd_mem, has_msg, msg_time, memory, last_update, D, winner);
  PFO_LAUNCH_CHECK();
  return PFO_OK;
}

// raw messages (tgn.py:357-378): event e = side*B + i; append order is all source-side messages in
// batch order, then all destination-side ones, so "last" == the largest e that names the node.
__global__ void msg_winner_max_kernel(const int32_t* __restrict__ src, const int32_t* __restrict__ dst, int B,
                                      int32_t* __restrict__ winner) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= 2 * B) return;
  atomicMax(&winner[e < B ? src[e] : dst[e - B]], e);
}
// winner == null: event e finds out by itself whether a LATER event of the batch names its node (64 ids per step over the
// 2B - e - 1 later events: the id lists are cache-resident) - no table, no atomics, no extra launch.  Used up to
// MSG_INLINE_MAX events; beyond that the quadratic scan loses to the atomicMax table.
#define MSG_INLINE_MAX 16384
__global__ void msg_write_kernel(const int32_t* __restrict__ src, const int32_t* __restrict__ dst,
                                 const double* __restrict__ ts, const int32_t* __restrict__ eidx, int B,
                                 const float* __restrict__ memory, const float* __restrict__ last_update,
                                 const float* __restrict__ edge_feat, const float* __restrict__ tw,
                                 const float* __restrict__ tb, int D, int Ef, float* __restrict__ msg_table,
                                 float* __restrict__ msg_time, uint8_t* __restrict__ has_msg,
                                 const int32_t* __restrict__ winner) {
  const int e = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (e >= 2 * B) return;
  const int i = e < B ? e : e - B;
  const int X = e < B ? src[i] : dst[i];
  const int O = e < B ? dst[i] : src[i];
  if (winner) {
    if (winner[X] != e) return;
  } else {
    bool later = false;
    for (int e2 = e + 1 + lane; e2 < 2 * B; e2 += 64) later = later || ((e2 < B ? src[e2] : dst[e2 - B]) == X);
    if (__ballot(later) != 0ull) return;
  }
  const int M = 3 * D + Ef;
  float* out = msg_table + (int64_t)X * M;
  const float t = (float)ts[i];                       // tgn.py:359
  const float delta = t - last_update[X];             // tgn.py:367 (fp32)
  for (int d = lane; d < D; d += 64) {
    out[d] = memory[(int64_t)X * D + d];
    out[D + d] = memory[(int64_t)O * D + d];
    out[2 * D + Ef + d] = pfo_cosf(pfo_time_arg(delta, tw[d], tb[d]));
  }
  if (lane < Ef) out[2 * D + lane] = edge_feat[(int64_t)eidx[i] * Ef + lane];
  if (lane == 0) {
    msg_time[X] = t;
    has_msg[X] = 1;
  }
}

bool pfo_msg_store_needs_winner(int B) { return 2 * (int64_t)B > MSG_INLINE_MAX; }
int pfo_msg_store_launch(const int32_t* src, const int32_t* dst, const double* ts, const int32_t* eidx, int B,
                         const float* memory, const float* last_update, const float* edge_feat, const float* tw,
                         const float* tb, int D, int Ef, float* msg_table, float* msg_time, uint8_t* has_msg,
                         int32_t* winner, hipStream_t stream) {
  if (pfo_msg_store_needs_winner(B)) {
    PFO_REQUIRE(winner, "large batches need the winner table");
    const unsigned nb = (unsigned)pfo_ceil_div(2 * B, 256);
    PFO_KLAUNCH(msg_winner_max_kernel, dim3(nb), dim3(256), 0, stream, src, dst, B, winner);
  } else {
    winner = nullptr;
  }
  PFO_KLAUNCH(msg_write_kernel, dim3((unsigned)pfo_ceil_div(2 * B, 4)), dim3(256), 0, stream, src, dst, ts, eidx,
                     B, memory, last_update, edge_feat, tw, tb, D, Ef, msg_table, msg_time, has_msg, winner);
  PFO_LAUNCH_CHECK();
  return PFO_OK;
}

// zeroes rows [0, *n_rows) of n_rep replicas (each rep_stride floats apart): the touched part of the gradient table
__global__ __launch_bounds__(256) void zero_rows_kernel(float* __restrict__ dst, const int32_t* __restrict__ n_rows, int D,
                                                        int n_rep, int64_t rep_stride) {
  const int64_t per = (int64_t)(*n_rows) * D / 4;            // D % 4 == 0
  const int64_t total = per * n_rep;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t q = e / per, o = e - q * per;
    reinterpret_cast<float4*>(dst + q * rep_stride)[o] = float4{0.f, 0.f, 0.f, 0.f};
  }
}
int pfo_zero_rows_launch(float* dst, const int32_t* n_rows, int cap_rows, int D, int n_rep, int64_t rep_stride, hipStream_t stream) {
  PFO_REQUIRE((D % 4) == 0 && (rep_stride % 4) == 0, "row length must be a multiple of 4");
  const int nb = (int)std::min<int64_t>(2048, pfo_ceil_div((int64_t)cap_rows * D / 4 * n_rep, 256));
  PFO_KLAUNCH(zero_rows_kernel, dim3(std::max(nb, 1)), dim3(256), 0, stream, dst, n_rows, D, n_rep, rep_stride);
  PFO_LAUNCH_CHECK();
  return PFO_OK;
}
