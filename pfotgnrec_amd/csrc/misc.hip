// Error plumbing and the small element-wise kernels of the step: TimeEncode, BPR loss, Adam,
// row scatter-add, the folded query-bias backward, partial-slab folds.
#include "memory.hpp"
#include <string.h>
#include <algorithm>
#include <stdlib.h>

static thread_local char g_err[512] = "";

void pfo_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* pfo_last_error(void) { return g_err; }

// ---------------------------------------------------------------------------------------------
// event-pair profiler
#include <vector>
namespace {
// `pin` >= 0: the launch's extent is a device-side count (touched rows); it is copied to slot `pin` of a pinned host ring
// right behind the kernel and the record's work is work * min(count, cap), evaluated in pfo_prof_collect
struct ProfRec { int kind; double work; hipEvent_t a, b; int pin; int cap; };
bool g_prof_on = false;
std::vector<ProfRec> g_recs;
std::vector<hipEvent_t> g_pool;
hipEvent_t g_pending = nullptr;
const int PIN_SLOTS = 8192;
int32_t* g_pin = nullptr;
int g_pin_next = 0;
hipEvent_t prof_event() {
  if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
  hipEvent_t e = nullptr;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return e;
}
}  // namespace
bool pfo_prof_on() { return g_prof_on; }
void pfo_prof_begin(hipStream_t s) {
  if (!g_prof_on) return;
  g_pending = prof_event();
  if (g_pending) (void)hipEventRecord(g_pending, s);
}
void pfo_prof_end(int kind, double work, hipStream_t s) {
  if (!g_prof_on || !g_pending) return;
  hipEvent_t b = prof_event();
  if (!b) return;
  (void)hipEventRecord(b, s);
  g_recs.push_back(ProfRec{kind, work, g_pending, b, -1, 0});
  g_pending = nullptr;
}
void pfo_prof_end_dev(int kind, double work_per_unit, const int32_t* units_dev, int units_cap, hipStream_t s) {
  if (!g_prof_on || !g_pending) return;
  if (!units_dev) { pfo_prof_end(kind, work_per_unit * units_cap, s); return; }
  hipEvent_t b = prof_event();
  if (!b) return;
  (void)hipEventRecord(b, s);
  if (!g_pin && hipHostMalloc(reinterpret_cast<void**>(&g_pin), PIN_SLOTS * sizeof(int32_t), hipHostMallocDefault) != hipSuccess) g_pin = nullptr;
  int pin = -1;
  if (g_pin && g_pin_next < PIN_SLOTS) {
    pin = g_pin_next++;
    g_pin[pin] = -1;
    if (hipMemcpyAsync(&g_pin[pin], units_dev, sizeof(int32_t), hipMemcpyDeviceToHost, s) != hipSuccess) pin = -1;
  }
  // (ring exhausted or copy refused: the record keeps its time and counts zero work)
  g_recs.push_back(ProfRec{kind, pin >= 0 ? work_per_unit : 0.0, g_pending, b, pin, units_cap});
  g_pending = nullptr;
}
extern "C" int pfo_prof_enable(int32_t on) {
  g_prof_on = on != 0;         // records accumulate across enable/disable toggles until pfo_prof_collect drains them
  return PFO_OK;
}
extern "C" int pfo_prof_collect(double* ms, double* work, int64_t* count) {
  PFO_REQUIRE(ms && work && count, "null output");
  for (int k = 0; k < PFO_PROF_KINDS; ++k) { ms[k] = 0; work[k] = 0; count[k] = 0; }
  bool synced = false;
  for (auto& r : g_recs) {
    if (hipEventSynchronize(r.b) != hipSuccess) { pfo_set_error("pfo_prof_collect: event sync failed"); return PFO_ERR_HIP; }
    float t = 0.f;
    if (hipEventElapsedTime(&t, r.a, r.b) != hipSuccess) continue;
    double w = r.work;
    if (r.pin >= 0) {
      // the count's copy was queued behind event b on the same stream: one device-wide wait covers all of them
      if (!synced) { (void)hipDeviceSynchronize(); synced = true; }
      const int units = g_pin[r.pin];
      w = units >= 0 ? r.work * (double)(units < r.cap ? units : r.cap) : 0.0;
    }
    ms[r.kind] += t; work[r.kind] += w; count[r.kind] += 1;
    g_pool.push_back(r.a); g_pool.push_back(r.b);
  }
  g_recs.clear();
  g_pin_next = 0;
  return PFO_OK;
}
// ---------------------------------------------------------------------------------------------
// stop events bound to a launch (common.hpp)
namespace {
thread_local hipEvent_t g_stop_event = nullptr;
thread_local int g_stop_skip = 0;
const bool g_stop_enabled = !(getenv("PFO_STOP_EVENTS") && getenv("PFO_STOP_EVENTS")[0] == '0');   // A/B switch
}  // namespace
void pfo_stop_event_arm(hipEvent_t e, int skip) { g_stop_event = e; g_stop_skip = skip; }
bool pfo_stop_event_take(hipEvent_t* e) {
  if (!g_stop_event || !g_stop_enabled) return false;
  if (g_stop_skip > 0) { --g_stop_skip; return false; }
  *e = g_stop_event;
  g_stop_event = nullptr;
  return true;
}
void pfo_stop_event_disarm(hipStream_t stream) {
  if (g_stop_event) { (void)hipEventRecord(g_stop_event, stream); g_stop_event = nullptr; }
}
void pfo_stop_event_cancel() { g_stop_event = nullptr; g_stop_skip = 0; }   // error path: nothing may stay armed for an unrelated launch

// ---------------------------------------------------------------------------------------------
// milestones (include/pfotgn.h)
#include <map>
#include <string>
namespace {
struct MarkRec { const char* name; hipEvent_t ev; };
bool g_marks_on = false;
std::vector<MarkRec> g_marks;
std::vector<hipEvent_t> g_mark_pool;
}  // namespace
bool pfo_marks_on() { return g_marks_on; }
void pfo_mark_at(const char* name, hipStream_t s) {
  if (!g_marks_on || g_marks.size() >= 200000) return;
  hipEvent_t e = nullptr;
  if (!g_mark_pool.empty()) { e = g_mark_pool.back(); g_mark_pool.pop_back(); }
  else if (hipEventCreate(&e) != hipSuccess) return;
  if (hipEventRecord(e, s) != hipSuccess) { g_mark_pool.push_back(e); return; }
  g_marks.push_back(MarkRec{name, e});
}
extern "C" int pfo_marks_enable(int32_t on) { g_marks_on = on != 0; return PFO_OK; }
extern "C" int pfo_mark(const char* name, void* stream) { pfo_mark_at(name, (hipStream_t)stream); return PFO_OK; }
extern "C" int64_t pfo_marks_dump(char* out, int64_t cap) {
  if (!out || cap <= 0) return 0;
  (void)hipDeviceSynchronize();
  std::vector<std::string> order;
  std::map<std::string, std::pair<double, int64_t>> acc;
  // names that start with '@' are marks on OTHER streams: they take no part in the chain of consecutive differences and
  // are reported, like every mark, as an offset from the latest "step.begin"
  std::vector<std::string> off_order;
  std::map<std::string, std::pair<double, int64_t>> off;
  long prev = -1, begin = -1;
  for (size_t i = 0; i < g_marks.size(); ++i) {
    const bool side = g_marks[i].name[0] == '@';
    float ms = 0.f;
    if (!strcmp(g_marks[i].name, "step.begin")) begin = (long)i;
    else if (begin >= 0 && hipEventElapsedTime(&ms, g_marks[begin].ev, g_marks[i].ev) == hipSuccess) {
      auto it = off.find(g_marks[i].name);
      if (it == off.end()) { off_order.push_back(g_marks[i].name); off[g_marks[i].name] = {ms, 1}; }
      else { it->second.first += ms; it->second.second += 1; }
    }
    if (side) continue;
    if (prev >= 0 && hipEventElapsedTime(&ms, g_marks[prev].ev, g_marks[i].ev) == hipSuccess) {
      std::string key = std::string(g_marks[prev].name) + " -> " + g_marks[i].name;
      auto it = acc.find(key);
      if (it == acc.end()) { order.push_back(key); acc[key] = {ms, 1}; }
      else { it->second.first += ms; it->second.second += 1; }
    }
    prev = (long)i;
  }
  for (auto& m : g_marks) g_mark_pool.push_back(m.ev);
  g_marks.clear();
  int64_t n = 0;
  for (auto& k : order) {
    char line[256];
    const int len = snprintf(line, sizeof(line), "%-44s %9.2f us  n=%lld\n", k.c_str(), 1e3 * acc[k].first / (double)acc[k].second,
                             (long long)acc[k].second);
    if (len <= 0 || n + len >= cap) break;
    memcpy(out + n, line, (size_t)len);
    n += len;
  }
  std::sort(off_order.begin(), off_order.end(), [&](const std::string& a, const std::string& b) {
    return off[a].first / (double)off[a].second < off[b].first / (double)off[b].second; });
  for (size_t k = 0; k <= off_order.size() && !off_order.empty(); ++k) {
    char line[256];
    const int len = k == 0 ? snprintf(line, sizeof(line), "offsets from step.begin (@ = another stream):\n")
                           : snprintf(line, sizeof(line), "  %-42s %9.2f us  n=%lld\n", off_order[k - 1].c_str(),
                                      1e3 * off[off_order[k - 1]].first / (double)off[off_order[k - 1]].second,
                                      (long long)off[off_order[k - 1]].second);
    if (len <= 0 || n + len >= cap) break;
    memcpy(out + n, line, (size_t)len);
    n += len;
  }
  out[n] = 0;
  return n;
}

extern "C" int pfo_shader_clock(double* ghz_out, int32_t reset) {
  PFO_REQUIRE(ghz_out != nullptr, "null output");
  double ct[2 * PFO_CLOCK_KERNELS] = {0};
  if (int rc = pfo_attn_clock_read(ct, reset)) return rc;
  if (int rc = pfo_gemm_clock_read(ct + 4, reset)) return rc;
  for (int i = 0; i < PFO_CLOCK_KERNELS; ++i) ghz_out[i] = ct[2 * i + 1] > 0 ? ct[2 * i] / (ct[2 * i + 1] * 10.0) : 0.0;   // cycles per ns
  return PFO_OK;
}
extern "C" int pfo_abi_version(void) { return 6; }   // 3: pfo_tgn_batch.dropout_keep, pfo_attn_dropout_mask, PFO_PROF_GRU_FUSED; 4: pfo_segment_sum, pfo_tgn_side_stream, defer_join / pcache fields; 5: PFO_PROF_KINDS 12 -> 16 (pfo_prof_collect arrays); 6: PFO_PROF_GEMM_TN_BX8 (17 kinds), pfo_shader_clock

// ---------------------------------------------------------------------------------------------
// roctx ranges (common.hpp)
#include <dlfcn.h>
namespace {
typedef int (*roctx_push_t)(const char*);
typedef int (*roctx_pop_t)(void);
roctx_push_t g_roctx_push = nullptr;
roctx_pop_t g_roctx_pop = nullptr;
bool g_roctx_looked = false;
void roctx_lookup() {
  g_roctx_looked = true;
  const char* off = getenv("PFO_ROCTX");
  if (off && off[0] == '0') return;
  void* push = dlsym(RTLD_DEFAULT, "roctxRangePushA");
  void* pop = dlsym(RTLD_DEFAULT, "roctxRangePop");
  if (!push || !pop) {
    // PFO_ROCTX=1: load the marker library even when no profiler preloaded it (a tool attached later sees the ranges)
    if (!(off && off[0] == '1')) return;
    void* h = dlopen("librocprofiler-sdk-roctx.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return;
    push = dlsym(h, "roctxRangePushA");
    pop = dlsym(h, "roctxRangePop");
  }
  if (push && pop) { g_roctx_push = (roctx_push_t)push; g_roctx_pop = (roctx_pop_t)pop; }
}
}  // namespace
void pfo_range_push(const char* name) {
  if (!g_roctx_looked) roctx_lookup();
  if (g_roctx_push) (void)g_roctx_push(name);
}
void pfo_range_pop() {
  if (g_roctx_pop) (void)g_roctx_pop();
}

// ---------------------------------------------------------------------------------------------
__global__ void time_encode_kernel(const float* __restrict__ t, int64_t n, const float* __restrict__ w,
                                   const float* __restrict__ b, int D, float* __restrict__ out) {
  const int64_t total = n * D;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = e / D;
    const int d = (int)(e - i * D);
    out[e] = pfo_cosf(pfo_time_arg(t[i], w[d], b[d]));
  }
}
extern "C" int pfo_time_encode(const float* t, int64_t n, const float* w, const float* b, int32_t D, float* out,
                               void* stream) {
  PFO_REQUIRE(n >= 0 && D > 0, "bad sizes");
  if (n == 0) return PFO_OK;
  PFO_REQUIRE(t && w && b && out, "null input");
  const int nb = (int)std::min<int64_t>(4096, pfo_ceil_div(n * D, 256));
  PFO_KLAUNCH(time_encode_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, t, n, w, b, (int)D, out);
  PFO_LAUNCH_CHECK();
  return PFO_OK;
}

// ---------------------------------------------------------------------------------------------
// BPR (main.py:321-337): one wavefront per interaction
// ticket != null: the workgroup that finishes LAST also takes the mean of the per-interaction losses, in index order
// (reproducible) - no second launch.  The ticket word is zero before the first use and resets itself.
__device__ __forceinline__ void bpr_mean_tail(const float* loss_part, int64_t B, float* loss_out, int* ticket) {
  __shared__ int s_last;
  __shared__ float s_sum[4];
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) s_last = (atomicAdd(ticket, 1) == (int)gridDim.x - 1) ? 1 : 0;
  __syncthreads();
  if (!s_last) return;
  __threadfence();
  float s = 0.f;
  for (int64_t i = threadIdx.x; i < B; i += 256) s += __hip_atomic_load(&loss_part[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  s = pfo_wave_sum(s);
  if ((threadIdx.x & 63) == 0) s_sum[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    *loss_out = ((s_sum[0] + s_sum[1]) + (s_sum[2] + s_sum[3])) / (float)B;
    *ticket = 0;
  }
}
__global__ __launch_bounds__(256) void bpr_kernel(const float* __restrict__ emb, int64_t B, int D, int64_t pos_off, int64_t neg_off, int n_neg,
                           int64_t R, float scale, float* __restrict__ loss_part, float* __restrict__ d_emb,
                           float* __restrict__ loss_out, int* ticket) {
  const int64_t b = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (b < B) {
  const float* s = emb + b * D;
  const float* p = emb + (pos_off + b) * D;
  float pos = 0.f;
  for (int d = lane; d < D; d += 64) pos = fmaf(s[d], p[d], pos);
  pos = pfo_wave_sum(pos);
  float dsum = 0.f;
  for (int k = 0; k < n_neg; ++k) {
    const float* nk = emb + (neg_off + b * n_neg + k) * D;
    float ns = 0.f;
    for (int d = lane; d < D; d += 64) ns = fmaf(s[d], nk[d], ns);
    dsum += pos - pfo_wave_sum(ns);                              // score_diff (main.py:334)
  }
  const float dm = dsum / (float)n_neg;                          // mean over negatives (main.py:335)
  const float sg = 1.f / (1.f + expf(-dm));
  if (lane == 0) loss_part[b] = -logf(sg);                       // log(sigmoid(.)) (main.py:336)
  if (d_emb) {
    const float ddm = -(1.f - sg) / (float)B * scale;
    const float dneg = -ddm / (float)n_neg;
    for (int d = lane; d < D; d += 64) {
      float acc = ddm * p[d];
      const float sd = s[d];
      for (int k = 0; k < n_neg; ++k) {
        const int64_t r = (neg_off + b * n_neg + k) * D + d;
        acc = fmaf(dneg, emb[r], acc);
        d_emb[r] = dneg * sd;
      }
      d_emb[b * D + d] = acc;
      d_emb[(pos_off + b) * D + d] = ddm * sd;
    }
    // rows no pair touches (the destination block of the `ours` layout [src | dst | p_pos | neg], anything after the
    // negatives) get their zero here: gap row g is written by interaction g mod B
    const int64_t gap1 = pos_off - B, gap2 = neg_off - (pos_off + B), gap3 = R - (neg_off + B * n_neg);
    for (int64_t gidx = b; gidx < gap1 + gap2 + gap3; gidx += B) {
      const int64_t row = gidx < gap1 ? B + gidx : (gidx < gap1 + gap2 ? pos_off + B + (gidx - gap1) : neg_off + B * n_neg + (gidx - gap1 - gap2));
      for (int d = lane; d < D; d += 64) d_emb[row * D + d] = 0.f;
    }
  }
  }
  if (ticket) bpr_mean_tail(loss_part, B, loss_out, ticket);
}
__global__ void bpr_mean_kernel(const float* __restrict__ loss_part, int64_t B, float* __restrict__ loss_out) {
  // single wavefront, fixed order: deterministic
  float s = 0.f;
  for (int64_t i = threadIdx.x; i < B; i += 64) s += loss_part[i];
  s = pfo_wave_sum(s);
  if (threadIdx.x == 0) *loss_out = s / (float)B;
}
int pfo_mean_launch(const float* src, int64_t n, float* out, hipStream_t stream) {
  PFO_KLAUNCH(bpr_mean_kernel, dim3(1), dim3(64), 0, stream, src, n, out);
  PFO_LAUNCH_CHECK();
  return PFO_OK;
}
static int bpr_launch(const float* emb, int64_t B, int32_t D, int64_t pos_off, int64_t neg_off, int32_t n_neg, int64_t R,
                      float scale, float* loss_out, float* d_emb, float* workspace, int32_t* ticket, void* stream) {
  PFO_REQUIRE(emb && loss_out && workspace, "null input");
  PFO_REQUIRE(B > 0 && D > 0 && n_neg > 0, "bad sizes");
  PFO_REQUIRE(pos_off >= B && pos_off + B <= R && neg_off + B * n_neg <= R && neg_off >= pos_off + B, "bad offsets");
  hipStream_t s = (hipStream_t)stream;
  PFO_KLAUNCH(bpr_kernel, dim3((unsigned)pfo_ceil_div(B, 4)), dim3(256), 0, s, emb, B, (int)D, pos_off, neg_off,
                     (int)n_neg, R, scale, workspace, d_emb, loss_out, ticket);
  if (!ticket) PFO_KLAUNCH(bpr_mean_kernel, dim3(1), dim3(64), 0, s, workspace, B, loss_out);
  PFO_LAUNCH_CHECK();
  return PFO_OK;
}
extern "C" int pfo_bpr_loss(const float* emb, int64_t B, int32_t D, int64_t pos_off, int64_t neg_off, int32_t n_neg,
                            int64_t R, float scale, float* loss_out, float* d_emb, float* workspace, void* stream) {
  return bpr_launch(emb, B, D, pos_off, neg_off, n_neg, R, scale, loss_out, d_emb, workspace, nullptr, stream);
}
extern "C" int pfo_bpr_loss_fused(const float* emb, int64_t B, int32_t D, int64_t pos_off, int64_t neg_off, int32_t n_neg,
                                  int64_t R, float scale, float* loss_out, float* d_emb, float* workspace, int32_t* ticket,
                                  void* stream) {
  PFO_REQUIRE(ticket, "null ticket");
  return bpr_launch(emb, B, D, pos_off, neg_off, n_neg, R, scale, loss_out, d_emb, workspace, ticket, stream);
}
extern "C" int pfo_bpr_loss_parts(const float* emb, int64_t B, int32_t D, int64_t pos_off, int64_t neg_off, int32_t n_neg,
                                  int64_t R, float scale, float* loss_parts, float* d_emb, void* stream) {
  PFO_REQUIRE(emb && loss_parts && d_emb, "null input");
  PFO_REQUIRE(B > 0 && D > 0 && n_neg > 0, "bad sizes");
  PFO_REQUIRE(pos_off >= B && pos_off + B <= R && neg_off + B * n_neg <= R && neg_off >= pos_off + B, "bad offsets");
  PFO_KLAUNCH(bpr_kernel, dim3((unsigned)pfo_ceil_div(B, 4)), dim3(256), 0, (hipStream_t)stream, emb, B, (int)D, pos_off,
                     neg_off, (int)n_neg, R, scale, loss_parts, d_emb, (float*)nullptr, (int*)nullptr);
  PFO_LAUNCH_CHECK();
  return PFO_OK;
}

// ---------------------------------------------------------------------------------------------
// evaluation ranking (evaluation.py:114-145): one wavefront per interaction
__global__ void rank_metrics_kernel(const float* __restrict__ emb, int64_t B, int D, int n_items, int32_t* __restrict__ rank_out,
                                    float* __restrict__ hits, float* __restrict__ ndcg) {
  const int64_t b = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (b >= B) return;
  const float* s = emb + b * D;
  const float* p = emb + (B + b) * D;
  float pos = 0.f;
  for (int d = lane; d < D; d += 64) pos = fmaf(s[d], p[d], pos);
  pos = pfo_wave_sum(pos);
  int rank = 0;
  for (int k = 0; k < n_items; ++k) {
    const float* nk = emb + (2 * B + b * (int64_t)n_items + k) * D;
    float ns = 0.f;
    for (int d = lane; d < D; d += 64) ns = fmaf(s[d], nk[d], ns);
    ns = pfo_wave_sum(ns);
    rank += (ns >= pos) ? 1 : 0;
  }
  if (lane == 0) {
    if (rank_out) rank_out[b] = rank;
    const int ks[3] = {1, 3, 5};
    for (int i = 0; i < 3; ++i) {
      const bool hit = rank < ks[i];
      if (hits) hits[b * 3 + i] = hit ? 1.f : 0.f;                           // recall@k with one test item
      if (ndcg) ndcg[b * 3 + i] = hit ? 1.f / log2f((float)rank + 2.f) : 0.f; // idcg = 1
    }
  }
}
extern "C" int pfo_rank_metrics(const float* emb, int64_t B, int32_t D, int32_t n_items, int32_t* rank_out, float* hits_out,
                                float* ndcg_out, void* stream) {
  PFO_REQUIRE(emb && B > 0 && D > 0 && n_items > 0, "bad arguments");
  PFO_KLAUNCH(rank_metrics_kernel, dim3((unsigned)pfo_ceil_div(B, 4)), dim3(256), 0, (hipStream_t)stream, emb, B, (int)D,
                     (int)n_items, rank_out, hits_out, ndcg_out);
  PFO_LAUNCH_CHECK();
  return PFO_OK;
}

// ---------------------------------------------------------------------------------------------
// Adam (torch.optim.Adam, amsgrad off, weight_decay 0)
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, int64_t n, float lr, float b1, float b2, float eps, float bc1,
                            float bc2_sqrt) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float gi = g[i];
    const float mi = m[i] + (gi - m[i]) * (1.f - b1);            // lerp form used by torch
    const float vi = v[i] * b2 + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] -= (lr / bc1) * (mi / denom);
  }
}
extern "C" int pfo_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                             float beta1, float beta2, float eps, int32_t step, void* stream) {
  PFO_REQUIRE(param && grad && exp_avg && exp_avg_sq && n > 0 && step >= 1, "bad arguments");
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  const int nb = (int)std::min<int64_t>(2048, pfo_ceil_div(n, 256));
  PFO_KLAUNCH(adam_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, n, lr,
                     beta1, beta2, eps, (float)bc1, (float)sqrt(bc2));
  PFO_LAUNCH_CHECK();
  return PFO_OK;
}

struct AdamRanges { int64_t lo[PFO_ADAM_MAX_RANGES], hi[PFO_ADAM_MAX_RANGES]; float lr_bc1[PFO_ADAM_MAX_RANGES], bc2_sqrt[PFO_ADAM_MAX_RANGES]; int n; };
template <bool ZERO>
__global__ void adam_ranges_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                   float* __restrict__ v, const AdamRanges r, float b1, float b2, float eps) {
  for (int q = 0; q < r.n; ++q) {
    const float step_size = r.lr_bc1[q], bc2s = r.bc2_sqrt[q];
    for (int64_t i = r.lo[q] + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < r.hi[q]; i += (int64_t)gridDim.x * blockDim.x) {
      const float gi = g[i];
      const float mi = m[i] + (gi - m[i]) * (1.f - b1);
      const float vi = v[i] * b2 + (1.f - b2) * gi * gi;
      m[i] = mi;
      v[i] = vi;
      p[i] -= step_size * (mi / (sqrtf(vi) / bc2s + eps));
      if (ZERO) g[i] = 0.f;         // optimizer.zero_grad() folded in: the next backward clears nothing on its critical path
    }
  }
}
int pfo_adam_step_ranges_impl(float* param, float* grad, float* exp_avg, float* exp_avg_sq, int32_t n_ranges,
                              const int64_t* lo, const int64_t* hi, const int32_t* step, float lr, float beta1,
                              float beta2, float eps, bool zero_grad, void* stream);
extern "C" int pfo_adam_step_ranges(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int32_t n_ranges,
                                    const int64_t* lo, const int64_t* hi, const int32_t* step, float lr, float beta1,
                                    float beta2, float eps, void* stream) {
  return pfo_adam_step_ranges_impl(param, const_cast<float*>(grad), exp_avg, exp_avg_sq, n_ranges, lo, hi, step, lr, beta1, beta2, eps, false, stream);
}
int pfo_adam_step_ranges_impl(float* param, float* grad, float* exp_avg, float* exp_avg_sq, int32_t n_ranges,
                              const int64_t* lo, const int64_t* hi, const int32_t* step, float lr, float beta1,
                              float beta2, float eps, bool zero_grad, void* stream) {
  PFO_REQUIRE(param && grad && exp_avg && exp_avg_sq, "null buffer");
  PFO_REQUIRE(n_ranges >= 0 && n_ranges <= PFO_ADAM_MAX_RANGES, "too many ranges");
  if (n_ranges == 0) return PFO_OK;
  PFO_REQUIRE(lo && hi && step, "null range arrays");
  AdamRanges r;
  r.n = n_ranges;
  int64_t longest = 0;
  for (int q = 0; q < n_ranges; ++q) {
    PFO_REQUIRE(lo[q] >= 0 && hi[q] >= lo[q] && step[q] >= 1, "bad range");
    r.lo[q] = lo[q]; r.hi[q] = hi[q];
    const double bc1 = 1.0 - pow((double)beta1, (double)step[q]);
    const double bc2 = 1.0 - pow((double)beta2, (double)step[q]);
    r.lr_bc1[q] = lr / (float)bc1;
    r.bc2_sqrt[q] = (float)sqrt(bc2);
    longest = std::max(longest, hi[q] - lo[q]);
  }
  const int nb = (int)std::min<int64_t>(2048, std::max<int64_t>(1, pfo_ceil_div(longest, 256)));
  if (zero_grad) PFO_KLAUNCH(adam_ranges_kernel<true>, dim3(nb), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, r, beta1, beta2, eps);
  else PFO_KLAUNCH(adam_ranges_kernel<false>, dim3(nb), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, r, beta1, beta2, eps);
  PFO_LAUNCH_CHECK();
  return PFO_OK;
}

// the same with the step counts on the device: t_r = base_r + *step_dev (graph-captured steps)
struct AdamRangesDev { int64_t lo[PFO_ADAM_MAX_RANGES], hi[PFO_ADAM_MAX_RANGES]; int base[PFO_ADAM_MAX_RANGES]; int n; };
__global__ void adam_ranges_dev_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                       float* __restrict__ v, const AdamRangesDev r, const int32_t* __restrict__ step_dev,
                                       float lr, float b1, float b2, float eps) {
  const int sd = *step_dev;
  for (int q = 0; q < r.n; ++q) {
    const double t = (double)(r.base[q] + sd);
    const float step_size = lr / (float)(1.0 - pow((double)b1, t));
    const float bc2s = (float)sqrt(1.0 - pow((double)b2, t));
    for (int64_t i = r.lo[q] + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < r.hi[q]; i += (int64_t)gridDim.x * blockDim.x) {
      const float gi = g[i];
      const float mi = m[i] + (gi - m[i]) * (1.f - b1);
      const float vi = v[i] * b2 + (1.f - b2) * gi * gi;
      m[i] = mi;
      v[i] = vi;
      p[i] -= step_size * (mi / (sqrtf(vi) / bc2s + eps));
    }
  }
}
extern "C" int pfo_adam_step_ranges_dev(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int32_t n_ranges,
                                        const int64_t* lo, const int64_t* hi, const int32_t* step, const int32_t* step_dev,
                                        float lr, float beta1, float beta2, float eps, void* stream) {
  PFO_REQUIRE(param && grad && exp_avg && exp_avg_sq && step_dev, "null buffer");
  PFO_REQUIRE(n_ranges >= 0 && n_ranges <= PFO_ADAM_MAX_RANGES, "too many ranges");
  if (n_ranges == 0) return PFO_OK;
  PFO_REQUIRE(lo && hi && step, "null range arrays");
  AdamRangesDev r;
  r.n = n_ranges;
  int64_t longest = 0;
  for (int q = 0; q < n_ranges; ++q) {
    PFO_REQUIRE(lo[q] >= 0 && hi[q] >= lo[q] && step[q] >= 0, "bad range");
    r.lo[q] = lo[q]; r.hi[q] = hi[q]; r.base[q] = step[q];
    longest = std::max(longest, hi[q] - lo[q]);
  }
  const int nb = (int)std::min<int64_t>(2048, std::max<int64_t>(1, pfo_ceil_div(longest, 256)));
  PFO_KLAUNCH(adam_ranges_dev_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, r,
                     step_dev, lr, beta1, beta2, eps);
  PFO_LAUNCH_CHECK();
  return PFO_OK;
}

// one workgroup per time dimension d, every layer in turn: d Wq[:, D+d] += gq * cos(b_d);  d tb[d] += -sin(b_d) * sum_e Wq[e, D+d] gq[e]
// The time-bias term either goes to `tb_part` (stored: a partial launch for the top layer, whose other gradients are then
// final - see pfo_tgn_backward_ev) or is added to d_tb together with an earlier launch's `tb_add`.  `dtime` (optional):
// the step's fp64 time-encoder partial sums [n_bins][2D] (attention backward), folded here in bin order into d_tw / d_tb.
struct CqBwdDev {
  const float* gq[PFO_MAX_LAYERS]; const float* Wq[PFO_MAX_LAYERS]; float* d_bq[PFO_MAX_LAYERS]; float* d_Wq[PFO_MAX_LAYERS];
  int n;
};
__global__ __launch_bounds__(256) void cq_backward_kernel(const CqBwdDev q, const float* __restrict__ tb, int D, float* __restrict__ d_tb,
                                                          float* __restrict__ tb_part, const float* __restrict__ tb_add,
                                                          const double* __restrict__ dtime, int n_bins, float* __restrict__ d_tw) {
  __shared__ float s_part[4];
  __shared__ double s_bins[2][4];
  const int d = blockIdx.x, E = 2 * D;
  float sb, cb;
  pfo_sincosf(tb[d], sb, cb);                                     // query time feature is cos(fma(0, w, b)) = cos(b)
  float total = 0.f;
  for (int l = 0; l < q.n; ++l) {
    const float* __restrict__ gq = q.gq[l];
    const float* __restrict__ Wq = q.Wq[l];
    float* __restrict__ d_Wq = q.d_Wq[l];
    float* __restrict__ d_bq = q.d_bq[l];
    float part = 0.f;
    for (int e = threadIdx.x; e < E; e += 256) {
      const float g = gq[e];
      d_Wq[(int64_t)e * E + D + d] += g * cb;
      part = fmaf(Wq[(int64_t)e * E + D + d], g, part);
      if (d == 0) d_bq[e] += g;
    }
    part = pfo_wave_sum(part);
    __syncthreads();                                              // the previous layer's partials have been read
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = part;
    __syncthreads();
    if (threadIdx.x == 0) total += (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]);
  }
  // time-encoder partial sums of this column: wavefront 0 folds the d_tw bins, wavefront 1 the d_tb bins, each in four
  // interleaved chains of ascending bins and a fixed tree (reproducible)
  double fold_w = 0.0, fold_b = 0.0;
  if (dtime) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave < 2 && lane < 4) {
      double acc = 0.0;
      for (int p = lane; p < n_bins; p += 4) acc += dtime[(int64_t)p * 2 * D + wave * D + d];
      s_bins[wave][lane] = acc;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      fold_w = (s_bins[0][0] + s_bins[0][1]) + (s_bins[0][2] + s_bins[0][3]);
      fold_b = (s_bins[1][0] + s_bins[1][1]) + (s_bins[1][2] + s_bins[1][3]);
    }
  }
  if (threadIdx.x == 0) {
    const float mine = -sb * total;
    if (tb_part) tb_part[d] = mine;
    else {
      if (dtime) {
        d_tw[d] = (float)((double)d_tw[d] + fold_w);
        d_tb[d] = (float)((double)d_tb[d] + fold_b);
      }
      d_tb[d] += mine + (tb_add ? tb_add[d] : 0.f);
    }
  }
}
int pfo_cq_backward_launch(const float* const* gq, const float* const* Wq, int n_layers, const float* tb, int D, float* const* d_bq,
                           float* const* d_Wq, float* d_tb, float* tb_part, const float* tb_add, const double* dtime, int n_bins,
                           float* d_tw, hipStream_t stream) {
  PFO_REQUIRE(n_layers >= 1 && n_layers <= PFO_MAX_LAYERS, "bad layer count");
  PFO_REQUIRE(!dtime || (d_tw && !tb_part), "the time-partial fold belongs to the final launch");
  CqBwdDev q;
  q.n = n_layers;
  for (int l = 0; l < n_layers; ++l) { q.gq[l] = gq[l]; q.Wq[l] = Wq[l]; q.d_bq[l] = d_bq[l]; q.d_Wq[l] = d_Wq[l]; }
  PFO_KLAUNCH(cq_backward_kernel, dim3(D), dim3(256), 0, stream, q, tb, D, d_tb, tb_part, tb_add, dtime, n_bins, d_tw);
  PFO_LAUNCH_CHECK();
  return PFO_OK;
}

// out[c] (+)= sum_p parts[p][c] in fp64, fixed summation order (deterministic).  Two levels in ONE launch: workgroup
// (x, y) folds the parts p = 4y + lane-group (mod 4 * FOLD_SLICES) of 64 columns into slice y of `scratch`, takes a
// ticket, and the last workgroup of a column block folds the FOLD_SLICES slices in slice order and resets the ticket.
#define FOLD_SLICES 32
__global__ __launch_bounds__(256) void fold_parts_kernel(const double* __restrict__ parts, int n_parts, int n,
                                                         float* __restrict__ out, int accumulate, double* scratch,
                                                         int* tickets, double* __restrict__ out64) {
  __shared__ double s_red[4][64];
  __shared__ int s_last;
  const int c = threadIdx.x & 63, pl = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + c;
  double s = 0.0;
  if (col < n) {
    // eight part rows in flight per thread (they lie 4 * FOLD_SLICES rows apart: one at a time, every load is a full
    // memory round trip); fixed order -> reproducible
    int p = blockIdx.y * 4 + pl;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    for (; p + 7 * 4 * FOLD_SLICES < n_parts; p += 8 * 4 * FOLD_SLICES) {
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = parts[(int64_t)(p + u * 4 * FOLD_SLICES) * n + col];
      s0 += v[0]; s1 += v[1]; s2 += v[2]; s3 += v[3];
      s0 += v[4]; s1 += v[5]; s2 += v[6]; s3 += v[7];
    }
    for (; p < n_parts; p += 4 * FOLD_SLICES) s0 += parts[(int64_t)p * n + col];
    s = (s0 + s1) + (s2 + s3);
  }
  s_red[pl][c] = s;
  __syncthreads();
  if (pl == 0 && col < n) scratch[(int64_t)blockIdx.y * n + col] = (s_red[0][c] + s_red[1][c]) + (s_red[2][c] + s_red[3][c]);
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) s_last = (atomicAdd(&tickets[blockIdx.x], 1) == FOLD_SLICES - 1) ? 1 : 0;
  __syncthreads();
  if (!s_last) return;
  __threadfence();
  if (pl == 0 && col < n) {
    double t0 = 0.0, t1 = 0.0, t2 = 0.0, t3 = 0.0;
#pragma unroll
    for (int y = 0; y < FOLD_SLICES; y += 4) {
      t0 += scratch[(int64_t)y * n + col]; t1 += scratch[(int64_t)(y + 1) * n + col];
      t2 += scratch[(int64_t)(y + 2) * n + col]; t3 += scratch[(int64_t)(y + 3) * n + col];
    }
    const double t = (t0 + t1) + (t2 + t3);
    if (out64) out64[col] = t;
    else out[col] = accumulate ? (float)((double)out[col] + t) : (float)t;
  }
  if (threadIdx.x == 0) tickets[blockIdx.x] = 0;
}
int64_t pfo_fold_parts_scratch_doubles(int n) { return (int64_t)FOLD_SLICES * n; }
int pfo_fold_parts_launch(const double* parts, int n_parts, int n, float* out, int accumulate, double* scratch, int* tickets,
                          hipStream_t stream, double* out64) {
  PFO_REQUIRE(n <= 64 * 64, "too many columns for the ticket array");
  PFO_KLAUNCH(fold_parts_kernel, dim3((unsigned)pfo_ceil_div(n, 64), FOLD_SLICES), dim3(256), 0, stream, parts, n_parts,
                     n, out, accumulate, scratch, tickets, out64);
  PFO_LAUNCH_CHECK();
  return PFO_OK;
}

// ---------------------------------------------------------------------------------------------
// Root list of one batch shard (tgn.py:118-124 / 237-239): [src | dst | group 0 | group 1 ...] with the edge time of the
// interaction each root belongs to; group g holds reps[g] nodes per interaction, row-major.
struct RootsDev { const int32_t* grp[PFO_MAX_ROOT_GROUPS]; int rep[PFO_MAX_ROOT_GROUPS]; int n_groups; };
__global__ __launch_bounds__(256) void roots_assemble_kernel(const int32_t* __restrict__ src, const int32_t* __restrict__ dst,
                                                             const double* __restrict__ ts, int lo, int hi, const RootsDev g,
                                                             int32_t* __restrict__ roots, double* __restrict__ root_ts) {
  const int b = hi - lo;
  int64_t total = 2 * (int64_t)b;
  for (int q = 0; q < g.n_groups; ++q) total += (int64_t)b * g.rep[q];
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    int64_t o = e;
    int node, i;
    if (o < b) { i = (int)o; node = src[lo + i]; }
    else if (o < 2 * (int64_t)b) { i = (int)(o - b); node = dst[lo + i]; }
    else {
      o -= 2 * (int64_t)b;
      int q = 0;
      while (q + 1 < g.n_groups && o >= (int64_t)b * g.rep[q]) { o -= (int64_t)b * g.rep[q]; ++q; }
      i = (int)(o / g.rep[q]);
      node = g.grp[q][(int64_t)(lo + i) * g.rep[q] + (o - (int64_t)i * g.rep[q])];
    }
    roots[e] = node;
    root_ts[e] = ts[lo + i];
  }
}
extern "C" int pfo_roots_assemble(const int32_t* src, const int32_t* dst, const double* ts, int32_t lo, int32_t hi,
                                  const int32_t* const* groups, const int32_t* reps, int32_t n_groups, int32_t* roots,
                                  double* root_ts, void* stream) {
  PFO_REQUIRE(src && dst && ts && roots && root_ts && hi > lo && lo >= 0, "bad arguments");
  PFO_REQUIRE(n_groups >= 0 && n_groups <= PFO_MAX_ROOT_GROUPS && (n_groups == 0 || (groups && reps)), "bad groups");
  RootsDev g;
  memset(&g, 0, sizeof(g));
  g.n_groups = n_groups;
  int64_t total = 2 * (int64_t)(hi - lo);
  for (int q = 0; q < n_groups; ++q) {
    PFO_REQUIRE(groups[q] && reps[q] >= 1, "bad group");
    g.grp[q] = groups[q]; g.rep[q] = reps[q];
    total += (int64_t)(hi - lo) * reps[q];
  }
  PFO_KLAUNCH(roots_assemble_kernel, dim3((unsigned)std::min<int64_t>(1024, pfo_ceil_div(total, 256))), dim3(256), 0,
                     (hipStream_t)stream, src, dst, ts, lo, hi, g, roots, root_ts);
  PFO_LAUNCH_CHECK();
  return PFO_OK;
}
