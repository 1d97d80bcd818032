// What a cross-stream wait costs the waiting queue (GPU box): K1 -> [wait] -> K2 chains on one stream, 400 links, device time per link.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/wait_cost.hip -o /tmp/wait_cost && /tmp/wait_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void spin(int* p, int n) { int v = 0; for (int i = 0; i < n; ++i) v += __builtin_amdgcn_s_memtime() & 1; if (v == -1) *p = v; }
__global__ void tiny(int* p) { if (threadIdx.x == 1000) *p = 1; }
int main() {
  hipStream_t a, b;
  CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
  CK(hipStreamCreateWithPriority(&b, hipStreamNonBlocking, -1));
  int* d; CK(hipMalloc(&d, 64));
  uint32_t* flag; CK(hipMalloc(&flag, 64)); CK(hipMemset(flag, 0, 64));
  int can = 0; (void)hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0);
  printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
  const int N = 400;
  hipEvent_t t0, t1; CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
  std::vector<hipEvent_t> ev(N);
  for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  auto run = [&](const char* name, int mode) -> int {
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
      CK(hipMemset(flag, 0, 64));
      if (mode == 1 || mode == 4) { for (int i = 0; i < N; ++i) CK(hipEventRecord(ev[i], b)); CK(hipStreamSynchronize(b)); }   // fired long ago
      if (mode == 3) { CK(hipStreamWriteValue32(b, flag, 1, 0)); CK(hipStreamSynchronize(b)); }
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(t0, a));
      for (int i = 0; i < N; ++i) {
        hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, a, d, 200);
        if (mode == 1) CK(hipStreamWaitEvent(a, ev[i], 0));                       // event of another stream, complete
        if (mode == 2) { hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, b, d); CK(hipEventRecord(ev[i], b)); CK(hipStreamWaitEvent(a, ev[i], 0)); }   // fresh event
        if (mode == 3) CK(hipStreamWaitValue32(a, flag, 1, hipStreamWaitValueGte, 0xffffffffu));   // satisfied memory word
        if (mode == 5) CK(hipEventRecord(ev[i], a));                              // a record on the waiting queue itself
        hipLaunchKernelGGL(tiny, dim3(256), dim3(256), 0, a, d);
      }
      CK(hipEventRecord(t1, a));
      CK(hipDeviceSynchronize());
      float ms = 0; CK(hipEventElapsedTime(&ms, t0, t1));
      if (ms < best) best = ms;
    }
    printf("%-58s %7.2f us per link\n", name, 1e3f * best / N);
    return 0;
  };
  if (run("K1 -> K2 (no wait)", 0)) return 1;
  if (run("K1 -> wait(event of another stream, fired long ago) -> K2", 1)) return 1;
  if (run("K1 -> wait(event recorded just now on another stream) -> K2", 2)) return 1;
  if (can && run("K1 -> hipStreamWaitValue32(satisfied word) -> K2", 3)) return 1;
  if (run("K1 -> hipEventRecord on the same stream -> K2", 5)) return 1;
  return 0;
}
