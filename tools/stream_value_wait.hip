// Cross-stream dependency cost on MI355X: hipEventRecord / hipStreamWaitEvent against hipStreamWriteValue32 / hipStreamWaitValue32
// (stream memory operations on signal memory).  Two streams ping-pong N tiny kernels; reported: microseconds per hop.
//   hipcc --offload-arch=gfx950 -O2 tools/stream_value_wait.hip -o /tmp/svw && /tmp/svw
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void tiny(int* p) { if (threadIdx.x == 0) atomicAdd(p, 1); }
__global__ void spin(long long cycles) { const long long t0 = clock64(); while (clock64() - t0 < cycles) {} }
int main() {
  int can = 0;
  CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
  printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
  hipStream_t s[2];
  CK(hipStreamCreateWithFlags(&s[0], hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s[1], hipStreamNonBlocking));
  int* cnt; CK(hipMalloc(&cnt, 4)); CK(hipMemset(cnt, 0, 4));
  const int N = 400;
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
  // ---- same stream
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipDeviceSynchronize());
    auto t0 = now();
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s[0], cnt);
    CK(hipDeviceSynchronize());
    if (rep) printf("one stream:                 %.2f us per kernel\n", us(t0, now()) / N);
  }
  // ---- events
  std::vector<hipEvent_t> ev(N);
  for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipDeviceSynchronize());
    auto t0 = now();
    for (int i = 0; i < N; ++i) {
      hipStream_t a = s[i & 1], b = s[(i + 1) & 1];
      hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, a, cnt);
      CK(hipEventRecord(ev[i], a));
      CK(hipStreamWaitEvent(b, ev[i], 0));
    }
    CK(hipDeviceSynchronize());
    if (rep) printf("ping-pong, events:          %.2f us per hop\n", us(t0, now()) / N);
  }
  // ---- stream memory operations
  if (can) {
    uint64_t* sig = nullptr;
    CK(hipExtMallocWithFlags((void**)&sig, 8, hipMallocSignalMemory));
    CK(hipMemset(sig, 0, 8));
    unsigned base = 0;
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipDeviceSynchronize());
      auto t0 = now();
      for (int i = 0; i < N; ++i) {
        hipStream_t a = s[i & 1], b = s[(i + 1) & 1];
        hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, a, cnt);
        CK(hipStreamWriteValue32(a, sig, base + i + 1, 0));
        CK(hipStreamWaitValue32(b, sig, base + i + 1, hipStreamWaitValueGte, 0xffffffffu));
      }
      CK(hipDeviceSynchronize());
      if (rep) printf("ping-pong, write/wait value: %.2f us per hop\n", us(t0, now()) / N);
      base += N;
    }
  }
  // ---- GPU side alone: the whole ping-pong is enqueued behind a ~30 ms spin kernel, timed with events on the device
  {
    hipEvent_t e0, e1, gate;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreateWithFlags(&gate, hipEventDisableTiming));
    uint64_t* sig = nullptr;
    if (can) { CK(hipExtMallocWithFlags((void**)&sig, 8, hipMallocSignalMemory)); CK(hipMemset(sig, 0, 8)); }
    for (int mode = 0; mode < (can ? 3 : 2); ++mode) {      // 0 one stream, 1 events, 2 write/wait value
      CK(hipDeviceSynchronize());
      if (sig) CK(hipMemset(sig, 0, 8));
      hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s[0], 60000000LL);
      CK(hipEventRecord(gate, s[0]));
      CK(hipStreamWaitEvent(s[1], gate, 0));
      CK(hipEventRecord(e0, s[0]));
      for (int i = 0; i < N; ++i) {
        hipStream_t a = mode ? s[i & 1] : s[0], b = mode ? s[(i + 1) & 1] : s[0];
        hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, a, cnt);
        if (mode == 1) { CK(hipEventRecord(ev[i], a)); CK(hipStreamWaitEvent(b, ev[i], 0)); }
        if (mode == 2) { CK(hipStreamWriteValue32(a, sig, i + 1, 0)); CK(hipStreamWaitValue32(b, sig, i + 1, hipStreamWaitValueGte, 0xffffffffu)); }
      }
      hipStream_t last = mode ? s[N & 1] : s[0];
      hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, last, cnt);
      CK(hipEventRecord(e1, last));
      CK(hipDeviceSynchronize());
      float ms = 0.f;
      CK(hipEventElapsedTime(&ms, e0, e1));
      const char* names[3] = {"one stream", "ping-pong, events", "ping-pong, write/wait value"};
      printf("GPU side alone (pre-enqueued), %-28s %.2f us per hop\n", names[mode], ms * 1e3 / N);
    }
  }
  int h = 0; CK(hipMemcpy(&h, cnt, 4, hipMemcpyDeviceToHost));
  printf("kernels run: %d\n", h);
  return 0;
}
