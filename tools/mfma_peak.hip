// Practical fp32-MFMA ceiling on this box: v_mfma_f32_32x32x2_f32 with 4 independent accumulators,
// no memory traffic.  hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o tools/mfma_peak && tools/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int WITH_LDS>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
  __shared__ float lds[4096];
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = a0 + threadIdx.x, b = b0;
  if (WITH_LDS) lds[threadIdx.x] = a, lds[threadIdx.x + 256] = b;
  __syncthreads();
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    if (WITH_LDS) { a = lds[(threadIdx.x + it) & 1023]; b = lds[(threadIdx.x + it + 7) & 1023]; }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  long long t1 = clock64();
  float s = 0;
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (float)(t1 - t0);
}

int main() {
  float* d;
  hipMalloc(&d, 256 * 2048 * 4);
  const int iters = 20000;
  for (int lds = 0; lds < 2; ++lds)
    for (int bpc = 1; bpc <= 2; ++bpc) {
      hipEvent_t e0, e1;
      hipEventCreate(&e0);
      hipEventCreate(&e1);
      auto kern = lds ? k<1> : k<0>;
      hipLaunchKernelGGL(kern, dim3(256 * bpc), dim3(256), 0, 0, d, 100, 1.f, 2.f);
      hipEventRecord(e0);
      hipLaunchKernelGGL(kern, dim3(256 * bpc), dim3(256), 0, 0, d, iters, 1.f, 2.f);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      float cyc;
      hipMemcpy(&cyc, d, 4, hipMemcpyDeviceToHost);
      double flops = 2.0 * 32 * 32 * 2 * 16.0 * iters * 4 /*waves*/ * 256 * bpc;
      printf("lds=%d blocks/CU=%d: %.3f ms  %.1f TFLOP/s  block0 clock64 ticks=%.0f (%.1f per MFMA)  wall/ticks -> %.0f MHz\n",
             lds, bpc, ms, flops / ms / 1e9, cyc, cyc / (16.0 * iters), cyc / ms / 1e3);
    }
  return 0;
}
