// What does a v_mfma_f32_16x16x4_f32 stream of conv_wino3_kernel's shape cost on gfx950?
//   hipcc --offload-arch=gfx950 -O3 tools/mfma16_stream.hip -o build_tools/m16 && build_tools/m16
// One wave per SIMD (256 threads), groups of 8 MFMAs on two alternating accumulators (dependency distance 2) or on
// DEP independent accumulators, with F filler instructions of a kind in front of every group:
//   kind 0 none, 1 v_pk_add_f32, 2 v_add_f32, 3 ds_read_b128, 4 v_pk_add_f32 + 2 ds_read_b128 (the kernel's mix)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int KIND, int F, int NACC>
__global__ __launch_bounds__(256, 1) void stream(float* out, long long* ticks, int iters, float a0) {
  __shared__ f32x4 lds[2048];
  for (int i = threadIdx.x; i < 2048; i += 256) lds[i] = f32x4{a0, a0, a0, a0};
  __syncthreads();
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float a = a0 + threadIdx.x, b = 2.f;
  f32x2 p[8];
  float x[8];
  f32x4 q[4];
  for (int i = 0; i < 8; ++i) { p[i] = f32x2{a0, a0 + i}; x[i] = a0 + i; }
  for (int i = 0; i < 4; ++i) q[i] = f32x4{a0, a0, a0, a0};
  const unsigned la = (threadIdx.x & 63) * 16;
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < 8; ++g) {
#pragma unroll
      for (int f = 0; f < F; ++f) {
        if (KIND == 1 || KIND == 4) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[f & 7]) : "v"(p[(f + 1) & 7]));
        if (KIND == 2) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[f & 7]) : "v"(a0));
        if (KIND == 3) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q[f & 3]) : "v"(la), "i"(4096));
      }
      if (KIND == 4) {
        asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(q[0]) : "v"(la));
        asm volatile("ds_read_b128 %0, %1 offset:8192" : "=v"(q[1]) : "v"(la));
      }
#pragma unroll
      for (int e = 0; e < 8; ++e)
        asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[e % NACC]) : "v"(a), "v"(b));
    }
    if (KIND >= 3) asm volatile("s_waitcnt lgkmcnt(0)");
  }
  long long t1 = clock64();
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][3];
  for (int i = 0; i < 8; ++i) s += p[i][0] + x[i];
  for (int i = 0; i < 4; ++i) s += q[i][0];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}

static float* d;
static long long* t;
template <int KIND, int F, int NACC>
void run(const char* what) {
  const int iters = 500;
  hipLaunchKernelGGL((stream<KIND, F, NACC>), dim3(256), dim3(256), 0, 0, d, t, iters, 1.f);
  (void)hipDeviceSynchronize();
  long long h;
  (void)hipMemcpy(&h, t, sizeof h, hipMemcpyDeviceToHost);
  printf("%-28s x%2d per 8 MFMAs, %d accumulators: %6.1f cycles per group of 8 (256 = matrix pipe only)\n", what, F, NACC,
         (double)h / (8.0 * iters));
}

int main() {
  (void)hipMalloc(&d, 256 * 256 * 4);
  (void)hipMalloc(&t, 16);
  run<0, 0, 1>("none");
  run<0, 0, 2>("none");
  run<0, 0, 4>("none");
  run<0, 0, 8>("none");
  run<1, 4, 2>("v_pk_add_f32");
  run<1, 8, 2>("v_pk_add_f32");
  run<1, 8, 8>("v_pk_add_f32");
  run<1, 16, 2>("v_pk_add_f32");
  run<2, 8, 2>("v_add_f32");
  run<2, 16, 2>("v_add_f32");
  run<2, 16, 8>("v_add_f32");
  run<3, 2, 2>("ds_read_b128");
  run<3, 4, 2>("ds_read_b128");
  run<4, 8, 2>("8 pk + 2 ds_read (kernel)");
  run<4, 8, 4>("8 pk + 2 ds_read (kernel)");
  run<4, 8, 8>("8 pk + 2 ds_read (kernel)");
  return 0;
}
