// Issue-slot calibration for fp32 MFMA streams on gfx950 (v_mfma_f32_32x32x2_f32, 64 cycles per SIMD):
//   hipcc --offload-arch=gfx950 -O3 tools/valu_beside_mfma.hip -o /tmp/vbm && /tmp/vbm
// (1) partner test: workgroup = 8 waves (2 per SIMD); one half streams MFMAs, the other half runs independent
//     v_add_f32 at s_setprio 3 and reports how long that took.
// (2) filler test: workgroup = 4 waves (1 per SIMD); each wave issues F filler instructions (v_add_f32,
//     v_fma_f32 on 4 registers, ds_read_b128, or buffer-style global loads) behind every MFMA.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int OLDER_MFMA>
__global__ __launch_bounds__(512, 1) void partner(float* out, long long* ticks, int mfma_iters, int valu_iters, float a0) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool is_mfma = OLDER_MFMA ? wave < 4 : wave >= 4;
  if (is_mfma) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = a0 + threadIdx.x, b = 2.f;
    long long t0 = clock64();
    for (int it = 0; it < mfma_iters; ++it) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
    }
    long long t1 = clock64();
    float s = 0;
    for (int i = 0; i < 4; ++i)
      for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if ((threadIdx.x & 255) == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
  } else {
    __builtin_amdgcn_s_setprio(3);
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = a0 + i + threadIdx.x;
    long long t0 = clock64();
    for (int it = 0; it < valu_iters; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a0));
    }
    long long t1 = clock64();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += x[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if ((threadIdx.x & 255) == 0 && blockIdx.x == 0) ticks[1] = t1 - t0;
  }
}

// KIND 0: v_add_f32, 1: ds_read_b128, 2: global_load_dwordx4 (L2-resident 16 KiB), 3: v_fma_f32
template <int KIND, int F>
__global__ __launch_bounds__(256, 1) void filler(float* out, long long* ticks, const f32x4* src, int mfma_iters, float a0) {
  __shared__ f32x4 lds[1024];
  lds[threadIdx.x] = f32x4{a0, a0, a0, a0};
  __syncthreads();
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = a0 + threadIdx.x, b = 2.f;
  float x[8];
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  f32x2 p2[5];
  for (int i = 0; i < 5; ++i) p2[i] = f32x2{a0, a0 + i};
  f32x4 q[8];
  for (int i = 0; i < 8; ++i) { x[i] = a0 + i; q[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  const f32x4* lp = lds + (threadIdx.x & 63);
  const f32x4* gp = src + threadIdx.x;
  long long t0 = clock64();
  for (int it = 0; it < mfma_iters; ++it) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
#pragma unroll
        for (int f = 0; f < F; ++f) {
          if (KIND == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[f & 7]) : "v"(a0));
          if (KIND == 3) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[f & 7]) : "v"(a0));
          if (KIND == 4) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p2[f & 3]) : "v"(p2[4]));
          if (KIND == 5) asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(x[f & 7]) : "v"(a0));
          if (KIND == 6) asm volatile("ds_write_b128 %0, %1 offset:2048" : : "v"((unsigned)((threadIdx.x & 63) * 16)), "v"(q[f & 7]) : "memory");
          if (KIND == 1) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q[f & 7]) : "v"((unsigned)(size_t)lp * 0 + (unsigned)((threadIdx.x & 63) * 16)), "i"(1024 * (F > 0 ? 1 : 1)));
          if (KIND == 2) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(q[f & 7]) : "v"(gp));
        }
      }
    if (KIND == 1) asm volatile("s_waitcnt lgkmcnt(0)");
    if (KIND == 2) asm volatile("s_waitcnt vmcnt(0)");
  }
  long long t1 = clock64();
  float s = 0;
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  for (int i = 0; i < 8; ++i) s += x[i] + q[i][0];
  for (int i = 0; i < 4; ++i) s += p2[i][0] + p2[i][1];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}

static float* d;
static long long* t;
static f32x4* src;

template <int KIND, int F>
void run_filler(const char* kind) {
  const int iters = 2000;
  hipLaunchKernelGGL((filler<KIND, F>), dim3(256), dim3(256), 0, 0, d, t, src, iters, 1.f);
  (void)hipDeviceSynchronize();
  long long h[2];
  (void)hipMemcpy(h, t, sizeof h, hipMemcpyDeviceToHost);
  printf("filler %-22s x%2d per MFMA: %6.1f ticks/MFMA\n", kind, F, (double)h[0] / (16.0 * iters));
}

int main() {
  (void)hipMalloc(&d, 256 * 512 * 4);
  (void)hipMalloc(&t, 16);
  (void)hipMalloc(&src, 1 << 20);
  (void)hipMemset(src, 0, 1 << 20);
  long long h[2];
  hipLaunchKernelGGL((partner<1>), dim3(256), dim3(512), 0, 0, d, t, 4000, 200, 1.f);
  (void)hipDeviceSynchronize();
  (void)hipMemcpy(h, t, sizeof h, hipMemcpyDeviceToHost);
  printf("partner (MFMA waves older):   %6.1f ticks/MFMA, partner %7.1f ticks/VALU (5.8 alone)\n", h[0] / 64000.0, h[1] / 3200.0);
  hipLaunchKernelGGL((partner<0>), dim3(256), dim3(512), 0, 0, d, t, 4000, 200, 1.f);
  (void)hipDeviceSynchronize();
  (void)hipMemcpy(h, t, sizeof h, hipMemcpyDeviceToHost);
  printf("partner (MFMA waves younger): %6.1f ticks/MFMA, partner %7.1f ticks/VALU\n", h[0] / 64000.0, h[1] / 3200.0);
  run_filler<0, 0>("none");
  run_filler<0, 1>("v_add_f32");
  run_filler<0, 2>("v_add_f32");
  run_filler<0, 4>("v_add_f32");
  run_filler<0, 8>("v_add_f32");
  run_filler<0, 12>("v_add_f32");
  run_filler<3, 4>("v_fma_f32");
  run_filler<4, 1>("v_pk_add_f32");
  run_filler<4, 2>("v_pk_add_f32");
  run_filler<4, 4>("v_pk_add_f32");
  run_filler<6, 1>("ds_write_b128");
  run_filler<6, 2>("ds_write_b128");
  run_filler<6, 4>("ds_write_b128");
  run_filler<1, 1>("ds_read_b128");
  run_filler<1, 2>("ds_read_b128");
  run_filler<1, 4>("ds_read_b128");
  run_filler<2, 1>("global_load_dwordx4");
  run_filler<2, 2>("global_load_dwordx4");
  return 0;
}
