// Does an anti-phase second wave set pay on gfx950 fp32 MFMA streams?  (VERDICT r1 item 4 option i)
//   hipcc --offload-arch=gfx950 -O3 tools/pingpong_mfma.hip -o /tmp/ppm && /tmp/ppm
// A Winograd F(2x2,3x3) work item of the 32->32 layers is two phases per compute wave:
//   P1 "matrix":  4 steps x (8 ds_read_b128 + 16 v_pk + 4 weight loads + 16 MFMA 32x32x2) + x-fold (64 VALU) + park
//                 (8 ds_write_b128)
//   P2 "memory":  y-fold (12 ds_read_b128 + 16 VALU), bias/residual/ReLU (48 VALU, 4 residual loads), 4 x 16-byte
//                 stores, next item's first window rows (8 ds_read_b128 + 16 v_pk)
// separated by workgroup barriers.  "solo" kernels: 4 waves (one per SIMD) running P1 / P2 alone (today's kernel is
// P1 + P2 back to back).  MODE 0: 8 waves (two per SIMD), both sets in phase.  MODE 1: set A runs P1 while set B runs
// P2 and vice versa.  Reported: cycles per item; two waves per SIMD process two items per round.
// MODE 2: like MODE 1, but the partner only runs the memory instructions of P2 (no VALU) - separates VALU blocking
// from LDS/VMEM contention.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct Regs {
  f32x16 acc[4];
  f32x4 d[8], v[4], bq[4], rv[4];
  f32x2 one;
};

__device__ __forceinline__ void p1(Regs& r, unsigned lds_addr, const f32x4* wp, bool park) {
#pragma unroll
  for (int s = 0; s < 4; ++s) {
#pragma unroll
    for (int px = 0; px < 4; ++px) {
      if (px == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
          asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r.d[i]) : "v"(lds_addr), "i"(i * 1024));
      }
      if (px == 2) {
        asm volatile("s_waitcnt lgkmcnt(0)");
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          f32x2 lo = {r.d[i][0], r.d[i][1]}, hi = {r.d[i][2], r.d[i][3]};
          const f32x2 lo2 = {r.d[4 + i][0], r.d[4 + i][1]}, hi2 = {r.d[4 + i][2], r.d[4 + i][3]};
          asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(lo) : "v"(r.one), "v"(lo2));
          asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(hi) : "v"(r.one), "v"(hi2));
          r.d[i] = f32x4{lo[0], lo[1], hi[0], hi[1]};
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
        asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(r.acc[px]) : "v"(r.bq[px][j]), "v"(r.v[px][j]));
      if (px >= 2) {
#pragma unroll
        for (int i = 0; i < (px == 2 ? 3 : 1); ++i) {
          const int k = px == 2 ? i : 3;
          f32x2 lo = {r.d[k][0], r.d[k][1]}, hi = {r.d[k][2], r.d[k][3]};
          asm volatile("v_pk_fma_f32 %0, %1, %0, %0" : "+v"(lo) : "v"(r.one));
          asm volatile("v_pk_fma_f32 %0, %1, %0, %0" : "+v"(hi) : "v"(r.one));
          r.v[k] = f32x4{lo[0], lo[1], hi[0], hi[1]};
        }
      }
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r.bq[px]) : "v"(wp + px * 64));
    }
  }
  if (park) {   // x-fold (64 VALU) + park
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      asm volatile("v_add_f32 %0, %0, %1" : "+v"(r.acc[0][e]) : "v"(r.acc[1][e]));
      asm volatile("v_add_f32 %0, %0, %1" : "+v"(r.acc[0][e]) : "v"(r.acc[2][e]));
      asm volatile("v_add_f32 %0, %0, %1" : "+v"(r.acc[3][e]) : "v"(r.acc[2][e]));
      asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r.acc[3][e]) : "v"(r.one[0]), "v"(r.acc[1][e]));
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 a = {r.acc[0][4 * q], r.acc[0][4 * q + 1], r.acc[0][4 * q + 2], r.acc[0][4 * q + 3]};
      const f32x4 b = {r.acc[3][4 * q], r.acc[3][4 * q + 1], r.acc[3][4 * q + 2], r.acc[3][4 * q + 3]};
      asm volatile("ds_write_b128 %0, %1 offset:%2" : : "v"(lds_addr), "v"(a), "i"(16384 + q * 2048) : "memory");
      asm volatile("ds_write_b128 %0, %1 offset:%2" : : "v"(lds_addr), "v"(b), "i"(16384 + 1024 + q * 2048) : "memory");
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
}

template <bool VALU>
__device__ __forceinline__ void p2(Regs& r, unsigned lds_addr, const f32x4* resp, f32x4* outp) {
  f32x4 pp[12];
#pragma unroll
  for (int i = 0; i < 4; ++i) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r.rv[i]) : "v"(resp + i * 64));
#pragma unroll
  for (int i = 0; i < 12; ++i)
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(pp[i]) : "v"(lds_addr), "i"(16384 + i * 1024));
  asm volatile("s_waitcnt lgkmcnt(0)");
  if (VALU) {
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(pp[g][e]) : "v"(pp[4 + g][e]));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(pp[g][e]) : "v"(r.one[0]), "v"(pp[8 + g][e]));
      }
  }
  asm volatile("s_waitcnt vmcnt(0)");
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    if (VALU) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(pp[g][e]) : "v"(r.rv[g][e]));
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(pp[g][e]) : "v"(r.one[1]));
        asm volatile("v_max_f32 %0, %0, %1" : "+v"(pp[g][e]) : "v"(r.one[1]));
      }
    }
    asm volatile("global_store_dwordx4 %0, %1, off nt" : : "v"(outp + g * 64), "v"(pp[g]) : "memory");
  }
  // next item's first window rows + transform
#pragma unroll
  for (int i = 0; i < 8; ++i)
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r.d[i]) : "v"(lds_addr), "i"(i * 1024));
  asm volatile("s_waitcnt lgkmcnt(0)");
  if (VALU) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f32x2 lo = {r.d[i][0], r.d[i][1]}, hi = {r.d[i][2], r.d[i][3]};
      const f32x2 lo2 = {r.d[4 + i][0], r.d[4 + i][1]}, hi2 = {r.d[4 + i][2], r.d[4 + i][3]};
      asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(lo) : "v"(r.one), "v"(lo2));
      asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(hi) : "v"(r.one), "v"(hi2));
      asm volatile("v_pk_fma_f32 %0, %1, %0, %0" : "+v"(lo) : "v"(r.one));
      asm volatile("v_pk_fma_f32 %0, %1, %0, %0" : "+v"(hi) : "v"(r.one));
      r.v[i] = f32x4{lo[0], lo[1], hi[0], hi[1]};
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
}

template <int MODE, int PRIO>
__global__ __launch_bounds__(512, 1) void pingpong(float* out, long long* ticks, const f32x4* src, f32x4* dst, int items, float a0) {
  extern __shared__ f32x4 lds4[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < 2560; i += blockDim.x) lds4[i] = f32x4{a0, a0, a0, a0};
  Regs r;
  for (int i = 0; i < 4; ++i) {
    for (int e = 0; e < 16; ++e) r.acc[i][e] = 0.f;
    r.v[i] = f32x4{a0, a0, a0, a0};
    r.bq[i] = f32x4{a0, 1.f, a0, 1.f};
    r.rv[i] = r.v[i];
  }
  for (int i = 0; i < 8; ++i) r.d[i] = f32x4{a0, a0, a0, a0};
  r.one = f32x2{1.f, 1.f};
  asm volatile("" : "+v"(r.one));
  const unsigned lds_addr = (unsigned)(lane * 16);
  const f32x4* wp = src + (blockIdx.x & 7) * 4096 + lane;
  const f32x4* resp = src + 65536 + ((size_t)blockIdx.x * 8 + wave) * 256 + lane;
  f32x4* outp = dst + ((size_t)blockIdx.x * 8 + wave) * 256 + lane;
  const bool setB = wave >= 4;
  if (MODE != 0 && setB && PRIO) __builtin_amdgcn_s_setprio(PRIO);
  __syncthreads();
  const long long t0 = clock64();
  if (MODE == 0) {
    for (int it = 0; it < items; ++it) {
      p1(r, lds_addr, wp, true);
      __syncthreads();
      p2<true>(r, lds_addr, resp, outp);
      __syncthreads();
    }
  } else {
    for (int it = 0; it < items; ++it) {   // each set processes `items` items
      if (!setB) p1(r, lds_addr, wp, true); else if (MODE == 1) p2<true>(r, lds_addr, resp, outp); else p2<false>(r, lds_addr, resp, outp);
      __syncthreads();
      if (setB) p1(r, lds_addr, wp, true); else if (MODE == 1) p2<true>(r, lds_addr, resp, outp); else p2<false>(r, lds_addr, resp, outp);
      __syncthreads();
    }
  }
  const long long t1 = clock64();
  float s = 0;
  for (int i = 0; i < 4; ++i)
    for (int e = 0; e < 16; ++e) s += r.acc[i][e];
  for (int i = 0; i < 4; ++i) s += r.v[i][0] + r.d[i][0];
  out[blockIdx.x * 512 + tid] = s;
  if (tid == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}

// P1 only / P2 only, 4 waves: the phase lengths on their own
template <int WHICH>
__global__ __launch_bounds__(256, 1) void solo(float* out, long long* ticks, const f32x4* src, f32x4* dst, int items, float a0) {
  extern __shared__ f32x4 lds4[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 2560; i += blockDim.x) lds4[i] = f32x4{a0, a0, a0, a0};
  Regs r;
  for (int i = 0; i < 4; ++i) {
    for (int e = 0; e < 16; ++e) r.acc[i][e] = 0.f;
    r.v[i] = f32x4{a0, a0, a0, a0};
    r.bq[i] = f32x4{a0, 1.f, a0, 1.f};
    r.rv[i] = r.v[i];
  }
  for (int i = 0; i < 8; ++i) r.d[i] = f32x4{a0, a0, a0, a0};
  r.one = f32x2{1.f, 1.f};
  asm volatile("" : "+v"(r.one));
  const unsigned lds_addr = (unsigned)(lane * 16);
  const f32x4* wp = src + (blockIdx.x & 7) * 4096 + lane;
  const f32x4* resp = src + 65536 + ((size_t)blockIdx.x * 8 + wave) * 256 + lane;
  f32x4* outp = dst + ((size_t)blockIdx.x * 8 + wave) * 256 + lane;
  __syncthreads();
  const long long t0 = clock64();
  for (int it = 0; it < items; ++it) {
    if (WHICH == 0) p1(r, lds_addr, wp, false);
    if (WHICH == 1) p1(r, lds_addr, wp, true);
    if (WHICH == 2) p2<true>(r, lds_addr, resp, outp);
    __syncthreads();
  }
  const long long t1 = clock64();
  float s = 0;
  for (int i = 0; i < 4; ++i)
    for (int e = 0; e < 16; ++e) s += r.acc[i][e];
  for (int i = 0; i < 4; ++i) s += r.v[i][0] + r.d[i][0];
  out[blockIdx.x * 512 + tid] = s;
  if (tid == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}

int main() {
  float* d;
  long long* t;
  f32x4 *src, *dst;
  (void)hipMalloc(&d, 256 * 512 * 4);
  (void)hipMalloc(&t, 16);
  (void)hipMalloc(&src, (65536 + 256 * 8 * 256) * 16);
  (void)hipMalloc(&dst, 256 * 8 * 256 * 16);
  (void)hipMemset(src, 0, (65536 + 256 * 8 * 256) * 16);
  const int items = 200, lds = 2560 * 16 * 2;
  long long h[2];
  auto report = [&](const char* name, double per) {
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h, t, sizeof h, hipMemcpyDeviceToHost);
    printf("%-58s %8.0f cycles per item (4096 = MFMA only)\n", name, (double)h[0] / items / per);
  };
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL((solo<0>), dim3(256), dim3(256), lds, 0, d, t, src, dst, items, 1.f);
    report("P1 without x-fold/park, 4 waves", 1);
    hipLaunchKernelGGL((solo<1>), dim3(256), dim3(256), lds, 0, d, t, src, dst, items, 1.f);
    report("P1 with x-fold/park, 4 waves", 1);
    hipLaunchKernelGGL((solo<2>), dim3(256), dim3(256), lds, 0, d, t, src, dst, items, 1.f);
    report("P2 alone, 4 waves", 1);
    hipLaunchKernelGGL((pingpong<0, 0>), dim3(256), dim3(512), lds, 0, d, t, src, dst, items, 1.f);
    report("MODE 0: both sets IN phase (P1|P2 together), per item", 2);
    hipLaunchKernelGGL((pingpong<1, 0>), dim3(256), dim3(512), lds, 0, d, t, src, dst, items, 1.f);
    report("MODE 1: anti-phase sets, per item (2 items per round)", 2);
    hipLaunchKernelGGL((pingpong<1, 1>), dim3(256), dim3(512), lds, 0, d, t, src, dst, items, 1.f);
    report("MODE 1 + s_setprio 1 on set B", 2);
    hipLaunchKernelGGL((pingpong<1, 3>), dim3(256), dim3(512), lds, 0, d, t, src, dst, items, 1.f);
    report("MODE 1 + s_setprio 3 on set B", 2);
    hipLaunchKernelGGL((pingpong<2, 0>), dim3(256), dim3(512), lds, 0, d, t, src, dst, items, 1.f);
    report("MODE 2: partner without VALU, per item", 2);
  }
  return 0;
}
