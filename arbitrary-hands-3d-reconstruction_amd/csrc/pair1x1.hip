// Two chained 1x1 convolutions of HRNet's layer1 in one kernel (acr/model.py:519-539, Bottleneck): the END of block i,
//     y = relu(W3 t2 + b3 + x)          64 -> 256 channels + residual          (conv3 / bn3 / += residual / relu)
// and the START of block i + 1,
//     t = relu(W1 y + b1)               256 -> 64 channels                     (conv1 / bn1 / relu)
// y is written once (it is the next block's residual) and never read back by this pair: as two launches the second
// re-reads the 1.07 GB (batch 64) the first has just written - both are HBM-bound (4.7 TB/s of minimum traffic).
//
// No LDS round trip between the two GEMMs.  Both run as D[cout][pixel] on v_mfma_f32_32x32x2_f32 (A = weights, B =
// activations): accumulator register j of a 32x32 tile holds, for the lane's pixel, cout row 8 (j / 4) + j % 4 in lanes
// 0-31 and that row + 4 in lanes 32-63 - which is exactly the shape of a B operand of one K = 2 step (lanes 0-31 supply
// k0, lanes 32-63 k1).  So after bias + residual + ReLU in registers, register j of the first GEMM's tile IS the B
// operand of step j of the second GEMM over that 32-channel chunk, with the second weight matrix packed in that k order
// (packer.pack_pair1x1).  Both weight matrices (2 x 64 KiB) live in LDS for the whole launch.
//
// Work item = 256 consecutive pixels of the flattened [B*H*W] map (a 1x1 convolution has no spatial structure): 8 waves x
// one pixel tile of 32 - two waves per SIMD, so that one wave's epilogue (accumulator reads, adds, stores) and LDS waits
// sit under the other's MFMAs.  Measured at batch 64: 0.736 ms per launch (the two launches it replaces: 0.515 + 0.390;
// 4 waves x 2 tiles: 0.82; 12 waves: 0.73; with nontemporal loads, which defeat the L1 reuse of a lane's 16-byte pieces of a
// line: 1.76).  What is left above the 0.57 ms its traffic costs at the 4.7 TB/s of the 1x1 kernels: the accumulator layout
// makes every residual load and y store a set of 64 separate 16-byte pieces (two lanes per 32 bytes); 128-byte segments
// would need two more LDS transposes per chunk and the LDS is full of weights.  Per 32-channel chunk c of y (8 chunks): 2 x 32 MFMAs (W3 chunk x t2), epilogue in registers
// (residual quads requested before the MFMAs), 16-byte stores of y, 2 x 2 x 16 MFMAs into the two 32-cout accumulators of
// t.  The lane's 32 t2 channels per pixel tile (k order: lane half h holds channels 32 h .. 32 h + 31) stay in registers
// for all chunks; the next item's are requested while this one computes.
#include "conv_frame.h"

namespace acrmi {

constexpr int PR_C1 = 64, PR_C2 = 256, PR_C3 = 64;      // t2 channels, y channels, t channels
constexpr int PR_PIX = 256;                              // pixels per work item
constexpr int PR_MT = 1, PR_WAVES = PR_PIX / (32 * PR_MT);  // pixel tiles per wave, waves per workgroup
constexpr int PR_LDS_FLOATS = PR_C2 * PR_C1 + PR_C3 * PR_C2 + PR_C2 + PR_C3;

__global__ __launch_bounds__(PR_WAVES * 64, 1) void pair1x1_kernel(const float* __restrict__ t2, int t2_cs, int t2_coff,
                                                         const float* __restrict__ x, int x_cs, int x_coff,
                                                         float* __restrict__ y, int y_cs, int y_coff,
                                                         float* __restrict__ t, int t_cs, int t_coff,
                                                         const float* __restrict__ wpk, long npix) {
  extern __shared__ f32x4 smem4[];
  // LDS: A1 [c 8][s4 8][lane 64][4] | A2 [c 8][nt 2][j4 4][lane 64][4] | b3 [c 8][g 4][h 2][4] | b1 [nt 2][g 4][h 2][4]
  const f32x4* a1 = smem4;
  const f32x4* a2 = smem4 + PR_C2 * PR_C1 / 4;
  const f32x4* b3 = a2 + PR_C3 * PR_C2 / 4;
  const f32x4* b1 = b3 + PR_C2 / 4;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  for (int i = tid; i < PR_LDS_FLOATS / 4; i += PR_WAVES * 64) smem4[i] = reinterpret_cast<const f32x4*>(wpk)[i];

  const long items = (npix + PR_PIX - 1) / PR_PIX;
  // this lane's pixel(s) of an item: tile p of wave w = pixels 32 PR_MT w + 32 p + li (clamped at the end of the map; the
  // stores of clamped lanes are masked)
  auto pix_of = [&](long item, int p) -> long { return item * PR_PIX + 32 * PR_MT * wave + 32 * p + li; };
  f32x4 bq[PR_MT][8], bn[PR_MT][8];      // t2 fragments of this / the next item: [pixel tile][8 x 4 channels of the lane's half]
  auto request_t2 = [&](long item, f32x4 (&dst)[PR_MT][8]) {
#pragma unroll
    for (int p = 0; p < PR_MT; ++p) {
      long px = pix_of(item, p);
      px = px < npix ? px : npix - 1;
      const f32x4* src = reinterpret_cast<const f32x4*>(t2 + (size_t)px * t2_cs + t2_coff + 32 * lh);
#pragma unroll
      for (int q = 0; q < 8; ++q) dst[p][q] = src[q];      // (cached: the 8 quads of a lane share a 128-byte line)
    }
  };
  long item = blockIdx.x;
  if (item < items) request_t2(item, bq);
  __syncthreads();      // the weights are in LDS
  for (; item < items; item += gridDim.x) {
    const long nitem = item + gridDim.x;
    if (nitem < items) request_t2(nitem, bn);
    long px[PR_MT];
    bool ok[PR_MT];
#pragma unroll
    for (int p = 0; p < PR_MT; ++p) {
      px[p] = pix_of(item, p);
      ok[p] = px[p] < npix;
      px[p] = ok[p] ? px[p] : npix - 1;
    }
    f32x16 acc2[PR_MT][2];      // [pixel tile][n-tile of t]
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int c = 0; c < 8; ++c) {
      // the residual quads of this chunk: x[pixel][32 c + 8 g + 4 lh .. + 3]
      f32x4 rv[PR_MT][4];
#pragma unroll
      for (int p = 0; p < PR_MT; ++p) {
        const float* xr = x + (size_t)px[p] * x_cs + x_coff + 32 * c + 4 * lh;
#pragma unroll
        for (int g = 0; g < 4; ++g) rv[p][g] = *reinterpret_cast<const f32x4*>(xr + 8 * g);
      }
      // ---- GEMM 1: y chunk c = W3[32 c .. +31][:] t2
      f32x16 acc1[PR_MT];
#pragma unroll
      for (int s4 = 0; s4 < 8; ++s4) {
        const f32x4 af = a1[(c * 8 + s4) * 64 + lane];
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int p = 0; p < PR_MT; ++p) {
            if (s4 == 0 && e == 0) acc1[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[e], bq[p][s4][e], zero, 0, 0, 0);
            else acc1[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[e], bq[p][s4][e], acc1[p], 0, 0, 0);
          }
      }
      // ---- epilogue 1 in registers: + b3 + x, ReLU; y out; the same registers are GEMM 2's B operands
      f32x4 yv[PR_MT][4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 bv = b3[(c * 4 + g) * 2 + lh];
#pragma unroll
        for (int p = 0; p < PR_MT; ++p) {
          f32x4 o = f32x4{acc1[p][4 * g], acc1[p][4 * g + 1], acc1[p][4 * g + 2], acc1[p][4 * g + 3]} + bv + rv[p][g];
          o[0] = fmaxf(o[0], 0.f); o[1] = fmaxf(o[1], 0.f); o[2] = fmaxf(o[2], 0.f); o[3] = fmaxf(o[3], 0.f);
          yv[p][g] = o;
          if (ok[p])
            *reinterpret_cast<f32x4*>(y + (size_t)px[p] * y_cs + y_coff + 32 * c + 8 * g + 4 * lh) = o;   // (16-byte pieces: L2 merges the line)
        }
      }
      // ---- GEMM 2: t += W1[:, 32 c .. +31] y chunk (step j = 4 g + e: channels 32 c + 8 g + e (+ 4 in the upper lanes))
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 af = a2[((c * 2 + nt) * 4 + g) * 64 + lane];
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int p = 0; p < PR_MT; ++p) {
              if (c == 0 && g == 0 && e == 0) acc2[p][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[e], yv[p][g][e], zero, 0, 0, 0);
              else acc2[p][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[e], yv[p][g][e], acc2[p][nt], 0, 0, 0);
            }
        }
    }
    // ---- epilogue 2: t = relu(acc2 + b1)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 bv = b1[(nt * 4 + g) * 2 + lh];
#pragma unroll
        for (int p = 0; p < PR_MT; ++p) {
          f32x4 o = f32x4{acc2[p][nt][4 * g], acc2[p][nt][4 * g + 1], acc2[p][nt][4 * g + 2], acc2[p][nt][4 * g + 3]} + bv;
          o[0] = fmaxf(o[0], 0.f); o[1] = fmaxf(o[1], 0.f); o[2] = fmaxf(o[2], 0.f); o[3] = fmaxf(o[3], 0.f);
          if (ok[p])
            *reinterpret_cast<f32x4*>(t + (size_t)px[p] * t_cs + t_coff + 32 * nt + 8 * g + 4 * lh) = o;
        }
      }
#pragma unroll
    for (int p = 0; p < PR_MT; ++p)
#pragma unroll
      for (int q = 0; q < 8; ++q) bq[p][q] = bn[p][q];
  }
}

hipError_t launch_pair1x1(const float* t2, int t2_cs, int t2_coff, const float* x, int x_cs, int x_coff, float* y, int y_cs,
                          int y_coff, float* t, int t_cs, int t_coff, const float* wpk, long npix, hipStream_t s) {
  if (!t2 || !x || !y || !t || !wpk || npix <= 0 || t2_cs % 4 || x_cs % 4 || y_cs % 4 || t_cs % 4 || t2_coff % 4 || x_coff % 4 ||
      y_coff % 4 || t_coff % 4)
    return hipErrorInvalidValue;
  constexpr size_t lds = (size_t)PR_LDS_FLOATS * sizeof(float);
  static_assert(lds <= 160 * 1024, "both weight matrices must fit the LDS");
  std::lock_guard<std::recursive_mutex> lock(launch_mutex());
  static unsigned char init[MAX_DEVICES] = {};
  if (first_use_on_device(init)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(pair1x1_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    if ((e = conv_ensure_device_info()) != hipSuccess) return e;
  }
  const long items = (npix + PR_PIX - 1) / PR_PIX;
  const long cus = conv_num_cus();
  const unsigned grid = (unsigned)(items < cus ? items : cus);
  hipLaunchKernelGGL(pair1x1_kernel, dim3(grid), dim3(PR_WAVES * 64), lds, s, t2, t2_cs, t2_coff, x, x_cs, x_coff, y, y_cs, y_coff, t, t_cs,
                     t_coff, wpk, npix);
  return hipGetLastError();
}

}  // namespace acrmi
