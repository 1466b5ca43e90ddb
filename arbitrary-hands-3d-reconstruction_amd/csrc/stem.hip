// Stem: uint8 RGB frame -> relu(bn1(conv1(x / 255 * 2 - 1))), the first convolution of the backbone read straight from
// the uint8 image (acr/model.py:832 normalisation + acr/model.py:589-603 conv1/bn1/relu: 3 -> 64 channels, 3x3, stride 2,
// pad 1).  Replaces u8norm_kernel + the generic direct kernel, which pads K = 27 to 9 taps x 8 channels (72) and
// writes / re-reads a 4-channel fp32 copy of the image (0.675 ms together at batch 64, 23 TF-equivalent).
//
// GEMM view: D[cout 64][pixel] = W[cout][k] x X[k][pixel], k = (ky*3 + kx)*3 + c < 27, padded to 28 = 14 steps of
// v_mfma_f32_32x32x2_f32 per 32-cout tile.  A workgroup (4 waves) walks a strip of 4 tiles of 8x16 output pixels:
//  * per tile the 17 x 33 x 3 uint8 halo patch is turned into fp32 in LDS through a 256-entry table of
//    (q / 255) * 2 - 1 (the exact expression of u8norm_kernel, evaluated once per workgroup); pixels outside the
//    image are 0 AFTER normalisation (the conv pads the normalised map);
//  * a wave owns 2 output rows x 16 columns = 32 pixels (MFMA columns) x 64 couts; lane (pixel, k parity) reads its
//    14 patch values with ds_read_b32 (k -> patch offset is a per-lane constant), the 28 weight fragments stay in
//    registers for the whole strip;
//  * epilogue as in conv_ws2_kernel: through a wave-private LDS tile [32 pixels][36] so that 8 lanes write one
//    pixel's 128-byte half line (bias, ReLU, non-temporal dwordx4 stores).
// HBM-bound by construction: 256 bytes written per output pixel, 3 read.
#include "kernels.h"

namespace acrmi {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int ST_TH = 8, ST_TW = 16, ST_STRIP = 4;
constexpr int ST_PR = 2 * ST_TH + 1;                             // 17 rows x 33 columns of input pixels per tile
constexpr int ST_ROW = 100;                                       // floats per patch row: 33 * 3 + 1 zero slot
constexpr int ST_PSTR = 36;

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// DT = ACRMI_DT_*: 0 fp32 output; 1 / 2: the same fp32 result rounded once (nearest even) to f16 / bf16 - `out` then
// points at 16-bit elements and out_cs / out_coff count them (16-bit programs, include/acrmi.h)
template <int DT>
__global__ __launch_bounds__(256) void stem_kernel(const uint8_t* __restrict__ img, int H, int W,
                                                   const float* __restrict__ wpk, const float* __restrict__ bias,
                                                   float* __restrict__ out, int out_cs, int out_coff, int relu) {
  __shared__ float lut[256];
  __shared__ float patch[2][ST_PR * ST_ROW];
  __shared__ float epi[4][32 * ST_PSTR];
  const int Ho = H / 2, Wo = W / 2;
  const int strips_x = Wo / (ST_TW * ST_STRIP), tiles_y = Ho / ST_TH;
  int bid = blockIdx.x;
  const int strip = bid % strips_x;
  bid /= strips_x;
  const int ty = bid % tiles_y, b = bid / tiles_y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int p8 = lane >> 3, q8 = lane & 7;

  lut[tid] = ((float)tid / 255.f) * 2.0f - 1.0f;
  // weight fragments: lane (cout row li, k parity lh) of step s, n-tile n
  float wf[14][2];
#pragma unroll
  for (int s = 0; s < 14; ++s)
#pragma unroll
    for (int n = 0; n < 2; ++n) wf[s][n] = wpk[(s * 2 + n) * 64 + lane];
  // store side: fp32 - lane = (pixel p8 of 8, cout quad q8), 4 rounds per 32-pixel tile; 16-bit - lane = (pixel of 16,
  // cout octet), 2 rounds
  const int ps = DT == 0 ? p8 : lane >> 2, qs = DT == 0 ? q8 : lane & 3;
  f32x4 bv[2][2];
#pragma unroll
  for (int n = 0; n < 2; ++n) {
    if (DT == 0) {
      bv[n][0] = *reinterpret_cast<const f32x4*>(bias + n * 32 + 4 * q8);
      bv[n][1] = bv[n][0];
    } else {
      bv[n][0] = *reinterpret_cast<const f32x4*>(bias + n * 32 + 8 * qs);
      bv[n][1] = *reinterpret_cast<const f32x4*>(bias + n * 32 + 8 * qs + 4);
    }
  }
  // patch offset of k = 2 s + lh for this lane's pixel (row 2 (2 wave + li / 16) + ky, column (2 (li % 16) + kx) * 3 + c);
  // k = 27 (lh = 1, s = 13) points at the row's zero slot
  int koff[14];
  {
    const int prow = 2 * (2 * wave + (li >> 4)), pcol = 2 * (li & 15) * 3;
#pragma unroll
    for (int s = 0; s < 14; ++s) {
      const int k = 2 * s + lh;
      koff[s] = k < 27 ? (prow + k / 9) * ST_ROW + pcol + k % 9 : prow * ST_ROW + (ST_ROW - 1);
    }
  }
  const uint8_t* frame = img + (size_t)b * H * W * 3;
  float* outb = out + (size_t)b * Ho * Wo * out_cs + out_coff;      // (fp32 output)
  unsigned short* outh = reinterpret_cast<unsigned short*>(out) + (size_t)b * Ho * Wo * out_cs + out_coff;   // (16-bit)
  const int iy0 = 2 * ty * ST_TH - 1;
  // the patch of strip tile t -> LDS buffer t & 1: element e = r * 99 + j (j = column * 3 + channel).  The bytes are
  // requested before tile t - 1 is computed and converted / written behind it (one HBM latency per tile otherwise)
  constexpr int NE = (ST_PR * 99 + 255) / 256;
  int qv[NE];
  auto request = [&](int t) {
    const int ix0 = 2 * (strip * ST_STRIP + t) * ST_TW - 1;
#pragma unroll
    for (int i = 0; i < NE; ++i) {
      const int e = tid + 256 * i;
      const int r = e / 99, j = e - r * 99, cc = j / 3;
      const int iy = iy0 + r, ix = ix0 + cc;
      const bool ok = e < ST_PR * 99 && iy >= 0 && iy < H && ix >= 0 && ix < W;
      qv[i] = ok ? frame[((size_t)iy * W + ix) * 3 + (j - cc * 3)] : -1;
    }
  };
  auto stage = [&](int t) {
    float* dst = patch[t & 1];
#pragma unroll
    for (int i = 0; i < NE; ++i) {
      const int e = tid + 256 * i;
      const int r = e / 99, j = e - r * 99;
      if (e < ST_PR * 99) dst[r * ST_ROW + j] = qv[i] >= 0 ? lut[qv[i]] : 0.f;
    }
    if (tid < ST_PR) dst[tid * ST_ROW + ST_ROW - 1] = 0.f;
  };
  request(0);
  __syncthreads();   // the table is complete
  stage(0);
  for (int t = 0; t < ST_STRIP; ++t) {
    __syncthreads();   // patch t is complete; every wave has left patch t - 1
    if (t + 1 < ST_STRIP) request(t + 1);
    const float* src = patch[t & 1];
    f32x16 acc[2];
#pragma unroll
    for (int s = 0; s < 14; ++s) {
      const float x = src[koff[s]];
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        if (s == 0) {
          const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[s][n], x, zero, 0, 0, 0);
        } else {
          acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[s][n], x, acc[n], 0, 0, 0);
        }
      }
    }
    // D[cout][pixel]: lane (li, lh) holds pixel li, couts 8q+4lh..+3 in register quad q
    float* ep = epi[wave];
    const int tx0 = (strip * ST_STRIP + t) * ST_TW, oy0 = ty * ST_TH + 2 * wave;
#pragma unroll
    for (int n = 0; n < 2; ++n) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<f32x4*>(ep + li * ST_PSTR + 8 * q + 4 * lh) =
            f32x4{acc[n][4 * q], acc[n][4 * q + 1], acc[n][4 * q + 2], acc[n][4 * q + 3]};
      if constexpr (DT == 0) {
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const int p = 8 * gq + p8;                       // pixel of the wave's 2 x 16 block
          f32x4 o4 = *reinterpret_cast<const f32x4*>(ep + p * ST_PSTR + 4 * q8) + bv[n][0];
          if (relu) {
            o4[0] = fmaxf(o4[0], 0.f); o4[1] = fmaxf(o4[1], 0.f); o4[2] = fmaxf(o4[2], 0.f); o4[3] = fmaxf(o4[3], 0.f);
          }
          const int oy = oy0 + (p >> 4), ox = tx0 + (p & 15);
          __builtin_nontemporal_store(o4, reinterpret_cast<f32x4*>(outb + ((size_t)oy * Wo + ox) * out_cs + n * 32 + 4 * q8));
        }
      } else {
#pragma unroll
        for (int gq = 0; gq < 2; ++gq) {
          const int p = 16 * gq + ps;
          const f32x4 lo = *reinterpret_cast<const f32x4*>(ep + p * ST_PSTR + 8 * qs) + bv[n][0];
          const f32x4 hi = *reinterpret_cast<const f32x4*>(ep + p * ST_PSTR + 8 * qs + 4) + bv[n][1];
          float o8[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          if (relu) {
#pragma unroll
            for (int e = 0; e < 8; ++e) o8[e] = fmaxf(o8[e], 0.f);
          }
          f32x4 pk;
          if constexpr (DT == 2) {
            bf16x8 h;
#pragma unroll
            for (int e = 0; e < 8; ++e) h[e] = (__bf16)o8[e];
            pk = __builtin_bit_cast(f32x4, h);
          } else {
            f16x8 h;
#pragma unroll
            for (int e = 0; e < 8; ++e) h[e] = (_Float16)o8[e];
            pk = __builtin_bit_cast(f32x4, h);
          }
          const int oy = oy0 + (p >> 4), ox = tx0 + (p & 15);
          __builtin_nontemporal_store(pk, reinterpret_cast<f32x4*>(outh + ((size_t)oy * Wo + ox) * out_cs + n * 32 + 8 * qs));
        }
      }
    }
    if (t + 1 < ST_STRIP) stage(t + 1);
  }
}

bool stem_shape_ok(int H, int W, int out_cs, int out_coff) {
  return H > 0 && W > 0 && H % (2 * ST_TH) == 0 && W % (2 * ST_TW * ST_STRIP) == 0 && out_cs % 4 == 0 && out_coff % 4 == 0 &&
         out_coff + 64 <= out_cs;
}

hipError_t launch_stem(const uint8_t* img, int B, int H, int W, const float* wpk, const float* bias, float* out,
                       int out_cs, int out_coff, int relu, hipStream_t s) {
  if (!stem_shape_ok(H, W, out_cs, out_coff) || B <= 0) return hipErrorInvalidValue;
  const long grid = (long)B * (H / 2 / ST_TH) * (W / 2 / (ST_TW * ST_STRIP));
  if (grid > 0x7fffffffL) return hipErrorInvalidValue;
  hipLaunchKernelGGL(stem_kernel<0>, dim3((unsigned)grid), dim3(256), 0, s, img, H, W, wpk, bias, out, out_cs, out_coff, relu);
  return hipGetLastError();
}

hipError_t launch_stem_h16(const uint8_t* img, int B, int H, int W, const float* wpk, const float* bias, void* out,
                           int out_cs, int out_coff, int relu, int dtype, hipStream_t s) {
  // (16-byte vectors of 8 elements: strides / offsets are multiples of 8)
  if (!stem_shape_ok(H, W, out_cs, out_coff) || out_cs % 8 || out_coff % 8 || B <= 0 || (dtype != 1 && dtype != 2))
    return hipErrorInvalidValue;
  const long grid = (long)B * (H / 2 / ST_TH) * (W / 2 / (ST_TW * ST_STRIP));
  if (grid > 0x7fffffffL) return hipErrorInvalidValue;
  float* o = reinterpret_cast<float*>(out);
  if (dtype == 2)
    hipLaunchKernelGGL(stem_kernel<2>, dim3((unsigned)grid), dim3(256), 0, s, img, H, W, wpk, bias, o, out_cs, out_coff, relu);
  else
    hipLaunchKernelGGL(stem_kernel<1>, dim3((unsigned)grid), dim3(256), 0, s, img, H, W, wpk, bias, o, out_cs, out_coff, relu);
  return hipGetLastError();
}

}  // namespace acrmi
