// ResNet stem: uint8 RGB frame -> relu(bn1(conv1(x / 255 * 2 - 1))) with conv1 = 7x7, stride 2, pad 3, 3 -> 64 channels
// (torchvision resnet.py conv1/bn1/relu; BASELINE.json configs[1]'s backbone - build-defined: the reference's
// `--backbone resnet50` is a dead flag, acr/config.py:95).  Same frame as stem_kernel (stem.hip): the image is read as
// uint8, normalised through a 256-entry table into an LDS patch, a wave owns 2 x 16 output pixels x 64 couts on
// v_mfma_f32_32x32x2_f32 with the reduction k = (ky*7 + kx)*3 + c < 147 padded to 148 = 74 steps; the 148 weight
// fragments stay in registers for the whole strip, the k -> patch offset is two compile-time constants per step.
#include "kernels.h"

namespace acrmi {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int S7_TH = 8, S7_TW = 16, S7_STRIP = 4;
constexpr int S7_PR = 2 * S7_TH + 5, S7_PC = 2 * S7_TW + 5;       // 21 x 37 input pixels per tile
constexpr int S7_ROW = 112;                                       // floats per patch row: 37 * 3 + 1 zero slot
constexpr int S7_RW = S7_PC * 3;                                  // 111 image floats per patch row
constexpr int S7_STEPS = 74;                                      // ceil(147 / 2)
constexpr int S7_PSTR = 36;

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// DT = ACRMI_DT_*: 0 fp32 output; 1 / 2: the same fp32 result rounded once (nearest even) to f16 / bf16 - `out` then
// points at 16-bit elements and out_cs / out_coff count them (16-bit programs, include/acrmi.h)
template <int DT>
__global__ __launch_bounds__(256) void stem7_kernel(const uint8_t* __restrict__ img, int H, int W,
                                                   const float* __restrict__ wpk, const float* __restrict__ bias,
                                                   float* __restrict__ out, int out_cs, int out_coff, int relu) {
  __shared__ float lut[256];
  __shared__ float patch[2][S7_PR * S7_ROW];
  __shared__ float epi[4][32 * S7_PSTR];
  const int Ho = H / 2, Wo = W / 2;
  const int strips_x = Wo / (S7_TW * S7_STRIP), tiles_y = Ho / S7_TH;
  int bid = blockIdx.x;
  const int strip = bid % strips_x;
  bid /= strips_x;
  const int ty = bid % tiles_y, b = bid / tiles_y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int p8 = lane >> 3, q8 = lane & 7;

  lut[tid] = ((float)tid / 255.f) * 2.0f - 1.0f;
  // weight fragments: lane (cout row li, k parity lh) of step s, n-tile n
  float wf[S7_STEPS][2];
#pragma unroll
  for (int s = 0; s < S7_STEPS; ++s)
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
  // patch offset of k = 2 s + lh for this lane's pixel (row 2 (2 wave + li / 16) + ky, column (2 (li % 16) + kx) * 3 + c):
  // base + (k / 21) * ROW + k % 21, both k of a step are compile-time constants; k = 147 (lh = 1, s = 73) points at the
  // row's zero slot
  const int prow = 2 * (2 * wave + (li >> 4)), pcol = 2 * (li & 15) * 3;
  const int kbase = prow * S7_ROW + pcol, kzero = prow * S7_ROW + (S7_ROW - 1);
  const uint8_t* frame = img + (size_t)b * H * W * 3;
  float* outb = out + (size_t)b * Ho * Wo * out_cs + out_coff;      // (fp32 output)
  unsigned short* outh = reinterpret_cast<unsigned short*>(out) + (size_t)b * Ho * Wo * out_cs + out_coff;   // (16-bit)
  const int iy0 = 2 * ty * S7_TH - 3;
  // the patch of strip tile t -> LDS buffer t & 1: element e = r * 111 + j (j = column * 3 + channel).  The bytes are
  // requested before tile t - 1 is computed and converted / written behind it (one HBM latency per tile otherwise)
  constexpr int NE = (S7_PR * S7_RW + 255) / 256;
  int qv[NE];
  auto request = [&](int t) {
    const int ix0 = 2 * (strip * S7_STRIP + t) * S7_TW - 3;
#pragma unroll
    for (int i = 0; i < NE; ++i) {
      const int e = tid + 256 * i;
      const int r = e / S7_RW, j = e - r * S7_RW, cc = j / 3;
      const int iy = iy0 + r, ix = ix0 + cc;
      const bool ok = e < S7_PR * S7_RW && iy >= 0 && iy < H && ix >= 0 && ix < W;
      qv[i] = ok ? frame[((size_t)iy * W + ix) * 3 + (j - cc * 3)] : -1;
    }
  };
  auto stage = [&](int t) {
    float* dst = patch[t & 1];
#pragma unroll
    for (int i = 0; i < NE; ++i) {
      const int e = tid + 256 * i;
      const int r = e / S7_RW, j = e - r * S7_RW;
      if (e < S7_PR * S7_RW) dst[r * S7_ROW + j] = qv[i] >= 0 ? lut[qv[i]] : 0.f;
    }
    if (tid < S7_PR) dst[tid * S7_ROW + S7_ROW - 1] = 0.f;
  };
  request(0);
  __syncthreads();   // the table is complete
  stage(0);
  for (int t = 0; t < S7_STRIP; ++t) {
    __syncthreads();   // patch t is complete; every wave has left patch t - 1
    if (t + 1 < S7_STRIP) request(t + 1);
    const float* src = patch[t & 1];
    f32x16 acc[2];
#pragma unroll
    for (int s = 0; s < S7_STEPS; ++s) {
      const int k0 = 2 * s, k1 = 2 * s + 1;
      const int o0 = (k0 / 21) * S7_ROW + k0 % 21, o1 = (k1 / 21) * S7_ROW + k1 % 21;
      const float x = src[k1 < 147 ? kbase + o0 + lh * (o1 - o0) : (lh ? kzero : kbase + o0)];
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
    const int tx0 = (strip * S7_STRIP + t) * S7_TW, oy0 = ty * S7_TH + 2 * wave;
#pragma unroll
    for (int n = 0; n < 2; ++n) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<f32x4*>(ep + li * S7_PSTR + 8 * q + 4 * lh) =
            f32x4{acc[n][4 * q], acc[n][4 * q + 1], acc[n][4 * q + 2], acc[n][4 * q + 3]};
      if constexpr (DT == 0) {
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const int p = 8 * gq + p8;                       // pixel of the wave's 2 x 16 block
          f32x4 o4 = *reinterpret_cast<const f32x4*>(ep + p * S7_PSTR + 4 * q8) + bv[n][0];
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
          const f32x4 lo = *reinterpret_cast<const f32x4*>(ep + p * S7_PSTR + 8 * qs) + bv[n][0];
          const f32x4 hi = *reinterpret_cast<const f32x4*>(ep + p * S7_PSTR + 8 * qs + 4) + bv[n][1];
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
    if (t + 1 < S7_STRIP) stage(t + 1);
  }
}

bool stem7_shape_ok(int H, int W, int out_cs, int out_coff) {
  return H > 0 && W > 0 && H % (2 * S7_TH) == 0 && W % (2 * S7_TW * S7_STRIP) == 0 && out_cs % 4 == 0 && out_coff % 4 == 0 &&
         out_coff + 64 <= out_cs;
}

hipError_t launch_stem7(const uint8_t* img, int B, int H, int W, const float* wpk, const float* bias, float* out,
                       int out_cs, int out_coff, int relu, hipStream_t s) {
  if (!stem7_shape_ok(H, W, out_cs, out_coff) || B <= 0) return hipErrorInvalidValue;
  const long grid = (long)B * (H / 2 / S7_TH) * (W / 2 / (S7_TW * S7_STRIP));
  if (grid > 0x7fffffffL) return hipErrorInvalidValue;
  hipLaunchKernelGGL(stem7_kernel<0>, dim3((unsigned)grid), dim3(256), 0, s, img, H, W, wpk, bias, out, out_cs, out_coff, relu);
  return hipGetLastError();
}

hipError_t launch_stem7_h16(const uint8_t* img, int B, int H, int W, const float* wpk, const float* bias, void* out,
                           int out_cs, int out_coff, int relu, int dtype, hipStream_t s) {
  // (16-byte vectors of 8 elements: strides / offsets are multiples of 8)
  if (!stem7_shape_ok(H, W, out_cs, out_coff) || out_cs % 8 || out_coff % 8 || B <= 0 || (dtype != 1 && dtype != 2))
    return hipErrorInvalidValue;
  const long grid = (long)B * (H / 2 / S7_TH) * (W / 2 / (S7_TW * S7_STRIP));
  if (grid > 0x7fffffffL) return hipErrorInvalidValue;
  float* o = reinterpret_cast<float*>(out);
  if (dtype == 2)
    hipLaunchKernelGGL(stem7_kernel<2>, dim3((unsigned)grid), dim3(256), 0, s, img, H, W, wpk, bias, o, out_cs, out_coff, relu);
  else
    hipLaunchKernelGGL(stem7_kernel<1>, dim3((unsigned)grid), dim3(256), 0, s, img, H, W, wpk, bias, o, out_cs, out_coff, relu);
  return hipGetLastError();
}

}  // namespace acrmi
