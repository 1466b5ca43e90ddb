// HBM-bound helper kernels of the ACR path: 16-byte coalesced accesses, grid-stride loops.
#include "kernels.h"

namespace acrmi {

typedef float f32x4 __attribute__((ext_vector_type(4)));

static inline int grid_for(long n, int block) {
  long g = (n + block - 1) / block;
  const long cap = 256L * 32;   // 256 CUs x 32 workgroups is plenty for a grid-stride loop
  return (int)(g < cap ? (g > 0 ? g : 1) : cap);
}

// uint8 RGB -> fp32 (x/255)*2-1, 4th channel zero (acr/model.py:832 + acr/utils.py:226-231 without the NCHW copy)
__global__ __launch_bounds__(256) void u8norm_kernel(const uint8_t* __restrict__ img, long n_pixels,
                                                     f32x4* __restrict__ out) {
  for (long p = blockIdx.x * 256L + threadIdx.x; p < n_pixels; p += (long)gridDim.x * 256) {
    const uint8_t* q = img + p * 3;
    f32x4 v;
    v[0] = ((float)q[0] / 255.f) * 2.0f - 1.0f;
    v[1] = ((float)q[1] / 255.f) * 2.0f - 1.0f;
    v[2] = ((float)q[2] / 255.f) * 2.0f - 1.0f;
    v[3] = 0.f;
    out[p] = v;
  }
}
hipError_t launch_u8norm(const uint8_t* img, long n_pixels, float* out, hipStream_t s) {
  hipLaunchKernelGGL(u8norm_kernel, dim3(grid_for(n_pixels, 256)), dim3(256), 0, s, img, n_pixels,
                     reinterpret_cast<f32x4*>(out));
  return hipGetLastError();
}

// fp32 NCHW -> a channel slice of an NHWC buffer.  A workgroup moves a [C <= 64][64 pixels] tile through LDS: reads are
// coalesced along the pixels of a channel plane, writes along the channels of a pixel.
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ in, int B, int C, long HW,
                                                           float* __restrict__ out, int cs, int coff) {
  __shared__ float tile[64][65];
  const long tiles_per_frame = (HW + 63) / 64;
  for (long t = blockIdx.x; t < (long)B * tiles_per_frame; t += gridDim.x) {
    const int b = (int)(t / tiles_per_frame);
    const long p0 = (t % tiles_per_frame) * 64;
    for (int c0 = 0; c0 < C; c0 += 64) {
      const int px = threadIdx.x & 63;
      for (int c = threadIdx.x >> 6; c < 64; c += 4)
        if (c0 + c < C && p0 + px < HW) tile[c][px] = in[((long)b * C + c0 + c) * HW + p0 + px];
      __syncthreads();
      const int ch = threadIdx.x & 63;
      for (int q = threadIdx.x >> 6; q < 64; q += 4)
        if (c0 + ch < C && p0 + q < HW) out[((long)b * HW + p0 + q) * cs + coff + c0 + ch] = tile[ch][q];
      __syncthreads();
    }
  }
}
hipError_t launch_nchw_to_nhwc(const float* in, int B, int C, int H, int W, float* out, int cs, int coff, hipStream_t s) {
  const long HW = (long)H * W;
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(grid_for((long)B * ((HW + 63) / 64) * 256, 256)), dim3(256), 0, s, in, B, C, HW, out,
                     cs, coff);
  return hipGetLastError();
}

// bilinear x2, align_corners=True (F.interpolate at acr/model.py:432): src = dst*(in-1)/(out-1)
__global__ __launch_bounds__(256) void bilinear2x_kernel(const float* __restrict__ in, int B, int H, int W, int in_cs,
                                                         int in_coff, int C4, float* __restrict__ out, int out_cs,
                                                         int out_coff) {
  const int Ho = 2 * H, Wo = 2 * W;
  const float sh = (float)(H - 1) / (float)(Ho - 1), sw = (float)(W - 1) / (float)(Wo - 1);
  const long n = (long)B * Ho * Wo * C4;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int c4 = i % C4;
    long p = i / C4;
    const int ox = p % Wo;
    p /= Wo;
    const int oy = p % Ho;
    const int b = p / Ho;
    const float fy = sh * oy, fx = sw * ox;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < H - 1), x1 = x0 + (x0 < W - 1);
    const float ly = fy - y0, lx = fx - x0, hy = 1.f - ly, hx = 1.f - lx;
    const float* base = in + (size_t)b * H * W * in_cs + in_coff + c4 * 4;
    const f32x4 v00 = *reinterpret_cast<const f32x4*>(base + ((size_t)y0 * W + x0) * in_cs);
    const f32x4 v01 = *reinterpret_cast<const f32x4*>(base + ((size_t)y0 * W + x1) * in_cs);
    const f32x4 v10 = *reinterpret_cast<const f32x4*>(base + ((size_t)y1 * W + x0) * in_cs);
    const f32x4 v11 = *reinterpret_cast<const f32x4*>(base + ((size_t)y1 * W + x1) * in_cs);
    f32x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = hy * (hx * v00[e] + lx * v01[e]) + ly * (hx * v10[e] + lx * v11[e]);
    *reinterpret_cast<f32x4*>(out + (((size_t)b * Ho + oy) * Wo + ox) * out_cs + out_coff + c4 * 4) = r;
  }
}
hipError_t launch_bilinear2x(const float* in, int B, int H, int W, int in_cs, int in_coff, int C, float* out,
                             int out_cs, int out_coff, hipStream_t s) {
  const long n = (long)B * 4 * H * W * (C / 4);
  hipLaunchKernelGGL(bilinear2x_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, in, B, H, W, in_cs, in_coff, C / 4,
                     out, out_cs, out_coff);
  return hipGetLastError();
}

// HR-module fusion: out = [relu](t0 + up(t1) + ...), summed in the reference's order (acr/model.py:677-684)
__global__ __launch_bounds__(256) void fuse_sum_kernel(const FuseArgs a) {
  const int C4 = a.C / 4;
  const long n = (long)a.B * a.H * a.W * C4;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int c4 = i % C4;
    long p = i / C4;
    const int x = p % a.W;
    p /= a.W;
    const int y = p % a.H;
    const int b = p / a.H;
    f32x4 acc;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      if (t < a.nterms) {
        const int sh = a.shift[t];
        const int h = a.H >> sh, w = a.W >> sh;
        const f32x4 v = *reinterpret_cast<const f32x4*>(
            a.term[t] + (((size_t)b * h + (y >> sh)) * w + (x >> sh)) * a.cs[t] + c4 * 4);
        if (t == 0) acc = v; else acc += v;
      }
    }
    if (a.relu) {
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] = fmaxf(acc[e], 0.f);
    }
    *reinterpret_cast<f32x4*>(a.out + (((size_t)b * a.H + y) * a.W + x) * a.out_cs + c4 * 4) = acc;
  }
}
hipError_t launch_fuse_sum(const FuseArgs& a, hipStream_t s) {
  const long n = (long)a.B * a.H * a.W * (a.C / 4);
  hipLaunchKernelGGL(fuse_sum_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, a);
  return hipGetLastError();
}

// cam scale channel: x := 1.1 ** x (acr/model.py:95-96)
__global__ __launch_bounds__(256) void pow11_kernel(float* buf, long n_pixels, int cs, int ch) {
  for (long p = blockIdx.x * 256L + threadIdx.x; p < n_pixels; p += (long)gridDim.x * 256)
    buf[p * cs + ch] = powf(1.1f, buf[p * cs + ch]);
}
hipError_t launch_pow11(float* buf, long n_pixels, int cs, int ch, hipStream_t s) {
  hipLaunchKernelGGL(pow11_kernel, dim3(grid_for(n_pixels, 256)), dim3(256), 0, s, buf, n_pixels, cs, ch);
  return hipGetLastError();
}

// coord maps (acr/model.py:340-369): channel coff = x (along W), coff+1 = y (along H), i/(size-1)*2-1
__global__ __launch_bounds__(256) void coordfill_kernel(float* buf, int B, int H, int W, int cs, int coff) {
  const long n = (long)B * H * W;
  for (long p = blockIdx.x * 256L + threadIdx.x; p < n; p += (long)gridDim.x * 256) {
    const int x = p % W, y = (p / W) % H;
    buf[p * cs + coff] = ((float)x / (float)(W - 1)) * 2.f - 1.f;
    buf[p * cs + coff + 1] = ((float)y / (float)(H - 1)) * 2.f - 1.f;
  }
}
hipError_t launch_coordfill(float* buf, int B, int H, int W, int cs, int coff, hipStream_t s) {
  hipLaunchKernelGGL(coordfill_kernel, dim3(grid_for((long)B * H * W, 256)), dim3(256), 0, s, buf, B, H, W, cs, coff);
  return hipGetLastError();
}


// ------------------------------------------------------------------------------------------------
// 16-bit storage variants (f16 / bf16 programs, include/acrmi.h ACRMI_DT_*): the same arithmetic in fp32 on values
// read from / rounded once (nearest even) to the storage type, 16-byte vectors of 8 elements.
// ------------------------------------------------------------------------------------------------
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <bool BF>
__device__ __forceinline__ void unpack8(const f32x4& v, float (&f)[8]) {
  if constexpr (BF) {
    const bf16x8 h = __builtin_bit_cast(bf16x8, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = (float)h[e];
  } else {
    const f16x8 h = __builtin_bit_cast(f16x8, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = (float)h[e];
  }
}
template <bool BF>
__device__ __forceinline__ f32x4 pack8(const float (&f)[8]) {
  if constexpr (BF) {
    bf16x8 h;
#pragma unroll
    for (int e = 0; e < 8; ++e) h[e] = (__bf16)f[e];
    return __builtin_bit_cast(f32x4, h);
  } else {
    f16x8 h;
#pragma unroll
    for (int e = 0; e < 8; ++e) h[e] = (_Float16)f[e];
    return __builtin_bit_cast(f32x4, h);
  }
}
template <bool BF>
__device__ __forceinline__ unsigned short to16(float f) {
  if constexpr (BF) return __builtin_bit_cast(unsigned short, (__bf16)f);
  else return __builtin_bit_cast(unsigned short, (_Float16)f);
}
template <bool BF>
__device__ __forceinline__ float from16(unsigned short u) {
  if constexpr (BF) return (float)__builtin_bit_cast(__bf16, u);
  else return (float)__builtin_bit_cast(_Float16, u);
}

template <bool BF>
__global__ __launch_bounds__(256) void bilinear2x_h16_kernel(const unsigned short* __restrict__ in, int B, int H, int W,
                                                             int in_cs, int in_coff, int C8, unsigned short* __restrict__ out,
                                                             int out_cs, int out_coff) {
  const int Ho = 2 * H, Wo = 2 * W;
  const float sh = (float)(H - 1) / (float)(Ho - 1), sw = (float)(W - 1) / (float)(Wo - 1);
  const long n = (long)B * Ho * Wo * C8;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int c8 = i % C8;
    long p = i / C8;
    const int ox = p % Wo;
    p /= Wo;
    const int oy = p % Ho;
    const int b = p / Ho;
    const float fy = sh * oy, fx = sw * ox;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < H - 1), x1 = x0 + (x0 < W - 1);
    const float ly = fy - y0, lx = fx - x0, hy = 1.f - ly, hx = 1.f - lx;
    const unsigned short* base = in + (size_t)b * H * W * in_cs + in_coff + c8 * 8;
    float v00[8], v01[8], v10[8], v11[8], r[8];
    unpack8<BF>(*reinterpret_cast<const f32x4*>(base + ((size_t)y0 * W + x0) * in_cs), v00);
    unpack8<BF>(*reinterpret_cast<const f32x4*>(base + ((size_t)y0 * W + x1) * in_cs), v01);
    unpack8<BF>(*reinterpret_cast<const f32x4*>(base + ((size_t)y1 * W + x0) * in_cs), v10);
    unpack8<BF>(*reinterpret_cast<const f32x4*>(base + ((size_t)y1 * W + x1) * in_cs), v11);
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] = hy * (hx * v00[e] + lx * v01[e]) + ly * (hx * v10[e] + lx * v11[e]);
    *reinterpret_cast<f32x4*>(out + (((size_t)b * Ho + oy) * Wo + ox) * out_cs + out_coff + c8 * 8) = pack8<BF>(r);
  }
}
hipError_t launch_bilinear2x_h16(const void* in, int B, int H, int W, int in_cs, int in_coff, int C, void* out, int out_cs,
                                 int out_coff, int dtype, hipStream_t s) {
  const long n = (long)B * 4 * H * W * (C / 8);
  auto i16 = reinterpret_cast<const unsigned short*>(in);
  auto o16 = reinterpret_cast<unsigned short*>(out);
  if (dtype == 2)
    hipLaunchKernelGGL(bilinear2x_h16_kernel<true>, dim3(grid_for(n, 256)), dim3(256), 0, s, i16, B, H, W, in_cs, in_coff,
                       C / 8, o16, out_cs, out_coff);
  else
    hipLaunchKernelGGL(bilinear2x_h16_kernel<false>, dim3(grid_for(n, 256)), dim3(256), 0, s, i16, B, H, W, in_cs, in_coff,
                       C / 8, o16, out_cs, out_coff);
  return hipGetLastError();
}

template <bool BF>
__global__ __launch_bounds__(256) void fuse_sum_h16_kernel(const FuseArgs a) {
  const int C8 = a.C / 8;
  const long n = (long)a.B * a.H * a.W * C8;
  unsigned short* out = reinterpret_cast<unsigned short*>(a.out);
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int c8 = i % C8;
    long p = i / C8;
    const int x = p % a.W;
    p /= a.W;
    const int y = p % a.H;
    const int b = p / a.H;
    float acc[8];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      if (t < a.nterms) {
        const int sh = a.shift[t];
        const int h = a.H >> sh, w = a.W >> sh;
        float v[8];
        unpack8<BF>(*reinterpret_cast<const f32x4*>(reinterpret_cast<const unsigned short*>(a.term[t]) +
                                                    (((size_t)b * h + (y >> sh)) * w + (x >> sh)) * a.cs[t] + c8 * 8), v);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = t == 0 ? v[e] : acc[e] + v[e];
      }
    }
    if (a.relu) {
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] = fmaxf(acc[e], 0.f);
    }
    *reinterpret_cast<f32x4*>(out + (((size_t)b * a.H + y) * a.W + x) * a.out_cs + c8 * 8) = pack8<BF>(acc);
  }
}
// a.term / a.out are 16-bit tensors (pointers carried as float*), strides in elements
hipError_t launch_fuse_sum_h16(const FuseArgs& a, int dtype, hipStream_t s) {
  const long n = (long)a.B * a.H * a.W * (a.C / 8);
  if (dtype == 2) hipLaunchKernelGGL(fuse_sum_h16_kernel<true>, dim3(grid_for(n, 256)), dim3(256), 0, s, a);
  else hipLaunchKernelGGL(fuse_sum_h16_kernel<false>, dim3(grid_for(n, 256)), dim3(256), 0, s, a);
  return hipGetLastError();
}

template <bool BF>
__global__ __launch_bounds__(256) void pow11_h16_kernel(unsigned short* buf, long n_pixels, int cs, int ch) {
  for (long p = blockIdx.x * 256L + threadIdx.x; p < n_pixels; p += (long)gridDim.x * 256)
    buf[p * cs + ch] = to16<BF>(powf(1.1f, from16<BF>(buf[p * cs + ch])));
}
hipError_t launch_pow11_h16(void* buf, long n_pixels, int cs, int ch, int dtype, hipStream_t s) {
  auto b16 = reinterpret_cast<unsigned short*>(buf);
  if (dtype == 2) hipLaunchKernelGGL(pow11_h16_kernel<true>, dim3(grid_for(n_pixels, 256)), dim3(256), 0, s, b16, n_pixels, cs, ch);
  else hipLaunchKernelGGL(pow11_h16_kernel<false>, dim3(grid_for(n_pixels, 256)), dim3(256), 0, s, b16, n_pixels, cs, ch);
  return hipGetLastError();
}

template <bool BF>
__global__ __launch_bounds__(256) void coordfill_h16_kernel(unsigned short* buf, int B, int H, int W, int cs, int coff) {
  const long n = (long)B * H * W;
  for (long p = blockIdx.x * 256L + threadIdx.x; p < n; p += (long)gridDim.x * 256) {
    const int x = p % W, y = (p / W) % H;
    buf[p * cs + coff] = to16<BF>(((float)x / (float)(W - 1)) * 2.f - 1.f);
    buf[p * cs + coff + 1] = to16<BF>(((float)y / (float)(H - 1)) * 2.f - 1.f);
  }
}
hipError_t launch_coordfill_h16(void* buf, int B, int H, int W, int cs, int coff, int dtype, hipStream_t s) {
  auto b16 = reinterpret_cast<unsigned short*>(buf);
  const int g = grid_for((long)B * H * W, 256);
  if (dtype == 2) hipLaunchKernelGGL(coordfill_h16_kernel<true>, dim3(g), dim3(256), 0, s, b16, B, H, W, cs, coff);
  else hipLaunchKernelGGL(coordfill_h16_kernel<false>, dim3(g), dim3(256), 0, s, b16, B, H, W, cs, coff);
  return hipGetLastError();
}

}  // namespace acrmi

namespace acrmi {

// ------------------------------------------------------------------------------------------------
// Pre-processing (acr/utils.py:1315-1337): BGR uint8 frame [H,W,3] -> white-padded square (imgaug
// 0.4.0 Pad: the extra pixel goes to bottom/right) -> cv2.resize(..., (512,512), INTER_CUBIC) -> RGB uint8.
// The resize is OpenCV's uint8 path restated from its published source (modules/imgproc/src/resize.cpp), bit for bit:
//   fx = (float)((dx + 0.5) * scale - 0.5) with scale in double, sx = floor(fx), fx -= sx;
//   coefficients interpolateCubic(fx) with A = -0.75 in float, stored as short = round-half-even(c * 2048)
//   (INTER_RESIZE_COEF_BITS = 11); horizontal pass in int32 over 4 border-clamped columns; vertical pass in int32
//   over 4 border-clamped rows; dst = saturate((v + 2^21) >> 22)   (FixedPtCast<int, uchar, 22>).
// oracle/preprocess.py is the CPU statement of the same algorithm; tests require equality.
// One thread per output pixel; the 1080p source (6.2 MB/frame) is read once through L2.
// ------------------------------------------------------------------------------------------------
__device__ inline void cv_cubic_taps(int d, double scale, int& s0, int (&c)[4]) {
  float f = (float)(((double)d + 0.5) * scale - 0.5);
  const float fl = floorf(f);
  s0 = (int)fl;
  f -= fl;
  const float A = -0.75f;
  float k[4];
  k[0] = ((A * (f + 1.f) - 5.f * A) * (f + 1.f) + 8.f * A) * (f + 1.f) - 4.f * A;
  k[1] = ((A + 2.f) * f - (A + 3.f)) * f * f + 1.f;
  k[2] = ((A + 2.f) * (1.f - f) - (A + 3.f)) * (1.f - f) * (1.f - f) + 1.f;
  k[3] = 1.f - k[0] - k[1] - k[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int v = (int)rintf(k[i] * 2048.f);            // saturate_cast<short>(float): cvRound, then clamp
    c[i] = v < -32768 ? -32768 : (v > 32767 ? 32767 : v);
  }
}

__global__ __launch_bounds__(256) void preprocess_kernel(const uint8_t* __restrict__ bgr, int n, int H, int W, int S,
                                                         int pad_top, int pad_left, int out_size,
                                                         uint8_t* __restrict__ out) {
  const long total = (long)n * out_size * out_size;
  const double scale = (double)S / (double)out_size;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int ox = i % out_size;
    const int oy = (i / out_size) % out_size;
    const int f = i / ((long)out_size * out_size);
    int sy, sx, cy[4], cx[4];
    cv_cubic_taps(oy, scale, sy, cy);
    cv_cubic_taps(ox, scale, sx, cx);
    int acc[3] = {0, 0, 0};
    const uint8_t* src = bgr + (size_t)f * H * W * 3;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      int yy = sy - 1 + a;
      yy = yy < 0 ? 0 : (yy >= S ? S - 1 : yy);      // border rows / columns of the padded square are clamped
      const int iy = yy - pad_top;
      int row[3] = {0, 0, 0};
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        int xx = sx - 1 + b;
        xx = xx < 0 ? 0 : (xx >= S ? S - 1 : xx);
        const int ix = xx - pad_left;
        int v0 = 255, v1 = 255, v2 = 255;           // white padding (acr/utils.py:1303-1308)
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) {
          const uint8_t* p = src + ((size_t)iy * W + ix) * 3;
          v0 = p[2]; v1 = p[1]; v2 = p[0];          // BGR -> RGB (acr/utils.py:1318)
        }
        row[0] += cx[b] * v0; row[1] += cx[b] * v1; row[2] += cx[b] * v2;
      }
      acc[0] += cy[a] * row[0]; acc[1] += cy[a] * row[1]; acc[2] += cy[a] * row[2];
    }
    uint8_t* o = out + (size_t)i * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int r = (acc[c] + (1 << 21)) >> 22;
      o[c] = (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r));
    }
  }
}

// The same arithmetic with PER-FRAME geometry (acr/utils.py:1315-1337 is per image; folder mode, acr/main.py:144-205, mixes
// sizes): up to PRE_FRAMES_PER_LAUNCH frames per launch, their {pointer, H, W} by value in the kernel arguments (no device
// table to allocate or upload).  256 consecutive output pixels never straddle a frame (512 * 512 % 256 == 0).
__global__ __launch_bounds__(256) void preprocess_frames_kernel(const PreBatch pb, int n, int out_size, uint8_t* __restrict__ out) {
  const long total = (long)n * out_size * out_size;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int ox = i % out_size;
    const int oy = (i / out_size) % out_size;
    const int f = i / ((long)out_size * out_size);
    const int H = pb.f[f].H, W = pb.f[f].W;
    // imgaug compute_paddings_to_reach_aspect_ratio(shape, 1.0): pad the shorter side, the extra pixel bottom / right
    const int S = H > W ? H : W;
    const int pad_top = H < W ? (W - H) / 2 : 0, pad_left = W < H ? (H - W) / 2 : 0;
    const double scale = (double)S / (double)out_size;
    int sy, sx, cy[4], cx[4];
    cv_cubic_taps(oy, scale, sy, cy);
    cv_cubic_taps(ox, scale, sx, cx);
    int acc[3] = {0, 0, 0};
    const uint8_t* src = pb.f[f].bgr;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      int yy = sy - 1 + a;
      yy = yy < 0 ? 0 : (yy >= S ? S - 1 : yy);
      const int iy = yy - pad_top;
      int row[3] = {0, 0, 0};
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        int xx = sx - 1 + b;
        xx = xx < 0 ? 0 : (xx >= S ? S - 1 : xx);
        const int ix = xx - pad_left;
        int v0 = 255, v1 = 255, v2 = 255;
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) {
          const uint8_t* p = src + ((size_t)iy * W + ix) * 3;
          v0 = p[2]; v1 = p[1]; v2 = p[0];
        }
        row[0] += cx[b] * v0; row[1] += cx[b] * v1; row[2] += cx[b] * v2;
      }
      acc[0] += cy[a] * row[0]; acc[1] += cy[a] * row[1]; acc[2] += cy[a] * row[2];
    }
    uint8_t* o = out + (size_t)i * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int r = (acc[c] + (1 << 21)) >> 22;
      o[c] = (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r));
    }
  }
}

hipError_t launch_preprocess_frames(const PreBatch& pb, int n, int out_size, uint8_t* out, hipStream_t s) {
  const long total = (long)n * out_size * out_size;
  long g = (total + 255) / 256;
  if (g > 256L * 32) g = 256L * 32;
  hipLaunchKernelGGL(preprocess_frames_kernel, dim3((unsigned)g), dim3(256), 0, s, pb, n, out_size, out);
  return hipGetLastError();
}

hipError_t launch_preprocess(const uint8_t* bgr, int n, int H, int W, int S, int pad_top, int pad_left, int out_size,
                             uint8_t* out, hipStream_t s) {
  const long total = (long)n * out_size * out_size;
  long g = (total + 255) / 256;
  if (g > 256L * 32) g = 256L * 32;
  hipLaunchKernelGGL(preprocess_kernel, dim3((unsigned)g), dim3(256), 0, s, bgr, n, H, W, S, pad_top, pad_left, out_size, out);
  return hipGetLastError();
}

// Max pooling 3x3, stride 2, padding 1 (ResNet stem, torchvision resnet.py maxpool; padded positions do not take part):
// NHWC, 4 floats / 8 halfs per thread.  The maximum of values of the storage type is exact in every type.
__global__ __launch_bounds__(256) void maxpool3s2_kernel(const float* __restrict__ in, int B, int H, int W, int in_cs, int in_coff,
                                                         int C4, float* __restrict__ out, int out_cs, int out_coff) {
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const long n = (long)B * Ho * Wo * C4;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int c4 = i % C4;
    long p = i / C4;
    const int ox = p % Wo;
    p /= Wo;
    const int oy = p % Ho;
    const int b = p / Ho;
    const float* base = in + (size_t)b * H * W * in_cs + in_coff + c4 * 4;
    f32x4 r = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
      for (int dx = -1; dx <= 1; ++dx) {
        const int y = 2 * oy + dy, x = 2 * ox + dx;
        if (y < 0 || y >= H || x < 0 || x >= W) continue;
        const f32x4 v = *reinterpret_cast<const f32x4*>(base + ((size_t)y * W + x) * in_cs);
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] = fmaxf(r[e], v[e]);
      }
    *reinterpret_cast<f32x4*>(out + (((size_t)b * Ho + oy) * Wo + ox) * out_cs + out_coff + c4 * 4) = r;
  }
}
hipError_t launch_maxpool3s2(const float* in, int B, int H, int W, int in_cs, int in_coff, int C, float* out, int out_cs,
                             int out_coff, hipStream_t s) {
  const long n = (long)B * ((H - 1) / 2 + 1) * ((W - 1) / 2 + 1) * (C / 4);
  hipLaunchKernelGGL(maxpool3s2_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, in, B, H, W, in_cs, in_coff, C / 4, out,
                     out_cs, out_coff);
  return hipGetLastError();
}

template <bool BF>
__global__ __launch_bounds__(256) void maxpool3s2_h16_kernel(const unsigned short* __restrict__ in, int B, int H, int W, int in_cs,
                                                             int in_coff, int C8, unsigned short* __restrict__ out, int out_cs,
                                                             int out_coff) {
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const long n = (long)B * Ho * Wo * C8;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int c8 = i % C8;
    long p = i / C8;
    const int ox = p % Wo;
    p /= Wo;
    const int oy = p % Ho;
    const int b = p / Ho;
    const unsigned short* base = in + (size_t)b * H * W * in_cs + in_coff + c8 * 8;
    float r[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] = -INFINITY;
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
      for (int dx = -1; dx <= 1; ++dx) {
        const int y = 2 * oy + dy, x = 2 * ox + dx;
        if (y < 0 || y >= H || x < 0 || x >= W) continue;
        float v[8];
        unpack8<BF>(*reinterpret_cast<const f32x4*>(base + ((size_t)y * W + x) * in_cs), v);
#pragma unroll
        for (int e = 0; e < 8; ++e) r[e] = fmaxf(r[e], v[e]);
      }
    *reinterpret_cast<f32x4*>(out + (((size_t)b * Ho + oy) * Wo + ox) * out_cs + out_coff + c8 * 8) = pack8<BF>(r);
  }
}
hipError_t launch_maxpool3s2_h16(const void* in, int B, int H, int W, int in_cs, int in_coff, int C, void* out, int out_cs,
                                 int out_coff, int dtype, hipStream_t s) {
  const long n = (long)B * ((H - 1) / 2 + 1) * ((W - 1) / 2 + 1) * (C / 8);
  auto i16 = reinterpret_cast<const unsigned short*>(in);
  auto o16 = reinterpret_cast<unsigned short*>(out);
  if (dtype == 2)
    hipLaunchKernelGGL(maxpool3s2_h16_kernel<true>, dim3(grid_for(n, 256)), dim3(256), 0, s, i16, B, H, W, in_cs, in_coff, C / 8,
                       o16, out_cs, out_coff);
  else
    hipLaunchKernelGGL(maxpool3s2_h16_kernel<false>, dim3(grid_for(n, 256)), dim3(256), 0, s, i16, B, H, W, in_cs, in_coff, C / 8,
                       o16, out_cs, out_coff);
  return hipGetLastError();
}

}  // namespace acrmi
