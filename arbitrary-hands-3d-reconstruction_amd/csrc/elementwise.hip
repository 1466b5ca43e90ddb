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

}  // namespace acrmi

namespace acrmi {

// ------------------------------------------------------------------------------------------------
// Pre-processing (acr/utils.py:1315-1337): BGR uint8 frame [H,W,3] -> white-padded square
// (imgaug Pad semantics: extra pixel goes to bottom/right) -> bicubic resize (a = -0.75, half-pixel centres,
// replicate border - the OpenCV INTER_CUBIC / torch bicubic kernel) to 512x512 RGB uint8.  One thread per
// output pixel; the 1080p source (6.2 MB/frame) is read once through L2.
// ------------------------------------------------------------------------------------------------
__device__ inline float cubic1(float x, float A) { return ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f; }
__device__ inline float cubic2(float x, float A) { return ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A; }

__global__ __launch_bounds__(256) void preprocess_kernel(const uint8_t* __restrict__ bgr, int n, int H, int W, int S,
                                                         int pad_top, int pad_left, int out_size,
                                                         uint8_t* __restrict__ out) {
  const long total = (long)n * out_size * out_size;
  const float scale = (float)S / (float)out_size;
  const float A = -0.75f;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int ox = i % out_size;
    const int oy = (i / out_size) % out_size;
    const int f = i / ((long)out_size * out_size);
    const float fy = scale * (oy + 0.5f) - 0.5f, fx = scale * (ox + 0.5f) - 0.5f;
    const float fly = floorf(fy), flx = floorf(fx);
    const float ty = fy - fly, tx = fx - flx;
    const int sy = (int)fly, sx = (int)flx;
    const float cy[4] = {cubic2(ty + 1.f, A), cubic1(ty, A), cubic1(1.f - ty, A), cubic2(2.f - ty, A)};
    const float cx[4] = {cubic2(tx + 1.f, A), cubic1(tx, A), cubic1(1.f - tx, A), cubic2(2.f - tx, A)};
    float acc[3] = {0.f, 0.f, 0.f};
    const uint8_t* src = bgr + (size_t)f * H * W * 3;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      int yy = sy - 1 + a;
      yy = yy < 0 ? 0 : (yy >= S ? S - 1 : yy);      // replicate border of the padded square
      const int iy = yy - pad_top;
      float row[3] = {0.f, 0.f, 0.f};
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        int xx = sx - 1 + b;
        xx = xx < 0 ? 0 : (xx >= S ? S - 1 : xx);
        const int ix = xx - pad_left;
        float v0 = 255.f, v1 = 255.f, v2 = 255.f;   // white padding (acr/utils.py:1305-1310)
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
      float r = rintf(acc[c]);
      r = r < 0.f ? 0.f : (r > 255.f ? 255.f : r);
      o[c] = (uint8_t)r;
    }
  }
}

hipError_t launch_preprocess(const uint8_t* bgr, int n, int H, int W, int S, int pad_top, int pad_left, int out_size,
                             uint8_t* out, hipStream_t s) {
  const long total = (long)n * out_size * out_size;
  long g = (total + 255) / 256;
  if (g > 256L * 32) g = 256L * 32;
  hipLaunchKernelGGL(preprocess_kernel, dim3((unsigned)g), dim3(256), 0, s, bgr, n, H, W, S, pad_top, pad_left, out_size, out);
  return hipGetLastError();
}

}  // namespace acrmi
