// Direct (im2col-free) NHWC fp32 convolution on the gfx950 matrix cores.
//
// GEMM view per workgroup: M = TH*TW output pixels of one frame, N = 32*NTW*WAVES_N output channels,
// K = ks*ks*Cin.  The input patch (with halo) of one Cin chunk is staged once in LDS as
// [PH][PW][CK+4] (the +4 float pad makes the per-lane ds_read_b128 of 32 neighbouring pixels
// bank-conflict free); weights are pre-packed on the host in MFMA B-fragment order so a wave
// fetches one 1 KiB line (global_load_dwordx4, L2 resident, shared by every workgroup) per
// (tap, 8-channel step, 32-cout tile).  v_mfma_f32_32x32x2_f32: exact fp32 products, fp32 accumulate:
//   A[i = lane&31][k = lane>>5] = patch[pixel i][ci = 8s + 4*(lane>>5) + j]
//   B[k = lane>>5][n = lane&31] = W[cout n][ci = 8s + 4*(lane>>5) + j]          j = 0..3
// Epilogue fuses folded-BN bias, residual add and ReLU (acr/model.py:483-499, 519-539).
#include "kernels.h"

namespace acrmi {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int KS, int S, int TH, int TW, int WAVES_M, int MT, int WAVES_N, int NTW, int CK>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64) void conv_mfma_kernel(const ConvArgs a) {
  constexpr int PH = (TH - 1) * S + KS, PW = (TW - 1) * S + KS, CP = CK + 4, PAD = KS / 2;
  constexpr int NT = WAVES_M * WAVES_N * 64;
  constexpr int TP = TH * TW;
  static_assert(TP == 32 * MT * WAVES_M, "tile pixels must equal 32*MT*WAVES_M");
  extern __shared__ f32x4 smem4[];
  float* patch = reinterpret_cast<float*>(smem4);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave % WAVES_M, wn = wave / WAVES_M;
  const int li = lane & 31, lh = lane >> 5;
  const int tiles_x = (a.Wo + TW - 1) / TW, tiles_y = (a.Ho + TH - 1) / TH;
  int t = blockIdx.x;
  const int b = t / (tiles_x * tiles_y);
  t -= b * tiles_x * tiles_y;
  const int ty0 = (t / tiles_x) * TH, tx0 = (t % tiles_x) * TW;
  const int g = blockIdx.z;
  const int n_tile0 = (blockIdx.y * WAVES_N + wn) * NTW;

  f32x16 acc[MT][NTW];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NTW; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

  int aoff[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int p = (wm * MT + m) * 32 + li;
    aoff[m] = ((p / TW) * S * PW + (p % TW) * S) * CP + 4 * lh;
  }
  const float* __restrict__ wbase = a.w + (size_t)g * KS * KS * a.cin8 * a.n_tiles * 256 + lane * 4;
  const float* __restrict__ inb = a.in + (size_t)b * a.H * a.W * a.in_cs + a.in_coff + g * a.Cin;
  const bool wave_active = n_tile0 < a.n_tiles;   // wave-uniform

  const int cin_pad = a.cin8 * 8;
  for (int c0 = 0; c0 < cin_pad; c0 += CK) {
    __syncthreads();
    // ---- stage the patch chunk: PH*PW pixels x CK channels, float4 per thread, zero halo ----
    for (int idx = tid; idx < PH * PW * (CK / 4); idx += NT) {
      const int pix = idx / (CK / 4), c4 = idx % (CK / 4);
      const int py = pix / PW, px = pix % PW;
      const int iy = ty0 * S - PAD + py, ix = tx0 * S - PAD + px;
      const int c = c0 + c4 * 4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W && c < a.Cin) {
        v = *reinterpret_cast<const f32x4*>(inb + ((size_t)iy * a.W + ix) * a.in_cs + c);
        if (c + 3 >= a.Cin) {
          if (c + 1 >= a.Cin) v[1] = 0.f;
          if (c + 2 >= a.Cin) v[2] = 0.f;
          v[3] = 0.f;
        }
      }
      *reinterpret_cast<f32x4*>(patch + pix * CP + c4 * 4) = v;
    }
    __syncthreads();
    if (!wave_active) continue;
    const int nsteps = (cin_pad - c0 < CK ? cin_pad - c0 : CK) / 8;
    const int s0 = c0 / 8;
    for (int s = 0; s < nsteps; ++s) {
#pragma unroll
      for (int tap = 0; tap < KS * KS; ++tap) {
        const int ky = tap / KS, kx = tap % KS;
        f32x4 av[MT], bv[NTW];
#pragma unroll
        for (int m = 0; m < MT; ++m)
          av[m] = *reinterpret_cast<const f32x4*>(patch + aoff[m] + (ky * PW + kx) * CP + s * 8);
#pragma unroll
        for (int n = 0; n < NTW; ++n)
          bv[n] = *reinterpret_cast<const f32x4*>(
              wbase + ((size_t)(tap * a.cin8 + s0 + s) * a.n_tiles + n_tile0 + n) * 256);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NTW; ++n)
              acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m][j], bv[n][j], acc[m][n], 0, 0, 0);
      }
    }
  }
  if (!wave_active) return;
  // ---- epilogue: C/D layout col = lane&31 (cout), row = (r&3) + 8*(r>>2) + 4*(lane>>5) (pixel) ----
  const float* __restrict__ bias = a.bias + (size_t)g * a.n_tiles * 32 + (size_t)b * a.bias_fstride;
#pragma unroll
  for (int n = 0; n < NTW; ++n) {
    const int co = (n_tile0 + n) * 32 + li;
    if (co >= a.Cout) continue;
    const float bv = bias[co];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int p = (wm * MT + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int oy = ty0 + p / TW, ox = tx0 + p % TW;
        if (oy < a.Ho && ox < a.Wo) {
          const size_t pix = ((size_t)b * a.Ho + oy) * a.Wo + ox;
          float v = acc[m][n][r] + bv;
          if (a.res) v += a.res[pix * a.res_cs + a.res_coff + g * a.Cout + co];
          if (a.relu) v = fmaxf(v, 0.f);
          a.out[pix * a.out_cs + a.out_coff + g * a.Cout + co] = v;
        }
      }
    }
  }
}

template <int KS, int S, int TH, int TW, int WAVES_M, int MT, int WAVES_N, int NTW, int CK>
static hipError_t launch_cfg(const ConvArgs& a, hipStream_t s) {
  constexpr int PH = (TH - 1) * S + KS, PW = (TW - 1) * S + KS;
  constexpr size_t lds = (size_t)PH * PW * (CK + 4) * sizeof(float);
  auto kern = conv_mfma_kernel<KS, S, TH, TW, WAVES_M, MT, WAVES_N, NTW, CK>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  const int tiles = ((a.Wo + TW - 1) / TW) * ((a.Ho + TH - 1) / TH) * a.B;
  const int nblk = (a.n_tiles + WAVES_N * NTW - 1) / (WAVES_N * NTW);
  dim3 grid(tiles, nblk, a.groups);
  hipLaunchKernelGGL(kern, grid, dim3(WAVES_M * WAVES_N * 64), lds, s, a);
  return hipGetLastError();
}

// Tile selection.  N32: one 32-cout tile per wave (Cout <= 32); N64: two.
// Small frames (<=16x16 outputs) take the 8x16 pixel tile so a batch still fills 256 CUs.
hipError_t launch_conv(ConvArgs a, hipStream_t s) {
  const bool n32 = a.n_tiles == 1;
  const bool small = (a.Ho * a.Wo <= 256) || (a.Ho % 16 != 0) || (a.Wo % 16 != 0);
  if (a.ks == 3 && a.stride == 1) {
    if (n32) return small ? launch_cfg<3, 1, 8, 16, 4, 1, 1, 1, 32>(a, s) : launch_cfg<3, 1, 16, 16, 4, 2, 1, 1, 32>(a, s);
    return small ? launch_cfg<3, 1, 8, 16, 2, 2, 2, 1, 32>(a, s) : launch_cfg<3, 1, 16, 16, 4, 2, 1, 2, 32>(a, s);
  }
  if (a.ks == 3 && a.stride == 2) {
    if (n32) return launch_cfg<3, 2, 8, 16, 4, 1, 1, 1, 16>(a, s);
    return launch_cfg<3, 2, 8, 16, 2, 2, 2, 1, 16>(a, s);
  }
  if (a.ks == 1 && a.stride == 1) {
    if (n32) return small ? launch_cfg<1, 1, 8, 16, 4, 1, 1, 1, 64>(a, s) : launch_cfg<1, 1, 16, 16, 4, 2, 1, 1, 64>(a, s);
    return small ? launch_cfg<1, 1, 8, 16, 2, 2, 2, 1, 64>(a, s) : launch_cfg<1, 1, 16, 16, 4, 2, 1, 2, 64>(a, s);
  }
  return hipErrorInvalidValue;
}

const char* conv_kernel_name(const ConvArgs& a) {
  if (a.ks == 3 && a.stride == 1) return "conv3x3s1_mfma_f32";
  if (a.ks == 3 && a.stride == 2) return "conv3x3s2_mfma_f32";
  return "conv1x1_mfma_f32";
}

}  // namespace acrmi
