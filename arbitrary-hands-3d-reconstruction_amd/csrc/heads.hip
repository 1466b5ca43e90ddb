// ACR head micro-kernels: part-attention pooling (MFMA), pare bias, center decode.
#include "kernels.h"
#include "../../include/acrmi.h"

namespace acrmi {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------
// Part attention (acr/model.py:103-113,126-136): pooled[b][p][c] = sum_pix softmax_pix(logit[b][p])[pix] * feat[b][pix][c]
// logits = segm channels 1..32 at even pixels (nearest /2 of the 256x256 map, acr/model.py:126-128).
// Pass 1: per (frame, 1/64 of the pixels) online max / sum-exp per part, lanes <-> parts (128 B coalesced rows).
// Pass 2: per (frame, 1/32 of the pixels): GEMM [32 parts x K pixels] x [K x C] on v_mfma_f32_32x32x2_f32,
//         A = exp(logit - max) computed on the fly, B = feature rows straight from HBM (read exactly once); each of
//         the four waves takes a quarter of the workgroup's pixels, the four accumulator sets are added through LDS
//         in a fixed order and one partial tile per workgroup goes to HBM.
// Pass 3: deterministic reduction of the 32 partial tiles, scaled by 1/sum-exp.
// The split is the same for every batch size, so a frame's result does not depend on the batch it is in (bit for
// bit); it is fine enough for a single frame (32 + 64 workgroups; the loops are latency bound there - round 1's
// 8 x 4 waves of 256 dependent trips took 0.365 ms per call at batch 1) and loads of four trips are requested before
// the first is used.
// ------------------------------------------------------------------------------------------------
constexpr int ATT_SCHUNKS = 64;
constexpr int ATT_KSPLIT = 32;

__global__ __launch_bounds__(256) void att_stats_kernel(const float* __restrict__ segm, int segm_cs, int H, int W,
                                                        float* __restrict__ ws) {
  const int b = blockIdx.x, chunk = blockIdx.y;
  const int part = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const int npix = H * W, per = npix / ATT_SCHUNKS;
  const float* base = segm + (size_t)b * (2 * H) * (2 * W) * segm_cs + 1 + part;
  float m = -INFINITY, s = 0.f;
  for (int q = chunk * per + grp; q < (chunk + 1) * per; q += 32) {   // per % 32 == 0
    float v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int qq = q + 8 * u, y = qq / W, x = qq % W;
      v[u] = base[((size_t)(2 * y) * (2 * W) + 2 * x) * segm_cs];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float nm = fmaxf(m, v[u]);
      s = s * expf(m - nm) + expf(v[u] - nm);
      m = nm;
    }
  }
  __shared__ float sm[8][32], ss[8][32];
  sm[grp][part] = m;
  ss[grp][part] = s;
  __syncthreads();
  if (threadIdx.x < 32) {
    float M = sm[0][part];
    for (int g = 1; g < 8; ++g) M = fmaxf(M, sm[g][part]);
    float S = 0.f;
    for (int g = 0; g < 8; ++g) S += ss[g][part] * expf(sm[g][part] - M);
    float* o = ws + (((size_t)b * ATT_SCHUNKS + chunk) * 32 + part) * 2;
    o[0] = M;
    o[1] = S;
  }
}

// FDT = storage type of the feature map (ACRMI_DT_*): 16-bit features (feat then points at 16-bit elements, feat_cs
// counts them) are widened to fp32 on load; the arithmetic is the fp32 one either way
template <int NTILES, int FDT = 0>
__global__ __launch_bounds__(256) void att_pool_kernel(const float* __restrict__ segm, int segm_cs,
                                                       const float* __restrict__ feat, int feat_cs, int H, int W,
                                                       const float* __restrict__ stats, float* __restrict__ part_ws) {
  const int b = blockIdx.x, ks = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 31, lh = lane >> 5;
  // global max of my part over the stat chunks
  float M = -INFINITY;
  {
    const float* st = stats + ((size_t)b * ATT_SCHUNKS * 32 + li) * 2;
#pragma unroll 8
    for (int c = 0; c < ATT_SCHUNKS; ++c) M = fmaxf(M, st[(size_t)c * 64]);
  }
  const int npix = H * W;
  const int per_wave = npix / (ATT_KSPLIT * 4);
  const int q0 = (ks * 4 + wave) * per_wave;
  const float* sbase = segm + (size_t)b * (2 * H) * (2 * W) * segm_cs + 1 + li;
  const float* fbase = feat + (size_t)b * npix * feat_cs + li;                                  // (fp32 features)
  const unsigned short* hbase = reinterpret_cast<const unsigned short*>(feat) + (size_t)b * npix * feat_cs + li;
  f32x16 acc[NTILES];
#pragma unroll
  for (int n = 0; n < NTILES; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
  constexpr int U = NTILES > 8 ? 2 : 4;   // trips in flight (registers: U * NTILES feature values)
  for (int q = q0 + lh; q < q0 + per_wave; q += 2 * U) {   // per_wave % 8 == 0
    float lv[U], bv[U][NTILES];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int qq = q + 2 * u, y = qq / W, x = qq % W;
      lv[u] = sbase[((size_t)(2 * y) * (2 * W) + 2 * x) * segm_cs];
      if constexpr (FDT == 0) {
        const float* f = fbase + (size_t)qq * feat_cs;
#pragma unroll
        for (int n = 0; n < NTILES; ++n) bv[u][n] = f[n * 32];
      } else {
        const unsigned short* f = hbase + (size_t)qq * feat_cs;
#pragma unroll
        for (int n = 0; n < NTILES; ++n) {
          if constexpr (FDT == 2) bv[u][n] = (float)__builtin_bit_cast(__bf16, f[n * 32]);
          else bv[u][n] = (float)__builtin_bit_cast(_Float16, f[n * 32]);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const float a = expf(lv[u] - M);
#pragma unroll
      for (int n = 0; n < NTILES; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv[u][n], acc[n], 0, 0, 0);
    }
  }
  // the four waves' accumulator sets -> one partial tile, ((w0 + w1) + w2) + w3 per element, NH n-tiles per pass
  // D layout: col = lane&31 (channel within tile), row = (r&3)+8*(r>>2)+4*(lane>>5) (part)
  constexpr int NH = NTILES < 4 ? NTILES : 4;
  __shared__ float red[4][NH * 1024];
  float* o = part_ws + (size_t)(b * ATT_KSPLIT + ks) * 32 * (NTILES * 32);
  for (int n0 = 0; n0 < NTILES; n0 += NH) {
#pragma unroll
    for (int n = 0; n < NTILES; ++n)
      if (n >= n0 && n < n0 + NH) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[wave][(n - n0) * 1024 + r * 64 + lane] = acc[n][r];
      }
    __syncthreads();
    const int cnt = (NTILES - n0 < NH ? NTILES - n0 : NH) * 1024;
    for (int idx = threadIdx.x; idx < cnt; idx += 256) {
      const float sum = ((red[0][idx] + red[1][idx]) + red[2][idx]) + red[3][idx];
      const int n = n0 + (idx >> 10), r = (idx >> 6) & 15, l = idx & 63;
      const int p = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
      o[(size_t)p * (NTILES * 32) + n * 32 + (l & 31)] = sum;
    }
    __syncthreads();
  }
}

// grid (B, 32 * C / 256): one pooled value per thread, partial tiles summed in a fixed order
__global__ __launch_bounds__(256) void att_reduce_kernel(const float* __restrict__ part_ws,
                                                         const float* __restrict__ stats, int C,
                                                         float* __restrict__ pooled) {
  const int b = blockIdx.x;
  __shared__ float inv[32];
  if (threadIdx.x < 32) {
    const int p = threadIdx.x;
    float M = -INFINITY;
    for (int c = 0; c < ATT_SCHUNKS; ++c) M = fmaxf(M, stats[(((size_t)b * ATT_SCHUNKS + c) * 32 + p) * 2]);
    float S = 0.f;
    for (int c = 0; c < ATT_SCHUNKS; ++c) {
      const float* st = stats + (((size_t)b * ATT_SCHUNKS + c) * 32 + p) * 2;
      S += st[1] * expf(st[0] - M);
    }
    inv[p] = 1.f / S;
  }
  __syncthreads();
  const int n = 32 * C;
  const int i = blockIdx.y * 256 + threadIdx.x;
  const float* src = part_ws + (size_t)b * ATT_KSPLIT * n + i;
  float s = 0.f;
  for (int k = 0; k < ATT_KSPLIT; k += 8) {
    float t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) t[u] = src[(size_t)(k + u) * n];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += t[u];
  }
  pooled[(size_t)b * n + i] = s * inv[i / C];
}

size_t attpool_ws_floats(int B, int C) {
  return (size_t)B * ATT_SCHUNKS * 32 * 2 + (size_t)B * ATT_KSPLIT * 32 * C;
}

hipError_t launch_attpool(const float* segm, int segm_cs, const float* feat, int feat_cs, int C, int B, int H, int W,
                          float* ws, float* pooled, hipStream_t s, int feat_dtype) {
  const int npix = H * W;
  if (npix % (ATT_KSPLIT * 4 * 8) != 0 || npix % (ATT_SCHUNKS * 32) != 0 || (32 * C) % 256 != 0) return hipErrorInvalidValue;
  float* stats_ws = ws;
  float* part_ws = ws + (size_t)B * ATT_SCHUNKS * 32 * 2;
  hipLaunchKernelGGL(att_stats_kernel, dim3(B, ATT_SCHUNKS), dim3(256), 0, s, segm, segm_cs, H, W, stats_ws);
  if (feat_dtype != 0) {   // 16-bit programs pool the 256 contact channels only (the shape conv is folded away)
    if (C != 256 || feat_dtype < 0 || feat_dtype > 2) return hipErrorInvalidValue;
    if (feat_dtype == 2)
      hipLaunchKernelGGL((att_pool_kernel<8, 2>), dim3(B, ATT_KSPLIT), dim3(256), 0, s, segm, segm_cs, feat, feat_cs, H, W,
                         stats_ws, part_ws);
    else
      hipLaunchKernelGGL((att_pool_kernel<8, 1>), dim3(B, ATT_KSPLIT), dim3(256), 0, s, segm, segm_cs, feat, feat_cs, H, W,
                         stats_ws, part_ws);
  } else if (C == 320)
    hipLaunchKernelGGL(att_pool_kernel<10>, dim3(B, ATT_KSPLIT), dim3(256), 0, s, segm, segm_cs, feat, feat_cs, H, W,
                       stats_ws, part_ws);
  else if (C == 256)
    hipLaunchKernelGGL(att_pool_kernel<8>, dim3(B, ATT_KSPLIT), dim3(256), 0, s, segm, segm_cs, feat, feat_cs, H, W,
                       stats_ws, part_ws);
  else if (C == 64)
    hipLaunchKernelGGL(att_pool_kernel<2>, dim3(B, ATT_KSPLIT), dim3(256), 0, s, segm, segm_cs, feat, feat_cs, H, W,
                       stats_ws, part_ws);
  else if (C == 32)
    hipLaunchKernelGGL(att_pool_kernel<1>, dim3(B, ATT_KSPLIT), dim3(256), 0, s, segm, segm_cs, feat, feat_cs, H, W,
                       stats_ws, part_ws);
  else
    return hipErrorInvalidValue;
  hipLaunchKernelGGL(att_reduce_kernel, dim3(B, 32 * C / 256), dim3(256), 0, s, part_ws, stats_ws, C, pooled);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// pare bias (acr/model.py:141-164): per frame and hand
//   offsets[j*6+o] = sum_c LC[o][c][j] * pooled[part0+j][c]            (LocallyConnected2d, :559-569)
//   shape[k]       = lin_b[k] + sum_{c,j} lin_w[k][c*16+j] * pooled[part0+j][shape_c0+c],  c < shape_nc
//   (C = 320: the reference's layout, 64 shape channels behind the 256 contact ones.  C = 256: the program's - the
//   1x1 shape conv (acr/model.py:132) is linear and the softmax weights of a part sum to 1, so it commutes with the
//   pooling and the packer folds it into lin_w / lin_b: the shape term then reads the 256 contact channels.)
//   bias[co]       = mix_b[co] + sum_k mix_wp[co][k] * [offsets|shape][k]     (pare is spatially constant)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void parebias_kernel(const PareArgs a) {
  const int b = blockIdx.x, tid = threadIdx.x;
  __shared__ float pl[16 * 320];
  const int shape_c0 = a.C == 320 ? 256 : 0, shape_n = (a.C - shape_c0) * 16;
  __shared__ float pare[112];
  __shared__ float red[10][4];
  for (int i = tid; i < 16 * a.C; i += 256) pl[i] = a.pooled[((size_t)b * 32 + a.part0) * a.C + i];
  __syncthreads();
  if (tid < 96) {
    const int j = tid / 6, o = tid % 6;
    float s = 0.f;
    for (int c = 0; c < 256; ++c) s += a.lc_w[(o * 256 + c) * 16 + j] * pl[j * a.C + c];
    pare[tid] = s;
  }
  float part[10];
#pragma unroll
  for (int k = 0; k < 10; ++k) part[k] = 0.f;
  for (int i = tid; i < shape_n; i += 256) {
    const float x = pl[(i & 15) * a.C + shape_c0 + (i >> 4)];
#pragma unroll
    for (int k = 0; k < 10; ++k) part[k] += a.lin_w[k * shape_n + i] * x;
  }
#pragma unroll
  for (int k = 0; k < 10; ++k) {
    float v = part[k];
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if ((tid & 63) == 0) red[k][tid >> 6] = v;
  }
  __syncthreads();
  if (tid < 10) pare[96 + tid] = a.lin_b[tid] + ((red[tid][0] + red[tid][1]) + (red[tid][2] + red[tid][3]));
  __syncthreads();
  if (tid < 109) {
    float s = a.mix_b[tid];
    for (int k = 0; k < 106; ++k) s += a.mix_wp[tid * 106 + k] * pare[k];
    a.out[(size_t)b * a.out_stride + tid] = s;
  } else if (tid < a.out_stride) {
    a.out[(size_t)b * a.out_stride + tid] = 0.f;
  }
}
hipError_t launch_parebias(const PareArgs& a, hipStream_t s) {
  if ((a.C != 320 && a.C != 256) || a.out_stride > 256) return hipErrorInvalidValue;
  hipLaunchKernelGGL(parebias_kernel, dim3(a.B), dim3(256), 0, s, a);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Center decode, one workgroup per frame (acr/result_parser.py:85-190,218-249; acr/utils.py:334-382,773-906)
// ------------------------------------------------------------------------------------------------
// rotation matrix -> axis-angle (acr/utils.py:334-360 -> :826-906 -> :773-823).  t = the TRANSPOSED matrix, row-major
// (the reference transposes first, :857).
__device__ inline void rotmat_t_to_aa(const float* t, float* aa) {
  const float t00 = t[0], t01 = t[1], t02 = t[2];
  const float t10 = t[3], t11 = t[4], t12 = t[5];
  const float t20 = t[6], t21 = t[7], t22 = t[8];
  float q[4], tr;
  if (t22 < 1e-6f) {
    if (t00 > t11) {
      tr = 1.f + t00 - t11 - t22;
      q[0] = t12 - t21; q[1] = tr; q[2] = t01 + t10; q[3] = t20 + t02;
    } else {
      tr = 1.f - t00 + t11 - t22;
      q[0] = t20 - t02; q[1] = t01 + t10; q[2] = tr; q[3] = t12 + t21;
    }
  } else {
    if (t00 < -t11) {
      tr = 1.f - t00 - t11 + t22;
      q[0] = t01 - t10; q[1] = t20 + t02; q[2] = t12 + t21; q[3] = tr;
    } else {
      tr = 1.f + t00 + t11 + t22;
      q[0] = tr; q[1] = t12 - t21; q[2] = t20 - t02; q[3] = t01 - t10;
    }
  }
  // reference: q /= sqrt(t); q *= 0.5
  const float rs = sqrtf(tr);
  q[0] = (q[0] / rs) * 0.5f; q[1] = (q[1] / rs) * 0.5f; q[2] = (q[2] / rs) * 0.5f; q[3] = (q[3] / rs) * 0.5f;
  const float s2 = q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  const float sn = sqrtf(s2), cs = q[0];
  const float two_theta = 2.0f * (cs < 0.f ? atan2f(-sn, -cs) : atan2f(sn, cs));
  const float k = s2 > 0.f ? two_theta / sn : 2.0f;
  for (int e = 0; e < 3; ++e) {
    float v = q[1 + e] * k;
    aa[e] = (v != v) ? 0.f : v;   // NaN -> 0 (acr/utils.py:359)
  }
}

__device__ inline void rot6d_to_aa(const float* x6, float* aa) {
  // x.view(3,2): b1 = (x0,x2,x4), a2 = (x1,x3,x5)  (acr/utils.py:362-376)
  float b1[3] = {x6[0], x6[2], x6[4]}, a2[3] = {x6[1], x6[3], x6[5]};
  float n1 = fmaxf(sqrtf(b1[0] * b1[0] + b1[1] * b1[1] + b1[2] * b1[2]), 1e-6f);
  b1[0] /= n1; b1[1] /= n1; b1[2] /= n1;
  const float d = b1[0] * a2[0] + b1[1] * a2[1] + b1[2] * a2[2];
  float b2[3] = {a2[0] - d * b1[0], a2[1] - d * b1[1], a2[2] - d * b1[2]};
  float n2 = fmaxf(sqrtf(b2[0] * b2[0] + b2[1] * b2[1] + b2[2] * b2[2]), 1e-6f);
  b2[0] /= n2; b2[1] /= n2; b2[2] /= n2;
  const float b3[3] = {b1[1] * b2[2] - b1[2] * b2[1], b1[2] * b2[0] - b1[0] * b2[2], b1[0] * b2[1] - b1[1] * b2[0]};
  // R = [b1 b2 b3] as columns; t = R^T, so t[i][j] = R[j][i]: rows of t are b1, b2, b3
  const float t[9] = {b1[0], b1[1], b1[2], b2[0], b2[1], b2[2], b3[0], b3[1], b3[2]};
  rotmat_t_to_aa(t, aa);
}

// NMS + arg-max + threshold of both center maps of frame b (acr/result_parser.py:218-249) and the cross-hand prior
// gate (:42-47, :131-145).  Called by all 256 threads of a workgroup; the result is in *pk after the last barrier.
struct CenterPick {
  int flat[2], flag[2], prior;
  float score[2];
};
struct PickScratch {
  float cmap[2][64 * 64];
  float rmax[64 * 64];        // 5-wide row maximum of one hand's cmap (the 5x5 window maximum = its 5-high column maximum)
  float bestv[2][4];
  int besti[2][4];
};
__device__ inline void pick_centers(const float* const* center, int center_cs, int b, float thresh, PickScratch& sc,
                                    CenterPick& pk, const int* prior_gate = nullptr) {
  const int tid = threadIdx.x;
  {
    // channel 0 of the two center maps: 2 x 16 strided reads per thread, ALL requested before the first one is used (a frame
    // is one workgroup: with the loads issued one per loop trip their latencies added up to most of the kernel's 55 us)
    float v[2][16];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int k = 0; k < 16; ++k) v[h][k] = center[h][((size_t)b * 4096 + tid + 256 * k) * center_cs];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int k = 0; k < 16; ++k) sc.cmap[h][tid + 256 * k] = v[h][k];
  }
  __syncthreads();
  // 5x5 window maximum, separably (max is exact and associative: the same value as the 25-element maximum)
  for (int h = 0; h < 2; ++h) {
    if (h) __syncthreads();     // the other hand's row maxima have been read
    for (int i = tid; i < 4096; i += 256) {
      const int x = i & 63;
      float m = sc.cmap[h][i];
      for (int dx = -2; dx <= 2; ++dx)
        if (dx != 0 && x + dx >= 0 && x + dx < 64) m = fmaxf(m, sc.cmap[h][i + dx]);
      sc.rmax[i] = m;
    }
    __syncthreads();
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = tid; i < 4096; i += 256) {
      const int y = i >> 6;
      const float v = sc.cmap[h][i];
      float m = sc.rmax[i];
      for (int dy = -2; dy <= 2; ++dy)
        if (dy != 0 && y + dy >= 0 && y + dy < 64) m = fmaxf(m, sc.rmax[i + dy * 64]);
      const float det = (m == v) ? v : v * 0.f;      // x * (maxpool(x) == x)  (acr/result_parser.py:245-249)
      if (det > bv || (det == bv && i < bi)) { bv = det; bi = i; }
    }
    for (int off = 32; off > 0; off >>= 1) {
      const float ov = __shfl_down(bv, off, 64);
      const int oi = __shfl_down(bi, off, 64);
      if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if ((tid & 63) == 0) { sc.bestv[h][tid >> 6] = bv; sc.besti[h][tid >> 6] = bi; }
  }
  __syncthreads();
  if (tid == 0) {
    for (int h = 0; h < 2; ++h) {
      float bv = sc.bestv[h][0];
      int bi = sc.besti[h][0];
      for (int w = 1; w < 4; ++w)
        if (sc.bestv[h][w] > bv || (sc.bestv[h][w] == bv && sc.besti[h][w] < bi)) { bv = sc.bestv[h][w]; bi = sc.besti[h][w]; }
      pk.flag[h] = bv > thresh;                       // strict; centermap_conf_thresh (acr/result_parser.py:241)
      pk.flat[h] = pk.flag[h] ? bi : 0;               // placeholder samples pixel 0 (:106-120)
      pk.score[h] = bv;
    }
    int use = pk.flag[0] && pk.flag[1];
    if (use) {                                        // determine_coeff (:42-47): > 32 px apart -> no prior
      const float dy = (float)(pk.flat[0] >> 6) - (float)(pk.flat[1] >> 6);
      const float dx = (float)(pk.flat[0] & 63) - (float)(pk.flat[1] & 63);
      if (sqrtf(dy * dy + dx * dx) > 32.f) use = 0;
    }
    if (prior_gate && prior_gate[b] >= 0) use = (pk.flag[0] && pk.flag[1]) ? (prior_gate[b] != 0) : 0;   // caller's batch-wide decision
    pk.prior = use;
  }
  __syncthreads();
}

__global__ __launch_bounds__(256) void decode_kernel(const DecodeArgs a) {
  const int b = blockIdx.x, tid = threadIdx.x;
  __shared__ PickScratch sc;
  __shared__ CenterPick pk;
  __shared__ float pred[2][112];
  pick_centers(a.center, a.center_cs, b, a.thresh, sc, pk, a.prior_gate);
  const int* s_flat = pk.flat;
  const int* s_flag = pk.flag;
  const float* s_score = pk.score;
  const int s_prior = pk.prior;
  if (tid < 218) {
    const int h = tid / 109, c = tid % 109;
    float v = a.params[h][((size_t)b * 4096 + s_flat[h]) * a.params_cs + c];
    if (s_prior && c >= 3)                            // own prior map at the OTHER hand's center (:141-145)
      v += a.prior[h][((size_t)b * 4096 + s_flat[1 - h]) * a.prior_cs + (c - 3)];
    pred[h][c] = v;
  }
  __syncthreads();
  float* slot0 = a.slots + (size_t)b * 2 * ACRMI_SLOT;
  if (tid < 218) {
    const int h = tid / 109, c = tid % 109;
    float* sl = slot0 + h * ACRMI_SLOT;
    const float v = pred[h][c];
    sl[ACRMI_SLOT_PARAMS + c] = v;
    if (c < 3) sl[ACRMI_SLOT_CAM + c] = v;
    if (c >= 99) sl[ACRMI_SLOT_BETAS + (c - 99)] = v;
    if (c == 0) {
      sl[ACRMI_SLOT_FLAG] = (float)s_flag[h];
      sl[ACRMI_SLOT_FLATIND] = (float)s_flat[h];
      sl[ACRMI_SLOT_SCORE] = s_score[h];
      sl[173] = 0.f; sl[174] = 0.f; sl[175] = 0.f;
    }
  }
  if (tid >= 224 && tid < 256) {
    const int k = tid - 224, h = k >> 4, j = k & 15;
    float aa[3];
    rot6d_to_aa(&pred[h][3 + 6 * j], aa);
    float* sl = slot0 + h * ACRMI_SLOT + ACRMI_SLOT_POSES + 3 * j;
    sl[0] = aa[0]; sl[1] = aa[1]; sl[2] = aa[2];
  }
  if (a.poison && *a.poison) {      // (uniform) invalid program results: NaN everywhere instead of plausible numbers
    __syncthreads();
    for (int i = tid; i < 2 * ACRMI_SLOT; i += 256) slot0[i] = __builtin_nanf("");
  }
}
hipError_t launch_decode(const DecodeArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(decode_kernel, dim3(a.B), dim3(256), 0, s, a);
  return hipGetLastError();
}

// acr/result_parser.py:85-145 at batch > 1, the part that looks across frames: l_ids / r_ids = the frames whose left / right
// center passed the threshold, ascending; a side without any hit in the whole batch carries a placeholder row whose flag is
// False, and :131 wants every flag set - no prior anywhere; determine_coeff (:42-47) compares l_cyxs[0] with r_cyxs[0] - the
// FIRST left-detected frame's left center and the FIRST right-detected frame's right center, not necessarily one frame - and
// more than 32 map pixels apart zeroes both coefficients for the whole batch; otherwise the prior is added in exactly the
// frames that have both hands (all_hand_valid_batch_ids, :128).  A reduction over <= 2 B flags: one workgroup of 256 threads.
__global__ __launch_bounds__(256) void prior_gate_kernel(const float* __restrict__ slots, int B, int* __restrict__ gate) {
  __shared__ int first[2][4];
  const int tid = threadIdx.x;
  int f0 = 0x7fffffff, f1 = 0x7fffffff;      // lowest frame index with a left / right detection seen by this thread
  for (int b = tid; b < B; b += 256) {
    const float* sl = slots + (size_t)b * 2 * ACRMI_SLOT;
    if (sl[ACRMI_SLOT_FLAG] > 0.5f) f0 = min(f0, b);
    if (sl[ACRMI_SLOT + ACRMI_SLOT_FLAG] > 0.5f) f1 = min(f1, b);
  }
  for (int off = 32; off > 0; off >>= 1) {
    f0 = min(f0, __shfl_down(f0, off, 64));
    f1 = min(f1, __shfl_down(f1, off, 64));
  }
  if ((tid & 63) == 0) { first[0][tid >> 6] = f0; first[1][tid >> 6] = f1; }
  __syncthreads();
  f0 = min(min(first[0][0], first[0][1]), min(first[0][2], first[0][3]));
  f1 = min(min(first[1][0], first[1][1]), min(first[1][2], first[1][3]));
  int on = 0;
  if (f0 < B && f1 < B) {
    const int fl = (int)slots[((size_t)f0 * 2 + 0) * ACRMI_SLOT + ACRMI_SLOT_FLATIND];
    const int fr = (int)slots[((size_t)f1 * 2 + 1) * ACRMI_SLOT + ACRMI_SLOT_FLATIND];
    const float dy = (float)(fl >> 6) - (float)(fr >> 6), dx = (float)(fl & 63) - (float)(fr & 63);
    on = !(sqrtf(dy * dy + dx * dx) > 32.f);
  }
  for (int b = tid; b < B; b += 256) {
    const float* sl = slots + (size_t)b * 2 * ACRMI_SLOT;
    gate[b] = (on && sl[ACRMI_SLOT_FLAG] > 0.5f && sl[ACRMI_SLOT + ACRMI_SLOT_FLAG] > 0.5f) ? 1 : 0;
  }
}
hipError_t launch_prior_gate(const float* slots, int B, int* gate, hipStream_t s) {
  hipLaunchKernelGGL(prior_gate_kernel, dim3(1), dim3(256), 0, s, slots, B, gate);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Temporal smoothing between decode and MANO (acr/main.py:69-83, acr/utils.py:1466-1527): one One-Euro filter set
// per hand type (poses[3:48], betas, and the global orientation as a 3x3 rotation matrix), applied to the frames of
// ONE video stream in order.  State (per hand: x_raw / x_filt / dx_filt of 64 elements + an init flag) lives in the
// context.  Element e of a hand: 0..44 finger pose, 45..54 betas, 55..63 rotation-matrix entries.
// Arithmetic follows torch's: python scalars meet float32 tensors as float32; the first sample passes through.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(128) void smooth_kernel(const SmoothArgs a) {
  const int tid = threadIdx.x, h = tid >> 6, e = tid & 63;
  __shared__ float rot[2][9];
  float* st = a.state + h * 3 * 64;     // [x_raw | x_filt | dx_filt][64]
  float x_raw = st[e], x_filt = st[64 + e], dx_filt = st[128 + e];
  int init = a.init[h];
  const float mincut = (e >= 45 && e < 55) ? a.mincutoff_betas : a.mincutoff;
  for (int b = 0; b < a.B; ++b) {
    float* sl = a.slots + ((size_t)b * 2 + h) * ACRMI_SLOT;
    const bool on = sl[ACRMI_SLOT_FLAG] > 0.5f;     // block-uniform per hand (wave = hand)
    if (on && e == 55) {      // batch_rodrigues (acr/utils.py:602-616) + quat2mat (:618-638) of the global orientation
      const float ax = sl[ACRMI_SLOT_POSES], ay = sl[ACRMI_SLOT_POSES + 1], az = sl[ACRMI_SLOT_POSES + 2];
      const float ex = ax + 1e-8f, ey = ay + 1e-8f, ez = az + 1e-8f;
      const float angle = sqrtf(ex * ex + ey * ey + ez * ez);
      const float nx = ax / angle, ny = ay / angle, nz = az / angle;
      const float half = angle * 0.5f;
      const float sn = sinf(half);
      float w = cosf(half), x = sn * nx, y = sn * ny, z = sn * nz;
      const float qn = sqrtf(w * w + x * x + y * y + z * z);
      w /= qn; x /= qn; y /= qn; z /= qn;
      const float w2 = w * w, x2 = x * x, y2 = y * y, z2 = z * z;
      const float wx = w * x, wy = w * y, wz = w * z, xy = x * y, xz = x * z, yz = y * z;
      float* R = rot[h];
      R[0] = w2 + x2 - y2 - z2; R[1] = 2 * xy - 2 * wz;    R[2] = 2 * wy + 2 * xz;
      R[3] = 2 * wz + 2 * xy;   R[4] = w2 - x2 + y2 - z2;  R[5] = 2 * yz - 2 * wx;
      R[6] = 2 * xz - 2 * wy;   R[7] = 2 * wx + 2 * yz;    R[8] = w2 - x2 - y2 + z2;
    }
    __syncthreads();
    if (on) {
      const float x = e < 45 ? sl[ACRMI_SLOT_POSES + 3 + e] : (e < 55 ? sl[ACRMI_SLOT_BETAS + (e - 45)] : rot[h][e - 55]);
      float s;
      if (!init) {
        s = x;
        dx_filt = 0.f;
      } else {
        const float dx = (x - x_raw) * a.freq;
        const float edx = a.alpha_d * dx + a.one_minus_alpha_d * dx_filt;
        const float cutoff = mincut + a.beta * fabsf(edx);
        const float tau = 1.0f / (a.two_pi * cutoff);
        const float al = 1.0f / (1.0f + tau / a.te);
        s = al * x + (1.0f - al) * x_filt;
        dx_filt = edx;
      }
      x_raw = x;
      x_filt = s;
      if (e < 45) sl[ACRMI_SLOT_POSES + 3 + e] = s;
      else if (e < 55) sl[ACRMI_SLOT_BETAS + (e - 45)] = s;
      else rot[h][e - 55] = s;
    }
    __syncthreads();
    if (on && e == 55) {      // rotation_matrix_to_angle_axis (acr/utils.py:334-360); the reference transposes first
      const float* R = rot[h];
      const float t[9] = {R[0], R[3], R[6], R[1], R[4], R[7], R[2], R[5], R[8]};
      float aa[3];
      rotmat_t_to_aa(t, aa);
      sl[ACRMI_SLOT_POSES] = aa[0]; sl[ACRMI_SLOT_POSES + 1] = aa[1]; sl[ACRMI_SLOT_POSES + 2] = aa[2];
    }
    if (on) init = 1;
    __syncthreads();
  }
  st[e] = x_raw; st[64 + e] = x_filt; st[128 + e] = dx_filt;
  if (e == 0) a.init[h] = init;
}
hipError_t launch_smooth(const SmoothArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(smooth_kernel, dim3(1), dim3(128), 0, s, a);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Point heads.  ResultParser reads the 109-ch params map at ONE pixel per hand (its own center,
// acr/result_parser.py:49-57,105,115) and the 106-ch prior map at ONE pixel (the other hand's center, :141-145), so
// after the center heads have run the other six head towers (acr/model.py:71-99,288-313: 3x3 s2 entry conv, two
// BasicBlocks, 1x1 exit) only need their 9x9 receptive field around that pixel: per (frame, side) three "tower
// points" (params @ own center, cam @ own center, prior @ other center) of ~4.7 MMAC each instead of three 64x64
// towers of 1.2 GMAC.  Zero padding is the map's, not the window's: every intermediate position outside the 64x64
// map is forced to 0 exactly as the dense convolution sees it.
//   pick   : centers of both hands per frame -> picks[b] = {flat_l, flat_r, prior gate, 0}
//   tower  : one workgroup (8 waves) per tower point, fp32 FMA (M is 81..1 pixels - far too small for MFMA tiles):
//            lane <-> output channel, wave <-> pixels; activations are LDS broadcasts, weights coalesced float4
//            rows [tap][cin/4][cout][4] from L2; window buffers never leave LDS
//   mix    : cam scale 1.1**x, 109x109 mix conv + per-frame pare bias (acr/model.py:95-96,160-164) at the pixel
// The results are written into the pixels of the dense maps that acrmi_decode reads, so decode is unchanged.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void center_pick_kernel(const PointArgs a) {
  __shared__ PickScratch sc;
  __shared__ CenterPick pk;
  pick_centers(a.center, a.center_cs, blockIdx.x, a.thresh, sc, pk);
  if (threadIdx.x == 0) {
    int* o = a.picks + blockIdx.x * 4;
    o[0] = pk.flat[0]; o[1] = pk.flat[1]; o[2] = a.prior_when_both ? (pk.flag[0] && pk.flag[1]) : pk.prior; o[3] = 0;
  }
}

constexpr int TP_WAVES = 8;
constexpr int TP_XW = 19;                      // input window of the stride-2 entry conv for a 9x9 tower window
constexpr int TP_XC = 36;                      // 34 channels + 2 zero
constexpr int TP_XIN = TP_XW * TP_XW * TP_XC;  // floats
constexpr int TP_LDS_FLOATS = TP_XIN + 81 * 64;

__device__ __forceinline__ float dot4(const f32x4& a, const f32x4& b, float acc) {
  acc = fmaf(a[0], b[0], acc); acc = fmaf(a[1], b[1], acc);
  acc = fmaf(a[2], b[2], acc); return fmaf(a[3], b[3], acc);
}

// 3x3 conv 64->64 + bias [+ residual] + ReLU on an NIN x NIN window in LDS -> (NIN-2)^2 window, positions outside
// the 64x64 map zeroed.  (oy0, ox0) = map coordinates of the output window's origin.
template <int NIN, int NRES>
__device__ __forceinline__ void tp_conv64(const float* in, float* out, const float* __restrict__ wgt,
                                          const float* __restrict__ bias, const float* res, int oy0, int ox0, int co,
                                          int wv, float* red) {
  constexpr int NOUT = NIN - 2, NP = NOUT * NOUT;
  const f32x4* W = reinterpret_cast<const f32x4*>(wgt);   // [(tap*16 + c4)*64 + co]
  float v0 = 0.f;   // value of output pixel (this wave's first), NP == 1 path
  if constexpr (NP == 1) {
    // one pixel: split K over the waves by tap (wave 0 also takes tap 8), reduce through LDS
    float acc = 0.f;
    for (int tap = wv; tap < 9; tap += TP_WAVES) {
      const float* src = in + ((tap / 3) * NIN + tap % 3) * 64;
      f32x4 w[16];
#pragma unroll
      for (int c4 = 0; c4 < 16; ++c4) w[c4] = W[(tap * 16 + c4) * 64 + co];
#pragma unroll
      for (int c4 = 0; c4 < 16; ++c4) acc = dot4(w[c4], *reinterpret_cast<const f32x4*>(src + c4 * 4), acc);
    }
    red[wv * 64 + co] = acc;
    __syncthreads();
    if (wv == 0) {
#pragma unroll
      for (int k = 0; k < TP_WAVES; ++k) v0 += red[k * 64 + co];
      float v = v0 + bias[co];
      if constexpr (NRES > 0) v += res[((NRES / 2) * NRES + NRES / 2) * 64 + co];
      out[co] = fmaxf(v, 0.f);     // the center pixel is always inside the map
    }
  } else {
    constexpr int PER = (NP + TP_WAVES - 1) / TP_WAVES;
    float acc[PER];
    int base[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int p = wv + TP_WAVES * i < NP ? wv + TP_WAVES * i : NP - 1;
      base[i] = ((p / NOUT) * NIN + p % NOUT) * 64;
      acc[i] = 0.f;
    }
    constexpr int U = 4, G = 9 * 16 / U;       // weight rows per group, groups
    f32x4 wn[U];
#pragma unroll
    for (int u = 0; u < U; ++u) wn[u] = W[u * 64 + co];
    for (int g = 0; g < G; ++g) {
      f32x4 wc[U];
#pragma unroll
      for (int u = 0; u < U; ++u) wc[u] = wn[u];
      const int gn = g + 1 < G ? g + 1 : g;    // (unconditional prefetch: the last group re-reads itself)
#pragma unroll
      for (int u = 0; u < U; ++u) wn[u] = W[(gn * U + u) * 64 + co];
      const int tap = g / (16 / U), c4 = (g % (16 / U)) * U;
      const int off = ((tap / 3) * NIN + tap % 3) * 64 + c4 * 4;
#pragma unroll
      for (int i = 0; i < PER; ++i)
#pragma unroll
        for (int u = 0; u < U; ++u)
          acc[i] = dot4(wc[u], *reinterpret_cast<const f32x4*>(in + base[i] + off + u * 4), acc[i]);
    }
    const float bv = bias[co];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int p = wv + TP_WAVES * i;
      if (p < NP) {
        const int oy = p / NOUT, ox = p % NOUT;
        float v = acc[i] + bv;
        if constexpr (NRES > 0) v += res[((oy + (NRES - NOUT) / 2) * NRES + ox + (NRES - NOUT) / 2) * 64 + co];
        const bool inside = (unsigned)(oy0 + oy) < 64u && (unsigned)(ox0 + ox) < 64u;
        out[p * 64 + co] = inside ? fmaxf(v, 0.f) : 0.f;
      }
    }
  }
  (void)v0;
  __syncthreads();
}

__global__ __launch_bounds__(TP_WAVES * 64) void tower_point_kernel(const PointArgs a) {
  extern __shared__ float tp_lds[];
  const int b = blockIdx.x / 3, t = blockIdx.x % 3, tid = threadIdx.x;
  const int co = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int* pk = a.picks + b * 4;
  if (t == 2 && !pk[2]) return;                  // the prior is only read when the gate is open
  const int flat = t == 2 ? pk[1 - a.side] : pk[a.side];
  const int cy = flat >> 6, cx = flat & 63;
  float* xin = tp_lds;                           // [19*19][36], dead after the entry conv
  float* t0 = tp_lds + TP_XIN;                   // [9*9][64]   tower input (block 0 residual)
  float* fa = xin;                               // [7*7][64]
  float* fb = fa + 49 * 64;                      // [5*5][64]   (block 1 residual)
  float* fc = fb + 25 * 64;                      // [3*3][64]
  float* fd = fc + 9 * 64;                       // [64]
  float* red = fd + 64;                          // [8][64]
  const float* wt = a.w + (size_t)t * TP_TOWER_FLOATS;
  // ---- x34 window: rows/cols 2*(c-4)-1 .. 2*(c+4)+1 ----
  const float* xb = a.x34 + (size_t)b * 128 * 128 * a.x_cs;
  for (int i = tid; i < TP_XW * TP_XW * 9; i += TP_WAVES * 64) {
    const int pix = i / 9, c4 = i - pix * 9;
    const int iy = 2 * (cy - 4) - 1 + pix / TP_XW, ix = 2 * (cx - 4) - 1 + pix % TP_XW;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if ((unsigned)iy < 128u && (unsigned)ix < 128u) {
      v = *reinterpret_cast<const f32x4*>(xb + ((size_t)iy * 128 + ix) * a.x_cs + c4 * 4);
      if (c4 == 8) { v[2] = 0.f; v[3] = 0.f; }
    }
    *reinterpret_cast<f32x4*>(xin + pix * TP_XC + c4 * 4) = v;
  }
  __syncthreads();
  // ---- entry: 3x3 stride 2, 34 -> 64, ReLU (acr/model.py:300-303) on the 9x9 window ----
  {
    constexpr int PER = (81 + TP_WAVES - 1) / TP_WAVES;
    const f32x4* W = reinterpret_cast<const f32x4*>(wt);   // [(tap*9 + c4)*64 + co]
    float acc[PER];
    int base[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int p = wv + TP_WAVES * i < 81 ? wv + TP_WAVES * i : 80;
      base[i] = ((2 * (p / 9)) * TP_XW + 2 * (p % 9)) * TP_XC;
      acc[i] = 0.f;
    }
    constexpr int U = 3, G = 81 / U;
    f32x4 wn[U];
#pragma unroll
    for (int u = 0; u < U; ++u) wn[u] = W[u * 64 + co];
    for (int g = 0; g < G; ++g) {
      f32x4 wc[U];
#pragma unroll
      for (int u = 0; u < U; ++u) wc[u] = wn[u];
      const int gn = g + 1 < G ? g + 1 : g;
#pragma unroll
      for (int u = 0; u < U; ++u) wn[u] = W[(gn * U + u) * 64 + co];
      const int tap = g / 3, c4 = (g % 3) * U;
      const int off = ((tap / 3) * TP_XW + tap % 3) * TP_XC + c4 * 4;
#pragma unroll
      for (int i = 0; i < PER; ++i)
#pragma unroll
        for (int u = 0; u < U; ++u)
          acc[i] = dot4(wc[u], *reinterpret_cast<const f32x4*>(xin + base[i] + off + u * 4), acc[i]);
    }
    const float bv = wt[TP_ENTRY_W + co];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int p = wv + TP_WAVES * i;
      if (p < 81) {
        const bool inside = (unsigned)(cy - 4 + p / 9) < 64u && (unsigned)(cx - 4 + p % 9) < 64u;
        t0[p * 64 + co] = inside ? fmaxf(acc[i] + bv, 0.f) : 0.f;
      }
    }
  }
  __syncthreads();
  // ---- two BasicBlocks (acr/model.py:483-499): relu(bn(conv)) -> bn(conv) + x -> relu ----
  const float* wc = wt + TP_ENTRY_W + 64;
  constexpr int CS = TP_CONV_W + 64;
  tp_conv64<9, 0>(t0, fa, wc, wc + TP_CONV_W, nullptr, cy - 3, cx - 3, co, wv, red);
  tp_conv64<7, 9>(fa, fb, wc + CS, wc + CS + TP_CONV_W, t0, cy - 2, cx - 2, co, wv, red);
  tp_conv64<5, 0>(fb, fc, wc + 2 * CS, wc + 2 * CS + TP_CONV_W, nullptr, cy - 1, cx - 1, co, wv, red);
  tp_conv64<3, 5>(fc, fd, wc + 3 * CS, wc + 3 * CS + TP_CONV_W, fb, cy, cx, co, wv, red);
  // ---- 1x1 exit (acr/model.py:305-311) -> the pixel of the dense map ----
  const float* we = wc + 4 * CS;                 // [64][112], bias [112]
  const int nout = t == 1 ? 3 : 106;
  if (tid < nout) {
    float s = we[64 * TP_EXIT_N + tid];
#pragma unroll 8
    for (int ci = 0; ci < 64; ++ci) s = fmaf(we[ci * TP_EXIT_N + tid], fd[ci], s);
    if (t == 2) a.prior[((size_t)b * 4096 + flat) * a.prior_cs + tid] = s;
    else a.p109[((size_t)b * 4096 + flat) * a.p109_cs + (t == 0 ? 3 : 0) + tid] = s;
  }
}

__global__ __launch_bounds__(128) void point_mix_kernel(const PointArgs a) {
  const int b = blockIdx.x, tid = threadIdx.x;
  const int flat = a.picks[b * 4 + a.side];
  __shared__ float p[112];
  if (tid < 109) {
    float v = a.p109[((size_t)b * 4096 + flat) * a.p109_cs + tid];
    p[tid] = tid == 0 ? powf(1.1f, v) : v;         // cam scale (acr/model.py:95-96)
  }
  __syncthreads();
  if (tid < 109) {
    float s = a.bias[(size_t)b * a.bias_stride + tid];
    for (int ci = 0; ci < 109; ++ci) s = fmaf(a.mix_w[ci * TP_EXIT_N + tid], p[ci], s);
    a.final_[((size_t)b * 4096 + flat) * a.final_cs + tid] = s;
  }
}

hipError_t launch_point_heads(const PointArgs& a, hipStream_t s) {
  static unsigned char attr_set[MAX_DEVICES] = {};
  constexpr int LDS_BYTES = TP_LDS_FLOATS * (int)sizeof(float);
  static_assert(TP_XIN >= (49 + 25 + 9 + 1 + TP_WAVES) * 64, "window buffers alias the dead x34 window");
  std::lock_guard<std::recursive_mutex> lock(launch_mutex());
  if (first_use_on_device(attr_set)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(tower_point_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(center_pick_kernel, dim3(a.B), dim3(256), 0, s, a);
  hipLaunchKernelGGL(tower_point_kernel, dim3(a.B * 3), dim3(TP_WAVES * 64), LDS_BYTES, s, a);
  hipLaunchKernelGGL(point_mix_kernel, dim3(a.B), dim3(128), 0, s, a);
  return hipGetLastError();
}

}  // namespace acrmi
