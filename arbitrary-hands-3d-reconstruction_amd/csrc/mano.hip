// Fused MANO forward: one workgroup per hand (mano/manolayer.py:104-276 with use_pca=False,
// flat_hand_mean=False, axis-angle root) + weak-perspective projection (acr/utils.py:384-412).
// Rodrigues (via quaternion) -> shape blend -> joint regression (wave reductions) -> pose blend ->
// 16-joint kinematic chain -> rest-pose removal -> linear blend skinning of 778 vertices ->
// fingertips / joint reorder / root alignment.  Everything between the 58 input floats and the
// 2397 output floats lives in LDS; the 1.46 MB of per-side tables are read coalesced
// (blend-shape tables pre-transposed on the host) and stay L2 resident across hands.
#include "kernels.h"

namespace acrmi {

__constant__ int c_parent[16] = {-1, 0, 1, 2, 0, 4, 5, 0, 7, 8, 0, 10, 11, 0, 13, 14};
__constant__ int c_depth[16] = {0, 1, 2, 3, 1, 2, 3, 1, 2, 3, 1, 2, 3, 1, 2, 3};
__constant__ int c_reorder[21] = {0, 13, 14, 15, 16, 1, 2, 3, 17, 4, 5, 6, 18, 10, 11, 12, 19, 7, 8, 9, 20};
__constant__ int c_tips[2][5] = {{745, 317, 445, 556, 673}, {745, 317, 444, 556, 673}};  // [left, right]

constexpr int NV = 778, NV3 = 2334;

// H16 = ACRMI_OPT_MANO_FP16 (BASELINE.json configs[4] "fp16 MANO LBS"): the blend-shape tables (shapedirs, posedirs)
// and the skinning weights are read as f16 copies (0.73 instead of 1.46 MB of L2 traffic per side and hand), every
// product and sum stays fp32; v_template, the joint regressor and the kinematic chain are untouched.  Round 6: the two
// small tables (skinning weights 25 KB, shape blend shapes 47 KB) are f16 PAIRS hi + lo (~22 bits) - the plain-f16 skinning
// weights were all of r5's 6.4e-5 m; the pose blend table, 86 % of the bytes, stays plain f16.
template <bool H16>
__global__ __launch_bounds__(256) void mano_kernel(const ManoArgs a) {
  // a.slices workgroups per hand (launch_mano): at small batches a hand's 120 us - one CU streaming the 1.26 MB pose-blend
  // table - are on the call's critical path; the slices split that table by vertex range.  Every vertex and joint is
  // computed by exactly one slice with the arithmetic of the one-workgroup form: results do not depend on a.slices.
  const int row = blockIdx.x / a.slices, slice = blockIdx.x - row * a.slices;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int vper = (NV + a.slices - 1) / a.slices;
  const int v0 = slice * vper, v1 = v0 + vper < NV ? v0 + vper : NV;
  const int side = a.side ? a.side[row] : (row & 1);
  const ManoTables& T = a.t[side];
  __shared__ float sR[16][9];
  __shared__ float sPoseMap[136];
  __shared__ float sBeta[10];
  __shared__ float sV[NV3];
  __shared__ float sJ[16][3];
  __shared__ float sG[16][12];   // rows of [R | t]
  __shared__ float sA[16][12];
  __shared__ float sJtr[21][3];
  __shared__ float sCenter[3];

  const float* pose = a.poses + (size_t)row * a.pose_stride;
  if (tid < 10) sBeta[tid] = a.betas[(size_t)row * a.beta_stride + tid];
  if (a.pose_rotmat) {
    // joint_rot_mode='rotmat' (mano/manolayer.py:151-162): the caller's 16 rotation matrices (already projected onto SO(3) by
    // batch_rotprojs on the host, like the reference does on the CPU), root first; th_hands_mean plays no part
    if (tid < 144) sR[tid / 9][tid % 9] = pose[tid];
  } else if (tid < 16) {
    // batch_rodrigues (mano/manolayer.py:423-434) + quat2mat (:396-421)
    float ax = pose[3 * tid], ay = pose[3 * tid + 1], az = pose[3 * tid + 2];
    if (tid > 0) {
      ax += T.hands_mean[3 * (tid - 1)];
      ay += T.hands_mean[3 * (tid - 1) + 1];
      az += T.hands_mean[3 * (tid - 1) + 2];
    }
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
    float* R = sR[tid];
    R[0] = w2 + x2 - y2 - z2; R[1] = 2 * xy - 2 * wz;    R[2] = 2 * wy + 2 * xz;
    R[3] = 2 * wz + 2 * xy;   R[4] = w2 - x2 + y2 - z2;  R[5] = 2 * yz - 2 * wx;
    R[6] = 2 * xz - 2 * wy;   R[7] = 2 * wx + 2 * yz;    R[8] = w2 - x2 - y2 + z2;
  }
  __syncthreads();
  if (tid < 135) {
    const int k9 = tid % 9;
    sPoseMap[tid] = sR[1 + tid / 9][k9] - ((k9 == 0 || k9 == 4 || k9 == 8) ? 1.f : 0.f);
  }
  // shape blend: v_shaped = shapedirs . beta + v_template
  for (int i = tid; i < NV3; i += 256) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 10; ++k)
      s += (H16 ? (float)__builtin_bit_cast(_Float16, T.shapedirs_h[k * NV3 + i]) + (float)__builtin_bit_cast(_Float16, T.shapedirs_l[k * NV3 + i])
                : T.shapedirs_t[k * NV3 + i]) * sBeta[k];
    sV[i] = s + T.v_template[i];
  }
  __syncthreads();
  // joint regression: 48 dot products of length 778, 12 per wave, wave-level reduction
  for (int o = wave * 12; o < wave * 12 + 12; ++o) {
    const int j = o / 3, d = o % 3;
    float s = 0.f;
    for (int v = lane; v < NV; v += 64) s += T.jreg[j * NV + v] * sV[v * 3 + d];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if (lane == 0) sJ[j][d] = s;
  }
  __syncthreads();
  // pose blend (needs v_shaped complete for the regression above, so done after it)
  for (int i = 3 * v0 + tid; i < 3 * v1; i += 256) {
    float s = 0.f;
    for (int k = 0; k < 135; ++k)
      s += (H16 ? (float)__builtin_bit_cast(_Float16, T.posedirs_h[k * NV3 + i]) : T.posedirs_t[k * NV3 + i]) * sPoseMap[k];
    sV[i] += s;
  }
  // kinematic chain, three levels below the root (mano/manolayer.py:187-223)
  for (int level = 0; level < 4; ++level) {
    if (tid < 16 && c_depth[tid] == level) {
      const int j = tid, p = c_parent[j];
      const float* R = sR[j];
      float* G = sG[j];
      if (p < 0) {
        for (int r = 0; r < 3; ++r) {
          G[r * 4 + 0] = R[r * 3]; G[r * 4 + 1] = R[r * 3 + 1]; G[r * 4 + 2] = R[r * 3 + 2];
          G[r * 4 + 3] = sJ[0][r];
        }
      } else {
        const float* P = sG[p];
        const float t0 = sJ[j][0] - sJ[p][0], t1 = sJ[j][1] - sJ[p][1], t2 = sJ[j][2] - sJ[p][2];
        for (int r = 0; r < 3; ++r) {
          const float p0 = P[r * 4], p1 = P[r * 4 + 1], p2 = P[r * 4 + 2], p3 = P[r * 4 + 3];
          G[r * 4 + 0] = p0 * R[0] + p1 * R[3] + p2 * R[6];
          G[r * 4 + 1] = p0 * R[1] + p1 * R[4] + p2 * R[7];
          G[r * 4 + 2] = p0 * R[2] + p1 * R[5] + p2 * R[8];
          G[r * 4 + 3] = p0 * t0 + p1 * t1 + p2 * t2 + p3;
        }
      }
    }
    __syncthreads();
  }
  if (tid < 16) {   // rest-pose removal: A = G - [0 | G.[J;0]]  (mano/manolayer.py:226-228)
    const float* G = sG[tid];
    float* A = sA[tid];
    for (int r = 0; r < 3; ++r) {
      A[r * 4] = G[r * 4]; A[r * 4 + 1] = G[r * 4 + 1]; A[r * 4 + 2] = G[r * 4 + 2];
      A[r * 4 + 3] = G[r * 4 + 3] - (G[r * 4] * sJ[tid][0] + G[r * 4 + 1] * sJ[tid][1] + G[r * 4 + 2] * sJ[tid][2]);
    }
  }
  __syncthreads();
  // linear blend skinning (mano/manolayer.py:230-240)
  for (int v = v0 + tid; v < v1; v += 256) {
    float Tm[12];
#pragma unroll
    for (int e = 0; e < 12; ++e) Tm[e] = 0.f;
    const float* wv = T.weights + v * 16;
    const unsigned short* wh = T.weights_h + v * 16;
    const unsigned short* wl = T.weights_l + v * 16;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const float w = H16 ? (float)__builtin_bit_cast(_Float16, wh[j]) + (float)__builtin_bit_cast(_Float16, wl[j]) : wv[j];
#pragma unroll
      for (int e = 0; e < 12; ++e) Tm[e] += sA[j][e] * w;
    }
    const float x = sV[v * 3], y = sV[v * 3 + 1], z = sV[v * 3 + 2];
    const float ox = Tm[0] * x + Tm[1] * y + Tm[2] * z + Tm[3];
    const float oy = Tm[4] * x + Tm[5] * y + Tm[6] * z + Tm[7];
    const float oz = Tm[8] * x + Tm[9] * y + Tm[10] * z + Tm[11];
    sV[v * 3] = ox; sV[v * 3 + 1] = oy; sV[v * 3 + 2] = oz;
  }
  __syncthreads();
  if (tid < 21) {   // joints: 16 chain translations + 5 fingertip vertices, reordered (:241-254)
    const int src = c_reorder[tid];
    for (int d = 0; d < 3; ++d)
      sJtr[tid][d] = src < 16 ? sG[src][d * 4 + 3] : sV[c_tips[side][src - 16] * 3 + d];
  }
  __syncthreads();
  if (tid < 3) sCenter[tid] = a.center_idx >= 0 ? sJtr[a.center_idx][tid] : 0.f;
  __syncthreads();
  const float cx = sCenter[0], cy = sCenter[1], cz = sCenter[2];
  float cs = 0.f, ctx = 0.f, cty = 0.f, padw = 0.f, padh = 0.f, ltx = 0.f, lty = 0.f;
  const bool proj = a.cam != nullptr;
  if (proj) {
    const float* cam = a.cam + (size_t)row * a.cam_stride;
    cs = cam[0]; ctx = cam[1]; cty = cam[2];
    if (a.offsets) {
      const float* of = a.offsets + (size_t)(row / a.off_div) * 10;
      padw = of[0]; padh = of[1];
      ltx = of[5] - of[9];      // crop_trbl[3] - pad_trbl[3]
      lty = of[2] - of[6];      // crop_trbl[0] - pad_trbl[0]
    }
  }
  for (int i = 3 * v0 + tid; i < 3 * v1; i += 256) {
    const int d = i % 3;
    const float v = sV[i] - (d == 0 ? cx : (d == 1 ? cy : cz));
    a.verts[(size_t)row * NV3 + i] = v;
    if (proj && a.verts_camed) a.verts_camed[(size_t)row * NV3 + i] = d == 2 ? v : v * cs + (d == 0 ? ctx : cty);
  }
  // joint j: a chain joint is written by slice 0, a fingertip by the slice that skinned its vertex
  const int jsrc = tid < 63 ? c_reorder[tid / 3] : 0;
  const int jtip = jsrc >= 16 ? c_tips[side][jsrc - 16] : -1;
  if (tid < 63 && (jtip < 0 ? slice == 0 : (jtip >= v0 && jtip < v1))) {
    const int j = tid / 3, d = tid % 3;
    const float v = sJtr[j][d] - (d == 0 ? cx : (d == 1 ? cy : cz));
    a.joints[(size_t)row * 63 + tid] = v;
    if (proj && d < 2) {
      const float pj = v * cs + (d == 0 ? ctx : cty);
      if (a.pj2d) a.pj2d[(size_t)row * 42 + j * 2 + d] = pj;
      if (a.pj2d_org && a.offsets)
        a.pj2d_org[(size_t)row * 42 + j * 2 + d] = (pj + 1.f) * (d == 0 ? padw : padh) / 2.f + (d == 0 ? ltx : lty);
    }
  }
  if (a.center && tid < 3 && slice == 0) a.center[(size_t)row * 3 + tid] = sCenter[tid];
}

hipError_t launch_mano(const ManoArgs& a0, hipStream_t s) {
  if (a0.H <= 0) return hipSuccess;
  ManoArgs a = a0;
  // workgroups per hand: enough to give every CU one (2 hands: 8 slices; 128 hands: 2).  A root joint that is a FINGERTIP
  // (center_idx 4, 8, 12, 16, 20: a skinned vertex, mano/manolayer.py:241-262) is known only to the slice that skinned
  // it: one slice then.
  const bool tip_center = a.center_idx >= 0 && a.center_idx % 4 == 0 && a.center_idx > 0;
  a.slices = tip_center ? 1 : (a.H >= 256 ? 1 : (256 / a.H > 8 ? 8 : 256 / a.H));
  if (a.lbs_f16) {
    if (!a.t[0].posedirs_h || !a.t[1].posedirs_h) return hipErrorInvalidValue;
    hipLaunchKernelGGL(mano_kernel<true>, dim3(a.H * a.slices), dim3(256), 0, s, a);
  } else {
    hipLaunchKernelGGL(mano_kernel<false>, dim3(a.H * a.slices), dim3(256), 0, s, a);
  }
  return hipGetLastError();
}

// Camera translation per hand: the reference's closed-form least squares (acr/utils.py:430-472,
// estimate_translation_np with unit confidences): for every joint (X, Y, Z) with 2-D target (u, v) = (pj2d+1)*img/2
//   f*tx + (c-u)*tz = (u-c)*Z - f*X,   f*ty + (c-v)*tz = (v-c)*Z - f*Y,   c = img/2
// solved through the 3x3 normal equations in fp64 (the reference does the same in numpy fp64).  One thread per hand.
__global__ void cam_trans_kernel(const float* __restrict__ joints, const float* __restrict__ pj2d, int n, double f,
                                 double img, float* __restrict__ out) {
  const int h = blockIdx.x * blockDim.x + threadIdx.x;
  if (h >= n) return;
  const double c = img / 2;
  double A[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, b[3] = {0, 0, 0};
  for (int j = 0; j < 21; ++j) {
    const double X = joints[(size_t)h * 63 + j * 3], Y = joints[(size_t)h * 63 + j * 3 + 1], Z = joints[(size_t)h * 63 + j * 3 + 2];
    const double u = ((double)pj2d[(size_t)h * 42 + j * 2] + 1) * c, v = ((double)pj2d[(size_t)h * 42 + j * 2 + 1] + 1) * c;
    const double r0[3] = {f, 0, c - u}, r1[3] = {0, f, c - v};
    const double c0 = (u - c) * Z - f * X, c1 = (v - c) * Z - f * Y;
    for (int p = 0; p < 3; ++p) {
      for (int q = 0; q < 3; ++q) A[p][q] += r0[p] * r0[q] + r1[p] * r1[q];
      b[p] += r0[p] * c0 + r1[p] * c1;
    }
  }
  // Gaussian elimination with partial pivoting
  double M[3][4] = {{A[0][0], A[0][1], A[0][2], b[0]}, {A[1][0], A[1][1], A[1][2], b[1]}, {A[2][0], A[2][1], A[2][2], b[2]}};
  for (int col = 0; col < 3; ++col) {
    int piv = col;
    for (int r = col + 1; r < 3; ++r)
      if (fabs(M[r][col]) > fabs(M[piv][col])) piv = r;
    for (int q = 0; q < 4; ++q) { const double t = M[col][q]; M[col][q] = M[piv][q]; M[piv][q] = t; }
    for (int r = col + 1; r < 3; ++r) {
      const double m = M[r][col] / M[col][col];
      for (int q = col; q < 4; ++q) M[r][q] -= m * M[col][q];
    }
  }
  double t[3];
  for (int r = 2; r >= 0; --r) {
    double s = M[r][3];
    for (int q = r + 1; q < 3; ++q) s -= M[r][q] * t[q];
    t[r] = s / M[r][r];
  }
  for (int d = 0; d < 3; ++d) out[(size_t)h * 3 + d] = (float)t[d];
}

hipError_t launch_cam_trans(const float* joints, const float* pj2d, int n, float focal, float img, float* out, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(cam_trans_kernel, dim3((n + 63) / 64), dim3(64), 0, s, joints, pj2d, n, (double)focal, (double)img, out);
  return hipGetLastError();
}

}  // namespace acrmi
