// Direct (im2col-free) NHWC fp32 convolution on the gfx950 matrix cores.
//
// GEMM view per work item: M = TH*TW output pixels of one frame, N = 32*NTW*WAVES_N output channels,
// K = ks*ks*Cin.  The input patch (with halo) of one Cin chunk is staged in LDS as [PH][PW][CK+4]
// (the +4 float pad makes the per-lane ds_read_b128 of 32 neighbouring pixels bank-conflict free);
// weights are pre-packed on the host in MFMA B-fragment order so a wave fetches one 1 KiB line
// (global_load_dwordx4, L2 resident, shared by every workgroup) per (tap, 8-channel step, 32-cout tile).
// v_mfma_f32_32x32x2_f32: exact fp32 products, fp32 accumulate:
//   A[i = lane&31][k = lane>>5] = patch[pixel i][ci = 8s + 4*(lane>>5) + j]
//   B[k = lane>>5][n = lane&31] = W[cout n][ci = 8s + 4*(lane>>5) + j]          j = 0..3
//
// Persistent, software-pipelined workgroups: the grid is k workgroups per CU (k chosen so the work
// items divide evenly), each walks its (tile, N-block, group) items; the global loads of the NEXT
// (item, Cin-chunk) patch are issued into registers before the MFMAs of the current chunk and only
// written to LDS after them, so HBM latency never stalls the matrix pipe even when all workgroups
// of a CU run in lock-step; inside a chunk the A/B fragments of step i+1 are fetched before the
// 4*MT*NTW MFMAs of step i.  Epilogue fuses folded-BN bias, residual add and ReLU
// (acr/model.py:483-499, 519-539).
#include "kernels.h"

namespace acrmi {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

static int g_force_cfg = -1;
void conv_force_cfg(int cfg) { g_force_cfg = cfg; }
static long long* g_dbg = nullptr;
static int g_phase_delay = 0;
void conv_set_phase_delay(int cycles) { g_phase_delay = cycles; }
void conv_set_debug(long long* dbg) { g_dbg = dbg; }

struct ConvWork {
  int tiles_x, tiles_per_frame, n_tiles_total, nblk, total;
};

template <int KS, int S, int TH, int TW, int WAVES_M, int MT, int WAVES_N, int NTW, int CK>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64, 2) void conv_mfma_kernel(const ConvArgs a, const ConvWork wk) {
  constexpr int PH = (TH - 1) * S + KS, PW = (TW - 1) * S + KS, CP = CK + 4, PAD = KS / 2;
  constexpr int NT = WAVES_M * WAVES_N * 64;
  constexpr int TP = TH * TW;
  constexpr int NLOAD = PH * PW * (CK / 4);
  constexpr int NLD = (NLOAD + NT - 1) / NT;
  constexpr int TAPS = KS * KS;
  static_assert(TP == 32 * MT * WAVES_M, "tile pixels must equal 32*MT*WAVES_M");
  extern __shared__ f32x4 smem4[];
  float* patch = reinterpret_cast<float*>(smem4);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave % WAVES_M, wn = wave / WAVES_M;
  const int li = lane & 31, lh = lane >> 5;

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
  const size_t tap_stride = (size_t)a.cin8 * a.n_tiles * 256;   // floats between taps
  const size_t step_stride = (size_t)a.n_tiles * 256;
  const int cin_pad = a.cin8 * 8;

  f32x4 stage[NLD];
  auto issue_loads = [&](int w, int c0) {
    const int tile = w % wk.n_tiles_total;
    const int g = (w / wk.n_tiles_total) / wk.nblk;
    const int b = tile / wk.tiles_per_frame;
    const int t = tile - b * wk.tiles_per_frame;
    const int ty0 = (t / wk.tiles_x) * TH, tx0 = (t % wk.tiles_x) * TW;
    const float* __restrict__ inb = a.in + (size_t)b * a.H * a.W * a.in_cs + a.in_coff + g * a.Cin;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int idx = tid + i * NT;
      const int pix = idx / (CK / 4), c4 = idx % (CK / 4);
      const int iy = ty0 * S - PAD + pix / PW, ix = tx0 * S - PAD + pix % PW;
      const int c = c0 + c4 * 4;
      const bool ok = (idx < NLOAD) && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W && c < a.Cin;
      const int iyc = iy < 0 ? 0 : (iy >= a.H ? a.H - 1 : iy), ixc = ix < 0 ? 0 : (ix >= a.W ? a.W - 1 : ix);
      const int cc = c < a.Cin ? c : 0;
      const int off = (iyc * a.W + ixc) * a.in_cs + cc;   // per-frame offset < 2^31 floats
      f32x4 v = *reinterpret_cast<const f32x4*>(inb + off);
      if (c + 1 >= a.Cin) v[1] = 0.f;
      if (c + 2 >= a.Cin) v[2] = 0.f;
      if (c + 3 >= a.Cin) v[3] = 0.f;
      if (!ok) v = f32x4{0.f, 0.f, 0.f, 0.f};
      stage[i] = v;
    }
  };

  int w = blockIdx.x, c0 = 0;
  if (w < wk.total) issue_loads(w, 0);
  while (w < wk.total) {
    __syncthreads();   // fragment reads of the previous chunk are done
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int idx = tid + i * NT;
      if (idx < NLOAD) *reinterpret_cast<f32x4*>(patch + (idx / (CK / 4)) * CP + (idx % (CK / 4)) * 4) = stage[i];
    }
    __syncthreads();
    // next (item, chunk) of this workgroup: its loads fly while the MFMAs below run
    int nw = w, nc = c0 + CK;
    if (nc >= cin_pad) { nc = 0; nw = w + gridDim.x; }
    if (nw < wk.total) issue_loads(nw, nc);

    const int rest = w / wk.n_tiles_total;
    const int g = rest / wk.nblk;
    const int n_tile0 = ((rest % wk.nblk) * WAVES_N + wn) * NTW;
    const bool wave_active = n_tile0 < a.n_tiles;   // wave-uniform
    if (wave_active) {
      const int nsteps = (cin_pad - c0 < CK ? cin_pad - c0 : CK) / 8;
      const float* __restrict__ wchunk = a.w + (size_t)g * TAPS * tap_stride + (size_t)n_tile0 * 256 + lane * 4 +
                                         (size_t)(c0 / 8) * step_stride;
      f32x4 av[2][MT], bv[2][NTW];
#pragma unroll
      for (int m = 0; m < MT; ++m) av[0][m] = *reinterpret_cast<const f32x4*>(patch + aoff[m]);
#pragma unroll
      for (int n = 0; n < NTW; ++n) bv[0][n] = *reinterpret_cast<const f32x4*>(wchunk + (size_t)n * 256);
      for (int s = 0; s < nsteps; ++s) {
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
          const int cur = tap & 1, nxt = cur ^ 1;
          // prefetch fragments of the next (s, tap); the last step re-reads a valid address
          const int ntap = tap + 1 < TAPS ? tap + 1 : 0;
          const int ns = tap + 1 < TAPS ? s : (s + 1 < nsteps ? s + 1 : s);
          const int nky = ntap / KS, nkx = ntap % KS;
#pragma unroll
          for (int m = 0; m < MT; ++m)
            av[nxt][m] = *reinterpret_cast<const f32x4*>(patch + aoff[m] + (nky * PW + nkx) * CP + ns * 8);
#pragma unroll
          for (int n = 0; n < NTW; ++n)
            bv[nxt][n] = *reinterpret_cast<const f32x4*>(wchunk + (size_t)ntap * tap_stride +
                                                         (size_t)ns * step_stride + (size_t)n * 256);
          // keep the prefetch ahead of this step's MFMAs (the scheduler otherwise sinks the loads to
          // just before their first use and the L2 latency is exposed every step)
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
              for (int n = 0; n < NTW; ++n)
                acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur][m][j], bv[cur][n][j], acc[m][n], 0, 0, 0);
        }
        if (TAPS & 1) {   // odd tap count: the double buffer parity flips every s; re-align
#pragma unroll
          for (int m = 0; m < MT; ++m) av[0][m] = av[1][m];
#pragma unroll
          for (int n = 0; n < NTW; ++n) bv[0][n] = bv[1][n];
        }
      }
      if (c0 + CK >= cin_pad) {
        // ---- epilogue: C/D layout col = lane&31 (cout), row = (r&3) + 8*(r>>2) + 4*(lane>>5) (pixel) ----
        const int tile = w % wk.n_tiles_total;
        const int b = tile / wk.tiles_per_frame;
        const int t = tile - b * wk.tiles_per_frame;
        const int ty0 = (t / wk.tiles_x) * TH, tx0 = (t % wk.tiles_x) * TW;
        const float* __restrict__ bias = a.bias + (size_t)g * a.n_tiles * 32 + (size_t)b * a.bias_fstride;
        const bool full_tile = (ty0 + TH <= a.Ho) && (tx0 + TW <= a.Wo);
        const bool has_res = a.res != nullptr;
        // per-frame bases (uniform -> SGPRs) + 32-bit per-lane offsets
        float* __restrict__ outb = a.out + (size_t)b * a.Ho * a.Wo * a.out_cs + a.out_coff + g * a.Cout;
        const float* __restrict__ resb =
            has_res ? a.res + (size_t)b * a.Ho * a.Wo * a.res_cs + a.res_coff + g * a.Cout : nullptr;
#pragma unroll
        for (int n = 0; n < NTW; ++n) {
          const int co = (n_tile0 + n) * 32 + li;
          const bool cok = co < a.Cout;
          const float bvv = cok ? bias[co] : 0.f;
          const int coc = cok ? co : 0;
#pragma unroll
          for (int m = 0; m < MT; ++m) {
#pragma unroll
            for (int half = 0; half < 2; ++half) {
              int pix[8];
              bool ok[8];
#pragma unroll
              for (int q = 0; q < 8; ++q) {
                const int r = half * 8 + q;
                const int p = (wm * MT + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                int oy = ty0 + p / TW, ox = tx0 + p % TW;
                ok[q] = cok && (full_tile || (oy < a.Ho && ox < a.Wo));
                oy = oy < a.Ho ? oy : a.Ho - 1;
                ox = ox < a.Wo ? ox : a.Wo - 1;
                pix[q] = oy * a.Wo + ox;
              }
              float rv[8];
              if (has_res) {
#pragma unroll
                for (int q = 0; q < 8; ++q) rv[q] = resb[pix[q] * a.res_cs + coc];
              }
#pragma unroll
              for (int q = 0; q < 8; ++q) {
                const int r = half * 8 + q;
                float v = acc[m][n][r] + bvv;
                if (has_res) v += rv[q];
                if (a.relu) v = fmaxf(v, 0.f);
                if (ok[q]) outb[pix[q] * a.out_cs + co] = v;
                acc[m][n][r] = 0.f;
              }
            }
          }
        }
      }
    }
    w = nw;
    c0 = nc;
  }
}

static int g_num_cus = 0;

template <int KS, int S, int TH, int TW, int WAVES_M, int MT, int WAVES_N, int NTW, int CK>
static hipError_t launch_cfg(const ConvArgs& a, hipStream_t s) {
  constexpr int PH = (TH - 1) * S + KS, PW = (TW - 1) * S + KS;
  constexpr size_t lds = (size_t)PH * PW * (CK + 4) * sizeof(float);
  constexpr int NTHREADS = WAVES_M * WAVES_N * 64;
  auto kern = conv_mfma_kernel<KS, S, TH, TW, WAVES_M, MT, WAVES_N, NTW, CK>;
  static int occ = 0;
  if (!occ) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    if (!g_num_cus) {
      int dev = 0;
      hipDeviceProp_t prop;
      if ((e = hipGetDevice(&dev)) != hipSuccess) return e;
      if ((e = hipGetDeviceProperties(&prop, dev)) != hipSuccess) return e;
      g_num_cus = prop.multiProcessorCount;
    }
    int o = 0;
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&o, kern, NTHREADS, lds);
    if (e != hipSuccess) return e;
    occ = o < 1 ? 1 : (o > 4 ? 4 : o);
  }
  ConvWork wk;
  wk.tiles_x = (a.Wo + TW - 1) / TW;
  wk.tiles_per_frame = wk.tiles_x * ((a.Ho + TH - 1) / TH);
  wk.n_tiles_total = wk.tiles_per_frame * a.B;
  wk.nblk = (a.n_tiles + WAVES_N * NTW - 1) / (WAVES_N * NTW);
  wk.total = wk.n_tiles_total * wk.nblk * a.groups;
  // k workgroups per CU, k <= occupancy, minimising the busiest CU's item count (ties -> larger k)
  int best_k = 1;
  long best_cost = -1;
  for (int k = 1; k <= occ; ++k) {
    const long slots = (long)g_num_cus * k;
    const long cost = ((wk.total + slots - 1) / slots) * k;
    if (best_cost < 0 || cost <= best_cost) { best_cost = cost; best_k = k; }
  }
  long grid = (long)g_num_cus * best_k;
  if (grid > wk.total) grid = wk.total;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NTHREADS), lds, s, a, wk);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Wave-specialised variant: NCW = WAVES_M*WAVES_N compute waves + NLW loader waves per workgroup, one
// persistent workgroup per CU, two LDS patch buffers.  Loader waves stream the next (item, Cin-chunk)
// patch HBM -> registers -> LDS while the compute waves run the MFMAs of the current chunk; the compute
// waves' vmcnt queue therefore only ever holds B-fragment loads (the in-order vmcnt counter otherwise
// makes every fragment wait also wait for the bulk patch loads).  One workgroup barrier per chunk.
// ------------------------------------------------------------------------------------------------
template <int KS, int S, int TH, int TW, int WAVES_M, int MT, int WAVES_N, int NTW, int CK, int NLW>
__global__ __launch_bounds__((WAVES_M * WAVES_N + NLW) * 64, 1) void conv_ws_kernel(const ConvArgs a, const ConvWork wk) {
  constexpr int PH = (TH - 1) * S + KS, PW = (TW - 1) * S + KS, CP = CK + 4, PAD = KS / 2;
  constexpr int NCW = WAVES_M * WAVES_N;
  constexpr int NLT = NLW * 64;
  constexpr int TP = TH * TW;
  constexpr int NLOAD = PH * PW * (CK / 4);
  constexpr int NLD = (NLOAD + NLT - 1) / NLT;
  constexpr int TAPS = KS * KS;
  constexpr int BUF = PH * PW * CP;   // floats per LDS buffer
  static_assert(TP == 32 * MT * WAVES_M, "tile pixels must equal 32*MT*WAVES_M");
  extern __shared__ f32x4 smem4[];
  float* lds = reinterpret_cast<float*>(smem4);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cin_pad = a.cin8 * 8;
  const int nchunks = (cin_pad + CK - 1) / CK;
  const int my_items = wk.total > (int)blockIdx.x ? (wk.total - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
  const int ktotal = my_items * nchunks;
  if (a.phase_delay > 0 && blockIdx.x * 2 >= gridDim.x) {
    const long long t_end = clock64() + a.phase_delay;
    while (clock64() < t_end) __builtin_amdgcn_s_sleep(32);
  }

  if (wave >= NCW) {
    // ================================ loader waves ================================
    const int ltid = tid - NCW * 64;
    // loader waves are the younger waves on their SIMD: without priority their address VALU and VMEM
    // issue starve behind the MFMA stream of the compute wave they share the SIMD with
    __builtin_amdgcn_s_setprio(3);
    int w = blockIdx.x, c0 = 0;
    // per-item geometry (pixel offsets, halo validity) is computed once per work item; per chunk only
    // the channel offset changes, so a chunk costs the loader NLD loads + NLD LDS writes and little VALU
    int off[NLD];
    unsigned pixok = 0;    // bit i: element i's pixel is inside the image (and i is a real element)
    const float* __restrict__ inb = a.in;
    const int c4off = (ltid % (CK / 4)) * 4;   // NLT is a multiple of CK/4: same channel slot for every i
    static_assert(NLT % (CK / 4) == 0 && NLD <= 32, "loader geometry");
    for (int k = 0; k < ktotal; ++k) {
      if (c0 == 0) {
        const int tile = w % wk.n_tiles_total;
        const int g = (w / wk.n_tiles_total) / wk.nblk;
        const int b = tile / wk.tiles_per_frame;
        const int t = tile - b * wk.tiles_per_frame;
        const int ty0 = (t / wk.tiles_x) * TH, tx0 = (t % wk.tiles_x) * TW;
        inb = a.in + (size_t)b * a.H * a.W * a.in_cs + a.in_coff + g * a.Cin;
        pixok = 0;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
          const int idx = ltid + i * NLT;
          const int pix = idx / (CK / 4);
          const int iy = ty0 * S - PAD + pix / PW, ix = tx0 * S - PAD + pix % PW;
          const bool ok = (idx < NLOAD) && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
          const int iyc = iy < 0 ? 0 : (iy >= a.H ? a.H - 1 : iy), ixc = ix < 0 ? 0 : (ix >= a.W ? a.W - 1 : ix);
          off[i] = (iyc * a.W + ixc) * a.in_cs;
          pixok |= (ok ? 1u : 0u) << i;
        }
      }
      // every load in flight, then mask + LDS writes.  The sched_barriers keep hipcc from interleaving
      // waits between the loads (it otherwise serialises them in 4-5 rounds).
      const int c = c0 + c4off;
      const bool cok = c < a.Cin;
      const int cc = cok ? c : 0;
      f32x4 stage[NLD];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < NLD; ++i) stage[i] = *reinterpret_cast<const f32x4*>(inb + off[i] + cc);
      __builtin_amdgcn_sched_barrier(0);
      float* dst = lds + (k & 1) * BUF;
      const bool ragged_c = cok && (c + 3 >= a.Cin);   // this float4 straddles Cin
#pragma unroll
      for (int i = 0; i < NLD; ++i) {
        const int idx = ltid + i * NLT;
        f32x4 v = stage[i];
        if (ragged_c) {
          if (c + 1 >= a.Cin) v[1] = 0.f;
          if (c + 2 >= a.Cin) v[2] = 0.f;
          v[3] = 0.f;
        }
        if (!cok || !((pixok >> i) & 1u)) v = f32x4{0.f, 0.f, 0.f, 0.f};
        if (idx < NLOAD) *reinterpret_cast<f32x4*>(dst + (idx / (CK / 4)) * CP + c4off) = v;
      }
      // barrier k: buffer k&1 is full; the compute waves have finished reading it two chunks ago
      __syncthreads();
      c0 += CK;
      if (c0 >= cin_pad) { c0 = 0; w += gridDim.x; }
    }
    __syncthreads();   // matches the compute waves' final barrier
    return;
  }

  // ================================ compute waves ================================
  const int wm = wave % WAVES_M, wn = wave / WAVES_M;
  const int li = lane & 31, lh = lane >> 5;
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
  const size_t tap_stride = (size_t)a.cin8 * a.n_tiles * 256;
  const size_t step_stride = (size_t)a.n_tiles * 256;

  int w = blockIdx.x, c0 = 0;
  const bool stamp = a.dbg && blockIdx.x == 0 && tid == 0;
  int ns_ = 0;
  if (stamp) a.dbg[ns_++] = clock64();
  __syncthreads();   // barrier 0: chunk 0 is in buffer 0
  if (stamp) a.dbg[ns_++] = clock64();
  for (int k = 0; k < ktotal; ++k) {
    const float* patch = lds + (k & 1) * BUF;
    const int rest = w / wk.n_tiles_total;
    const int g = rest / wk.nblk;
    const int n_tile0 = ((rest % wk.nblk) * WAVES_N + wn) * NTW;
    const bool wave_active = n_tile0 < a.n_tiles;
    const bool last_chunk = c0 + CK >= cin_pad;
    if (wave_active) {
      const int nsteps = (cin_pad - c0 < CK ? cin_pad - c0 : CK) / 8;
      const float* __restrict__ wchunk = a.w + (size_t)g * TAPS * tap_stride + (size_t)n_tile0 * 256 + lane * 4 +
                                         (size_t)(c0 / 8) * step_stride;
      f32x4 av[2][MT], bv[2][NTW];
#pragma unroll
      for (int m = 0; m < MT; ++m) av[0][m] = *reinterpret_cast<const f32x4*>(patch + aoff[m]);
#pragma unroll
      for (int n = 0; n < NTW; ++n) bv[0][n] = *reinterpret_cast<const f32x4*>(wchunk + (size_t)n * 256);
      for (int s = 0; s < nsteps; ++s) {
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
          const int cur = tap & 1, nxt = cur ^ 1;
          const int ntap = tap + 1 < TAPS ? tap + 1 : 0;
          const int ns = tap + 1 < TAPS ? s : (s + 1 < nsteps ? s + 1 : s);
          const int nky = ntap / KS, nkx = ntap % KS;
#pragma unroll
          for (int m = 0; m < MT; ++m)
            av[nxt][m] = *reinterpret_cast<const f32x4*>(patch + aoff[m] + (nky * PW + nkx) * CP + ns * 8);
#pragma unroll
          for (int n = 0; n < NTW; ++n)
            bv[nxt][n] = *reinterpret_cast<const f32x4*>(wchunk + (size_t)ntap * tap_stride +
                                                         (size_t)ns * step_stride + (size_t)n * 256);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
              for (int n = 0; n < NTW; ++n)
                acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur][m][j], bv[cur][n][j], acc[m][n], 0, 0, 0);
        }
        if (TAPS & 1) {
#pragma unroll
          for (int m = 0; m < MT; ++m) av[0][m] = av[1][m];
#pragma unroll
          for (int n = 0; n < NTW; ++n) bv[0][n] = bv[1][n];
        }
      }
    }
    if (stamp && ns_ < 60) a.dbg[ns_++] = clock64();
    // barrier k+1: this buffer may be refilled (chunk k+2), and buffer (k+1)&1 holds chunk k+1
    __syncthreads();
    if (stamp && ns_ < 60) a.dbg[ns_++] = clock64();
    if (wave_active && last_chunk) {
      const int tile = w % wk.n_tiles_total;
      const int b = tile / wk.tiles_per_frame;
      const int t = tile - b * wk.tiles_per_frame;
      const int ty0 = (t / wk.tiles_x) * TH, tx0 = (t % wk.tiles_x) * TW;
      const float* __restrict__ bias = a.bias + (size_t)g * a.n_tiles * 32 + (size_t)b * a.bias_fstride;
      const bool full_tile = (ty0 + TH <= a.Ho) && (tx0 + TW <= a.Wo);
      const bool has_res = a.res != nullptr;
      float* __restrict__ outb = a.out + (size_t)b * a.Ho * a.Wo * a.out_cs + a.out_coff + g * a.Cout;
      const float* __restrict__ resb =
          has_res ? a.res + (size_t)b * a.Ho * a.Wo * a.res_cs + a.res_coff + g * a.Cout : nullptr;
      // Fast path: 4x4 transposes inside lane quads (DPP) turn "lane = cout, reg = pixel" into
      // "lane = pixel, 4 regs = 4 consecutive couts", so bias/residual/output move as dwordx4
      // (4x fewer VMEM instructions, 128 B contiguous per 8 lanes).  Needs 16-byte aligned channel slices
      // and a full 32-cout tile; otherwise the scalar path below runs.
      const bool vec_ok = ((a.out_coff + g * a.Cout) % 4 == 0) && (a.out_cs % 4 == 0) &&
                          (!has_res || (((a.res_coff + g * a.Cout) % 4 == 0) && (a.res_cs % 4 == 0))) &&
                          ((n_tile0 + NTW) * 32 <= a.Cout);
      if (vec_ok) {
        const int lq = li >> 2, lj = li & 3;
        int pixv[MT][4];
        unsigned okv[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          okv[m] = 0;
#pragma unroll
          for (int gq = 0; gq < 4; ++gq) {
            const int p = (wm * MT + m) * 32 + 8 * gq + 4 * lh + lj;
            int oy = ty0 + p / TW, ox = tx0 + p % TW;
            okv[m] |= ((full_tile || (oy < a.Ho && ox < a.Wo)) ? 1u : 0u) << gq;
            oy = oy < a.Ho ? oy : a.Ho - 1;
            ox = ox < a.Wo ? ox : a.Wo - 1;
            pixv[m][gq] = oy * a.Wo + ox;
          }
        }
        f32x4 rv4[MT][NTW][4];
        if (has_res) {
#pragma unroll
          for (int n = 0; n < NTW; ++n)
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
              for (int gq = 0; gq < 4; ++gq)
                rv4[m][n][gq] = *reinterpret_cast<const f32x4*>(resb + pixv[m][gq] * a.res_cs + (n_tile0 + n) * 32 + 4 * lq);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int n = 0; n < NTW; ++n) {
          const f32x4 b4 = *reinterpret_cast<const f32x4*>(bias + (n_tile0 + n) * 32 + 4 * lq);
#pragma unroll
          for (int m = 0; m < MT; ++m) {
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
              float x0 = acc[m][n][4 * gq], x1 = acc[m][n][4 * gq + 1], x2 = acc[m][n][4 * gq + 2], x3 = acc[m][n][4 * gq + 3];
              // stage 1: exchange with lane^1 (quad_perm [1,0,3,2] = 0xB1)
              {
                const float s01 = (lj & 1) ? x0 : x1, s23 = (lj & 1) ? x2 : x3;
                const float r01 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s01), 0xB1, 0xF, 0xF, true));
                const float r23 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s23), 0xB1, 0xF, 0xF, true));
                if (lj & 1) { x0 = r01; x2 = r23; } else { x1 = r01; x3 = r23; }
              }
              // stage 2: exchange with lane^2 (quad_perm [2,3,0,1] = 0x4E)
              {
                const float s02 = (lj & 2) ? x0 : x2, s13 = (lj & 2) ? x1 : x3;
                const float r02 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s02), 0x4E, 0xF, 0xF, true));
                const float r13 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s13), 0x4E, 0xF, 0xF, true));
                if (lj & 2) { x0 = r02; x1 = r13; } else { x2 = r02; x3 = r13; }
              }
              f32x4 v = {x0 + b4[0], x1 + b4[1], x2 + b4[2], x3 + b4[3]};
              if (has_res) v += rv4[m][n][gq];
              if (a.relu) {
                v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
              }
              if ((okv[m] >> gq) & 1u)
                *reinterpret_cast<f32x4*>(outb + pixv[m][gq] * a.out_cs + (n_tile0 + n) * 32 + 4 * lq) = v;
              acc[m][n][4 * gq] = 0.f; acc[m][n][4 * gq + 1] = 0.f; acc[m][n][4 * gq + 2] = 0.f; acc[m][n][4 * gq + 3] = 0.f;
            }
          }
        }
      } else {
      // scalar path: every residual load in flight before the first store
      int pixo[MT][16];
      unsigned okm[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        okm[m] = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int p = (wm * MT + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          int oy = ty0 + p / TW, ox = tx0 + p % TW;
          okm[m] |= ((full_tile || (oy < a.Ho && ox < a.Wo)) ? 1u : 0u) << r;
          oy = oy < a.Ho ? oy : a.Ho - 1;
          ox = ox < a.Wo ? ox : a.Wo - 1;
          pixo[m][r] = oy * a.Wo + ox;
        }
      }
      float rv[MT][NTW][16];
      if (has_res) {
#pragma unroll
        for (int n = 0; n < NTW; ++n) {
          const int co = (n_tile0 + n) * 32 + li;
          const int coc = co < a.Cout ? co : 0;
#pragma unroll
          for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) rv[m][n][r] = resb[pixo[m][r] * a.res_cs + coc];
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int n = 0; n < NTW; ++n) {
        const int co = (n_tile0 + n) * 32 + li;
        const bool cok = co < a.Cout;
        const float bvv = cok ? bias[co] : 0.f;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float v = acc[m][n][r] + bvv;
            if (has_res) v += rv[m][n][r];
            if (a.relu) v = fmaxf(v, 0.f);
            if (cok && ((okm[m] >> r) & 1u)) outb[pixo[m][r] * a.out_cs + co] = v;
            acc[m][n][r] = 0.f;
          }
        }
      }
      }   // scalar path
    }
    if (stamp && last_chunk && ns_ < 60) a.dbg[ns_++] = clock64();
    c0 += CK;
    if (c0 >= cin_pad) { c0 = 0; w += gridDim.x; }
  }
  if (stamp) a.dbg[63] = ns_;
}

template <int KS, int S, int TH, int TW, int WAVES_M, int MT, int WAVES_N, int NTW, int CK, int NLW>
static hipError_t launch_ws(const ConvArgs& a, hipStream_t s) {
  constexpr int PH = (TH - 1) * S + KS, PW = (TW - 1) * S + KS;
  constexpr size_t lds = 2 * (size_t)PH * PW * (CK + 4) * sizeof(float);
  static_assert(lds <= 160 * 1024, "two patch buffers must fit the 160 KiB LDS");
  constexpr int NTHREADS = (WAVES_M * WAVES_N + NLW) * 64;
  auto kern = conv_ws_kernel<KS, S, TH, TW, WAVES_M, MT, WAVES_N, NTW, CK, NLW>;
  static bool init = false;
  if (!init) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    if (!g_num_cus) {
      int dev = 0;
      hipDeviceProp_t prop;
      if ((e = hipGetDevice(&dev)) != hipSuccess) return e;
      if ((e = hipGetDeviceProperties(&prop, dev)) != hipSuccess) return e;
      g_num_cus = prop.multiProcessorCount;
    }
    init = true;
  }
  ConvWork wk;
  wk.tiles_x = (a.Wo + TW - 1) / TW;
  wk.tiles_per_frame = wk.tiles_x * ((a.Ho + TH - 1) / TH);
  wk.n_tiles_total = wk.tiles_per_frame * a.B;
  wk.nblk = (a.n_tiles + WAVES_N * NTW - 1) / (WAVES_N * NTW);
  wk.total = wk.n_tiles_total * wk.nblk * a.groups;
  // k persistent workgroups per CU (k = 2 only if two double-buffers fit the LDS), minimising the busiest
  // CU's item count; ties go to the larger k (more waves to hide fragment latency)
  const int max_k = lds <= 78 * 1024 ? 2 : 1;
  int best_k = 1;
  long best_cost = -1;
  for (int k = 1; k <= max_k; ++k) {
    const long slots = (long)g_num_cus * k;
    const long cost = ((wk.total + slots - 1) / slots) * k;
    if (best_cost < 0 || cost <= best_cost) { best_cost = cost; best_k = k; }
  }
  long grid = (long)g_num_cus * best_k;
  if (grid > wk.total) grid = wk.total;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NTHREADS), lds, s, a, wk);
  return hipGetLastError();
}

// Tile selection.  N32: one 32-cout tile per wave (Cout <= 32); N64: two.
// Small frames (<=16x16 outputs) take the 8x16 pixel tile so a batch still fills 256 CUs.
hipError_t launch_conv(ConvArgs a, hipStream_t s) {
  a.dbg = g_dbg;
  a.phase_delay = g_phase_delay;
  const bool n32 = a.n_tiles == 1;
  const bool small = (a.Ho * a.Wo <= 256) || (a.Ho % 16 != 0) || (a.Wo % 16 != 0);
  if (g_force_cfg != 100) {
    if (a.ks == 3 && a.stride == 1) {
      if (n32) return small ? launch_ws<3, 1, 8, 16, 4, 1, 1, 1, 32, 2>(a, s) : launch_ws<3, 1, 16, 16, 4, 2, 1, 1, 32, 2>(a, s);
      if (g_force_cfg == 201) return launch_ws<3, 1, 16, 16, 4, 2, 1, 2, 32, 4>(a, s);
      if (g_force_cfg == 202) return launch_ws<3, 1, 8, 16, 2, 2, 2, 1, 32, 2>(a, s);
      return small ? launch_ws<3, 1, 8, 16, 2, 2, 2, 1, 32, 2>(a, s) : launch_ws<3, 1, 16, 16, 4, 2, 1, 2, 32, 2>(a, s);
    }
    if (a.ks == 3 && a.stride == 2) {
      if (n32) return launch_ws<3, 2, 8, 16, 4, 1, 1, 1, 16, 2>(a, s);
      return launch_ws<3, 2, 8, 16, 2, 2, 2, 1, 16, 2>(a, s);
    }
    if (a.ks == 1 && a.stride == 1) {
      if (n32) return small ? launch_ws<1, 1, 8, 16, 4, 1, 1, 1, 64, 2>(a, s) : launch_ws<1, 1, 16, 16, 4, 2, 1, 1, 64, 4>(a, s);
      return small ? launch_ws<1, 1, 8, 16, 2, 2, 2, 1, 64, 2>(a, s) : launch_ws<1, 1, 16, 16, 4, 2, 1, 2, 64, 4>(a, s);
    }
    return hipErrorInvalidValue;
  }
  if (a.ks == 3 && a.stride == 1) {
    if (n32) return small ? launch_cfg<3, 1, 8, 16, 4, 1, 1, 1, 32>(a, s) : launch_cfg<3, 1, 16, 16, 4, 2, 1, 1, 32>(a, s);
    if (g_force_cfg == 1) return launch_cfg<3, 1, 8, 16, 2, 2, 2, 1, 32>(a, s);
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
