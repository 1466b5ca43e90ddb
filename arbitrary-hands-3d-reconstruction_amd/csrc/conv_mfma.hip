// NHWC fp32 convolutions on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32 products,
// fp32 accumulate), im2col-free.
//
// Work item = (output tile of one frame, block of output channels, group).  Every kernel here is
// wave-specialised and persistent: a workgroup = NCW compute waves + NLW loader waves, one workgroup
// per CU walks the work items.  The loader waves stream the halo'd input patch of the next
// (item, Cin-chunk) HBM -> registers -> LDS (layout [PH][PW][CK+4]; the 4-float pad keeps the compute
// waves' ds_read_b128 of 32 neighbouring pixels (nearly) bank-conflict free) into the second of two LDS
// buffers while the compute waves run the MFMAs of the current chunk: the compute waves' in-order vmcnt
// queue then only ever holds weight-fragment loads.  One workgroup barrier per chunk.
// Weights are pre-packed on the host in MFMA B-fragment order ([tap][ci/8][cout/32][lane][4]) so a wave
// fetches one contiguous 1 KiB line per (tap, 8-channel step, 32-cout tile) straight from L2:
//   A[i = lane&31][k = lane>>5] = patch[pixel i][ci = 8s + 4*(lane>>5) + j]
//   B[k = lane>>5][n = lane&31] = W[cout n][ci = 8s + 4*(lane>>5) + j]          j = 0..3
// Fragments of step i+1 are fetched before the MFMAs of step i (sched_barrier pins that order).
// Epilogue: folded-BN bias (+ per-frame bias), residual, ReLU; 4x4 DPP transposes inside lane quads turn
// "lane = cout" into "lane = pixel, 4 regs = 4 couts" so residual/output move as dwordx4.
//
//  * conv_ws_kernel   : direct convolution, 3x3 (stride 1/2) and 1x1.
//  * conv_wino_kernel : 3x3 stride 1 as Winograd F(2,3) along x (1.5x fewer MFMAs): per output-pixel PAIR
//                       4 products instead of 6 per kernel row; input transform in registers on the A
//                       fragments (v0=d0-d2, v1=d1+d2, v2=d2-d1, v3=d1-d3), weights pre-transformed on the
//                       host (G g), output transform (y0=m0+m1+m2, y1=m1-m2-m3) in the epilogue.
#include "kernels.h"
#include <cstdio>
#include <cstdlib>
#include <type_traits>

namespace acrmi {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

static int g_force_cfg = -1;
void conv_force_cfg(int cfg) { g_force_cfg = cfg; }
static long long* g_dbg = nullptr;
void conv_set_debug(long long* dbg) { g_dbg = dbg; }
static int g_phase_delay = 0;
void conv_set_phase_delay(int cycles) { g_phase_delay = cycles; }
static int g_num_cus = 0;

struct ConvWork {
  int tiles_x, tiles_per_frame, n_tiles_total, nblk, total;
  // ceil(2^40 / d) for the four divisors above: item -> (n-block, tile, group, frame, tile row/col) on the scalar
  // unit (hipcc lowers a 32-bit division of uniform values to ~25 VALU instructions, and VALU slots next to a
  // saturated matrix pipe are the scarce resource).  Exact while n * d < 2^40.
  unsigned long long m_tiles_x, m_tiles_per_frame, m_n_tiles_total, m_nblk;
};

static unsigned long long div_magic(int d) { return ((1ull << 40) + (unsigned long long)d - 1) / (unsigned long long)d; }
static void set_magics(ConvWork& wk) {
  wk.m_tiles_x = div_magic(wk.tiles_x);
  wk.m_tiles_per_frame = div_magic(wk.tiles_per_frame);
  wk.m_n_tiles_total = div_magic(wk.n_tiles_total);
  wk.m_nblk = div_magic(wk.nblk);
}
__device__ __forceinline__ int fdiv(int n, unsigned long long magic) {
  return (int)(((unsigned long long)(unsigned)n * magic) >> 40);
}

// work item -> coordinates
struct ItemPos {
  int rest, g, b, ty, tx;   // n-block, group, frame, tile row, tile column
};
__device__ __forceinline__ ItemPos item_pos(const ConvWork& wk, int w) {
  ItemPos p;
  const int q1 = fdiv(w, wk.m_nblk);
  p.rest = w - q1 * wk.nblk;
  p.g = fdiv(q1, wk.m_n_tiles_total);
  const int tile = q1 - p.g * wk.n_tiles_total;
  p.b = fdiv(tile, wk.m_tiles_per_frame);
  const int t = tile - p.b * wk.tiles_per_frame;
  p.ty = fdiv(t, wk.m_tiles_x);
  p.tx = t - p.ty * wk.tiles_x;
  return p;
}

// ------------------------------------------------------------------------------------------------
// loader waves: fill LDS buffer (k & 1) with the patch of chunk k, one workgroup barrier per chunk
// ------------------------------------------------------------------------------------------------
struct NoHook {
  __device__ __forceinline__ void operator()(int, int, bool) const {}
};

// hook(item, item_seq, is_last_chunk) runs on the loader waves after chunk k's patch is written to LDS and
// before barrier k (used to pre-stage the item's residual tile next to its last chunk).
template <int KS, int S, int TH, int TW, int CK, int NLW, class Hook = NoHook>
__device__ __forceinline__ void ws_loader(const ConvArgs& a, const ConvWork& wk, float* lds, int ltid, int ktotal,
                                          int cin_pad, Hook hook = Hook()) {
  constexpr int PH = (TH - 1) * S + KS, PW = (TW - 1) * S + KS, CP = CK + 4, PAD = KS / 2;
  constexpr int NLT = NLW * 64;
  constexpr int NLOAD = PH * PW * (CK / 4);
  constexpr int NLD = (NLOAD + NLT - 1) / NLT;
  constexpr int BUF = PH * PW * CP;
  static_assert(NLT % (CK / 4) == 0 && NLD <= 32, "loader geometry");
  // loader waves are the younger waves on their SIMD: without priority their VMEM issue trails the
  // MFMA stream of the compute wave they share the SIMD with and the patch arrives late
  __builtin_amdgcn_s_setprio(3);
  int w = blockIdx.x, c0 = 0;
  // per-item geometry (pixel offsets, halo validity) is computed once per work item; per chunk only the
  // channel offset changes, so a chunk costs the loader NLD loads + NLD LDS writes and little VALU
  int off[NLD];
  unsigned pixok = 0;
  const float* __restrict__ inb = a.in;
  const int c4off = (ltid % (CK / 4)) * 4;
  const int nchunks_l = (cin_pad + CK - 1) / CK;
  for (int k = 0; k < ktotal; ++k) {
    if (c0 == 0) {
      const int tile = (w / wk.nblk) % wk.n_tiles_total;
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
        off[i] = (iyc * a.W + ixc) * a.in_cs;   // per-frame offset < 2^31 floats
        pixok |= (ok ? 1u : 0u) << i;
      }
    }
    const int c = c0 + c4off;
    const bool cok = c < a.Cin;
    const int cc = cok ? c : 0;
    f32x4 stage[NLD];
    // every load in flight before the first LDS write (hipcc otherwise serialises them in rounds)
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
    hook(w, k / nchunks_l, c0 + CK >= cin_pad);
    // barrier k: buffer k&1 is full; the compute waves finished reading it two chunks ago
    __syncthreads();
    c0 += CK;
    if (c0 >= cin_pad) { c0 = 0; w += gridDim.x; }
  }
  __syncthreads();   // matches the compute waves' final barrier
}

// ------------------------------------------------------------------------------------------------
// epilogue of one 32 (M slots) x 32 (couts) accumulator tile.  slot2pix(slot, &ok) -> oy*Wo+ox (clamped).
// ------------------------------------------------------------------------------------------------
struct EpiCtx {
  const float* bias;   // this frame/group's bias row
  const float* resb;   // per-frame/group residual base or null
  float* outb;         // per-frame/group output base
  int res_cs, out_cs, Cout, relu;
  bool vec_align;      // channel slices 16-byte aligned
};

__device__ __forceinline__ float dpp_xor1(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
}
__device__ __forceinline__ float dpp_xor2(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
}

// vector path, phase 1: residual float4 of the 4 (pixel, cout-quad) cells this lane will own
template <class Slot2Pix>
__device__ __forceinline__ void epi_load_res(const EpiCtx& e, Slot2Pix slot2pix, int slot_base, int co0, int li, int lh,
                                             f32x4 (&rv)[4]) {
  const int lq = li >> 2, lj = li & 3;
#pragma unroll
  for (int gq = 0; gq < 4; ++gq) {
    bool ok;
    const int pix = slot2pix(slot_base + 8 * gq + 4 * lh + lj, ok);
    rv[gq] = *reinterpret_cast<const f32x4*>(e.resb + pix * e.res_cs + co0 + 4 * lq);
  }
}

// vector path, phase 2: transpose, bias, residual, ReLU, dwordx4 store
template <class Slot2Pix>
__device__ __forceinline__ void epi_store_vec(const EpiCtx& e, Slot2Pix slot2pix, int slot_base, int co0, int li, int lh,
                                              const f32x16& acc, const f32x4 (&rv)[4], bool has_res) {
  const int lq = li >> 2, lj = li & 3;
  const f32x4 b4 = *reinterpret_cast<const f32x4*>(e.bias + co0 + 4 * lq);
#pragma unroll
  for (int gq = 0; gq < 4; ++gq) {
    float x0 = acc[4 * gq], x1 = acc[4 * gq + 1], x2 = acc[4 * gq + 2], x3 = acc[4 * gq + 3];
    {   // exchange with lane^1
      const float r01 = dpp_xor1((lj & 1) ? x0 : x1), r23 = dpp_xor1((lj & 1) ? x2 : x3);
      if (lj & 1) { x0 = r01; x2 = r23; } else { x1 = r01; x3 = r23; }
    }
    {   // exchange with lane^2
      const float r02 = dpp_xor2((lj & 2) ? x0 : x2), r13 = dpp_xor2((lj & 2) ? x1 : x3);
      if (lj & 2) { x0 = r02; x1 = r13; } else { x2 = r02; x3 = r13; }
    }
    f32x4 v = {x0 + b4[0], x1 + b4[1], x2 + b4[2], x3 + b4[3]};
    if (has_res) v += rv[gq];
    if (e.relu) {
      v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
    }
    bool ok;
    const int pix = slot2pix(slot_base + 8 * gq + 4 * lh + lj, ok);
    if (ok) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(e.outb + pix * e.out_cs + co0 + 4 * lq));
  }
}

// scalar path (ragged channel counts / unaligned channel slices)
template <class Slot2Pix>
__device__ __forceinline__ void epi_store_scalar(const EpiCtx& e, Slot2Pix slot2pix, int slot_base, int co0, int li, int lh,
                                                 const f32x16& acc, bool has_res) {
  const int co = co0 + li;
  const bool cok = co < e.Cout;
  const int coc = cok ? co : 0;
  const float bv = cok ? e.bias[co] : 0.f;
  int pix[16];
  unsigned okm = 0;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    bool ok;
    pix[r] = slot2pix(slot_base + (r & 3) + 8 * (r >> 2) + 4 * lh, ok);
    okm |= (ok ? 1u : 0u) << r;
  }
  float rv[16];
  if (has_res) {
#pragma unroll
    for (int r = 0; r < 16; ++r) rv[r] = e.resb[pix[r] * e.res_cs + coc];
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    float v = acc[r] + bv;
    if (has_res) v += rv[r];
    if (e.relu) v = fmaxf(v, 0.f);
    if (cok && ((okm >> r) & 1u)) e.outb[pix[r] * e.out_cs + co] = v;
  }
}

// ------------------------------------------------------------------------------------------------
// direct convolution
// ------------------------------------------------------------------------------------------------
template <int KS, int S, int TH, int TW, int WAVES_M, int MT, int WAVES_N, int NTW, int CK, int NLW>
__global__ __launch_bounds__((WAVES_M * WAVES_N + NLW) * 64, 1) void conv_ws_kernel(const ConvArgs a, const ConvWork wk) {
  constexpr int PH = (TH - 1) * S + KS, PW = (TW - 1) * S + KS, CP = CK + 4;
  constexpr int NCW = WAVES_M * WAVES_N;
  constexpr int TAPS = KS * KS;
  constexpr int BUF = PH * PW * CP;
  static_assert(TH * TW == 32 * MT * WAVES_M, "tile pixels must equal 32*MT*WAVES_M");
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
    ws_loader<KS, S, TH, TW, CK, NLW>(a, wk, lds, tid - NCW * 64, ktotal, cin_pad);
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
    // item order: N-block fastest, then tile, then group - the N-blocks of one tile run at the same time on
    // neighbouring workgroups, so the tile's input patch is fetched from HBM once and re-read from MALL/L2
    const int rest = w % wk.nblk;
    const int g = w / (wk.nblk * wk.n_tiles_total);
    const int n_tile0 = (rest * WAVES_N + wn) * NTW;
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
          // prefetch the fragments of the next (s, tap); the very last step re-reads a valid address
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
        if (TAPS & 1) {   // odd tap count: the double-buffer parity flips every s; re-align
#pragma unroll
          for (int m = 0; m < MT; ++m) av[0][m] = av[1][m];
#pragma unroll
          for (int n = 0; n < NTW; ++n) bv[0][n] = bv[1][n];
        }
      }
    }
    if (stamp && ns_ < 60) a.dbg[ns_++] = clock64();
    // barrier k+1: this buffer may be refilled (chunk k+2); buffer (k+1)&1 holds chunk k+1
    __syncthreads();
    if (stamp && ns_ < 60) a.dbg[ns_++] = clock64();
    if (wave_active && last_chunk) {
      const int tile = (w / wk.nblk) % wk.n_tiles_total;
      const int b = tile / wk.tiles_per_frame;
      const int t = tile - b * wk.tiles_per_frame;
      const int ty0 = (t / wk.tiles_x) * TH, tx0 = (t % wk.tiles_x) * TW;
      const bool has_res = a.res != nullptr;
      EpiCtx e;
      e.bias = a.bias + (size_t)g * a.n_tiles * 32 + (size_t)b * a.bias_fstride;
      e.outb = a.out + (size_t)b * a.Ho * a.Wo * a.out_cs + a.out_coff + g * a.Cout;
      e.resb = has_res ? a.res + (size_t)b * a.Ho * a.Wo * a.res_cs + a.res_coff + g * a.Cout : nullptr;
      e.res_cs = a.res_cs; e.out_cs = a.out_cs; e.Cout = a.Cout; e.relu = a.relu;
      e.vec_align = ((a.out_coff + g * a.Cout) % 4 == 0) && (a.out_cs % 4 == 0) &&
                    (!has_res || (((a.res_coff + g * a.Cout) % 4 == 0) && (a.res_cs % 4 == 0)));
      const bool full_tile = (ty0 + TH <= a.Ho) && (tx0 + TW <= a.Wo);
      auto slot2pix = [&](int p, bool& ok) -> int {
        int oy = ty0 + p / TW, ox = tx0 + p % TW;
        ok = full_tile || (oy < a.Ho && ox < a.Wo);
        oy = oy < a.Ho ? oy : a.Ho - 1;
        ox = ox < a.Wo ? ox : a.Wo - 1;
        return oy * a.Wo + ox;
      };
      f32x4 rv[MT][NTW][4];
      bool vec[NTW];
#pragma unroll
      for (int n = 0; n < NTW; ++n) {
        vec[n] = e.vec_align && ((n_tile0 + n + 1) * 32 <= a.Cout);
        if (vec[n] && has_res) {
#pragma unroll
          for (int m = 0; m < MT; ++m) epi_load_res(e, slot2pix, (wm * MT + m) * 32, (n_tile0 + n) * 32, li, lh, rv[m][n]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int n = 0; n < NTW; ++n)
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          if (vec[n]) epi_store_vec(e, slot2pix, (wm * MT + m) * 32, (n_tile0 + n) * 32, li, lh, acc[m][n], rv[m][n], has_res);
          else epi_store_scalar(e, slot2pix, (wm * MT + m) * 32, (n_tile0 + n) * 32, li, lh, acc[m][n], has_res);
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
        }
    }
    if (stamp && last_chunk && ns_ < 60) a.dbg[ns_++] = clock64();
    c0 += CK;
    if (c0 >= cin_pad) { c0 = 0; w += gridDim.x; }
  }
  if (stamp) a.dbg[63] = ns_;
}

// ------------------------------------------------------------------------------------------------
// 3x3 stride-1 convolution as Winograd F(2,3) along x.  M slots are output-pixel PAIRS (y, 2p | 2p+1);
// each compute wave owns 32 pairs x 32 couts with 4 position accumulators (64 registers).
// "taps" of the packed weights = 3 (ky) x 4 (positions): U[ky][v] = sum_kx G[v][kx] w[ky][kx].
// ------------------------------------------------------------------------------------------------
template <int TH, int TW, int WAVES_M, int WAVES_N, int CK, int NLW, int ABL = 0, int MINW = 1>
__global__ __launch_bounds__((WAVES_M * WAVES_N + NLW) * 64, MINW) void conv_wino_kernel(const ConvArgs a, const ConvWork wk) {
  constexpr int PH = TH + 2, PW = TW + 2, CP = CK + 4;
  constexpr int NCW = WAVES_M * WAVES_N;
  constexpr int BUF = PH * PW * CP;
  constexpr int PPR = TW / 2;   // pairs per tile row
  static_assert(TH * PPR == 32 * WAVES_M, "tile pairs must equal 32*WAVES_M");
  extern __shared__ f32x4 smem4[];
  float* lds = reinterpret_cast<float*>(smem4);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cin_pad = a.cin8 * 8;
  const int nchunks = (cin_pad + CK - 1) / CK;
  const int my_items = wk.total > (int)blockIdx.x ? (wk.total - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
  const int ktotal = my_items * nchunks;
  // Residual staging: [patch 0][patch 1][res 0][res 1].  With >= 2 Cin chunks per item the loader waves fetch the
  // item's residual tile (NCW waves x 2 outputs x 32 slots x 32 couts, rows XOR-swizzled on the 16-byte column)
  // next to its last chunk, so the compute waves' epilogue reads it from LDS instead of waiting ~2-3k cycles for
  // HBM.  (With one chunk per item the buffer of item i+2 would be refilled while item i is still being read.)
  constexpr int RES = NCW * 2 * 1024;
  float* res_base = lds + 2 * BUF;
  const bool has_res = a.res != nullptr;
  const bool stage_res = has_res && nchunks >= 2 && (a.out_cs % 4 == 0) && (a.res_cs % 4 == 0);
  if (wave >= NCW) {
    constexpr int NLT = NLW * 64;
    constexpr int NIT = (NCW * 512 + NLT - 1) / NLT;
    const int ltid = tid - NCW * 64;
    auto hook = [&](int w_item, int seq, bool last) {
      if (!stage_res || !last) return;
      const int tile = (w_item / wk.nblk) % wk.n_tiles_total;
      const int nb = w_item % wk.nblk;
      const int g = w_item / (wk.nblk * wk.n_tiles_total);
      if ((a.res_coff + g * a.Cout) % 4 != 0 || (a.out_coff + g * a.Cout) % 4 != 0) return;
      const int b = tile / wk.tiles_per_frame;
      const int t = tile - b * wk.tiles_per_frame;
      const int ty0 = (t / wk.tiles_x) * TH, tx0 = (t % wk.tiles_x) * TW;
      const float* __restrict__ resb = a.res + (size_t)b * a.Ho * a.Wo * a.res_cs + a.res_coff + g * a.Cout;
      float* dst = res_base + (seq & 1) * RES;
      f32x4 rv[NIT];
#pragma unroll
      for (int i = 0; i < NIT; ++i) {
        const int idx = ltid + i * NLT;
        const int wv = (idx >> 9) % NCW, o = (idx >> 8) & 1, slot = (idx >> 3) & 31, q = idx & 7;
        const int n_tile = nb * WAVES_N + wv / WAVES_M;
        const int p = (wv % WAVES_M) * 32 + slot;
        int oy = ty0 + p / PPR, ox = tx0 + 2 * (p % PPR) + o;
        oy = oy < a.Ho ? oy : a.Ho - 1;
        ox = ox < a.Wo ? ox : a.Wo - 1;
        const int co = ((n_tile + 1) * 32 <= a.Cout ? n_tile * 32 : 0) + 4 * q;
        rv[i] = *reinterpret_cast<const f32x4*>(resb + (oy * a.Wo + ox) * a.res_cs + co);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < NIT; ++i) {
        const int idx = ltid + i * NLT;
        const int slot = (idx >> 3) & 31, q = idx & 7;
        if (idx < NCW * 512) *reinterpret_cast<f32x4*>(dst + (idx >> 8) * 1024 + slot * 32 + 4 * (q ^ (slot & 7))) = rv[i];
      }
    };
    ws_loader<3, 1, TH, TW, CK, NLW>(a, wk, lds, ltid, ktotal, cin_pad, hook);
    return;
  }
  const int wm = wave % WAVES_M, wn = wave / WAVES_M;
  const int li = lane & 31, lh = lane >> 5;
  f32x16 acc[4];
#pragma unroll
  for (int v = 0; v < 4; ++v)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[v][r] = 0.f;
  const int pr = wm * 32 + li;                                   // this lane's pair inside the tile
  const int aoff = ((pr / PPR) * PW + 2 * (pr % PPR)) * CP + 4 * lh;   // window origin (row py, col 2*pxp)
  const size_t tap_stride = (size_t)a.cin8 * a.n_tiles * 256;
  const size_t step_stride = (size_t)a.n_tiles * 256;

  int w = blockIdx.x, c0 = 0;
  const bool stamp = a.dbg && blockIdx.x == 0 && tid == 0;
  int ns_ = 0;
  if (stamp) a.dbg[ns_++] = clock64();
  __syncthreads();   // barrier 0
  if (stamp) a.dbg[ns_++] = clock64();
  for (int k = 0; k < ktotal; ++k) {
    const float* patch = lds + (k & 1) * BUF;
    // item order: N-block fastest, then tile, then group - the N-blocks of one tile run at the same time on
    // neighbouring workgroups, so the tile's input patch is fetched from HBM once and re-read from MALL/L2
    const int rest = w % wk.nblk;
    const int g = w / (wk.nblk * wk.n_tiles_total);
    const int n_tile = rest * WAVES_N + wn;
    const bool wave_active = n_tile < a.n_tiles;
    const bool last_chunk = c0 + CK >= cin_pad;
    if (wave_active) {
      const int nsteps = (cin_pad - c0 < CK ? cin_pad - c0 : CK) / 8;
      const float* __restrict__ wchunk = a.w + (size_t)g * 12 * tap_stride + (size_t)n_tile * 256 + lane * 4 +
                                         (size_t)(c0 / 8) * step_stride;
      // Row groups g = (s, ky): three statically indexed fragment buffers (index = ky), LDS windows fetched one
      // group ahead, weight fragments two groups ahead; no register copies (the buffer of group g+2 is the one
      // group g-1 used).  All weight offsets are 32-bit: 12 loop-invariant (ky, position) offsets plus one
      // per-step offset, so a fragment address costs an s_add instead of a 64-bit multiply chain (the scalar
      // address arithmetic between MFMAs otherwise leaves bubbles in the matrix pipe).
      const int tap_i = (int)tap_stride, step_i = (int)step_stride;
      f32x4 d[3][4], bv[3][4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        d[0][c] = *reinterpret_cast<const f32x4*>(patch + aoff + c * CP);
        bv[0][c] = *reinterpret_cast<const f32x4*>(wchunk + c * tap_i);
        bv[1][c] = *reinterpret_cast<const f32x4*>(wchunk + (4 + c) * tap_i);
      }
      for (int s = 0; s < nsteps; ++s) {
        const int sn = s + 1 < nsteps ? s + 1 : s;      // next step (the tail re-reads a valid one)
        const int sb = s * step_i, snb = sn * step_i;   // weight offsets of this / the next step
        const int s8 = s * 8, sn8 = sn * 8;             // LDS channel offsets
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          {   // LDS window of the next group -> d[(ky+1)%3]
            const int k1 = ky == 2 ? 0 : ky + 1;
            const int so = ky == 2 ? sn8 : s8;
#pragma unroll
            for (int c = 0; c < 4; ++c)
              if (ABL != 2) d[(ky + 1) % 3][c] = *reinterpret_cast<const f32x4*>(patch + aoff + (k1 * PW + c) * CP + so);
          }
          {   // weight fragments two groups ahead -> bv[(ky+2)%3]
            const int k2 = (ky + 2) % 3;
            const int wo = ky == 0 ? sb : snb;
#pragma unroll
            for (int c = 0; c < 4; ++c)
              if (ABL != 1) bv[(ky + 2) % 3][c] = *reinterpret_cast<const f32x4*>(wchunk + (k2 * 4 + c) * tap_i + wo);
          }
          __builtin_amdgcn_sched_barrier(0);
          f32x4 v[4];
          if (ABL == 3) {
            v[0] = d[ky][0]; v[1] = d[ky][1]; v[2] = d[ky][2]; v[3] = d[ky][3];
          } else {
            v[0] = d[ky][0] - d[ky][2];
            v[1] = d[ky][1] + d[ky][2];
            v[2] = d[ky][2] - d[ky][1];
            v[3] = d[ky][1] - d[ky][3];
          }
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int p = 0; p < 4; ++p)
              acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[p][j], bv[ky][p][j], acc[p], 0, 0, 0);
        }
      }
    }
    if (stamp && ns_ < 60) a.dbg[ns_++] = clock64();
    __syncthreads();
    if (stamp && ns_ < 60) a.dbg[ns_++] = clock64();
    if (wave_active && last_chunk) {
      const int tile = (w / wk.nblk) % wk.n_tiles_total;
      const int b = tile / wk.tiles_per_frame;
      const int t = tile - b * wk.tiles_per_frame;
      const int ty0 = (t / wk.tiles_x) * TH, tx0 = (t % wk.tiles_x) * TW;
      const bool has_res = a.res != nullptr;
      EpiCtx e;
      e.bias = a.bias + (size_t)g * a.n_tiles * 32 + (size_t)b * a.bias_fstride;
      e.outb = a.out + (size_t)b * a.Ho * a.Wo * a.out_cs + a.out_coff + g * a.Cout;
      e.resb = has_res ? a.res + (size_t)b * a.Ho * a.Wo * a.res_cs + a.res_coff + g * a.Cout : nullptr;
      e.res_cs = a.res_cs; e.out_cs = a.out_cs; e.Cout = a.Cout; e.relu = a.relu;
      e.vec_align = ((a.out_coff + g * a.Cout) % 4 == 0) && (a.out_cs % 4 == 0) &&
                    (!has_res || (((a.res_coff + g * a.Cout) % 4 == 0) && (a.res_cs % 4 == 0)));
      const bool full_tile = (ty0 + TH <= a.Ho) && (tx0 + TW <= a.Wo);
      const bool vec = e.vec_align && ((n_tile + 1) * 32 <= a.Cout);
      // output transform: y0 = m0 + m1 + m2 (x = 2p), y1 = m1 - m2 - m3 (x = 2p + 1)
      f32x16 y[2];
      y[0] = acc[0] + acc[1] + acc[2];
      y[1] = acc[1] - acc[2] - acc[3];
      auto s2p0 = [&](int p, bool& ok) -> int {
        int oy = ty0 + p / PPR, ox = tx0 + 2 * (p % PPR);
        ok = full_tile || (oy < a.Ho && ox < a.Wo);
        oy = oy < a.Ho ? oy : a.Ho - 1;
        ox = ox < a.Wo ? ox : a.Wo - 1;
        return oy * a.Wo + ox;
      };
      auto s2p1 = [&](int p, bool& ok) -> int {
        int oy = ty0 + p / PPR, ox = tx0 + 2 * (p % PPR) + 1;
        ok = full_tile || (oy < a.Ho && ox < a.Wo);
        oy = oy < a.Ho ? oy : a.Ho - 1;
        ox = ox < a.Wo ? ox : a.Wo - 1;
        return oy * a.Wo + ox;
      };
      f32x4 rv[2][4];
      if (vec && has_res) {
        if (stage_res) {   // the loader waves parked the residual tile in LDS next to the last chunk
          const float* rs = res_base + ((k / nchunks) & 1) * RES + wave * 2048;
          const int lq = li >> 2, lj = li & 3;
#pragma unroll
          for (int o = 0; o < 2; ++o)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
              const int slot = 8 * gq + 4 * lh + lj;
              rv[o][gq] = *reinterpret_cast<const f32x4*>(rs + o * 1024 + slot * 32 + 4 * (lq ^ (slot & 7)));
            }
        } else {
          epi_load_res(e, s2p0, wm * 32, n_tile * 32, li, lh, rv[0]);
          epi_load_res(e, s2p1, wm * 32, n_tile * 32, li, lh, rv[1]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (vec) {
        epi_store_vec(e, s2p0, wm * 32, n_tile * 32, li, lh, y[0], rv[0], has_res);
        epi_store_vec(e, s2p1, wm * 32, n_tile * 32, li, lh, y[1], rv[1], has_res);
      } else {
        epi_store_scalar(e, s2p0, wm * 32, n_tile * 32, li, lh, y[0], has_res);
        epi_store_scalar(e, s2p1, wm * 32, n_tile * 32, li, lh, y[1], has_res);
      }
#pragma unroll
      for (int v = 0; v < 4; ++v)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[v][r] = 0.f;
    }
    if (stamp && last_chunk && ns_ < 60) a.dbg[ns_++] = clock64();
    c0 += CK;
    if (c0 >= cin_pad) { c0 = 0; w += gridDim.x; }
  }
  if (stamp) a.dbg[63] = ns_;
}

// ------------------------------------------------------------------------------------------------
// 3x3 stride-1 convolution as 2-D Winograd F(2x2,3x3): 16 products per 2x2 output block instead of 36
// (2.25x fewer MFMAs).  Work item = 8x16 output pixels (32 slots = 2x2 blocks) x 32*NT couts.
//  * loader waves do the whole input transform V = B^T d B on their way from HBM to LDS
//    (layout [16 positions][32 slots][CK+4]); the compute waves' inner loop is ds_read + weight load + MFMA only.
//  * the 16 positions are split over the 4 compute waves by row: wave py owns positions (py, 0..3), i.e.
//    4*NT accumulator tiles; A fragments are read from LDS once per workgroup, not once per N-wave.
//  * output transform: x-fold in registers, y-fold across the 4 waves through LDS (one exchange per item, hung on
//    the item's last chunk barrier); wave (oy, ox) then owns output parity (oy, ox) of every 2x2 block and runs the
//    usual bias/residual/ReLU/dwordx4-store epilogue on it.
// Weights: 16 "taps" U[py][px] = G g G^T (host, fp64).  Needs >= 2 Cin chunks per item (exchange-area reuse).
// ------------------------------------------------------------------------------------------------
template <int CK, int NLW>
__device__ __forceinline__ void wino2_loader(const ConvArgs& a, const ConvWork& wk, float* lds, int ltid, int ktotal,
                                             int cin_pad) {
  constexpr int CP = CK + 4, VPOS = 32 * CP, VBUF = 16 * VPOS;
  // Loader waves form NTEAM teams of NTASK lanes (one (slot, channel-quad) task per lane); team t owns the chunks
  // k = t (mod NTEAM) and requests the raw 4x4 windows of its next chunk as soon as it has written the current one,
  // so a request has NTEAM chunk periods (minus one transform) to come back from HBM/MALL.
  constexpr int NLT = NLW * 64, QPC = CK / 4, NTASK = 32 * QPC;
  static_assert(NLT % NTASK == 0, "loader waves must split into whole teams");
  constexpr int NTEAM = NLT / NTASK, NPT = 1;
  const int abl = a.phase_delay;   // timing ablations (tools/conv_bench.py --phase): 1 prio 0, 2 no transform, 3 no LDS writes
  if (abl == 1) __builtin_amdgcn_s_setprio(0);
  else __builtin_amdgcn_s_setprio(3);
  const int team = __builtin_amdgcn_readfirstlane(ltid / NTASK);
  ltid -= team * NTASK;
  // VALU issue slots are scarce next to a saturated matrix pipe (a VALU instruction of a co-resident wave costs
  // ~20 cycles here), so the loader keeps its per-chunk VALU work to the transform itself: raw buffer loads take
  // a per-item byte offset per window pixel (halo pixels get an out-of-range offset -> the load returns 0, no
  // masks) and the chunk's channel offset rides in the scalar offset; LDS stores use immediate offsets.
  constexpr unsigned POISON = 0x40000000u;   // > any in-frame byte offset (host checks the frame is < 1 GiB)
  int w = blockIdx.x, c0 = 0;
  unsigned off[NPT][16];
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in), 0, 0, 0x00020000);
  f32x4 cur[NPT][16];
  const int frame_bytes = a.H * a.W * a.in_cs * 4;
  auto advance = [&](int n) {
    for (int i = 0; i < n; ++i) {
      c0 += CK;
      if (c0 >= cin_pad) { c0 = 0; w += gridDim.x; }
    }
  };
  auto geometry = [&](int w_item) {
    const ItemPos ip = item_pos(wk, w_item);
    const int g = ip.g, b = ip.b;
    const int ty0 = ip.ty * 8, tx0 = ip.tx * 16;
    rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in + (size_t)b * a.H * a.W * a.in_cs), 0, frame_bytes,
                                             0x00020000);
#pragma unroll
    for (int i = 0; i < NPT; ++i) {
      const int idx = ltid + i * NLT;
      const int slot = (idx / QPC) & 31, q = idx % QPC;
      const int y0 = ty0 + 2 * (slot >> 3) - 1, x0 = tx0 + 2 * (slot & 7) - 1;
      const unsigned lane_b = (unsigned)(a.in_coff + g * a.Cin + 4 * q) * 4u;
      unsigned ry[4], cx[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int iy = y0 + r, ix = x0 + r;
        ry[r] = (iy >= 0 && iy < a.H) ? (unsigned)(iy * a.W * a.in_cs) * 4u + lane_b : POISON;
        cx[r] = (ix >= 0 && ix < a.W) ? (unsigned)(ix * a.in_cs) * 4u : POISON;
      }
#pragma unroll
      for (int t16 = 0; t16 < 16; ++t16) off[i][t16] = ry[t16 >> 2] + cx[t16 & 3];
    }
  };
  auto issue = [&](f32x4 (&dst)[NPT][16], int c0_) {
#pragma unroll
    for (int i = 0; i < NPT; ++i) {
#pragma unroll
      for (int t16 = 0; t16 < 16; ++t16) {
        unsigned vo = off[i][t16];
        if (c0_ + CK > a.Cin) {   // (uniform) only the chunk that crosses Cin: quads entirely past Cin read as 0
          if (c0_ + 4 * (int)((ltid + i * NLT) % QPC) >= a.Cin) vo = POISON;
        }
        dst[i][t16] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, vo, c0_ * 4, 0));
      }
    }
  };
  int kown = team;
  advance(team);
  const bool stamp = a.dbg && blockIdx.x == 0 && team == 0 && ltid == 0;   // per own chunk: start, data here, written, requested
  int ns_ = 0;
  if (stamp) a.dbg[64 + ns_++] = clock64();
  if (kown < ktotal) {
    geometry(w);
    issue(cur, c0);
  }
  for (int k = 0; k < ktotal; ++k) {
    if (stamp && ns_ < 60) a.dbg[64 + ns_++] = clock64();
    if (k == kown) {
    float* dst = lds + (k & 1) * VBUF;
    if (stamp && ns_ < 60) {
      const float probe = cur[0][15][0];   // wait for the last requested window
      asm volatile("" ::"v"(probe));
      a.dbg[64 + ns_++] = clock64();
    }
#pragma unroll
    for (int i = 0; i < NPT; ++i) {
      const int idx = ltid + i * NLT;
      const int slot = (idx / QPC) & 31, q = idx % QPC;
      const int c = c0 + q * 4;
      f32x4 d[16];
#pragma unroll
      for (int t16 = 0; t16 < 16; ++t16) d[t16] = cur[i][t16];
      if ((a.Cin & 3) && c0 + CK > a.Cin) {   // (uniform) ragged Cin: the quad that straddles Cin keeps its pad lanes 0
        if (c < a.Cin && c + 3 >= a.Cin) {
#pragma unroll
          for (int t16 = 0; t16 < 16; ++t16) {
            if (c + 1 >= a.Cin) d[t16][1] = 0.f;
            if (c + 2 >= a.Cin) d[t16][2] = 0.f;
            d[t16][3] = 0.f;
          }
        }
      }
      // rows: t[py][c] = B^T d ; then columns: v[py][px] = t B
      f32x4 t[16];
#pragma unroll
      for (int c4 = 0; c4 < 4; ++c4) {
        t[0 + c4] = d[0 + c4] - d[8 + c4];
        t[4 + c4] = d[4 + c4] + d[8 + c4];
        t[8 + c4] = d[8 + c4] - d[4 + c4];
        t[12 + c4] = d[4 + c4] - d[12 + c4];
      }
      if (abl == 2) {
#pragma unroll
        for (int t16 = 0; t16 < 16; ++t16) t[t16] = d[t16];
      }
      if (idx < NTASK && abl != 3) {
        float* o = dst + slot * CP + q * 4;
        if (abl == 2) {
#pragma unroll
          for (int t16 = 0; t16 < 16; ++t16) *reinterpret_cast<f32x4*>(o + t16 * VPOS) = t[t16];
        } else
#pragma unroll
        for (int py = 0; py < 4; ++py) {
          *reinterpret_cast<f32x4*>(o + (py * 4 + 0) * VPOS) = t[py * 4 + 0] - t[py * 4 + 2];
          *reinterpret_cast<f32x4*>(o + (py * 4 + 1) * VPOS) = t[py * 4 + 1] + t[py * 4 + 2];
          *reinterpret_cast<f32x4*>(o + (py * 4 + 2) * VPOS) = t[py * 4 + 2] - t[py * 4 + 1];
          *reinterpret_cast<f32x4*>(o + (py * 4 + 3) * VPOS) = t[py * 4 + 1] - t[py * 4 + 3];
        }
      }
    }
      // this team's next chunk: request its raw windows now, NTEAM barriers before they are needed
      const int wprev = w;
      advance(NTEAM);
      kown += NTEAM;
      __builtin_amdgcn_sched_barrier(0);
      if (stamp && ns_ < 60) a.dbg[64 + ns_++] = clock64();
      if (kown < ktotal) {
        if (w != wprev) geometry(w);
        issue(cur, c0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (stamp && ns_ < 60) a.dbg[64 + ns_++] = clock64();
    }
    __syncthreads();   // barrier k: buffer k&1 is full
  }
  __syncthreads();   // matches the compute waves' final barrier
  if (stamp) a.dbg[127] = ns_;
}

template <int NT, int CK, int NLW>
__global__ __launch_bounds__((4 + NLW) * 64, 1) void conv_wino2_kernel(const ConvArgs a, const ConvWork wk) {
  constexpr int CP = CK + 4, VPOS = 32 * CP, VBUF = 16 * VPOS, SPC = CK / 8;
  constexpr int PSTR = 36, PTILE = 32 * PSTR;   // parked tile: [32 slots][32 couts + 4 pad]
  constexpr int PART_W = 2 * NT * PTILE;        // floats one wave parks: [ox][nt] tiles
  static_assert(SPC % 2 == 0, "an even number of 8-channel steps per chunk keeps the fragment buffers static");
  extern __shared__ f32x4 smem4[];
  float* lds = reinterpret_cast<float*>(smem4);
  float* part = lds + 2 * VBUF;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: position row / output parity in SGPRs
  const int cin_pad = a.cin8 * 8;
  const int nchunks = (cin_pad + CK - 1) / CK;
  const int my_items = wk.total > (int)blockIdx.x ? (wk.total - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
  const int ktotal = my_items * nchunks;
  if (wave >= 4) {
    wino2_loader<CK, NLW>(a, wk, lds, tid - 256, ktotal, cin_pad);
    return;
  }
  const int py = wave;
  const int li = lane & 31, lh = lane >> 5;
  // The MFMA operands are swapped (weight fragment as A, activation fragment as B), so an accumulator tile is
  // D[cout][slot]: the lane owns ONE slot (li) and register 4*g + e holds cout 8*g + 4*lh + e - four consecutive
  // couts per register quad, which the epilogue parks in LDS with one ds_write_b128.
  // Store side: this wave writes output parity (oy, ox) of every 2x2 block; lane = (block column p8, cout quad q8),
  // four block rows gq per lane, so 8 lanes cover one pixel's 128-byte line.
  const int oy = wave >> 1, ox = wave & 1;
  const int p8 = lane >> 3, q8 = lane & 7;
  const int lane_pix = oy * a.Wo + 2 * p8 + ox;
  const int lane_out = lane_pix * a.out_cs + 4 * q8, lane_res = lane_pix * a.res_cs + 4 * q8;
  const int gs_out = 2 * a.Wo * a.out_cs, gs_res = 2 * a.Wo * a.res_cs;   // block row +1 = two pixel rows down
  const float ysign = oy ? -1.f : 1.f;
  f32x16 acc[4][NT];
  const int voff = (py * 4) * VPOS + li * CP + 4 * lh;

  // Weight fragments: raw buffer loads, lane part (lane * 16 bytes) in the vector offset, everything else - group,
  // position (py, p), 8-channel step, n-tile - in the SCALAR offset, so the inner loop issues no VALU instruction
  // at all.  Steps past cin8 re-read the last valid step (their activation fragments are zero).
  const int tap_b = a.cin8 * a.n_tiles * 1024, step_b = a.n_tiles * 1024;   // bytes per position / per step
  const __amdgpu_buffer_rsrc_t wrsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, a.groups * 16 * tap_b, 0x00020000);
  const unsigned wlane = lane * 16;
  int w = blockIdx.x, c0 = 0;
  struct WStream {          // weights of one item: byte offset of (group, position row py, n-block) + clamped n-tile offsets
    int base;
    int nto[NT];
  };
  auto wstream = [&](int w_item) -> WStream {
    const ItemPos ip = item_pos(wk, w_item);
    WStream ws;
    ws.base = (ip.g * 16 + py * 4) * tap_b + ip.rest * NT * 1024;
#pragma unroll
    for (int n = 0; n < NT; ++n) ws.nto[n] = (ip.rest * NT + n < a.n_tiles ? n : a.n_tiles - 1 - ip.rest * NT) * 1024;
    return ws;
  };
  // One fragment buffer per position, refreshed in place: as soon as the MFMAs of position p of step s are issued,
  // the activation fragment (LDS) and the NT weight fragments (L2) of position p of step s+1 are requested into the
  // same registers - a prefetch distance of 3/4 step at half the registers of a double buffer.  The weight stream
  // runs ahead across chunk and item boundaries (it does not depend on LDS); LDS fragments restart after each barrier.
  f32x4 aq[4], bq[4][NT];
  auto loadB = [&](int p, const WStream& ws, int step) {
    const int so = ws.base + p * tap_b + (step < a.cin8 ? step : a.cin8 - 1) * step_b;
#pragma unroll
    for (int n = 0; n < NT; ++n)
      bq[p][n] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wlane, so + ws.nto[n], 0));
  };
  WStream wsc = wstream(w);
  if (ktotal > 0) {
#pragma unroll
    for (int p = 0; p < 4; ++p) loadB(p, wsc, 0);
  }
  const bool stamp = a.dbg && blockIdx.x == 0 && tid == 0;   // per chunk: start, MFMAs done, parked, barrier, stored
  int ns_ = 0;
  if (stamp) a.dbg[ns_++] = clock64();
  __syncthreads();   // barrier 0
  // One Cin chunk.  The chunk body is instantiated three times - an item's first chunk (its very first MFMA per
  // accumulator takes C = 0, an inline constant, so accumulators are never cleared with VALU moves), middle chunks,
  // and the last chunk (which carries the epilogue) - and the item loop below strings them together, so no
  // fragment register ever flows through a control-flow merge (a merge makes hipcc double the fragment buffers
  // and copy between them).
  int k = 0;
  auto chunk = [&](auto first_tag, auto last_tag) {
    constexpr bool FIRST = decltype(first_tag)::value, LAST = decltype(last_tag)::value;
    constexpr bool last_chunk = LAST;
    if (stamp && ns_ < 60) a.dbg[ns_++] = clock64();
    const float* vb = lds + (k & 1) * VBUF + voff;
    const int nc0 = LAST ? 0 : c0 + CK, nw = LAST ? w + (int)gridDim.x : w;
    const bool more = k + 1 < ktotal;
    const WStream wsn = (LAST && more) ? wstream(nw) : wsc;
    const int nstep0 = nc0 / 8;
#pragma unroll
    for (int p = 0; p < 4; ++p) aq[p] = *reinterpret_cast<const f32x4*>(vb + p * VPOS);
    auto step = [&](auto s_tag) {
      constexpr int s = decltype(s_tag)::value;
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int n = 0; n < NT; ++n) {
            if (FIRST && s == 0 && j == 0) {
              const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
              acc[p][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(bq[p][n][j], aq[p][j], zero, 0, 0, 0);
            } else {
              acc[p][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(bq[p][n][j], aq[p][j], acc[p][n], 0, 0, 0);
            }
          }
        __builtin_amdgcn_sched_barrier(0);
        if (s + 1 < SPC) {
          aq[p] = *reinterpret_cast<const f32x4*>(vb + p * VPOS + (s + 1) * 8);
          loadB(p, wsc, c0 / 8 + s + 1);
        } else if (!last_chunk) {
          loadB(p, wsn, nstep0);   // first step of the next chunk; an item's last chunk leaves that to its epilogue
        }
      }
    };
    static_assert(SPC == 2, "steps per chunk");
    step(std::integral_constant<int, 0>());
    step(std::integral_constant<int, 1>());
    __builtin_amdgcn_sched_barrier(0);
    if (stamp && ns_ < 60) a.dbg[ns_++] = clock64();
    // ---- item epilogue, first half (before the chunk barrier): residual request, x-fold, park for the y-fold
    const bool has_res = a.res != nullptr;
    int rest = 0, ty0 = 0, tx0 = 0;
    bool full_tile = false;
    float* out_tile = nullptr;
    EpiCtx e;
    f32x4 rv[NT][4], bv[NT];
    bool vec[NT];
    auto s2p = [&](int p, bool& ok) -> int {
      int y = ty0 + 2 * (p >> 3) + oy, x = tx0 + 2 * (p & 7) + ox;
      ok = y < a.Ho && x < a.Wo;
      y = y < a.Ho ? y : a.Ho - 1;
      x = x < a.Wo ? x : a.Wo - 1;
      return y * a.Wo + x;
    };
    if (last_chunk) {
      const ItemPos ip = item_pos(wk, w);
      const int g = ip.g, b = ip.b;
      rest = ip.rest;
      ty0 = ip.ty * 8;
      tx0 = ip.tx * 16;
      e.bias = a.bias + (size_t)g * a.n_tiles * 32 + (size_t)b * a.bias_fstride;
      e.outb = a.out + (size_t)b * a.Ho * a.Wo * a.out_cs + a.out_coff + g * a.Cout;
      e.resb = has_res ? a.res + (size_t)b * a.Ho * a.Wo * a.res_cs + a.res_coff + g * a.Cout : nullptr;
      e.res_cs = a.res_cs; e.out_cs = a.out_cs; e.Cout = a.Cout; e.relu = a.relu;
      e.vec_align = ((a.out_coff + g * a.Cout) % 4 == 0) && (a.out_cs % 4 == 0) &&
                    (!has_res || (((a.res_coff + g * a.Cout) % 4 == 0) && (a.res_cs % 4 == 0)));
      full_tile = (ty0 + 8 <= a.Ho) && (tx0 + 16 <= a.Wo);
      // Tile-relative buffer descriptors: the lane part of an address is a kernel-constant VGPR, everything else a
      // scalar offset - no 64-bit address pairs in vector registers while the accumulators are still live.
      const int tile_pix = ty0 * a.Wo + tx0;
      out_tile = e.outb + (size_t)tile_pix * a.out_cs;
      const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<float*>(has_res ? e.resb + (size_t)tile_pix * a.res_cs : a.in), 0, -1, 0x00020000);
      const __amdgpu_buffer_rsrc_t brsrc =
          __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(e.bias), 0, -1, 0x00020000);
      // residual and bias of the item: requested first (the fragment registers are free - the last step did not
      // refresh them), consumed after the barrier
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const int n_tile = rest * NT + n;
        vec[n] = full_tile && e.vec_align && ((n_tile + 1) * 32 <= a.Cout);
        if (vec[n]) {
          bv[n] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(brsrc, 16 * q8, n_tile * 128, 0));
          if (has_res) {
#pragma unroll
            for (int gq = 0; gq < 4; ++gq)
              rv[n][gq] = __builtin_bit_cast(
                  f32x4, __builtin_amdgcn_raw_buffer_load_b128(rrsrc, lane_res * 4, (n_tile * 32 + gq * gs_res) * 4, 0));
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (stamp && ns_ < 60) a.dbg[ns_++] = clock64();
      // park the x-folded tiles as [slot][cout] (row stride 36 floats): the writer lane holds 4 consecutive couts
      // per register quad, the reader picks (pixel, cout quad) cells in NHWC store order - the LDS round trip the
      // y-fold needs anyway doubles as the register transpose
      float* pw = part + wave * PART_W + li * PSTR + 4 * lh;
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const f32x16 r0 = acc[0][n] + acc[1][n] + acc[2][n];
        const f32x16 r1 = acc[1][n] - acc[2][n] - acc[3][n];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          *reinterpret_cast<f32x4*>(pw + (0 * NT + n) * PTILE + 8 * q) = f32x4{r0[4 * q], r0[4 * q + 1], r0[4 * q + 2], r0[4 * q + 3]};
          *reinterpret_cast<f32x4*>(pw + (1 * NT + n) * PTILE + 8 * q) = f32x4{r1[4 * q], r1[4 * q + 1], r1[4 * q + 2], r1[4 * q + 3]};
        }
      }
      // the weight fragments of the next item's first step, held back by the last step above, go out now that the
      // accumulators are dead (the barrier and the second half cover their L2 latency)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int p = 0; p < 4; ++p) loadB(p, wsn, nstep0);
    }
    if (stamp && ns_ < 60) a.dbg[ns_++] = clock64();
    __syncthreads();   // barrier k+1
    if (stamp && ns_ < 60) a.dbg[ns_++] = clock64();
    // ---- second half: y-fold (oy = 0: R0+R1+R2, oy = 1: R1-R2-R3) of this wave's output parity, then store
    if (last_chunk) {
      const float* pr = part + (ox * NT) * PTILE + p8 * PSTR + 4 * q8;
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const int n_tile = rest * NT + n;
        if (n_tile >= a.n_tiles) continue;
        f32x4 y[4];   // block row gq, block column p8, couts 4*q8 .. 4*q8+3
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const int o = n * PTILE + 8 * gq * PSTR;
          const f32x4 p0 = *reinterpret_cast<const f32x4*>(pr + (oy + 0) * PART_W + o);
          const f32x4 p1 = *reinterpret_cast<const f32x4*>(pr + (oy + 1) * PART_W + o);
          const f32x4 p2 = *reinterpret_cast<const f32x4*>(pr + (oy + 2) * PART_W + o);
          y[gq] = p0 + ysign * (p1 + p2);
        }
        if (stamp && ns_ < 60) a.dbg[ns_++] = clock64();
        if (vec[n]) {   // full tile, aligned channel slices: bias, residual, ReLU, one full 128-byte line per 8 lanes
#pragma unroll
          for (int gq = 0; gq < 4; ++gq) {
            f32x4 v = y[gq] + bv[n];
            if (has_res) v += rv[n][gq];
            if (a.relu) {
              v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
            }
            // global (not buffer) store: on gfx950 a buffer_store_dwordx4 with an SGPR soffset was observed to pick
            // up a VALU write to its data registers issued right behind it (hipcc inserts no wait state there)
            char* ub = reinterpret_cast<char*>(out_tile + n_tile * 32 + gq * gs_out);   // uniform base
            __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(ub + (unsigned)(lane_out * 4)));
          }
        } else {        // ragged tile / ragged or unaligned channels: element-wise
#pragma unroll
          for (int gq = 0; gq < 4; ++gq) {
            bool ok;
            const int pix = s2p(8 * gq + p8, ok);
#pragma unroll
            for (int e4 = 0; e4 < 4; ++e4) {
              const int co = n_tile * 32 + 4 * q8 + e4;
              if (ok && co < a.Cout) {
                float v = y[gq][e4] + e.bias[co];
                if (has_res) v += e.resb[(size_t)pix * a.res_cs + co];
                if (a.relu) v = fmaxf(v, 0.f);
                e.outb[(size_t)pix * a.out_cs + co] = v;
              }
            }
          }
        }
      }
    }
    wsc = wsn;
    c0 = nc0;
    w = nw;
    ++k;
  };
  for (int item = 0; item < my_items; ++item) {   // nchunks >= 2 (checked by the host)
    chunk(std::true_type(), std::false_type());
    for (int ci = 2; ci < nchunks; ++ci) chunk(std::false_type(), std::false_type());
    chunk(std::false_type(), std::true_type());
  }
  if (stamp) a.dbg[63] = ns_;
}

// ------------------------------------------------------------------------------------------------
// launch helpers
// ------------------------------------------------------------------------------------------------
static hipError_t ensure_device_info() {
  if (g_num_cus) return hipSuccess;
  int dev = 0;
  hipDeviceProp_t prop;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  e = hipGetDeviceProperties(&prop, dev);
  if (e != hipSuccess) return e;
  g_num_cus = prop.multiProcessorCount;
  return hipSuccess;
}

// One persistent workgroup per CU.  (Two per CU were measured: the 384-thread workgroups do not become
// co-resident on gfx950 even when LDS and registers would allow it - the second half of the grid simply runs
// after the first - so k = 2 only adds a second prologue/tail; PMC: profiles/r01_pmc_wino_b2.txt.)
static long pick_grid(long total, size_t lds_bytes) {
  (void)lds_bytes;
  const long grid = g_num_cus;
  return grid > total ? total : grid;
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
    if ((e = ensure_device_info()) != hipSuccess) return e;
    init = true;
  }
  ConvWork wk;
  wk.tiles_x = (a.Wo + TW - 1) / TW;
  wk.tiles_per_frame = wk.tiles_x * ((a.Ho + TH - 1) / TH);
  wk.n_tiles_total = wk.tiles_per_frame * a.B;
  wk.nblk = (a.n_tiles + WAVES_N * NTW - 1) / (WAVES_N * NTW);
  wk.total = wk.n_tiles_total * wk.nblk * a.groups;
  hipLaunchKernelGGL(kern, dim3((unsigned)pick_grid(wk.total, lds)), dim3(NTHREADS), lds, s, a, wk);
  return hipGetLastError();
}

template <int TH, int TW, int WAVES_M, int WAVES_N, int CK, int NLW, int ABL = 0, int MINW = 1>
static hipError_t launch_wino(const ConvArgs& a, hipStream_t s) {
  constexpr size_t lds = 2 * (size_t)(TH + 2) * (TW + 2) * (CK + 4) * sizeof(float) +
                         2 * (size_t)(WAVES_M * WAVES_N) * 2 * 1024 * sizeof(float);   // 2 patch + 2 residual areas
  static_assert(lds <= 160 * 1024, "patch and residual buffers must fit the 160 KiB LDS");
  constexpr int NTHREADS = (WAVES_M * WAVES_N + NLW) * 64;
  auto kern = conv_wino_kernel<TH, TW, WAVES_M, WAVES_N, CK, NLW, ABL, MINW>;
  static bool init = false;
  if (!init) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    if ((e = ensure_device_info()) != hipSuccess) return e;
    if (getenv("ACRMI_DEBUG")) {
      int occ = -1;
      (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, NTHREADS, lds);
      hipFuncAttributes fa;
      (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(kern));
      fprintf(stderr, "[acrmi] conv_wino<%d,%d,%d,%d,%d,%d>: threads %d lds %zu regs %d occupancy(API) %d blocks/CU\n", TH, TW,
              WAVES_M, WAVES_N, CK, NLW, NTHREADS, lds, fa.numRegs, occ);
    }
    init = true;
  }
  ConvWork wk;
  wk.tiles_x = (a.Wo + TW - 1) / TW;
  wk.tiles_per_frame = wk.tiles_x * ((a.Ho + TH - 1) / TH);
  wk.n_tiles_total = wk.tiles_per_frame * a.B;
  wk.nblk = (a.n_tiles + WAVES_N - 1) / WAVES_N;
  wk.total = wk.n_tiles_total * wk.nblk * a.groups;
  long grid = pick_grid(wk.total, lds);
  if (MINW >= 3) {   // register budget allows two co-resident workgroups per CU
    const long g2 = 2L * g_num_cus;
    grid = g2 > wk.total ? wk.total : g2;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NTHREADS), lds, s, a, wk);
  return hipGetLastError();
}

template <int NT, int CK, int NLW>
static hipError_t launch_wino2(const ConvArgs& a, hipStream_t s) {
  constexpr size_t lds = (2 * (size_t)16 * 32 * (CK + 4) + 4 * (size_t)2 * NT * 32 * 36) * sizeof(float);
  static_assert(lds <= 160 * 1024, "two V buffers and the exchange area must fit the 160 KiB LDS");
  constexpr int NTHREADS = (4 + NLW) * 64;
  auto kern = conv_wino2_kernel<NT, CK, NLW>;
  static bool init = false;
  if (!init) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    if ((e = ensure_device_info()) != hipSuccess) return e;
    if (getenv("ACRMI_DEBUG")) {
      hipFuncAttributes fa;
      (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(kern));
      fprintf(stderr, "[acrmi] conv_wino2<%d,%d,%d>: threads %d lds %zu regs %d scratch %zu\n", NT, CK, NLW, NTHREADS,
              lds, fa.numRegs, (size_t)fa.localSizeBytes);
    }
    init = true;
  }
  if ((a.cin8 * 8 + CK - 1) / CK < 2) return hipErrorInvalidValue;   // exchange area is single-buffered
  if ((size_t)a.H * a.W * a.in_cs * 4 >= (1u << 30)) return hipErrorInvalidValue;   // loader's halo-poison offsets
  ConvWork wk;
  wk.tiles_x = (a.Wo + 15) / 16;
  wk.tiles_per_frame = wk.tiles_x * ((a.Ho + 7) / 8);
  wk.n_tiles_total = wk.tiles_per_frame * a.B;
  wk.nblk = (a.n_tiles + NT - 1) / NT;
  wk.total = wk.n_tiles_total * wk.nblk * a.groups;
  if ((unsigned long long)wk.total * (unsigned long long)wk.n_tiles_total >= (1ull << 40)) return hipErrorInvalidValue;
  set_magics(wk);
  hipLaunchKernelGGL(kern, dim3((unsigned)pick_grid(wk.total, lds)), dim3(NTHREADS), lds, s, a, wk);
  return hipGetLastError();
}

// Tile selection.  N32: one 32-cout tile per wave (Cout <= 32); N64: two.
// Small frames (<=16x16 outputs) take the 8x16 pixel tile so a batch still fills 256 CUs.
hipError_t launch_conv(ConvArgs a, hipStream_t s) {
  a.dbg = g_dbg;
  a.phase_delay = g_phase_delay;
  const bool n32 = a.n_tiles == 1;
  const bool small = (a.Ho * a.Wo <= 256) || (a.Ho % 16 != 0) || (a.Wo % 16 != 0);
  if (a.algo == 2) {   // Winograd F(2x2,3x3): 3x3 stride 1 only, weights packed with 16 taps
    if (a.ks != 3 || a.stride != 1) return hipErrorInvalidValue;
    if (g_force_cfg == 801) return n32 ? launch_wino2<1, 16, 2>(a, s) : launch_wino2<2, 16, 2>(a, s);   // one loader team
    return n32 ? launch_wino2<1, 16, 4>(a, s) : launch_wino2<2, 16, 4>(a, s);
  }
  if (a.algo == 1) {   // Winograd F(2,3) along x: 3x3 stride 1 only, weights packed with 12 taps
    if (a.ks != 3 || a.stride != 1) return hipErrorInvalidValue;
    if (n32) return small ? launch_wino<8, 16, 2, 1, 32, 2>(a, s) : launch_wino<16, 16, 4, 1, 32, 2>(a, s);
    // measured (tools/conv_bench.py --wino): the 8x16-pixel tile with 2x2 compute waves beats the 16x16 tile
    // with 4x2 waves (register-limited to 168 VGPRs, spills) on every N>=64 layer: 113-133 vs 93-110 TF-eq
    if (g_force_cfg == 303) return launch_wino<16, 16, 4, 1, 32, 2>(a, s);
    if (g_force_cfg == 701) return launch_wino<8, 16, 2, 2, 32, 2, 0, 3>(a, s);   // <=168 VGPRs: 2 workgroups per CU
    if (g_force_cfg == 601) return launch_wino<8, 16, 2, 2, 32, 2, 1>(a, s);   // timing ablations (wrong results)
    if (g_force_cfg == 602) return launch_wino<8, 16, 2, 2, 32, 2, 2>(a, s);
    if (g_force_cfg == 603) return launch_wino<8, 16, 2, 2, 32, 2, 3>(a, s);
    return launch_wino<8, 16, 2, 2, 32, 2>(a, s);
  }
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
    if (g_force_cfg == 401) return launch_ws<1, 1, 16, 16, 4, 2, 1, 2, 32, 2>(a, s);
    if (g_force_cfg == 402) return launch_ws<1, 1, 8, 16, 2, 2, 2, 1, 64, 2>(a, s);
    if (g_force_cfg == 403) return launch_ws<1, 1, 8, 16, 2, 2, 2, 1, 32, 2>(a, s);
    return small ? launch_ws<1, 1, 8, 16, 2, 2, 2, 1, 64, 2>(a, s) : launch_ws<1, 1, 16, 16, 4, 2, 1, 2, 64, 4>(a, s);
  }
  return hipErrorInvalidValue;
}

const char* conv_kernel_name(const ConvArgs& a) {
  if (a.algo == 2) return "conv3x3s1_wino2d_mfma_f32";
  if (a.algo == 1) return "conv3x3s1_wino_mfma_f32";
  if (a.ks == 3 && a.stride == 1) return "conv3x3s1_mfma_f32";
  if (a.ks == 3 && a.stride == 2) return "conv3x3s2_mfma_f32";
  return "conv1x1_mfma_f32";
}

}  // namespace acrmi
