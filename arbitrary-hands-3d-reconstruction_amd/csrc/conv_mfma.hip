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
// (conv_wino2_kernel / conv_ws2_kernel swap the operands: weights as A, activations as B.)
// Epilogue: folded-BN bias (+ per-frame bias), residual, ReLU, 16-byte stores.
//
//  * conv_wino2_kernel: every 3x3 stride-1 conv as 2-D Winograd F(2x2,3x3) (2.25x fewer MFMAs) - conv_wino2.inc.
//  * conv_ws2_kernel  : direct convolution, 3x3 stride 2, 1x1 (and 3x3 stride 1 on request) - conv_ws2.inc.
//  * conv_wino_kernel : the previous round's kernel, 3x3 stride 1 as Winograd F(2,3) along x (1.5x fewer MFMAs),
//                       kept selectable (algo 1): per output-pixel PAIR 4 products instead of 6 per kernel row;
//                       input transform in registers on the A fragments (v0=d0-d2, v1=d1+d2, v2=d2-d1, v3=d1-d3),
//                       weights pre-transformed on the host (G g), output transform (y0=m0+m1+m2, y1=m1-m2-m3) and
//                       4x4 DPP transposes ("lane = cout" -> "lane = pixel, 4 regs = 4 couts") in the epilogue.
#include "conv_frame.h"
#include "../../include/acrmi.h"
#include <cstdio>
#include <cstdlib>
#include <mutex>

namespace acrmi {

static int g_force_cfg = -1;
void conv_force_cfg(int cfg) { g_force_cfg = cfg; }
static long long* g_dbg = nullptr;
void conv_set_debug(long long* dbg) { g_dbg = dbg; }
static int g_phase_delay = 0;
void conv_set_phase_delay(int cycles) { g_phase_delay = cycles; }
static int g_xcd_swizzle = 1;
void conv_set_xcd_swizzle(int on) { g_xcd_swizzle = on; }
// per-device state: a process may hold contexts on several GPUs (acr.main.ACR(device=...)); kernel attributes
// (dynamic LDS size) and the CU count belong to a device, not to the process
static int g_num_cus_dev[MAX_DEVICES] = {};
// ------------------------------------------------------------------------------------------------
// loader waves with LDS-DMA (global_load_lds_dwordx4: HBM/L2 -> LDS without a register stop-over), same patch layout
// and the same barrier protocol as ws_loader.
// The register-staged ws_loader costs its SIMD ~NLD loads + NLD ds_write_b128 + ~10 address VALU per load and item,
// and the MFMA wave sharing the SIMD pays for every one of those issue slots (conv_wino3 stamps: 3.2k-cycle steps
// with the loader at priority 1-3, 2.9k at priority 0 - where the patch then arrives 1.8k cycles late).  Here a chunk
// costs a loader wave ceil(PH*PW*(CK/4+1) / (64*NLW)) DMA instructions and, for interior tiles and full chunks, one
// address add each:
//  * a DMA instruction writes 64 lanes x 16 bytes to CONSECUTIVE LDS addresses, so the patch [pixel][CK+4 floats] is
//    treated as a stream of 16-byte slots, CK/4 + 1 per pixel (the pad slot is never fetched - and never read);
//  * slot -> (patch row, patch column, channel quad) never changes: each lane keeps its tile-relative source offset;
//    per chunk only a uniform base pointer is new.  On border tiles (halo or ragged tile outside the image) and in a
//    ragged last chunk (channels >= Cin) the affected lanes fetch the 16-byte zero buffer instead.
// Needs Cin % 4 == 0 (a channel quad is wholly inside Cin or wholly zero).  One chunk is in flight at a time (two LDS
// buffers): enough for the MFMA-bound layers, the HBM-bound 1x1 layers keep the run-ahead register loader.
// ------------------------------------------------------------------------------------------------
template <int KS, int S, int TH, int TW, int CK, int NLW>
__device__ __forceinline__ void dma_loader(const ConvArgs& a, const ConvWork& wk, float* lds, int ltid, int my_items,
                                           int nchunks, int vb) {
  constexpr int PH = (TH - 1) * S + KS, PW = (TW - 1) * S + KS, CP = CK + 4, PAD = KS / 2, BUF = PH * PW * CP;
  constexpr int SPP = CK / 4 + 1;                   // 16-byte slots per pixel
  constexpr int NSLOT = PH * PW * SPP, NI = (NSLOT + NLW * 64 - 1) / (NLW * 64);
  const int lane = ltid & 63;
  const int lw = __builtin_amdgcn_readfirstlane(ltid >> 6);
  // A loader wave's VALU instructions only issue in the gaps of the MFMA stream it shares a SIMD with, so the per-slot
  // work is kept to: nothing but an address add (interior tile, full chunk), + one AND / compare / select against a
  // precomputed halo mask (border tile of an exactly tiled map), or the general bounds test (ragged tiles / chunks).
  int off[NI];          // source offset (floats) relative to the patch origin pixel, first channel of the chunk
  int halo[NI];         // bit 0/1/2/3: slot lies in the top/bottom/left/right PAD ring of the patch; bits 8.. = pyx below
  bool live[NI];        // slot is fetched at all (inside the patch, not the pad slot)
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int f = (i * NLW + lw) * 64 + lane;
    const int pix = f / SPP, c4 = f - pix * SPP;
    const int py = pix / PW, px = pix - py * PW;
    off[i] = (py * a.W + px) * a.in_cs + 4 * c4;
    halo[i] = (py < PAD ? 1 : 0) | (py >= PH - PAD ? 2 : 0) | (px < PAD ? 4 : 0) | (px >= PW - PAD ? 8 : 0) |
              (c4 << 8) | (px << 16) | (py << 24);
    live[i] = f < NSLOT && c4 < SPP - 1;
  }
  // the map is tiled exactly and the patches of its outermost tiles overhang by at most the PAD ring
  const bool ring_ok = a.Ho % TH == 0 && a.Wo % TW == 0 && (a.Ho - 1) * S - PAD + KS - a.H <= PAD &&
                       (a.Wo - 1) * S - PAD + KS - a.W <= PAD;
  const unsigned lds0 = (unsigned)(size_t)lds;      // LDS byte address of patch buffer 0
  auto dma = [&](const float* src, unsigned dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
  };
  int w = vb, k = 0;
  for (int item = 0; item < my_items; ++item, w += (int)gridDim.x) {
    const ItemPos ip = item_pos(wk, w);
    const int iy0 = ip.ty * TH * S - PAD, ix0 = ip.tx * TW * S - PAD;      // image coordinates of the patch origin
    const int edge = (iy0 < 0 ? 1 : 0) | (iy0 + PH > a.H ? 2 : 0) | (ix0 < 0 ? 4 : 0) | (ix0 + PW > a.W ? 8 : 0);
    // (the origin may lie outside the tensor: only lanes that pass the halo / bounds test use it)
    const float* origin = a.in + (((long long)ip.b * a.H + iy0) * a.W + ix0) * (long long)a.in_cs + a.in_coff + ip.g * a.Cin;
    for (int c = 0; c < nchunks; ++c, ++k) {
      const int c0 = c * CK;
      const int nq = (a.Cin - c0 < CK ? a.Cin - c0 : CK) / 4;              // channel quads of this chunk inside Cin
      const float* base = origin + c0;
      const unsigned dst0 = lds0 + (unsigned)((k & 1) * BUF * 4);
      if (edge == 0 && nq == CK / 4) {                  // uniform: interior tile, full chunk
#pragma unroll
        for (int i = 0; i < NI; ++i)
          if (live[i]) dma(base + off[i], dst0 + (unsigned)((i * NLW + lw) * 1024));   // lane l lands at dst + 16 l
      } else if (ring_ok && nq == CK / 4) {             // border tile: the halo ring outside the image fetches zeros
#pragma unroll
        for (int i = 0; i < NI; ++i)
          if (live[i]) dma((halo[i] & edge) ? a.zeros : base + off[i], dst0 + (unsigned)((i * NLW + lw) * 1024));
      } else {                                           // ragged tile and / or ragged chunk
        const int ylo = iy0 < 0 ? -iy0 : 0, yhi = a.H - 1 - iy0, xlo = ix0 < 0 ? -ix0 : 0, xhi = a.W - 1 - ix0;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
          const int py = halo[i] >> 24, px = (halo[i] >> 16) & 255, c4 = (halo[i] >> 8) & 255;
          const bool in = py >= ylo && py <= yhi && px >= xlo && px <= xhi && c4 < nq;
          if (live[i]) dma(in ? base + off[i] : a.zeros, dst0 + (unsigned)((i * NLW + lw) * 1024));
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (asm loads are not in hipcc's wait bookkeeping)
      __syncthreads();   // barrier k: chunk k is in buffer k & 1; the compute waves left buffer (k+1) & 1 at their barrier k
    }
  }
  __syncthreads();       // matches the compute waves' final barrier
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
      v[0] = vrelu1(v[0]); v[1] = vrelu1(v[1]); v[2] = vrelu1(v[2]); v[3] = vrelu1(v[3]);
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
    if (e.relu) v = vrelu1(v);
    if (cok && ((okm >> r) & 1u)) e.outb[pix[r] * e.out_cs + co] = v;
  }
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
  const int vb = virtual_block(a);
  const int my_items = wk.total > vb ? (wk.total - 1 - vb) / (int)gridDim.x + 1 : 0;
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
      const float* __restrict__ resb = a.res + res_frame_off(a, b) + a.res_coff + g * a.Cout;
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

  int w = vb, c0 = 0;
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
      e.resb = has_res ? a.res + res_frame_off(a, b) + a.res_coff + g * a.Cout : nullptr;
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
// launch helpers
// ------------------------------------------------------------------------------------------------
static int current_device() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) dev = 0;
  return dev;
}
// Lazily created per-device state (CU count, zero buffer, the per-kernel "dynamic LDS attribute set" flags) is guarded
// by one recursive mutex held for the whole of launch_conv / launch_point_heads: several contexts / host threads of
// one process may make their first launch at the same time (EnginePool, acr.main.ACR(device=...)), and a flag must not
// be visible before the attribute it stands for is set.  Uncontended cost per launch: one lock/unlock pair (~20 ns)
// next to a ~4 us launch.
static std::recursive_mutex g_init_mutex;
std::recursive_mutex& launch_mutex() { return g_init_mutex; }
static hipError_t ensure_device_info() {
  const int dev = current_device();
  std::lock_guard<std::recursive_mutex> lock(g_init_mutex);
  if (g_num_cus_dev[dev]) return hipSuccess;
  int n = 0;
  hipError_t e = hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
  if (e != hipSuccess) return e;
  g_num_cus_dev[dev] = n;
  return hipSuccess;
}
// true the first time `flags` (a per-kernel-instantiation array) is asked about the current device
bool first_use_on_device(unsigned char* flags) {
  const int dev = current_device();
  std::lock_guard<std::recursive_mutex> lock(g_init_mutex);
  if (flags[dev]) return false;
  flags[dev] = 1;
  return true;
}

// One persistent workgroup per CU.  (Two per CU were measured: the 384-thread workgroups do not become
// co-resident on gfx950 even when LDS and registers would allow it - the second half of the grid simply runs
// after the first - so k = 2 only adds a second prologue/tail; PMC: profiles/r01_pmc_wino_b2.txt.)
static long pick_grid(long total, size_t lds_bytes) {
  (void)lds_bytes;
  const long grid = g_num_cus_dev[current_device()];
  return grid > total ? total : grid;
}

int conv_forced_cfg() { return g_force_cfg; }
int conv_num_cus() { return g_num_cus_dev[current_device()]; }
int conv_current_device() { return current_device(); }
hipError_t conv_ensure_device_info() { return ensure_device_info(); }
long conv_pick_grid(long total) { return pick_grid(total, 0); }

template <int TH, int TW, int WAVES_M, int WAVES_N, int CK, int NLW, int ABL = 0, int MINW = 1>
static hipError_t launch_wino(const ConvArgs& a, hipStream_t s) {
  constexpr size_t lds = 2 * (size_t)(TH + 2) * (TW + 2) * (CK + 4) * sizeof(float) +
                         2 * (size_t)(WAVES_M * WAVES_N) * 2 * 1024 * sizeof(float);   // 2 patch + 2 residual areas
  static_assert(lds <= 160 * 1024, "patch and residual buffers must fit the 160 KiB LDS");
  constexpr int NTHREADS = (WAVES_M * WAVES_N + NLW) * 64;
  auto kern = conv_wino_kernel<TH, TW, WAVES_M, WAVES_N, CK, NLW, ABL, MINW>;
  static unsigned char init[MAX_DEVICES] = {};
  if (first_use_on_device(init)) {
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
  }
  ConvWork wk;
  wk.tiles_x = (a.Wo + TW - 1) / TW;
  wk.tiles_per_frame = wk.tiles_x * ((a.Ho + TH - 1) / TH);
  wk.n_tiles_total = wk.tiles_per_frame * a.B;
  wk.nblk = (a.n_tiles + WAVES_N - 1) / WAVES_N;
  wk.total = wk.n_tiles_total * wk.nblk * a.groups;
  long grid = pick_grid(wk.total, lds);
  if (MINW >= 3) {   // register budget allows two co-resident workgroups per CU
    const long g2 = 2L * g_num_cus_dev[current_device()];
    grid = g2 > wk.total ? wk.total : g2;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NTHREADS), lds, s, a, wk);
  return hipGetLastError();
}

#include "conv_wino2.inc"
#include "conv_ws2.inc"
#include "conv_wino3.inc"
#include "conv_wino24.inc"
#include "conv_wino24b.inc"
#include "conv_wino24c.inc"
#include "conv_pp2.inc"
#include "conv_x3.inc"
#include "conv_x3p.inc"
#include "conv_x3s2.inc"
#include "conv_p1.inc"

// Tile selection.  N32: one 32-cout tile per wave (Cout <= 32); N64: two.
// Small frames (<=16x16 outputs) take the 8x16 pixel tile so a batch still fills 256 CUs.
hipError_t launch_conv(ConvArgs a, hipStream_t s) {
  std::lock_guard<std::recursive_mutex> launch_lock(g_init_mutex);
  a.dbg = g_dbg;
  a.phase_delay = g_phase_delay;
  static const char* swz_env = experiment_env("ACRMI_XCD_SWIZZLE");      // A/B runs: 0 = round-robin item order
  a.xcd_swizzle = swz_env ? atoi(swz_env) : g_xcd_swizzle;
  {   // halo / pad-channel lanes of the LDS-DMA loaders read zeros from here (one small allocation per device)
    static float* zeros[MAX_DEVICES] = {};
    const int dev = current_device();
    std::lock_guard<std::recursive_mutex> lock(g_init_mutex);
    if (!zeros[dev]) {
      float* z = nullptr;
      hipError_t e = hipMalloc(&z, 256);
      if (e != hipSuccess) return e;
      // the memset runs on the null stream; the kernels that read the buffer run on non-blocking streams, which do
      // not order themselves behind it: wait for it here, once per device
      if ((e = hipMemset(z, 0, 256)) != hipSuccess || (e = hipDeviceSynchronize()) != hipSuccess) {
        (void)hipFree(z);
        return e;
      }
      zeros[dev] = z;
    }
    a.zeros = zeros[dev];
  }
  {
    hipError_t de = ensure_device_info();   // the CU count steers the item shape at small batches
    if (de != hipSuccess) return de;
  }
  if (a.dtype != ACRMI_DT_F32) return launch_conv_h16(a, s);   // f16 / bf16 storage: conv_h16.hip
  const bool n32 = a.n_tiles == 1;
  const bool small = (a.Ho * a.Wo <= 256) || (a.Ho % 16 != 0) || (a.Wo % 16 != 0);
  if (a.splitk && a.algo != 2) return hipErrorInvalidValue;
  if (a.nxt > 0 && a.algo != 5 && a.algo != 0 && !(a.algo == 3 && a.out2)) return hipErrorInvalidValue;      // extra residual terms: the stride-2 kernels, or conv_wino3's second output
  if (a.out2 && a.algo != 3) return hipErrorInvalidValue;
  if ((a.algo == 6 || a.algo == 7) && a.ks == 1) return launch_x3p(a, a.algo == 7, s);      // ... 1x1 (conv_x3p.inc)
  if ((a.algo == 6 || a.algo == 7) && a.stride == 2) return launch_x3s2(a, a.algo == 7, s);      // ... 3x3 stride 2 (conv_x3s2.inc)
  if (a.algo == 6 || a.algo == 7) return launch_x3(a, a.algo == 7, s);      // 3x3 stride 1, split f16 / bf16 operands on the 16-bit matrix pipe (conv_x3.inc)
  if (a.algo == 5) return launch_pp2(a, s);     // 3x3 stride 2, polyphase + F(2,2): weights packed with 4 x 7 taps (conv_pp2.inc)
  if (a.algo == 4)                              // F(2x4,3x3): weights packed with 4x6 taps; four-wave frame where it applies
    {
      if (wino24c_ok(a)) return launch_wino24c(a, s);      // all 24 positions per wave (round 6; conv_wino24c.inc)
      const int tw = wino24b_ok(a);
      return tw ? launch_wino24b(a, tw, s) : launch_wino24(a, s);
    }
  if (a.algo == 3) return launch_wino3(a, s);   // F(2x2,3x3), Cin <= 32, Cout = 32: weights packed for LDS residency
  if (a.algo == 2) {   // Winograd F(2x2,3x3): 3x3 stride 1 only, weights packed with 16 taps
    if (a.ks != 3 || a.stride != 1) return hipErrorInvalidValue;
    if (a.splitk) return launch_wino2<1, 32, 2>(a, s);      // K-slices as groups, one n-tile per item (conv_wino2.inc SPLIT)
    // two n-tiles per wave need two Cin chunks per item (single-buffered exchange area); Cin <= 32 runs one
    // n-tile per wave and one 32-cout block per work item instead
    // Cout = 33 with more than one Cin chunk: one regular tile + the 33rd channel on 4x4x1 MFMAs (conv_wino2.inc ODD)
    const bool odd33 = a.Cout == 33 && a.groups == 1 && a.cin8 * 8 > 32 && g_force_cfg != 804;
    // small batches: when two-n-tile items would not even fill the CUs once, one n-tile per item gives twice as many
    // items of half the length (256->256 at 16x16, one frame: 8 items of 85k cycles -> 16 of 45k; conv_bench --cfg 806
    // keeps two)
    const long items2 = (long)((a.Ho + 7) / 8) * ((a.Wo + 15) / 16) * a.B * ((a.n_tiles + 1) / 2) * a.groups;
    const bool few = items2 < g_num_cus_dev[current_device()] && g_force_cfg != 806;
    const bool nt1 = n32 || a.cin8 * 8 <= 32 || odd33 || few;
    // 8 < Cin <= 16 with Cout >= 64: two 8-channel chunks give the two-n-tile kernel the two barriers per item its
    // single-buffered exchange area needs (instead of one n-tile per item and the input transform once per 32 couts)
    if (!n32 && !odd33 && !few && a.cin8 == 2 && g_force_cfg != 807) return launch_wino2<2, 8, 2>(a, s);
    if (g_force_cfg == 801) return nt1 ? launch_wino2<1, 32, 4>(a, s) : launch_wino2<2, 32, 4>(a, s);
    return nt1 ? launch_wino2<1, 32, 2>(a, s) : launch_wino2<2, 32, 2>(a, s);
  }
  if (a.algo == 1) {   // Winograd F(2,3) along x: 3x3 stride 1 only, weights packed with 12 taps
    if (a.ks != 3 || a.stride != 1) return hipErrorInvalidValue;
    // (the previous round's kernel; the program lowers every 3x3 stride-1 conv to algo 2 now)
    if (n32) return small ? launch_wino<8, 16, 2, 1, 32, 2>(a, s) : launch_wino<16, 16, 4, 1, 32, 2>(a, s);
    return launch_wino<8, 16, 2, 2, 32, 2>(a, s);
  }
  if (a.nxt > 0) {      // extra residual terms (HR fuse): the polyphase kernel above, or the 32-cout stride-2 direct kernel
    if (a.ks != 3 || a.stride != 2 || a.cin8 * 8 <= 16 || a.nxt > 3) return hipErrorInvalidValue;
    return launch_ws2_impl<3, 2, 8, 16, 4, 1, 1, 1, 16, 2, false, true>(a, s);
  }
  // direct: template arguments <KS, S, TH, TW, WAVES_M, MT, WAVES_N, NTW, CK, NLW>
  if (a.ks == 3 && a.stride == 1) {
    if (n32) return small ? launch_ws2<3, 1, 8, 16, 4, 1, 1, 1, 32, 2>(a, s) : launch_ws2<3, 1, 16, 16, 4, 2, 1, 1, 32, 2>(a, s);
    return small ? launch_ws2<3, 1, 8, 16, 2, 2, 2, 1, 32, 2>(a, s) : launch_ws2<3, 1, 16, 16, 4, 2, 1, 2, 32, 2>(a, s);
  }
  // small batches (conv_bench --cfg 806 = the large-batch choice): the coarsest item shape that still gives every CU
  // an item, else the finest - a 256->64 stride-2 layer on one frame is 32 items of 147k cycles with 64-cout items
  const long cus = g_num_cus_dev[current_device()];
  const long tiles8 = (long)((a.Ho + 7) / 8) * ((a.Wo + 15) / 16) * a.B * a.groups;
  const long tiles16 = (long)((a.Ho + 15) / 16) * ((a.Wo + 15) / 16) * a.B * a.groups;
  const int nb2 = (a.n_tiles + 1) / 2;
  const bool fine = g_force_cfg != 806;
  if (a.ks == 3 && a.stride == 2) {
    if (n32 || (fine && tiles8 * nb2 < cus)) return launch_ws2<3, 2, 8, 16, 4, 1, 1, 1, 16, 2>(a, s);
    return launch_ws2<3, 2, 8, 16, 2, 2, 2, 1, 16, 2>(a, s);
  }
  if (a.ks == 1 && a.stride == 2) {
    // the projection shortcut of a strided ResNet block (torchvision resnet.py downsample: 1x1 stride 2) = the 1x1
    // stride-1 kernels on every other row and column of the input: only the loader's pixel coordinates change
    // (ConvArgs.in_sub).  (First form: the <1, 2, ...> instantiation, which stages the whole 15x31-pixel patch - 47 TF.)
    ConvArgs b = a;
    b.stride = 1;
    b.in_sub = 2;
    return launch_conv(b, s);
  }
  if (a.ks == 1 && a.stride == 1) {
    // the four-wave streaming frame (conv_p1.inc) wherever its 256-pixel x 64-cout items fill the chip twice (large batches;
    // a single frame's 64 -> 256 @128^2 is exactly 256 items: one per CU is slower than the finer items below);
    // conv_bench --cfg 809 keeps the eight-wave kernels (A/B)
    if (p1_ok(a) && g_force_cfg != 809 &&
        (long)((a.Ho * a.Wo) / 256) * a.B * a.groups * ((a.Cout / 32) / (a.Cout % 64 == 0 ? 2 : 1)) >= 2 * cus)
      return launch_p1(a, s);
    if (n32) return (small || (fine && tiles16 < cus)) ? launch_ws2<1, 1, 8, 16, 4, 1, 1, 1, 64, 2>(a, s)
                                                        : launch_ws2<1, 1, 16, 16, 4, 2, 1, 1, 64, 4>(a, s);
    // (a short contraction writing many channels - layer1's 64 -> 256 + residual - is latency-bound on one or two
    //  frames: 32-cout items keep four of them in flight per CU; 34 -> 24 us on one frame, 36 -> 27 on two, slower from
    //  four frames on; conv_bench --cfg 808 forces it)
    if ((fine && (tiles8 * nb2 < cus || (a.Cin <= 64 && tiles8 * nb2 <= 4 * cus))) || g_force_cfg == 808)
      return launch_ws2<1, 1, 8, 16, 4, 1, 1, 1, 64, 2>(a, s);
    return (small || (fine && tiles16 * nb2 < cus)) ? launch_ws2<1, 1, 8, 16, 2, 2, 2, 1, 64, 2>(a, s)
                                                    : launch_ws2<1, 1, 16, 16, 4, 2, 1, 2, 64, 4>(a, s);
  }
  return hipErrorInvalidValue;
}

const char* conv_kernel_name(const ConvArgs& a) {
  if (a.dtype != ACRMI_DT_F32) return a.dtype == ACRMI_DT_BF16 ? "conv_direct_mfma_bf16" : "conv_direct_mfma_f16";
  if (a.algo == 7) return a.ks == 1 ? "conv1x1_split_bf16x3_mfma" : (a.stride == 2 ? "conv3x3s2_split_bf16x3_mfma" : "conv3x3s1_split_bf16x3_mfma");
  if (a.algo == 6) return a.ks == 1 ? "conv1x1_split_f16x3_mfma" : (a.stride == 2 ? "conv3x3s2_split_f16x3_mfma" : "conv3x3s1_split_f16x3_mfma");
  if (a.algo == 5) return "conv3x3s2_polyphase_mfma_f32";
  if (a.algo == 4) return "conv3x3s1_wino2x4_mfma_f32";
  if (a.algo == 3) return "conv3x3s1_wino2d_lds_mfma_f32";
  if (a.algo == 2) return "conv3x3s1_wino2d_mfma_f32";
  if (a.algo == 1) return "conv3x3s1_wino_mfma_f32";
  if (a.ks == 3 && a.stride == 1) return "conv3x3s1_mfma_f32";
  if (a.ks == 3 && a.stride == 2) return "conv3x3s2_mfma_f32";
  return "conv1x1_mfma_f32";
}

}  // namespace acrmi
