// 16-bit convolutions (f16 / bf16 storage, fp32 accumulate).
//
// The reference's --model_precision fp16 branch (acr/model.py:18-19,33-37: autocast around backbone + heads) on the
// gfx950 matrix pipe: v_mfma_f32_32x32x16_{f16,bf16} runs at 16x the fp32 MFMA rate, so every layer of the network is
// HBM- or L2-bound and the convolution is the DIRECT form for every shape (3x3 stride 1 / 2, 1x1, groups): Winograd
// would trade arithmetic that is free for round-off that is not.
//
// Frame of the kernel = conv_ws2_kernel's (conv_ws2.inc): persistent one-workgroup-per-CU grid, NCW compute waves +
// NLW loader waves, raw halo'd patch [PH][PW][CKF+4 floats] double-buffered in LDS, one workgroup barrier per Cin
// chunk.  What makes the reuse exact: a 16-bit NHWC tensor IS an fp32 NHWC tensor with half the channels (two
// elements per float), and one 8-float step of the fp32 kernel = 16 halfs = the K of ONE 32x32x16 MFMA:
//   * the loader waves run ws_loader unchanged on the float view (in_cs/2, in_coff/2, ceil(Cin/2) floats; a ragged
//     Cin is masked per float = per element pair, the odd element of the last pair is a zero pad channel);
//   * lane (li, lh) reads its activation fragment as ONE ds_read_b128 at channel float 4*lh = half 8*lh of the step:
//     exactly the B operand (k = 8*(lane>>5) .. +7, n = lane&31 = pixel);
//   * weights are packed by packer.pack_conv_h16 as the A operands, [tap][step][n-tile][lane][8 halfs] = 1 KiB per
//     fragment - the byte strides of the fp32 pack - and arrive through the same scalar-offset buffer-load ring,
//     deepened to R = 9 / 4 units because a unit is now MT*NTW MFMAs of 32 cycles instead of 4x that of 64;
//   * accumulators are D[cout][pixel] as before; the epilogue sends each tile through the wave's LDS tile
//     [32 pixels][36] and reads it back with 4 lanes per pixel x 8 couts: bias, residual (16-bit, one 16-byte load),
//     ReLU in fp32, ONE rounding to the storage type, 16-byte non-temporal stores (64 bytes per pixel and n-tile).
//     OUTF32 = the head exits (center / params x mix / prior / segm maps): fp32 output and fp32 residual, the
//     reference's .float() (acr/model.py:56-62) without the intermediate rounding.

#include "conv_frame.h"
#include "../../include/acrmi.h"

namespace acrmi {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <bool BF>
__device__ __forceinline__ f32x16 mfma_h16(const f32x4& a, const f32x4& b, const f32x16& c) {
  if constexpr (BF)
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
// 8 storage elements (one 16-byte vector) <-> 8 floats
template <bool BF>
__device__ __forceinline__ void unpack_h16(const f32x4& v, float (&f)[8]) {
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
__device__ __forceinline__ f32x4 pack_h16(const float (&f)[8]) {   // round to nearest even
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
__device__ __forceinline__ unsigned short to_h16(float f) {
  if constexpr (BF) return __builtin_bit_cast(unsigned short, (__bf16)f);
  else return __builtin_bit_cast(unsigned short, (_Float16)f);
}
template <bool BF>
__device__ __forceinline__ float from_h16(unsigned short u) {
  if constexpr (BF) return (float)__builtin_bit_cast(__bf16, u);
  else return (float)__builtin_bit_cast(_Float16, u);
}

// a: `in`, in_cs, in_coff, Cin, cin8 are the FLOAT VIEW of the 16-bit input (see above; launch_conv_h16 builds it);
// out / res / out_cs / out_coff / res_cs / res_coff / Cout are in elements of the output type.
// ONE: every item is a single Cin chunk.
// RING: depth of the weight-fragment ring for 3x3 kernels (0 = by register budget: 9 units with one n-tile per wave, 3 with
// two).  RING = -1 (3x3, single-chunk items of <= 2 steps, one n-tile, one group): the layer's whole weight set - 9 taps x 2
// steps = 18 fragments = 72 registers - is loaded ONCE per wave and stays in registers for the launch.  That is HRNet
// branch 0 (32 -> 32 at 128x128, 64 launches per step): with the ring, 4 waves x 18 KiB of fragments per item went through
// the CU's L1 (64 B/clk: as long as the item's 1152 cycles of MFMAs) next to the loader waves' streaming activation loads
// that evict them - stamps: 5.5k cycles of "MFMA phase" per item.
template <int KS, int S, int TH, int TW, int WAVES_M, int MT, int WAVES_N, int NTW, int CK, int NLW, bool ONE, bool BF,
          bool OUTF32, int RING = 0>
__global__ __launch_bounds__((WAVES_M * WAVES_N + NLW) * 64, 1) void conv_h16_kernel(const ConvArgs a, const ConvWork wk) {
  constexpr int PH = (TH - 1) * S + KS, PW = (TW - 1) * S + KS, CP = CK + 4;
  constexpr int NCW = WAVES_M * WAVES_N;
  constexpr int TAPS = KS * KS;
  constexpr int BUF = PH * PW * CP;
  constexpr int SPC = CK / 8;                       // 16-element steps per full chunk
  constexpr bool WREG = RING < 0;                    // the layer's weights live in registers
  constexpr int SU = TAPS == 1 ? 4 : (WREG ? CK / 8 : 1); // steps per statically unrolled pass (WREG: the whole chunk)
  constexpr int UNITS = SU * TAPS;                  // (step, tap) units per pass
  // weight-fragment ring (L2 latency; a unit is MT*NTW MFMAs of 32 cycles): 9 units deep where the registers allow it
  // (two n-tiles per wave: 72 registers of fragments next to 64 of accumulators and the residual prefetch spill)
  constexpr int R = TAPS == 1 ? 4 : (WREG ? (CK / 8) * TAPS : (RING ? RING : (NTW >= 2 ? 3 : 9)));
  static_assert(!WREG || (ONE && TAPS == 9 && NTW == 1 && WAVES_N == 1 && (CK == 16 || CK == 32)),
                "register-resident weights: 3x3, one chunk of <= 4 steps, one n-tile per wave and item");
  // activation-fragment ring (LDS latency).  Measured: 6 / 9 deep instead of 3 changes nothing (the 195 cycles per
  // 64-cycle unit in the stamps are not LDS latency) and spills the 1x1 fp32-output kernels
  constexpr int RA = TAPS == 1 ? 2 : 3;
  constexpr int PSTR = 36, PTILE = 32 * PSTR;       // epilogue tile: [32 pixels][32 couts + 4 pad] floats
  constexpr int ESZ = OUTF32 ? 4 : 2;               // bytes per output / residual element
  constexpr int NGQ = OUTF32 ? 4 : 2;               // store rounds per 32-pixel tile (8 / 16 pixels each)
  static_assert(UNITS % R == 0 && UNITS % RA == 0, "the rings must divide a pass");
  static_assert(SPC % SU == 0, "a full chunk is a whole number of passes");
  static_assert(TH * TW == 32 * MT * WAVES_M, "tile pixels must equal 32*MT*WAVES_M");
  static_assert(TW == 16, "epilogue pixel mapping assumes 16-pixel tile rows");
  extern __shared__ f32x4 smem4[];
  float* lds = reinterpret_cast<float*>(smem4);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cin_pad = a.cin8 * 8;                   // floats
  const int nchunks = (cin_pad + CK - 1) / CK;
  const int vb = virtual_block(a);
  const int my_items = wk.total > vb ? (wk.total - 1 - vb) / (int)gridDim.x + 1 : 0;
  const int ktotal = my_items * nchunks;
  if (wave >= NCW) {
    ws_loader<KS, S, TH, TW, CK, NLW>(a, wk, lds, tid - NCW * 64, ktotal, cin_pad);
    return;
  }
  const int wm = wave % WAVES_M, wn = wave / WAVES_M;
  const int li = lane & 31, lh = lane >> 5;
  float* epi = lds + 2 * BUF + wave * PTILE;
  int aoff[MT];   // patch offset (floats) of this lane's pixel in M-tile m, tap (0,0), float 4*lh of the step
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int p = (wm * MT + m) * 32 + li;
    aoff[m] = ((p / TW) * S * PW + (p % TW) * S) * CP + 4 * lh;
  }
  // store side.  fp32 output: lane = (pixel p8 of a half row, cout quad q8), 4 half rows per tile;
  // 16-bit output: lane = (pixel ps of a 16-pixel row, cout octet qs), 2 rows per tile.
  const int ps = OUTF32 ? lane >> 3 : lane >> 2;
  const int qs = OUTF32 ? lane & 7 : lane & 3;
  constexpr int PPR = OUTF32 ? 8 : 16;              // pixels per store round
  constexpr int CPL = OUTF32 ? 4 : 8;               // couts per lane
  const bool has_res = a.res != nullptr;
  char* const out_c = reinterpret_cast<char*>(a.out);
  const char* const res_c = reinterpret_cast<const char*>(a.res);

  const int tap_b = a.cin8 * a.n_tiles * 1024, step_b = a.n_tiles * 1024;   // weight bytes per tap / per step
  const __amdgpu_buffer_rsrc_t wrsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, a.groups * TAPS * tap_b, 0x00020000);
  const unsigned wlane = lane * 16;
  f32x16 acc[MT][NTW];
  f32x4 act[RA][MT], wgt[R][NTW];
  int w = vb, c0 = 0;
  struct WStream {   // weights of one item: byte offset of (group, n-block of this wave) + clamped n-tile offsets
    int base;
    int nto[NTW];
    int active;
  };
  auto wstream = [&](int w_item, int nb) -> WStream {   // nb < 0: the item's own n-block
    const ItemPos ip = item_pos(wk, w_item);
    const int nt0 = ((nb < 0 ? ip.rest : nb) * WAVES_N + wn) * NTW;
    WStream ws;
    ws.active = nt0 < a.n_tiles;
    const int nt0c = ws.active ? nt0 : 0;
    ws.base = ip.g * TAPS * tap_b + nt0c * 1024;
#pragma unroll
    for (int n = 0; n < NTW; ++n) ws.nto[n] = (nt0c + n < a.n_tiles ? n : a.n_tiles - 1 - nt0c) * 1024;
    return ws;
  };
  auto load_w = [&](int r, const WStream& ws, int tap, int step) {
    const int so = ws.base + tap * tap_b + (step < a.cin8 ? step : a.cin8 - 1) * step_b;
#pragma unroll
    for (int n = 0; n < NTW; ++n)
      wgt[r][n] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wlane, so + ws.nto[n], 0));
  };
  const int nbn = ONE ? wk.nb_inner : 1;
  WStream wsc = wstream(w, nbn > 1 ? 0 : -1);
  if (ktotal > 0) {
#pragma unroll
    for (int u = 0; u < R; ++u) load_w(u, wsc, u % TAPS, u / TAPS);
  }
  const bool stamp = a.dbg && blockIdx.x == 0 && tid == 0;
  int ns_ = 0;
  if (stamp) a.dbg[ns_++] = clock64();
  __syncthreads();   // barrier 0: chunk 0 is in buffer 0
  int k = 0;
  auto chunk = [&](auto first_tag, auto last_tag) {
    constexpr bool FIRST = decltype(first_tag)::value, LAST = decltype(last_tag)::value;
    if (stamp && ns_ < 60) a.dbg[ns_++] = clock64();
    const float* patch = lds + (k & 1) * BUF;
    const int nc0 = LAST ? 0 : c0 + CK, nw = LAST ? w + (int)gridDim.x : w;
    const bool more = k + 1 < ktotal;
    const WStream wsn = (LAST && more) ? wstream(nw, nbn > 1 ? 0 : -1) : wsc;
    const int nsteps = (cin_pad - c0 < CK ? cin_pad - c0 : CK) / 8;
    // ---- an item's last chunk first requests the item's residual and bias (see conv_ws2_kernel)
    int nt0 = 0, item_rest = 0;
    char* outb = nullptr;
    const float* bias = nullptr;
    const char* resb = nullptr;
    bool vec_align = false;
    f32x4 rv[MT][NTW][NGQ], bv[NTW][OUTF32 ? 1 : 2];
    int pix[MT][NGQ];
    bool pok[MT][NGQ];
    __amdgpu_buffer_rsrc_t rrsrc = wrsrc, brsrc = wrsrc;
    if (LAST) {
      const ItemPos ip = item_pos(wk, w);
      const int g = ip.g, b = ip.b;
      const int ty0 = ip.ty * TH, tx0 = ip.tx * TW;
      item_rest = ip.rest;
      bias = a.bias + (size_t)g * a.n_tiles * 32 + (size_t)b * a.bias_fstride;
      outb = out_c + ((size_t)b * a.Ho * a.Wo * a.out_cs + a.out_coff + g * a.Cout) * ESZ;
      resb = has_res ? res_c + (res_frame_off(a, b) + a.res_coff + g * a.Cout) * ESZ : nullptr;
      vec_align = ((a.out_coff + g * a.Cout) % CPL == 0) && (a.out_cs % CPL == 0) &&
                  (!has_res || (((a.res_coff + g * a.Cout) % CPL == 0) && (a.res_cs % CPL == 0)));
      const bool full_tile = (ty0 + TH <= a.Ho) && (tx0 + TW <= a.Wo);
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int gq = 0; gq < NGQ; ++gq) {
          const int p = (wm * MT + m) * 32 + PPR * gq + ps;
          int y = ty0 + p / TW, x = tx0 + p % TW;
          pok[m][gq] = full_tile || (y < a.Ho && x < a.Wo);
          y = y < a.Ho ? y : a.Ho - 1;
          x = x < a.Wo ? x : a.Wo - 1;
          pix[m][gq] = y * a.Wo + x;
        }
      const size_t res_off = res_frame_off(a, b) + a.res_coff + g * a.Cout;
      const size_t res_left = (res_total(a) - res_off) * ESZ;
      rrsrc = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<char*>(has_res ? resb : reinterpret_cast<const char*>(a.in)), 0,
          (has_res && vec_align) ? (int)(unsigned)(res_left > 0xffffffffu ? 0xffffffffu : res_left) : 0, 0x00020000);
      brsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(bias), 0, bias_row_left(a, g), 0x00020000);
    }
    for (int nb = 0; nb < nbn; ++nb) {
    if (LAST) {
      nt0 = ((nbn > 1 ? nb : item_rest) * WAVES_N + wn) * NTW;
#pragma unroll
      for (int n = 0; n < NTW; ++n) {
        const int ntc = nt0 + n < a.n_tiles ? nt0 + n : a.n_tiles - 1;
        if constexpr (OUTF32) {
          bv[n][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(brsrc, 16 * qs, ntc * 128, 0));
        } else {
          bv[n][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(brsrc, 32 * qs, ntc * 128, 0));
          bv[n][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(brsrc, 32 * qs + 16, ntc * 128, 0));
        }
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int gq = 0; gq < NGQ; ++gq)
            rv[m][n][gq] = __builtin_bit_cast(
                f32x4, __builtin_amdgcn_raw_buffer_load_b128(rrsrc, (pix[m][gq] * a.res_cs + CPL * qs) * ESZ, ntc * 32 * ESZ,
                                                             RES_CACHE_DIRECT));
      }
    }
    // activation fragments of the first RA units (the patch only became visible at the barrier)
#pragma unroll
    for (int u = 0; u < RA; ++u)
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const int t = u % TAPS;
        act[u][m] = *reinterpret_cast<const f32x4*>(patch + aoff[m] + ((t / KS) * PW + t % KS) * CP + (u / TAPS) * 8);
      }
    // one pass = UNITS statically unrolled (step, tap) units starting at step s; one MFMA per unit and tile
    auto pass = [&](int s, auto zero_tag) {
      constexpr bool ZERO = decltype(zero_tag)::value;
      const float* ps_ = patch + s * 8;
#pragma unroll
      for (int u = 0; u < UNITS; ++u) {
        const int r = u % R, ra = u % RA;
        __builtin_amdgcn_sched_barrier(0);
        if (SU == 1 || s + u / TAPS < nsteps) {   // (ragged chunk: uniform branch around the MFMAs only)
#pragma unroll
          for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NTW; ++n) {
              if (ZERO && u == 0) {
                const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                acc[m][n] = mfma_h16<BF>(wgt[r][n], act[ra][m], zero);
              } else {
                acc[m][n] = mfma_h16<BF>(wgt[r][n], act[ra][m], acc[m][n]);
              }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // units u + RA (activations, LDS) and u + R (weights, L2) into the registers unit u just released
        const int ua = u + RA, ta = ua % TAPS, sa = ua / TAPS;
#pragma unroll
        for (int m = 0; m < MT; ++m)
          act[ra][m] = *reinterpret_cast<const f32x4*>(ps_ + aoff[m] + ((ta / KS) * PW + ta % KS) * CP + sa * 8);
        if constexpr (!WREG) {
          const int un = u + R, tn = un % TAPS, sn = un / TAPS;
          load_w(r, wsc, tn, c0 / 8 + s + sn);
        }
      }
    };
    if (FIRST) pass(0, std::true_type());
    else pass(0, std::false_type());
    for (int s = SU; s < nsteps; s += SU) pass(s, std::false_type());
    __builtin_amdgcn_sched_barrier(0);
    if (nb + 1 == nbn) {   // the patch is free once the item's last n-block has read it
      if (stamp && ns_ < 60) a.dbg[ns_++] = clock64();
      __syncthreads();   // barrier k+1
      if (stamp && ns_ < 60) a.dbg[ns_++] = clock64();
    }
    if (LAST) {
      // ---- epilogue: bias, residual, ReLU, one rounding, store; each tile goes through the wave's LDS tile
      if (wsc.active) {
#pragma unroll
        for (int n = 0; n < NTW; ++n) {
          if (nt0 + n >= a.n_tiles) continue;
          const int co_l = (nt0 + n) * 32 + CPL * qs;      // first cout of this lane's vector
#pragma unroll
          for (int m = 0; m < MT; ++m) {
            // D[cout][pixel]: lane (li, lh) holds pixel li, couts 8q+4lh..+3 in register quad q
#pragma unroll
            for (int q = 0; q < 4; ++q)
              *reinterpret_cast<f32x4*>(epi + li * PSTR + 8 * q + 4 * lh) =
                  f32x4{acc[m][n][4 * q], acc[m][n][4 * q + 1], acc[m][n][4 * q + 2], acc[m][n][4 * q + 3]};
            if constexpr (OUTF32) {
              f32x4 y[4];
#pragma unroll
              for (int gq = 0; gq < 4; ++gq) y[gq] = *reinterpret_cast<const f32x4*>(epi + (8 * gq + ps) * PSTR + 4 * qs);
              if (vec_align && co_l + 4 <= a.Cout) {
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                  f32x4 o4 = y[gq] + bv[n][0] + rv[m][n][gq];   // (rv is 0 without a residual)
                  if (a.relu) {
                    o4[0] = fmaxf(o4[0], 0.f); o4[1] = fmaxf(o4[1], 0.f); o4[2] = fmaxf(o4[2], 0.f); o4[3] = fmaxf(o4[3], 0.f);
                  }
                  if (pok[m][gq])
                    __builtin_nontemporal_store(
                        o4, reinterpret_cast<f32x4*>(outb + (nt0 + n) * 128 + (unsigned)((pix[m][gq] * a.out_cs + 4 * qs) * 4)));
                }
              } else {   // ragged or unaligned channels: element-wise
#pragma unroll
                for (int gq = 0; gq < 4; ++gq)
#pragma unroll
                  for (int e4 = 0; e4 < 4; ++e4) {
                    const int co = co_l + e4;
                    if (pok[m][gq] && co < a.Cout) {
                      float o1 = y[gq][e4] + bias[co];
                      if (has_res) o1 += reinterpret_cast<const float*>(resb)[(size_t)pix[m][gq] * a.res_cs + co];
                      if (a.relu) o1 = fmaxf(o1, 0.f);
                      reinterpret_cast<float*>(outb)[(size_t)pix[m][gq] * a.out_cs + co] = o1;
                    }
                  }
              }
            } else {
              f32x4 y[2][2];
#pragma unroll
              for (int gq = 0; gq < 2; ++gq) {
                y[gq][0] = *reinterpret_cast<const f32x4*>(epi + (16 * gq + ps) * PSTR + 8 * qs);
                y[gq][1] = *reinterpret_cast<const f32x4*>(epi + (16 * gq + ps) * PSTR + 8 * qs + 4);
              }
              if (vec_align && co_l + 8 <= a.Cout) {
#pragma unroll
                for (int gq = 0; gq < 2; ++gq) {
                  float r8[8], o8[8];
                  unpack_h16<BF>(rv[m][n][gq], r8);              // (0 without a residual: zero-record descriptor)
#pragma unroll
                  for (int e = 0; e < 8; ++e) {
                    const float v = y[gq][e >> 2][e & 3] + bv[n][e >> 2][e & 3] + r8[e];
                    o8[e] = a.relu ? fmaxf(v, 0.f) : v;
                  }
                  if (pok[m][gq])
                    __builtin_nontemporal_store(
                        pack_h16<BF>(o8),
                        reinterpret_cast<f32x4*>(outb + (nt0 + n) * 64 + (unsigned)((pix[m][gq] * a.out_cs + 8 * qs) * 2)));
                }
              } else {   // ragged or unaligned channels: element-wise
#pragma unroll
                for (int gq = 0; gq < 2; ++gq)
#pragma unroll
                  for (int e = 0; e < 8; ++e) {
                    const int co = co_l + e;
                    if (pok[m][gq] && co < a.Cout) {
                      float o1 = y[gq][e >> 2][e & 3] + bias[co];
                      if (has_res)
                        o1 += from_h16<BF>(reinterpret_cast<const unsigned short*>(resb)[(size_t)pix[m][gq] * a.res_cs + co]);
                      if (a.relu) o1 = fmaxf(o1, 0.f);
                      reinterpret_cast<unsigned short*>(outb)[(size_t)pix[m][gq] * a.out_cs + co] = to_h16<BF>(o1);
                    }
                  }
              }
            }
          }
        }
      }
      // weight fragments of the first units of the item's next n-block / of the next item
      __builtin_amdgcn_sched_barrier(0);
      const WStream wnext = nb + 1 < nbn ? wstream(w, nb + 1) : wsn;
      if constexpr (!WREG) {
#pragma unroll
        for (int u = 0; u < R; ++u) load_w(u, wnext, u % TAPS, u / TAPS);
      }
      wsc = wnext;
      if (stamp && ns_ < 60) a.dbg[ns_++] = clock64();
    }
    }   // n-blocks of the item
    if (!LAST) wsc = wsn;
    c0 = nc0;
    w = nw;
    ++k;
  };
  for (int item = 0; item < my_items; ++item) {
    if constexpr (ONE) {
      chunk(std::true_type(), std::true_type());
    } else {
      chunk(std::true_type(), std::false_type());
      for (int ci = 2; ci < nchunks; ++ci) chunk(std::false_type(), std::false_type());
      chunk(std::false_type(), std::true_type());
    }
  }
  if (stamp) a.dbg[63] = ns_;
}

template <int KS, int S, int TH, int TW, int WAVES_M, int MT, int WAVES_N, int NTW, int CK, int NLW, bool ONE, bool BF,
          bool OUTF32, int RING = 0>
static hipError_t launch_h16_impl(const ConvArgs& a, hipStream_t s) {
  constexpr int PH = (TH - 1) * S + KS, PW = (TW - 1) * S + KS;
  constexpr size_t lds = (2 * (size_t)PH * PW * (CK + 4) + (size_t)WAVES_M * WAVES_N * 32 * 36) * sizeof(float);
  static_assert(lds <= 160 * 1024, "two patch buffers and the epilogue tiles must fit the 160 KiB LDS");
  constexpr int NTHREADS = (WAVES_M * WAVES_N + NLW) * 64;
  auto kern = conv_h16_kernel<KS, S, TH, TW, WAVES_M, MT, WAVES_N, NTW, CK, NLW, ONE, BF, OUTF32, RING>;
  static unsigned char init[MAX_DEVICES] = {};
  if (first_use_on_device(init)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    if ((e = conv_ensure_device_info()) != hipSuccess) return e;
  }
  ConvWork wk;
  wk.tiles_x = (a.Wo + TW - 1) / TW;
  wk.tiles_per_frame = wk.tiles_x * ((a.Ho + TH - 1) / TH);
  wk.n_tiles_total = wk.tiles_per_frame * a.B;
  wk.nblk = (a.n_tiles + WAVES_N * NTW - 1) / (WAVES_N * NTW);
  if (ONE && wk.nblk > 1 && conv_forced_cfg() != 901) {   // all n-blocks of a tile from one patch
    wk.nb_inner = wk.nblk;
    wk.nblk = 1;
  }
  wk.total = wk.n_tiles_total * wk.nblk * a.groups;
  if ((unsigned long long)wk.total * (unsigned long long)wk.n_tiles_total >= (1ull << 40)) return hipErrorInvalidValue;
  set_magics(wk);
  hipLaunchKernelGGL(kern, dim3((unsigned)conv_pick_grid(wk.total)), dim3(NTHREADS), lds, s, a, wk);
  return hipGetLastError();
}

// F32OK: the shape also exists with fp32 output (the head exits); the other shapes only ever write 16-bit maps
template <int KS, int S, int TH, int TW, int WAVES_M, int MT, int WAVES_N, int NTW, int CK, int NLW, bool F32OK = false,
          int RING = 0>
static hipError_t launch_h16(const ConvArgs& a, hipStream_t s) {
  const bool one = (a.cin8 * 8 + CK - 1) / CK < 2;
  const bool bf = a.dtype == ACRMI_DT_BF16;
#define ACRMI_H16_CASE(ONE_, BF_, F32_) \
  return launch_h16_impl<KS, S, TH, TW, WAVES_M, MT, WAVES_N, NTW, CK, NLW, ONE_, BF_, F32_, RING>(a, s)
  if (a.out_f32) {
    if constexpr (F32OK) {
      if (one) { if (bf) ACRMI_H16_CASE(true, true, true); ACRMI_H16_CASE(true, false, true); }
      if (bf) ACRMI_H16_CASE(false, true, true);
      ACRMI_H16_CASE(false, false, true);
    } else {
      return hipErrorInvalidValue;
    }
  }
  if (one) { if (bf) ACRMI_H16_CASE(true, true, false); ACRMI_H16_CASE(true, false, false); }
  if constexpr (RING < 0) {
    return hipErrorInvalidValue;      // register-resident weights exist for single-chunk items only
  } else {
    if (bf) ACRMI_H16_CASE(false, true, false);
    ACRMI_H16_CASE(false, false, false);
  }
#undef ACRMI_H16_CASE
}

// 16-bit convolution: a.in / in_cs / in_coff / Cin arrive in ELEMENTS of the 16-bit input and are turned into the float
// view here; a.cin8 = ceil(Cin / 16) steps.  Tile selection mirrors the direct fp32 kernels; the loaders get four waves
// on the big tiles (the layers are HBM / L2 bound: two chunks of requests in flight instead of one).
// fp32 output (a.out_f32; output AND residual fp32) exists for 1x1 and 3x3 stride-1 shapes: the head exits.
hipError_t launch_conv_h16(ConvArgs a, hipStream_t s) {
  {
    hipError_t de = conv_ensure_device_info();
    if (de != hipSuccess) return de;
  }
  if ((a.in_cs & 1) || (a.in_coff & 1) || (a.groups > 1 && (a.Cin & 1))) return hipErrorInvalidValue;
  a.in_cs /= 2;
  a.in_coff /= 2;
  a.cin8 = (a.Cin + 15) / 16;
  a.Cin = (a.Cin + 1) / 2;
  const bool n32 = a.n_tiles == 1;
  const bool small = (a.Ho * a.Wo <= 256) || (a.Ho % 16 != 0) || (a.Wo % 16 != 0);
  const long cus = conv_num_cus();
  const long tiles8 = (long)((a.Ho + 7) / 8) * ((a.Wo + 15) / 16) * a.B * a.groups;
  const long tiles16 = (long)((a.Ho + 15) / 16) * ((a.Wo + 15) / 16) * a.B * a.groups;
  const int nb2 = (a.n_tiles + 1) / 2;
  // template arguments <KS, S, TH, TW, WAVES_M, MT, WAVES_N, NTW, CK (floats = 2 elements), NLW>
  if (a.ks == 3 && a.stride == 1) {
    // One n-tile per wave and per ITEM on the big maps, whatever Cout is: with two n-tiles per wave the weight-fragment
    // ring is 3 units deep (registers) - shorter than an L2 round trip, the 64- and 128-channel layers then wait for
    // weight fragments (branch 2: 1.86 -> 1.59 ms per step with the deeper ring) - and 4 m-tiles x 1 n-tile per wave
    // spills (188..540 bytes of scratch).  The n-blocks of a tile are consecutive work items of one XCD band: the
    // patch is fetched from HBM once and re-read from L2.
    if (small || tiles16 * (n32 ? 1 : nb2) < cus)
      return n32 ? launch_h16<3, 1, 8, 16, 4, 1, 1, 1, 32, 2>(a, s) : launch_h16<3, 1, 8, 16, 2, 2, 2, 1, 32, 2, true>(a, s);
    // (measured and dropped: 32x16-pixel items with a 16-float chunk and a 3-unit ring for the Cin, Cout <= 32 layers -
    // 4 m-tiles per wave spill 104 bytes in the single-chunk variant, branch 0 went from 3.95 to 4.17 ms per step)
    // Cin <= 32 elements, Cout <= 32, one group: the layer's 18 weight fragments stay in registers (RING = -1)
    // ... on EIGHT compute waves (two per SIMD, one m-tile each, 152 registers): these single-chunk items have nothing to
    // overlap a wave's epilogue and fragment latencies with except the SIMD's other wave - 0.055 -> 0.049 ms per launch
    // (conv_bench --cfg 904 = four waves).  The multi-chunk kernels lose with eight waves (64->64: 0.042 -> 0.048 ms).
    if (n32 && a.groups == 1 && a.cin8 <= 2 && !a.out_f32 && conv_forced_cfg() != 902)
      return conv_forced_cfg() == 904 ? launch_h16<3, 1, 16, 16, 4, 2, 1, 1, 16, 4, false, -1>(a, s)
                                      : launch_h16<3, 1, 16, 16, 8, 1, 1, 1, 16, 4, false, -1>(a, s);
    // (measured and dropped: the same for Cin <= 64 elements - 36 fragments = 144 registers, a persistent workgroup sees
    // the same n-tile in every item when grid % n_tiles == 0 - spills 76 bytes and is no faster: 64->64 0.042 ms either way)
    return launch_h16<3, 1, 16, 16, 4, 2, 1, 1, 32, 4, true>(a, s);
  }
  if (a.ks == 3 && a.stride == 2) {
    if (n32 || tiles8 * nb2 < cus) return launch_h16<3, 2, 8, 16, 4, 1, 1, 1, 16, 4>(a, s);
    return launch_h16<3, 2, 8, 16, 2, 2, 2, 1, 16, 4>(a, s);
  }
  if (a.ks == 1 && a.stride == 2) {      // projection shortcut of a strided ResNet block: see launch_conv (in_sub)
    a.stride = 1;                          // (a is already the float view: no second pass through this function)
    a.in_sub = 2;
  }
  if (a.ks == 1 && a.stride == 1) {
    if (n32) return (small || tiles16 < cus) ? launch_h16<1, 1, 8, 16, 4, 1, 1, 1, 32, 2, true>(a, s)
                                             : launch_h16<1, 1, 16, 16, 4, 2, 1, 1, 32, 4, true>(a, s);
    if (tiles8 * nb2 < cus) return launch_h16<1, 1, 8, 16, 4, 1, 1, 1, 32, 2, true>(a, s);
    return (small || tiles16 * nb2 < cus) ? launch_h16<1, 1, 8, 16, 2, 2, 2, 1, 32, 2, true>(a, s)
                                          : launch_h16<1, 1, 16, 16, 4, 2, 1, 2, 32, 4, true>(a, s);
  }
  return hipErrorInvalidValue;
}

}  // namespace acrmi
