// Shared device frame of the persistent convolution kernels (conv_mfma.hip: fp32; conv_h16.hip: f16 / bf16): work-item
// arithmetic, the XCD-banded item order and the register-staged loader waves.
#pragma once
#include "kernels.h"
#include <type_traits>

namespace acrmi {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((__vector_size__(16)));   // data operand of the raw buffer store builtins

// max(x, m) as ONE v_max_f32.  fmaxf() on a value hipcc cannot prove canonical (the result of an asm block, of a buffer
// load) costs two: v_max_f32 x, x, x first - and beside the fp32 MFMAs every VALU instruction of an epilogue is paid in full
// (DESIGN.md section 2).  Same result as fmaxf for every non-NaN input; a NaN yields m (fmaxf: the same).
__device__ __forceinline__ float vmax1(float x, float m) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(m));
  return r;
}
__device__ __forceinline__ float vrelu1(float x) {
  float r;
  asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(x));
  return r;
}

// host-side state shared by the launchers of both translation units (defined in conv_mfma.hip)
int conv_forced_cfg();                 // acrmi_tune key 0 (-1 = automatic)
int conv_num_cus();                    // CU count of the current device (after conv_ensure_device_info)
int conv_current_device();
hipError_t conv_ensure_device_info();
long conv_pick_grid(long total);

// Cache policy of the residual loads (aux operand of raw.buffer.load: 0 default, 2 slc = streaming): a residual
// element is read exactly once per launch.  Measured (batch 64): the HBM-bound 64->256 1x1 + residual layer 0.611 ->
// 0.583 ms with slc; the Winograd layers do not care (0.158 vs 0.161 ms), they keep the default.
constexpr int RES_CACHE_DIRECT = 2, RES_CACHE_WINO = 0;

struct ConvWork {
  int tiles_x, tiles_per_frame, n_tiles_total, nblk, total;
  int nb_inner = 1;   // conv_ws2_kernel, single-chunk items: n-blocks run per item from one LDS patch (nblk is then 1)
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

// Workgroup b of a launch lands on XCD b % 8 (observed dispatch order on gfx950; used for speed only, never for
// correctness).  A persistent workgroup walks items vb, vb + grid, vb + 2*grid, ...; with vb = b the 256 items in
// flight at any time are dealt round-robin over the XCDs, so the n-blocks of one tile (consecutive items) and the
// tiles whose halos overlap read the same input rows through eight different L2s (r01: FETCH 1.38x the algorithmic
// input bytes of the Winograd kernel).  With the swizzle XCD x works on the contiguous band of grid/8 items
// [k*grid + x*grid/8, ...): neighbours share an L2.
__device__ __forceinline__ int virtual_block(const ConvArgs& a) {
  const int b = (int)blockIdx.x, g = (int)gridDim.x;
  return (a.xcd_swizzle && (g & 7) == 0) ? (b & 7) * (g >> 3) + (b >> 3) : b;
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

// Residual addressing: a frame's residual map starts res_frame_off floats (elements) behind a.res; a BROADCAST residual
// (a.res_bcast: one [Ho][Wo][res_cs] map added to every frame - the position-bias map that stands for the coordinate
// channels of the head convs, packer.coord_bias_map) has a single frame.
__device__ __forceinline__ size_t res_frame_off(const ConvArgs& a, int b) {
  return a.res_bcast ? (size_t)0 : (size_t)b * a.Ho * a.Wo * a.res_cs;
}
__device__ __forceinline__ size_t res_total(const ConvArgs& a) {
  return (size_t)(a.res_bcast ? 1 : a.B) * a.Ho * a.Wo * a.res_cs;
}

// Bytes of bias behind group g's origin inside one bias row.  The epilogues read whole 32-channel n-tiles through a
// buffer descriptor; a per-frame bias row is only bias_fstride floats long (< n_tiles*32 for a ragged Cout), so the
// descriptor ends at the row: the tail channels read 0 (they are masked at the store) instead of running past the
// last frame's row and off the allocation.
__device__ __forceinline__ int bias_row_left(const ConvArgs& a, int g) {
  const int packed = a.groups * a.n_tiles * 32;
  const int row = (a.bias_fstride && a.bias_fstride < packed) ? a.bias_fstride : packed;
  const int left = row - g * a.n_tiles * 32;
  return left > 0 ? left * 4 : 0;
}

// ------------------------------------------------------------------------------------------------
// loader waves: fill LDS buffer (k & 1) with the patch of chunk k, one workgroup barrier per chunk
// ------------------------------------------------------------------------------------------------
struct NoHook {
  __device__ __forceinline__ void operator()(int, int, bool) const {}
};

// hook(item, item_seq, is_last_chunk) runs on the loader waves after chunk k's patch is written to LDS and
// before barrier k (used to pre-stage the item's residual tile next to its last chunk).
// PREP_SLACK: compute the next item's geometry in the slack before the barrier (direct kernels: their compute waves
// reach the barrier late) instead of right before its first request (Winograd kernel: there the loader is what the
// barrier waits for, and anything ahead of it delays every wave).
// EXTRA: workgroup barriers the compute waves execute inside an item's epilogue, BEHIND the barrier that ends the
// item's last chunk (conv_wino24_kernel parks its tiles in two rounds); the loader waves have to arrive at them too.
// Seen from here that barrier is the one behind the write of the NEXT item's first chunk (or the final one).
template <int KS, int S, int TH, int TW, int CK, int NLW, class Hook = NoHook, bool PREP_SLACK = true, int PRIO = 3, int EXTRA = 0>
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
  // (conv_wino3_kernel's loader has a whole item of slack and runs at priority 0: there the MFMA wave sharing the
  // SIMD should never lose an issue slot to it)
  if (a.phase_delay == 9) __builtin_amdgcn_s_setprio(0);   // tuning (conv_bench --phase 9): loader at priority 0
  else __builtin_amdgcn_s_setprio(PRIO);
  // Per-item geometry (pixel offsets, halo validity) is computed once per work item; per chunk only the channel
  // offset changes, so a chunk costs the loader NLD loads + NLD LDS writes and little VALU.
  // The loads run one chunk ahead of the LDS writes, in two statically named register sets: the requests of chunk
  // k+1 go out before chunk k is written and the barrier is waited on, so HBM requests stay in flight all the time
  // (with "load, wait, write, barrier" per chunk the memory-bound layers kept the queue empty half the time).
  struct Stage {
    f32x4 v[NLD];
    unsigned pixok;
    int c;   // first channel of this lane's float4
  };
  const int vb = virtual_block(a);
  int w = vb, c0 = 0;                // chunk being written
  int wn = vb, cn0 = 0;              // chunk being requested
  int off[NLD];
  unsigned pixok_n = 0;
  const float* __restrict__ inb = a.in;
  const int c4off = (ltid % (CK / 4)) * 4;
  const int nchunks_l = (cin_pad + CK - 1) / CK;
  const bool idle = a.phase_delay == 8;   // timing ablation: idle loader (wrong results)
  unsigned want_mask = 0;                 // this lane's element slots that exist (idx < NLOAD)
#pragma unroll
  for (int i = 0; i < NLD; ++i) want_mask |= (ltid + i * NLT < NLOAD ? 1u : 0u) << i;
  // request the chunk at (wn, cn0) and advance.  The loads are issued unconditionally (past the last chunk they
  // re-read the previous addresses): behind a branch hipcc cannot count the loads in flight any more and makes
  // every later wait a vmcnt(0), which would serialise the two register sets again.
  // geometry of the item the next request starts (if it starts one).  ~10 VALU instructions per load, and a loader
  // wave's VALU only issues while the MFMA wave it shares the SIMD with stalls - so this runs in the slack before a
  // barrier (see the loop below), never between a request and the LDS write the compute waves are waiting for.
  auto prepare = [&](bool valid) {
    if (valid && cn0 == 0) {
      const int tile = (wn / wk.nblk) % wk.n_tiles_total;
      const int g = (wn / wk.n_tiles_total) / wk.nblk;
      const int b = tile / wk.tiles_per_frame;
      const int t = tile - b * wk.tiles_per_frame;
      const int ty0 = (t / wk.tiles_x) * TH, tx0 = (t % wk.tiles_x) * TW;
      inb = a.in + (size_t)b * a.H * a.W * a.in_cs + a.in_coff + g * a.Cin;
      pixok_n = 0;
#pragma unroll
      for (int i = 0; i < NLD; ++i) {
        const int idx = ltid + i * NLT;
        const int pix = idx / (CK / 4);
        const int sub = a.in_sub > 1 ? a.in_sub : 1;      // (1x1 stride 2 as 1x1 stride 1 on every other row / column)
        const int iy = (ty0 * S - PAD + pix / PW) * sub, ix = (tx0 * S - PAD + pix % PW) * sub;
        const bool ok = (idx < NLOAD) && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
        const int iyc = iy < 0 ? 0 : (iy >= a.H ? a.H - 1 : iy), ixc = ix < 0 ? 0 : (ix >= a.W ? a.W - 1 : ix);
        off[i] = (iyc * a.W + ixc) * a.in_cs;   // per-frame offset < 2^31 floats
        pixok_n |= (ok ? 1u : 0u) << i;
      }
    }
  };
  auto request = [&](Stage& st, bool valid) {   // PREP_SLACK: prepare() for this position has run
    if constexpr (!PREP_SLACK) prepare(valid);
    st.pixok = pixok_n;
    st.c = cn0 + c4off;
    const int cc = st.c < a.Cin ? st.c : 0;
    // every load in flight before anything else (hipcc otherwise serialises them in rounds)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NLD; ++i) st.v[i] = *reinterpret_cast<const f32x4*>(inb + off[i] + cc);
    __builtin_amdgcn_sched_barrier(0);
    if (valid) {
      cn0 += CK;
      if (cn0 >= cin_pad) { cn0 = 0; wn += gridDim.x; }
    }
  };
  // chunk k = (w, c0) -> LDS buffer k&1, geometry for the next request (prep), then barrier k
  auto write = [&](const Stage& st, int k, bool prep) {
    float* dst = lds + (k & 1) * BUF;
    const int c = st.c;
    const bool cok = c < a.Cin;
    const bool ragged_c = cok && (c + 3 >= a.Cin);   // this float4 straddles Cin
    // Interior tiles of full-channel chunks (the common case) need no masking at all: a wave-uniform test sends them
    // down a path with no VALU per element - a loader wave's VALU only issues in the gaps of its SIMD's MFMA stream.
    const bool plain = __builtin_amdgcn_ballot_w64(!cok || ragged_c || (st.pixok & want_mask) != want_mask) == 0;
    if (!idle && plain) {
#pragma unroll
      for (int i = 0; i < NLD; ++i) {
        const int idx = ltid + i * NLT;
        if (idx < NLOAD) *reinterpret_cast<f32x4*>(dst + (idx / (CK / 4)) * CP + c4off) = st.v[i];
      }
    } else if (!idle) {
#pragma unroll
      for (int i = 0; i < NLD; ++i) {
        const int idx = ltid + i * NLT;
        f32x4 v = st.v[i];
        if (ragged_c) {
          if (c + 1 >= a.Cin) v[1] = 0.f;
          if (c + 2 >= a.Cin) v[2] = 0.f;
          v[3] = 0.f;
        }
        if (!cok || !((st.pixok >> i) & 1u)) v = f32x4{0.f, 0.f, 0.f, 0.f};
        if (idx < NLOAD) *reinterpret_cast<f32x4*>(dst + (idx / (CK / 4)) * CP + c4off) = v;
      }
    }
    hook(w, k / nchunks_l, c0 + CK >= cin_pad);
    if constexpr (PREP_SLACK) prepare(prep);
    // barrier k: buffer k&1 is full; the compute waves finished reading it two chunks ago
    __syncthreads();
    if constexpr (EXTRA > 0) {
      if (k > 0 && c0 == 0) {        // (uniform) chunk k opens an item: the previous item's epilogue runs behind this barrier
#pragma unroll
        for (int e = 0; e < EXTRA; ++e) __syncthreads();
      }
    }
    c0 += CK;
    if (c0 >= cin_pad) { c0 = 0; w += gridDim.x; }
  };
  // (the run-ahead needs 2 x NLD float4 registers: kernels with big patches or a residual hook stay sequential)
  constexpr bool RUN_AHEAD = NLD <= 16 && std::is_same<Hook, NoHook>::value;
  Stage s0, s1;
  if constexpr (RUN_AHEAD) {
    if (ktotal > 0) {
      if constexpr (PREP_SLACK) prepare(true);
      request(s0, true);
      if constexpr (PREP_SLACK) prepare(ktotal > 1);
      for (int k = 0; k < ktotal; k += 2) {
        request(s1, k + 1 < ktotal);
        write(s0, k, k + 2 < ktotal);
        if (k + 1 < ktotal) {
          request(s0, k + 2 < ktotal);
          write(s1, k + 1, k + 3 < ktotal);
        }
      }
    }
  } else {
    if constexpr (PREP_SLACK) prepare(ktotal > 0);
    for (int k = 0; k < ktotal; ++k) {
      request(s0, true);
      write(s0, k, k + 1 < ktotal);
    }
  }
  __syncthreads();   // matches the compute waves' final barrier
  if constexpr (EXTRA > 0) {
    if (ktotal > 0) {
#pragma unroll
      for (int e = 0; e < EXTRA; ++e) __syncthreads();
    }
  }
}


}  // namespace acrmi
