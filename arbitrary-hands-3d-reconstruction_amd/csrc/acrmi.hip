// libacrmi.so: context, program replay and the C ABI declared in include/acrmi.h.
#include "../../include/acrmi.h"
#include "kernels.h"

#include <dlfcn.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

using namespace acrmi;

// ---- op dependencies and stream lanes ----------------------------------------------------------------
constexpr int MAX_LANES = 8;
// ACRMI_OPT_LANES = 0: measured on MI355X (tools/lanes_check.py) - batch 1: 6.97 ms on one stream, 4.63 on four;
// batch 32: 23.2 / 22.5 / 22.2 ms on one / two / four; batch 64: 46.9 ms on one, 45.6 on two (the second lane fills the
// tails and pipeline fills of the first), 45.8 on three
constexpr int AUTO_LANES_SMALL = 4, AUTO_LANES_LARGE = 2, AUTO_SMALL_BATCH = 32;
struct Schedule {
  std::vector<int> order;               // active op indices, program order (a topological order)
  std::vector<std::vector<int>> deps;   // per op: earlier ops it must wait for (RAW / WAR / WAW on buffer ids)
  std::vector<char> leaf;               // per op: no later op depends on it
  // multi-stream form (ACRMI_OPT_LANES): lane per op, events of other lanes' ops to wait for, event to record
  int n_lanes = 0;
  std::vector<int> lane;
  std::vector<std::vector<int>> wait;
  std::vector<char> signal;
};
struct acrmi_ctx {
  int device = 0;
  std::string err;
  float* weights = nullptr;
  size_t n_weights = 0;
  std::vector<acrmi_buffer_desc> bufs;
  std::vector<float*> buf_ptr;
  std::vector<acrmi_op> ops;
  acrmi_head_layout heads{};
  bool have_program = false;
  int max_batch = 0;
  float* att_ws = nullptr;      // attention-pool workspace
  size_t att_ws_floats = 0;
  int* picks = nullptr;         // point heads: decoded centers per frame [max_batch,4]
  bool point_heads = false;     // ACRMI_OPT_POINT_HEADS
  Schedule sched[2][2];         // [point][0: small batches / 1: large batches]
  // ACRMI_OPT_LANES: independent chains of the program on parallel HIP streams (lane 0 = the caller's stream)
  int want_lanes = 0;           // 0 = by batch size (AUTO_LANES_*)
  // ACRMI_OPT_LANE_PLAN: lane per op from MEASURED op times (acrmi_profile_ops at a small batch stores them here), small-
  // batch schedules only; empty = the structural heuristic
  bool lane_plan = false;       // (explicit: Engine.tune_lanes / the host switches it on after profiling)
  std::vector<float> op_ms[2];  // [point]: per-op milliseconds of the last small-batch profile
  hipStream_t lanes[MAX_LANES] = {};
  hipEvent_t fork_ev = nullptr, join_ev[MAX_LANES] = {};
  std::vector<hipEvent_t> op_ev;
  // split-K convolutions (ACRMI_CONV_SPLITK): partial tiles + arrival counters, one set per lane (ops of one lane are
  // stream-ordered; ops of different lanes may overlap)
  float* split_ws[MAX_LANES] = {};
  unsigned* split_cnt[MAX_LANES] = {};
  size_t split_ws_floats = 0, split_counters = 0;
  ManoTables mano[2]{};
  bool have_mano[2] = {false, false};
  float* mano_allocs[2][9] = {};   // 6 fp32 tables + 3 f16 copies per side
  bool mano_f16 = false;           // ACRMI_OPT_MANO_FP16
  // options the reference reads from its config (acr/config.py): centermap_conf_thresh (acr/result_parser.py:241),
  // align_idx / mano_mesh_root_align (acr/mano_wrapper.py:19-33), -t temporal_optimization + smooth_coeff (acr/main.py:45-47)
  float conf_thresh = 0.35f;
  int center_idx = 9;           // < 0: no root alignment
  bool temporal = false;
  float smooth_coeff = 4.0f;
  float* smooth_state = nullptr;   // [2][3][64] floats + 2 ints (One-Euro state of one video stream)
  // multi-GPU: RCCL communicator created by acrmi_comm_init
  void* comm = nullptr;
  int comm_ranks = 0;
};

// Every entry point runs on the context's device and leaves the caller's current device as it found it (a process
// may hold contexts on several GPUs).
struct DeviceGuard {
  int prev = -1;
  bool switched = false;
  hipError_t err = hipSuccess;
  explicit DeviceGuard(int dev) {
    err = hipGetDevice(&prev);
    if (err == hipSuccess && prev != dev) {
      err = hipSetDevice(dev);
      switched = err == hipSuccess;
    }
  }
  ~DeviceGuard() {
    if (switched) (void)hipSetDevice(prev);
  }
};

static std::string g_err;

static int fail(acrmi_ctx* c, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (c) c->err = buf; else g_err = buf;
  return code;
}
#define ON_DEVICE(c)                                                                                   \
  DeviceGuard guard_((c)->device);                                                                     \
  if (guard_.err != hipSuccess) return fail(c, ACRMI_EHIP, "hipSetDevice(%d): %s", (c)->device, hipGetErrorString(guard_.err))
#define HIPCHK(c, expr)                                                                      \
  do {                                                                                       \
    hipError_t e_ = (expr);                                                                  \
    if (e_ != hipSuccess) return fail(c, ACRMI_EHIP, "%s: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

extern "C" {

int acrmi_version(void) { return ACRMI_VERSION; }

const char* acrmi_last_error(const acrmi_ctx* ctx) { return ctx ? ctx->err.c_str() : g_err.c_str(); }

int acrmi_create(acrmi_ctx** out, int device) {
  if (!out) return fail(nullptr, ACRMI_EINVAL, "acrmi_create: out is NULL");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    return fail(nullptr, ACRMI_EHIP, "acrmi_create: no HIP device (%s)", hipGetErrorString(e));
  if (device < 0 || device >= n) return fail(nullptr, ACRMI_EINVAL, "acrmi_create: device %d of %d", device, n);
  {
    DeviceGuard g(device);      // validates the device; the caller's current device is restored
    if (g.err != hipSuccess) return fail(nullptr, ACRMI_EHIP, "hipSetDevice: %s", hipGetErrorString(g.err));
  }
  acrmi_ctx* c = new acrmi_ctx();
  c->device = device;
  *out = c;
  return ACRMI_OK;
}

static void free_program(acrmi_ctx* c) {
  for (hipEvent_t e : c->op_ev)
    if (e) (void)hipEventDestroy(e);
  c->op_ev.clear();
  for (float* p : c->buf_ptr)
    if (p) (void)hipFree(p);
  c->buf_ptr.clear();
  c->bufs.clear();
  c->ops.clear();
  for (int l = 0; l < MAX_LANES; ++l) {
    if (c->split_ws[l]) (void)hipFree(c->split_ws[l]);
    if (c->split_cnt[l]) (void)hipFree(c->split_cnt[l]);
    c->split_ws[l] = nullptr; c->split_cnt[l] = nullptr;
  }
  c->split_ws_floats = c->split_counters = 0;
  if (c->att_ws) (void)hipFree(c->att_ws);
  c->att_ws = nullptr;
  if (c->picks) (void)hipFree(c->picks);
  c->picks = nullptr;
  c->have_program = false;
}

static void comm_destroy(acrmi_ctx* c);

void acrmi_destroy(acrmi_ctx* c) {
  if (!c) return;
  DeviceGuard guard_(c->device);
  comm_destroy(c);
  free_program(c);
  for (int l = 0; l < MAX_LANES; ++l) {
    if (c->lanes[l]) (void)hipStreamDestroy(c->lanes[l]);
    if (c->join_ev[l]) (void)hipEventDestroy(c->join_ev[l]);
  }
  if (c->fork_ev) (void)hipEventDestroy(c->fork_ev);
  if (c->weights) (void)hipFree(c->weights);
  if (c->smooth_state) (void)hipFree(c->smooth_state);
  for (auto& side : c->mano_allocs)
    for (float* p : side)
      if (p) (void)hipFree(p);
  delete c;
}

int acrmi_load_weights(acrmi_ctx* c, const float* blob, size_t n) {
  if (!c || !blob || n == 0) return fail(c, ACRMI_EINVAL, "acrmi_load_weights: bad arguments");
  ON_DEVICE(c);
  if (c->weights) (void)hipFree(c->weights);
  c->weights = nullptr;
  HIPCHK(c, hipMalloc(&c->weights, n * sizeof(float)));
  HIPCHK(c, hipMemcpy(c->weights, blob, n * sizeof(float), hipMemcpyHostToDevice));
  c->n_weights = n;
  return ACRMI_OK;
}

static int run_op(acrmi_ctx* c, const acrmi_op& op, const uint8_t* img, int B, hipStream_t s, int lane = 0) {
  auto ptr = [&](int id) -> float* { return id >= 0 ? c->buf_ptr[id] : nullptr; };
  auto desc = [&](int id) -> const acrmi_buffer_desc& { return c->bufs[id]; };
  // buffer `id` advanced by `coff` ELEMENTS of its storage type (the pointers stay typed float*: they are opaque here)
  auto eptr = [&](int id, int coff) -> float* {
    if (id < 0) return nullptr;
    return reinterpret_cast<float*>(reinterpret_cast<char*>(c->buf_ptr[id]) + (size_t)coff * (c->bufs[id].dtype ? 2 : 4));
  };
  switch (op.kind) {
    case ACRMI_OP_U8NORM: {
      const auto& d = desc(op.out_buf);
      HIPCHK(c, launch_u8norm(img, (long)B * d.h * d.w, ptr(op.out_buf), s));
      return ACRMI_OK;
    }
    case ACRMI_OP_STEM: {
      const auto& d = desc(op.out_buf);
      if (op.ksize == 7) {     // ResNet stem (stem7.hip)
        if (d.dtype)
          HIPCHK(c, launch_stem7_h16(img, B, 2 * d.h, 2 * d.w, c->weights + op.w_off, c->weights + op.b_off, ptr(op.out_buf),
                                     d.cs, op.out_coff, op.relu, d.dtype, s));
        else
          HIPCHK(c, launch_stem7(img, B, 2 * d.h, 2 * d.w, c->weights + op.w_off, c->weights + op.b_off, ptr(op.out_buf), d.cs,
                                 op.out_coff, op.relu, s));
        return ACRMI_OK;
      }
      if (d.dtype)
        HIPCHK(c, launch_stem_h16(img, B, 2 * d.h, 2 * d.w, c->weights + op.w_off, c->weights + op.b_off, ptr(op.out_buf),
                                  d.cs, op.out_coff, op.relu, d.dtype, s));
      else
        HIPCHK(c, launch_stem(img, B, 2 * d.h, 2 * d.w, c->weights + op.w_off, c->weights + op.b_off, ptr(op.out_buf), d.cs,
                              op.out_coff, op.relu, s));
      return ACRMI_OK;
    }
    case ACRMI_OP_CONV: {
      const auto& di = desc(op.in_buf);
      const auto& dout = desc(op.out_buf);
      ConvArgs a{};
      a.in = ptr(op.in_buf);
      a.w = c->weights + op.w_off;
      a.bias = op.bias_per_frame ? ptr(op.aux_buf) : c->weights + op.b_off;
      a.res = ptr(op.res_buf);
      a.out = ptr(op.out_buf);
      a.B = B; a.H = di.h; a.W = di.w; a.Ho = dout.h; a.Wo = dout.w;
      a.in_cs = di.cs; a.in_coff = op.in_coff; a.Cin = op.cin;
      a.out_cs = dout.cs; a.out_coff = op.out_coff; a.Cout = op.cout;
      a.res_cs = op.res_buf >= 0 ? desc(op.res_buf).cs : 0; a.res_coff = op.res_coff;
      if (op.flags & ACRMI_CONV_SPLITK) {     // groups = K-slices of one convolution; workspace of the lane this op runs on
        a.splitk = 1; a.split_ws = c->split_ws[lane]; a.split_cnt = c->split_cnt[lane];
      }
      if (op.flags & ACRMI_CONV_BIAS_MAP) {   // position-bias map in the weight blob, added to every frame
        a.res = c->weights + op.w_off2; a.res_cs = (op.groups * op.cout + 3) / 4 * 4; a.res_coff = 0; a.res_bcast = 1;
      }
      a.ks = op.ksize; a.stride = op.stride; a.relu = op.relu; a.groups = op.groups;
      a.cin8 = (op.cin + 7) / 8;
      a.n_tiles = op.cout <= 32 ? 1 : ((op.cout + 63) / 64) * 2;
      a.bias_fstride = op.bias_per_frame ? desc(op.aux_buf).cs : 0;
      a.algo = op.flags & 7;
      a.dtype = di.dtype;                                    // 16-bit input: conv_h16.hip
      a.out_f32 = di.dtype != ACRMI_DT_F32 && dout.dtype == ACRMI_DT_F32;
      HIPCHK(c, launch_conv(a, s));
      return ACRMI_OK;
    }
    case ACRMI_OP_FUSESUM: {
      const auto& dout = desc(op.out_buf);
      FuseArgs f{};
      f.nterms = op.nterms; f.B = B; f.H = dout.h; f.W = dout.w; f.C = op.cout; f.out_cs = dout.cs; f.relu = op.relu;
      f.out = eptr(op.out_buf, op.out_coff);
      for (int t = 0; t < op.nterms; ++t) {
        f.term[t] = eptr(op.term_buf[t], op.term_coff[t]);
        f.cs[t] = desc(op.term_buf[t]).cs;
        f.shift[t] = op.term_shift[t];
      }
      if (dout.dtype) HIPCHK(c, launch_fuse_sum_h16(f, dout.dtype, s));
      else HIPCHK(c, launch_fuse_sum(f, s));
      return ACRMI_OK;
    }
    case ACRMI_OP_BILINEAR2X: {
      const auto& di = desc(op.in_buf);
      if (di.dtype)
        HIPCHK(c, launch_bilinear2x_h16(ptr(op.in_buf), B, di.h, di.w, di.cs, op.in_coff, op.cin, ptr(op.out_buf),
                                        desc(op.out_buf).cs, op.out_coff, di.dtype, s));
      else
        HIPCHK(c, launch_bilinear2x(ptr(op.in_buf), B, di.h, di.w, di.cs, op.in_coff, op.cin, ptr(op.out_buf),
                                    desc(op.out_buf).cs, op.out_coff, s));
      return ACRMI_OK;
    }
    case ACRMI_OP_PAIR1X1: {
      const auto& di = desc(op.in_buf);
      HIPCHK(c, launch_pair1x1(ptr(op.in_buf), di.cs, op.in_coff, ptr(op.res_buf), desc(op.res_buf).cs, op.res_coff, ptr(op.out_buf),
                               desc(op.out_buf).cs, op.out_coff, ptr(op.aux_buf), desc(op.aux_buf).cs, 0, c->weights + op.w_off,
                               (long)B * di.h * di.w, s));
      return ACRMI_OK;
    }
    case ACRMI_OP_MAXPOOL: {
      const auto& di = desc(op.in_buf);
      if (di.dtype)
        HIPCHK(c, launch_maxpool3s2_h16(ptr(op.in_buf), B, di.h, di.w, di.cs, op.in_coff, op.cin, ptr(op.out_buf),
                                        desc(op.out_buf).cs, op.out_coff, di.dtype, s));
      else
        HIPCHK(c, launch_maxpool3s2(ptr(op.in_buf), B, di.h, di.w, di.cs, op.in_coff, op.cin, ptr(op.out_buf),
                                    desc(op.out_buf).cs, op.out_coff, s));
      return ACRMI_OK;
    }
    case ACRMI_OP_POW11: {
      const auto& d = desc(op.out_buf);
      if (d.dtype) HIPCHK(c, launch_pow11_h16(ptr(op.out_buf), (long)B * d.h * d.w, d.cs, op.out_coff, d.dtype, s));
      else HIPCHK(c, launch_pow11(ptr(op.out_buf), (long)B * d.h * d.w, d.cs, op.out_coff, s));
      return ACRMI_OK;
    }
    case ACRMI_OP_ATTPOOL: {
      const auto& ds = desc(op.in_buf);     // segm logits
      const auto& df = desc(op.res_buf);    // features
      HIPCHK(c, launch_attpool(ptr(op.in_buf), ds.cs, eptr(op.res_buf, op.res_coff), df.cs, op.cin, B, df.h, df.w,
                               c->att_ws, ptr(op.out_buf), s, df.dtype));
      return ACRMI_OK;
    }
    case ACRMI_OP_PAREBIAS: {
      PareArgs p{};
      p.pooled = ptr(op.in_buf);
      p.lc_w = c->weights + op.w_off;
      p.lin_w = c->weights + op.w_off2;
      p.lin_b = c->weights + op.b_off2;
      p.mix_wp = c->weights + op.w_off3;
      p.mix_b = c->weights + op.b_off;
      p.out = ptr(op.out_buf);
      p.B = B; p.C = op.cin; p.part0 = op.flags; p.out_stride = desc(op.out_buf).cs;
      HIPCHK(c, launch_parebias(p, s));
      return ACRMI_OK;
    }
    case ACRMI_OP_COORDFILL: {
      const auto& d = desc(op.out_buf);
      if (d.dtype) HIPCHK(c, launch_coordfill_h16(ptr(op.out_buf), c->max_batch, d.h, d.w, d.cs, op.out_coff, d.dtype, s));
      else HIPCHK(c, launch_coordfill(ptr(op.out_buf), c->max_batch, d.h, d.w, d.cs, op.out_coff, s));
      return ACRMI_OK;
    }
    case ACRMI_OP_POINTHEADS: {
      const acrmi_head_layout& h = c->heads;
      const int side = op.flags & 1;
      PointArgs p{};
      p.x34 = ptr(op.in_buf); p.x_cs = desc(op.in_buf).cs;
      p.center[0] = ptr(h.center_buf[0]); p.center[1] = ptr(h.center_buf[1]); p.center_cs = desc(h.center_buf[0]).cs;
      p.w = c->weights + op.w_off;
      p.mix_w = c->weights + op.w_off2;
      p.bias = ptr(op.aux_buf); p.bias_stride = desc(op.aux_buf).cs;
      p.p109 = ptr(op.res_buf); p.p109_cs = desc(op.res_buf).cs;
      p.prior = ptr(h.prior_buf[side]); p.prior_cs = desc(h.prior_buf[side]).cs;
      p.final_ = ptr(op.out_buf); p.final_cs = desc(op.out_buf).cs;
      p.picks = c->picks; p.side = side; p.B = B; p.thresh = c->conf_thresh;
      HIPCHK(c, launch_point_heads(p, s));
      return ACRMI_OK;
    }
    default:
      return fail(c, ACRMI_EINVAL, "unknown op kind %d", op.kind);
  }
}

// ops of the other head variant are skipped
static inline bool op_active(const acrmi_op& op, bool point) {
  return op.kind != ACRMI_OP_COORDFILL && op.mode != (point ? ACRMI_MODE_DENSE : ACRMI_MODE_POINT);
}

// Buffers an op reads / writes (whole buffers: channel slices of one buffer are ordered conservatively).
// Pseudo-buffers n_bufs / n_bufs+1 stand for the attention-pool and center-pick workspaces.
static void op_rw(const acrmi_ctx* c, const acrmi_op& op, std::vector<int>& R, std::vector<int>& W) {
  R.clear(); W.clear();
  const int n_bufs = (int)c->bufs.size();
  auto r = [&](int id) { if (id >= 0) R.push_back(id); };
  auto w = [&](int id) { if (id >= 0) W.push_back(id); };
  switch (op.kind) {
    case ACRMI_OP_U8NORM: case ACRMI_OP_STEM: w(op.out_buf); break;
    case ACRMI_OP_CONV: r(op.in_buf); r(op.res_buf); if (op.bias_per_frame) r(op.aux_buf); w(op.out_buf); break;
    case ACRMI_OP_FUSESUM: for (int t = 0; t < op.nterms; ++t) r(op.term_buf[t]); w(op.out_buf); break;
    case ACRMI_OP_BILINEAR2X: case ACRMI_OP_MAXPOOL: r(op.in_buf); w(op.out_buf); break;
    case ACRMI_OP_PAIR1X1: r(op.in_buf); r(op.res_buf); w(op.out_buf); w(op.aux_buf); break;
    case ACRMI_OP_POW11: r(op.out_buf); w(op.out_buf); break;
    case ACRMI_OP_ATTPOOL: r(op.in_buf); r(op.res_buf); w(op.out_buf); w(n_bufs); break;
    case ACRMI_OP_PAREBIAS: r(op.in_buf); w(op.out_buf); break;
    case ACRMI_OP_POINTHEADS:
      r(op.in_buf); r(op.aux_buf); r(c->heads.center_buf[0]); r(c->heads.center_buf[1]);
      w(op.res_buf); w(op.out_buf); w(c->heads.prior_buf[op.flags & 1]); w(n_bufs + 1);
      break;
    default: break;
  }
}

// Dependencies between ops: RAW/WAR/WAW hazards on buffer ids (ids are reused for disjoint lifetimes, which
// the WAR edges respect).  The program order is a topological order.
static void build_schedule(acrmi_ctx* c, bool point, bool large) {
  Schedule& S = c->sched[point ? 1 : 0][large ? 1 : 0];
  const int n = (int)c->ops.size(), nb = (int)c->bufs.size() + 2;
  S = Schedule();
  S.deps.assign(n, std::vector<int>());
  S.leaf.assign(n, 1);
  std::vector<int> last_writer(nb, -1), R, W;
  std::vector<std::vector<int>> readers(nb);
  for (int j = 0; j < n; ++j) {
    if (!op_active(c->ops[j], point)) continue;
    S.order.push_back(j);
    op_rw(c, c->ops[j], R, W);
    std::vector<int>& deps = S.deps[j];
    auto dep = [&](int i) { if (i >= 0 && i != j && std::find(deps.begin(), deps.end(), i) == deps.end()) deps.push_back(i); };
    for (int b : R) dep(last_writer[b]);
    for (int b : W) { dep(last_writer[b]); for (int i : readers[b]) dep(i); }
    for (int b : R) readers[b].push_back(j);
    for (int b : W) { last_writer[b] = j; readers[b].clear(); }
    for (int d : deps) S.leaf[d] = 0;
  }
  const int want = c->want_lanes > 0 ? c->want_lanes : (large ? AUTO_LANES_LARGE : AUTO_LANES_SMALL);
  const int max_lanes = std::max(1, std::min(want, MAX_LANES));
  S.lane.assign(n, 0);
  S.wait.assign(n, std::vector<int>());
  S.signal.assign(n, 0);
  std::vector<int> lane_tail(max_lanes, -1);
  const std::vector<float>& ms = c->op_ms[point ? 1 : 0];
  const bool planned = c->lane_plan && !large && max_lanes > 1 && (int)ms.size() == n;
  // Planned form (small batches, where a launch leaves most CUs idle and concurrent lanes really overlap): list
  // scheduling with the measured op times - every op, in program order, goes to the lane where it can START first.
  // Cost model measured on MI355X (tools/cross_stream_wait.py, tools/critical_path.py): a dependent kernel in the same
  // stream starts ~4.5 us after its producer ends, through an event on another stream ~21 us after (WAIT_MS below is the
  // difference); every other lane waited for costs the consumer's queue one barrier packet (SYNC_MS).
  constexpr float SYNC_MS = 0.002f, EVENT_MS = 0.002f;   // (EVENT_MS: what a profiled time includes)
  static const float WAIT_MS = getenv("ACRMI_PLAN_WAIT_US") ? 1e-3f * (float)atof(getenv("ACRMI_PLAN_WAIT_US")) : 0.016f;
  std::vector<float> fin(planned ? n : 0, 0.f), lane_free(max_lanes, 0.f);
  for (int j : S.order) {
    int lane = -1;
    if (planned) {
      float best = 0.f;
      bool best_prod = false;
      for (int l = 0; l < max_lanes; ++l) {
        float start = lane_free[l];
        unsigned others = 0;
        bool prod = false;
        for (int d : S.deps[j]) {
          if (S.lane[d] == l) { prod = true; continue; }
          others |= 1u << S.lane[d];
          start = std::max(start, fin[d] + WAIT_MS);
        }
        start += SYNC_MS * (float)__builtin_popcount(others);
        if (lane < 0 || start < best - 1e-6f || (start < best + 1e-6f && prod && !best_prod)) { lane = l; best = start; best_prod = prod; }
      }
      fin[j] = best + std::max(ms[j] - EVENT_MS, 0.002f);
      lane_free[lane] = fin[j] + 0.5f * SYNC_MS;
      S.n_lanes = std::max(S.n_lanes, lane + 1);
    } else {
      // structural heuristic: an op continues the lane of a producer that is still that lane's tail (the producer of
      // in_buf first), otherwise it opens a lane, or takes the one whose tail is oldest
      for (int d : S.deps[j])
        if (lane_tail[S.lane[d]] == d) { lane = S.lane[d]; break; }
      if (lane < 0) {
        if (S.n_lanes < max_lanes) lane = S.n_lanes++;
        else lane = (int)(std::min_element(lane_tail.begin(), lane_tail.end()) - lane_tail.begin());
      }
    }
    S.lane[j] = lane;
    std::vector<int> latest(max_lanes, -1);      // waiting for a lane's latest op covers its earlier ones
    for (int d : S.deps[j])
      if (S.lane[d] != lane && d > latest[S.lane[d]]) latest[S.lane[d]] = d;
    for (int l = 0; l < max_lanes; ++l)
      if (latest[l] >= 0) { S.wait[j].push_back(latest[l]); S.signal[latest[l]] = 1; }
    lane_tail[lane] = j;
  }
  if (getenv("ACRMI_DEBUG_SCHED")) {
    // depth of the DAG = kernels on the critical path (a single stream runs all of S.order in sequence)
    std::vector<int> depth(n, 0);
    int edges = 0, maxd = 0;
    for (int j : S.order) {
      for (int d : S.deps[j]) depth[j] = std::max(depth[j], depth[d] + 1);
      edges += (int)S.deps[j].size();
      maxd = std::max(maxd, depth[j] + 1);
    }
    int waits = 0;
    for (int j : S.order) waits += (int)S.wait[j].size();
    fprintf(stderr, "[acrmi] schedule(%s, %s batches, %s): %zu ops, %d edges, critical path %d ops; %d lanes, %d cross-lane waits\n",
            point ? "point" : "dense", large ? "large" : "small", planned ? "planned from measured op times" : "structural",
            S.order.size(), edges, maxd, S.n_lanes, waits);
  }
}

int acrmi_set_program(acrmi_ctx* c, const acrmi_buffer_desc* bufs, int n_bufs, const acrmi_op* ops, int n_ops,
                      const acrmi_head_layout* heads, int max_batch) {
  if (!c || !bufs || !ops || !heads || n_bufs <= 0 || n_ops <= 0 || max_batch <= 0)
    return fail(c, ACRMI_EINVAL, "acrmi_set_program: bad arguments");
  if (!c->weights) return fail(c, ACRMI_ESTATE, "acrmi_set_program: load weights first");
  ON_DEVICE(c);
  // ---- validate before anything is allocated: a malformed program must fail here, not fault on the device
  int prog_dt = ACRMI_DT_F32;      // the one 16-bit storage type of the program, if any
  for (int i = 0; i < n_bufs; ++i) {
    const auto& d = bufs[i];
    if (d.dtype < ACRMI_DT_F32 || d.dtype > ACRMI_DT_BF16) return fail(c, ACRMI_EINVAL, "buffer %d: unknown dtype %d", i, d.dtype);
    if (d.h <= 0 || d.w <= 0 || d.cs <= 0 || d.cs % (d.dtype ? 8 : 4)) return fail(c, ACRMI_EINVAL, "buffer %d: bad geometry", i);
    if (d.dtype) {
      if (prog_dt && prog_dt != d.dtype) return fail(c, ACRMI_EINVAL, "buffer %d: f16 and bf16 buffers in one program", i);
      prog_dt = d.dtype;
    }
  }
  auto bdt = [&](int id) { return id >= 0 ? bufs[id].dtype : (int)ACRMI_DT_F32; };
  auto buf_ok = [&](int id) { return id >= 0 && id < n_bufs; };
  auto w_ok = [&](long long off, long long n) { return off >= 0 && n >= 0 && (unsigned long long)(off + n) <= c->n_weights; };
  {
    const int hb[8] = {heads->center_buf[0], heads->center_buf[1], heads->params_buf[0], heads->params_buf[1],
                       heads->prior_buf[0], heads->prior_buf[1], heads->segm_buf, heads->backbone_buf};
    for (int id : hb)
      if (!buf_ok(id)) return fail(c, ACRMI_EINVAL, "head layout references buffer %d of %d", id, n_bufs);
    for (int k = 0; k < 7; ++k)      // decode / attention pooling / the host read these as fp32 (acr/model.py:56-62 .float())
      if (bufs[hb[k]].dtype != ACRMI_DT_F32) return fail(c, ACRMI_EINVAL, "head layout: head maps must be fp32 buffers");
    if (bufs[heads->params_buf[0]].cs < 109 || bufs[heads->params_buf[1]].cs < 109 || bufs[heads->prior_buf[0]].cs < 106 ||
        bufs[heads->prior_buf[1]].cs < 106)
      return fail(c, ACRMI_EINVAL, "head layout: params/prior buffers are too narrow");
    for (int k = 0; k < 6; ++k)
      if (bufs[hb[k]].h != 64 || bufs[hb[k]].w != 64) return fail(c, ACRMI_EINVAL, "head layout: head maps must be 64x64");
  }
  for (int i = 0; i < n_ops; ++i) {
    const acrmi_op& op = ops[i];
    const int ids[4] = {op.in_buf, op.out_buf, op.res_buf, op.aux_buf};
    for (int id : ids)
      if (id >= n_bufs || id < -1) return fail(c, ACRMI_EINVAL, "op %d references buffer %d of %d", i, id, n_bufs);
    if (op.mode < ACRMI_MODE_BOTH || op.mode > ACRMI_MODE_POINT) return fail(c, ACRMI_EINVAL, "op %d: bad mode", i);
    bool need_in = false, need_out = true;
    switch (op.kind) {
      case ACRMI_OP_U8NORM: case ACRMI_OP_POW11: case ACRMI_OP_COORDFILL: case ACRMI_OP_STEM: break;
      case ACRMI_OP_CONV: case ACRMI_OP_BILINEAR2X: case ACRMI_OP_MAXPOOL: case ACRMI_OP_PAIR1X1: case ACRMI_OP_ATTPOOL: case ACRMI_OP_PAREBIAS: case ACRMI_OP_POINTHEADS:
        need_in = true;
        break;
      case ACRMI_OP_FUSESUM: break;
      default: return fail(c, ACRMI_EINVAL, "op %d: unknown kind %d", i, op.kind);
    }
    if ((need_in && !buf_ok(op.in_buf)) || (need_out && !buf_ok(op.out_buf)))
      return fail(c, ACRMI_EINVAL, "op %d (kind %d): missing input/output buffer", i, op.kind);
    if (op.in_coff < 0 || op.out_coff < 0 || op.res_coff < 0) return fail(c, ACRMI_EINVAL, "op %d: negative channel offset", i);
    if (op.kind == ACRMI_OP_CONV) {
      const int idt = bufs[op.in_buf].dtype, odt = bufs[op.out_buf].dtype;
      if (op.in_coff % (idt ? 8 : 4) || (op.ksize != 1 && op.ksize != 3) || op.stride < 1 || op.stride > 2 || op.cin <= 0 || op.cout <= 0 ||
          op.groups <= 0)
        return fail(c, ACRMI_EINVAL, "op %d: unsupported conv geometry", i);
      const int algo = op.flags & 7;
      if (algo > 4) return fail(c, ACRMI_EINVAL, "op %d: unknown conv algo %d", i, algo);
      // 16-bit input: direct kernel only; the output is 16-bit too or fp32 (a head exit; 1x1 and 3x3 stride 1), a residual
      // has the type of the output.  fp32 input: everything fp32.
      if (idt ? (algo != 0 || (odt != idt && odt != ACRMI_DT_F32) || (odt == ACRMI_DT_F32 && op.stride != 1) ||
                 (op.groups > 1 && op.cin % 2))
              : odt != ACRMI_DT_F32)
        return fail(c, ACRMI_EINVAL, "op %d: conv buffer types do not fit (in %d, out %d, algo %d)", i, idt, odt, algo);
      if (op.res_buf >= 0 && bufs[op.res_buf].dtype != odt)
        return fail(c, ACRMI_EINVAL, "op %d: the residual must have the type of the output", i);
      const bool splitk = (op.flags & ACRMI_CONV_SPLITK) != 0;
      if (splitk && (algo != 2 || idt || op.groups < 2 || op.groups > 8 || op.cin % 32 || op.cin < 64 || op.cout == 33 ||
                     op.bias_per_frame))
        return fail(c, ACRMI_EINVAL, "op %d: split-K needs algo 2, fp32, 2..8 slices of Cin %% 32 == 0, Cin >= 64 channels each", i);
      if (op.flags & ACRMI_CONV_BIAS_MAP) {
        const long long mcs = (op.groups * op.cout + 3) / 4 * 4;
        if (splitk || op.res_buf >= 0 || idt || algo == 3 || !w_ok(op.w_off2, (long long)bufs[op.out_buf].h * bufs[op.out_buf].w * mcs))
          return fail(c, ACRMI_EINVAL, "op %d: a position-bias map needs an fp32 conv without a residual buffer (not algo 3) and "
                      "[Ho][Wo][round4(groups*Cout)] floats inside the blob at w_off2", i);
      }
      if (op.bias_per_frame && buf_ok(op.aux_buf) && bufs[op.aux_buf].dtype != ACRMI_DT_F32)
        return fail(c, ACRMI_EINVAL, "op %d: the per-frame bias must be fp32", i);
      if (algo != 0 && !(op.ksize == 3 && op.stride == 1))
        return fail(c, ACRMI_EINVAL, "op %d: algo %d needs a 3x3 stride-1 convolution", i, algo);
      if (algo == 3 && (op.groups != 1 || op.cin > 32 || op.cout != 32 || op.bias_per_frame || bufs[op.out_buf].h % 8 ||
                        bufs[op.out_buf].w % 16 || op.out_coff % 4 || op.res_coff % 4))
        return fail(c, ACRMI_EINVAL, "op %d: algo 3 needs groups 1, Cin <= 32, Cout = 32, a map of 8x16-pixel tiles", i);
      const int ogroups = splitk ? 1 : op.groups;      // the slices of a split-K conv share the output channels
      if (op.in_coff + op.groups * op.cin > bufs[op.in_buf].cs || op.out_coff + ogroups * op.cout > bufs[op.out_buf].cs ||
          (op.res_buf >= 0 && op.res_coff + ogroups * op.cout > bufs[op.res_buf].cs) ||
          (op.groups > 1 && op.cin % (bufs[op.in_buf].dtype == ACRMI_DT_F32 ? 4 : 8)))      // a group starts on a 16-byte vector
        return fail(c, ACRMI_EINVAL, "op %d: channel slice outside its buffer's channel stride", i);
      const int pad = op.ksize / 2;
      const int ho = (bufs[op.in_buf].h + 2 * pad - op.ksize) / op.stride + 1, wo = (bufs[op.in_buf].w + 2 * pad - op.ksize) / op.stride + 1;
      if (ho != bufs[op.out_buf].h || wo != bufs[op.out_buf].w ||
          (op.res_buf >= 0 && (bufs[op.res_buf].h != ho || bufs[op.res_buf].w != wo)))
        return fail(c, ACRMI_EINVAL, "op %d: output/residual buffer geometry does not match the convolution", i);
      const long long n_tiles = op.cout <= 32 ? 1 : ((op.cout + 63) / 64) * 2;
      if (algo == 4 && (op.cin < 32 || (op.cin == 32 && (op.cout % 32 || ho % 8 || wo % 32))))      // (Cin = 32: conv_wino24b_kernel only)
        return fail(c, ACRMI_EINVAL, "op %d: algo 4 needs Cin > 32, or Cin = 32 with Cout %% 32 = 0 on a map of 8x32-pixel tiles", i);
      const long long taps = algo == 4 ? 24 : (algo >= 2 ? 16 : (algo == 1 ? 12 : op.ksize * op.ksize));
      const long long ksteps = idt ? (op.cin + 15) / 16 : (op.cin + 7) / 8;      // 1 KiB weight fragments per tap and n-tile
      const long long wn = algo == 3 ? 16384 : (long long)op.groups * taps * ksteps * n_tiles * 256;
      if (!w_ok(op.w_off, wn)) return fail(c, ACRMI_EINVAL, "op %d: packed weights outside the blob", i);
      if (op.bias_per_frame) {
        if (!buf_ok(op.aux_buf) || bufs[op.aux_buf].cs < op.groups * op.cout)
          return fail(c, ACRMI_EINVAL, "op %d: per-frame bias buffer missing or too narrow", i);
      } else if (!w_ok(op.b_off, (long long)op.groups * n_tiles * 32)) {
        return fail(c, ACRMI_EINVAL, "op %d: bias outside the blob", i);
      }
    }
    if (op.kind == ACRMI_OP_STEM) {
      if (bufs[op.out_buf].dtype && (bufs[op.out_buf].cs % 8 || op.out_coff % 8))
        return fail(c, ACRMI_EINVAL, "op %d: a 16-bit stem output needs channel stride / offset in multiples of 8", i);
      if (op.ksize != 3 && op.ksize != 7) return fail(c, ACRMI_EINVAL, "op %d: the stem kernels are 3x3 and 7x7 (stride 2)", i);
      const bool ok = op.ksize == 7 ? stem7_shape_ok(2 * bufs[op.out_buf].h, 2 * bufs[op.out_buf].w, bufs[op.out_buf].cs, op.out_coff)
                                    : stem_shape_ok(2 * bufs[op.out_buf].h, 2 * bufs[op.out_buf].w, bufs[op.out_buf].cs, op.out_coff);
      if (op.cout != 64 || !ok)
        return fail(c, ACRMI_EINVAL, "op %d: the stem kernel needs 64 output channels and a map of 8x64-pixel strips", i);
      if (!w_ok(op.w_off, (op.ksize == 7 ? 74 : 14) * 2 * 64) || !w_ok(op.b_off, 64))
        return fail(c, ACRMI_EINVAL, "op %d: stem weights outside the blob", i);
    }
    if (op.kind == ACRMI_OP_FUSESUM) {
      const int vq = bufs[op.out_buf].dtype ? 8 : 4;      // elements per 16-byte vector
      if (op.nterms < 1 || op.nterms > 4 || op.cout <= 0 || op.cout % vq || op.out_coff % vq || op.out_coff + op.cout > bufs[op.out_buf].cs)
        return fail(c, ACRMI_EINVAL, "op %d: bad fuse-sum geometry", i);
      for (int t = 0; t < op.nterms; ++t) {
        if (!buf_ok(op.term_buf[t]) || op.term_coff[t] < 0 || bufs[op.term_buf[t]].dtype != bufs[op.out_buf].dtype ||
            op.term_coff[t] % vq || op.term_shift[t] < 0 || op.term_shift[t] > 3 ||
            op.term_coff[t] + op.cout > bufs[op.term_buf[t]].cs ||
            (bufs[op.term_buf[t]].h << op.term_shift[t]) != bufs[op.out_buf].h ||
            (bufs[op.term_buf[t]].w << op.term_shift[t]) != bufs[op.out_buf].w)
          return fail(c, ACRMI_EINVAL, "op %d: fuse-sum term %d does not fit the output", i, t);
      }
    }
    if (op.kind == ACRMI_OP_PAIR1X1) {
      if (!buf_ok(op.res_buf) || !buf_ok(op.aux_buf) || op.cin != 64 || op.cout != 256)
        return fail(c, ACRMI_EINVAL, "op %d: the 1x1 pair is 64 -> 256 (+ residual) -> 64 with in, res, out and aux buffers", i);
      const int ids4[4] = {op.in_buf, op.res_buf, op.out_buf, op.aux_buf};
      for (int id : ids4)
        if (bufs[id].dtype != ACRMI_DT_F32 || bufs[id].h != bufs[op.in_buf].h || bufs[id].w != bufs[op.in_buf].w)
          return fail(c, ACRMI_EINVAL, "op %d: the 1x1 pair's buffers must be fp32 maps of one size", i);
      if (op.in_coff % 4 || op.res_coff % 4 || op.out_coff % 4 || op.in_coff + 64 > bufs[op.in_buf].cs ||
          op.res_coff + 256 > bufs[op.res_buf].cs || op.out_coff + 256 > bufs[op.out_buf].cs || bufs[op.aux_buf].cs < 64 ||
          op.out_buf == op.in_buf || op.aux_buf == op.in_buf || op.aux_buf == op.out_buf || op.aux_buf == op.res_buf)
        return fail(c, ACRMI_EINVAL, "op %d: the 1x1 pair's channel slices do not fit / its buffers alias", i);
      if (!w_ok(op.w_off, PAIR1X1_FLOATS)) return fail(c, ACRMI_EINVAL, "op %d: pair weights outside the blob", i);
    }
    if (op.kind == ACRMI_OP_MAXPOOL) {
      const int vq = bufs[op.in_buf].dtype ? 8 : 4;
      if (op.cin <= 0 || bufs[op.in_buf].dtype != bufs[op.out_buf].dtype || op.cin % vq || op.in_coff % vq || op.out_coff % vq ||
          op.in_coff + op.cin > bufs[op.in_buf].cs || op.out_coff + op.cin > bufs[op.out_buf].cs ||
          bufs[op.out_buf].h != (bufs[op.in_buf].h - 1) / 2 + 1 || bufs[op.out_buf].w != (bufs[op.in_buf].w - 1) / 2 + 1)
        return fail(c, ACRMI_EINVAL, "op %d: bad max-pool geometry", i);
    }
    if (op.kind == ACRMI_OP_BILINEAR2X &&
        (op.cin <= 0 || bufs[op.in_buf].dtype != bufs[op.out_buf].dtype || op.cin % (bufs[op.in_buf].dtype ? 8 : 4) ||
         op.in_coff % (bufs[op.in_buf].dtype ? 8 : 4) || op.out_coff % (bufs[op.in_buf].dtype ? 8 : 4) || op.in_coff + op.cin > bufs[op.in_buf].cs ||
         op.out_coff + op.cin > bufs[op.out_buf].cs || bufs[op.out_buf].h != 2 * bufs[op.in_buf].h ||
         bufs[op.out_buf].w != 2 * bufs[op.in_buf].w))
      return fail(c, ACRMI_EINVAL, "op %d: bad bilinear geometry", i);
    if ((op.kind == ACRMI_OP_POW11 && op.out_coff >= bufs[op.out_buf].cs) ||
        (op.kind == ACRMI_OP_COORDFILL && op.out_coff + 2 > bufs[op.out_buf].cs))
      return fail(c, ACRMI_EINVAL, "op %d: channel outside the buffer", i);
    if (op.kind == ACRMI_OP_ATTPOOL) {
      if (buf_ok(op.res_buf) && (bufs[op.in_buf].dtype != ACRMI_DT_F32 || bufs[op.out_buf].dtype != ACRMI_DT_F32 ||
                                 (bufs[op.res_buf].dtype != ACRMI_DT_F32 && op.cin != 256)))
        return fail(c, ACRMI_EINVAL, "op %d: attention pooling reads fp32 logits, writes fp32, and pools 256 channels of a 16-bit map", i);
      if (!buf_ok(op.res_buf) || (op.cin != 32 && op.cin != 64 && op.cin != 256 && op.cin != 320) ||
          op.res_coff + op.cin > bufs[op.res_buf].cs || bufs[op.in_buf].cs < 33 || bufs[op.in_buf].h != 2 * bufs[op.res_buf].h ||
          bufs[op.in_buf].w != 2 * bufs[op.res_buf].w || (long long)bufs[op.out_buf].h * bufs[op.out_buf].w * bufs[op.out_buf].cs < 32LL * op.cin)
        return fail(c, ACRMI_EINVAL, "op %d: bad attention-pool geometry", i);
    }
    if ((op.kind == ACRMI_OP_PAREBIAS || op.kind == ACRMI_OP_POINTHEADS || op.kind == ACRMI_OP_U8NORM) &&
        (bdt(op.in_buf) || bdt(op.out_buf) || bdt(op.res_buf) || bdt(op.aux_buf)))
      return fail(c, ACRMI_EINVAL, "op %d (kind %d): fp32 buffers only", i, op.kind);
    if (op.kind == ACRMI_OP_PAREBIAS) {
      const long long shape_n = (op.cin == 320 ? 64 : 256) * 16;
      if ((op.cin != 256 && op.cin != 320) || (op.flags != 0 && op.flags != 16) || bufs[op.out_buf].cs < 109 || bufs[op.out_buf].cs > 256 ||
          (long long)bufs[op.in_buf].h * bufs[op.in_buf].w * bufs[op.in_buf].cs < 32LL * op.cin || !w_ok(op.w_off, 6 * 256 * 16) ||
          !w_ok(op.w_off2, 10 * shape_n) || !w_ok(op.b_off2, 10) || !w_ok(op.w_off3, 109 * 106) || !w_ok(op.b_off, 109))
        return fail(c, ACRMI_EINVAL, "op %d: bad pare-bias geometry or weights", i);
    }
    if (op.kind == ACRMI_OP_POINTHEADS) {
      const bool ok = buf_ok(op.res_buf) && buf_ok(op.aux_buf) &&
                      bufs[op.in_buf].h == 128 && bufs[op.in_buf].w == 128 && bufs[op.in_buf].cs == 36 &&
                      bufs[op.res_buf].h == 64 && bufs[op.out_buf].h == 64 && bufs[op.res_buf].cs >= 109 &&
                      bufs[op.out_buf].cs >= 109 && bufs[op.aux_buf].cs >= 109 && op.mode == ACRMI_MODE_POINT &&
                      w_ok(op.w_off, 3LL * TP_TOWER_FLOATS) && w_ok(op.w_off2, 109LL * TP_EXIT_N);
      if (!ok) return fail(c, ACRMI_EINVAL, "op %d: unsupported point-heads geometry", i);
    }
  }
  free_program(c);
  c->bufs.assign(bufs, bufs + n_bufs);
  c->ops.assign(ops, ops + n_ops);
  c->heads = *heads;
  c->max_batch = max_batch;
  c->buf_ptr.assign(n_bufs, nullptr);
  for (int i = 0; i < n_bufs; ++i) {
    const auto& d = bufs[i];
    const size_t bytes = (size_t)max_batch * d.h * d.w * d.cs * (d.dtype ? 2 : sizeof(float));
    hipError_t e = hipMalloc(&c->buf_ptr[i], bytes);
    if (e != hipSuccess) return fail(c, ACRMI_ENOMEM, "hipMalloc(%zu) for buffer %d: %s", bytes, i, hipGetErrorString(e));
    HIPCHK(c, hipMemset(c->buf_ptr[i], 0, bytes));
  }
  for (int i = 0; i < n_ops; ++i) {
    const acrmi_op& op = ops[i];
    if (op.kind != ACRMI_OP_CONV || !(op.flags & ACRMI_CONV_SPLITK)) continue;
    const auto& d = bufs[op.out_buf];
    c->split_ws_floats = std::max(c->split_ws_floats, conv_splitk_ws_floats(max_batch, d.h, d.w, op.cout, op.groups));
    c->split_counters = std::max(c->split_counters, conv_splitk_counters(max_batch, d.h, d.w, op.cout));
  }
  if (c->split_ws_floats)
    for (int l = 0; l < MAX_LANES; ++l) {
      HIPCHK(c, hipMalloc(&c->split_ws[l], c->split_ws_floats * sizeof(float)));
      HIPCHK(c, hipMalloc(&c->split_cnt[l], c->split_counters * sizeof(unsigned)));
      HIPCHK(c, hipMemset(c->split_cnt[l], 0, c->split_counters * sizeof(unsigned)));
    }
  c->att_ws_floats = attpool_ws_floats(max_batch, 320);
  HIPCHK(c, hipMalloc(&c->att_ws, c->att_ws_floats * sizeof(float)));
  HIPCHK(c, hipMalloc(&c->picks, (size_t)max_batch * 4 * sizeof(int)));
  c->op_ms[0].clear(); c->op_ms[1].clear();      // measured times belong to the previous program
  for (int v = 0; v < 4; ++v) build_schedule(c, v & 1, v & 2);
  c->op_ev.assign(n_ops, nullptr);
  c->have_program = true;
  // init-time ops (constants that live in persistent buffers)
  for (const acrmi_op& op : c->ops)
    if (op.kind == ACRMI_OP_COORDFILL) {
      int r = run_op(c, op, nullptr, max_batch, nullptr);
      if (r) return r;
    }
  HIPCHK(c, hipDeviceSynchronize());
  return ACRMI_OK;
}

int acrmi_load_mano(acrmi_ctx* c, int side, const float* v_template, const float* shapedirs, const float* posedirs,
                    const float* J_regressor, const float* weights, const float* hands_mean) {
  if (!c || side < 0 || side > 1 || !v_template || !shapedirs || !posedirs || !J_regressor || !weights || !hands_mean)
    return fail(c, ACRMI_EINVAL, "acrmi_load_mano: bad arguments");
  ON_DEVICE(c);
  // a reload replaces the side's tables: nothing of an earlier launch may still be reading the old ones
  HIPCHK(c, hipDeviceSynchronize());
  for (float*& p : c->mano_allocs[side]) {
    if (p) (void)hipFree(p);
    p = nullptr;
  }
  c->have_mano[side] = false;
  constexpr int NV3 = 2334;
  std::vector<float> sd_t((size_t)10 * NV3), pd_t((size_t)135 * NV3);
  for (int i = 0; i < NV3; ++i) {
    for (int k = 0; k < 10; ++k) sd_t[(size_t)k * NV3 + i] = shapedirs[(size_t)i * 10 + k];
    for (int k = 0; k < 135; ++k) pd_t[(size_t)k * NV3 + i] = posedirs[(size_t)i * 135 + k];
  }
  int n_up = 0;
  auto up = [&](const float* h, size_t n, const float** dst) -> int {
    float* d = nullptr;
    HIPCHK(c, hipMalloc(&d, n * sizeof(float)));
    c->mano_allocs[side][n_up++] = d;
    HIPCHK(c, hipMemcpy(d, h, n * sizeof(float), hipMemcpyHostToDevice));
    *dst = d;
    return ACRMI_OK;
  };
  ManoTables& t = c->mano[side];
  int r;
  if ((r = up(v_template, NV3, &t.v_template))) return r;
  if ((r = up(sd_t.data(), sd_t.size(), &t.shapedirs_t))) return r;
  if ((r = up(pd_t.data(), pd_t.size(), &t.posedirs_t))) return r;
  if ((r = up(J_regressor, 16 * 778, &t.jreg))) return r;
  if ((r = up(weights, 778 * 16, &t.weights))) return r;
  if ((r = up(hands_mean, 45, &t.hands_mean))) return r;
  // f16 copies of the blend-shape tables and the skinning weights (ACRMI_OPT_MANO_FP16), rounded to nearest even
  auto up16 = [&](const float* h, size_t n, const unsigned short** dst) -> int {
    std::vector<unsigned short> bits(n + (n & 1));
    for (size_t i = 0; i < n; ++i) {
      const _Float16 v = (_Float16)h[i];
      memcpy(&bits[i], &v, 2);
    }
    float* d = nullptr;
    HIPCHK(c, hipMalloc(&d, bits.size() * 2));
    c->mano_allocs[side][n_up++] = d;
    HIPCHK(c, hipMemcpy(d, bits.data(), bits.size() * 2, hipMemcpyHostToDevice));
    *dst = reinterpret_cast<const unsigned short*>(d);
    return ACRMI_OK;
  };
  if ((r = up16(sd_t.data(), sd_t.size(), &t.shapedirs_h))) return r;
  if ((r = up16(pd_t.data(), pd_t.size(), &t.posedirs_h))) return r;
  if ((r = up16(weights, 778 * 16, &t.weights_h))) return r;
  c->have_mano[side] = true;
  return ACRMI_OK;
}

// The program with its independent chains on parallel streams: lane 0 is the caller's stream, the other lanes fork
// from it (so they start after everything queued before this call) and join it at the end.
static unsigned lane_event_flags() {
  static const unsigned f = [] {
    const char* e = getenv("ACRMI_EVENT_FLAGS");      // experiment switch: extra hipEventCreateWithFlags bits (hex)
    if (e) fprintf(stderr, "[acrmi] WARNING: ACRMI_EVENT_FLAGS=%s - lane events created with non-default flags (experiment)\n", e);
    return hipEventDisableTiming | (e ? (unsigned)strtoul(e, nullptr, 16) : 0u);
  }();
  return f;
}

static int run_program_lanes(acrmi_ctx* c, const uint8_t* img, int B, hipStream_t user, bool point) {
  const Schedule& S = c->sched[point ? 1 : 0][B > AUTO_SMALL_BATCH ? 1 : 0];
  for (int l = 1; l < S.n_lanes; ++l) {
    if (!c->lanes[l]) HIPCHK(c, hipStreamCreateWithFlags(&c->lanes[l], hipStreamNonBlocking));
    if (!c->join_ev[l]) HIPCHK(c, hipEventCreateWithFlags(&c->join_ev[l], lane_event_flags()));
  }
  if (!c->fork_ev) HIPCHK(c, hipEventCreateWithFlags(&c->fork_ev, lane_event_flags()));
  auto st = [&](int l) { return l == 0 ? user : c->lanes[l]; };
  HIPCHK(c, hipEventRecord(c->fork_ev, user));
  for (int l = 1; l < S.n_lanes; ++l) HIPCHK(c, hipStreamWaitEvent(c->lanes[l], c->fork_ev, 0));
  int r = ACRMI_OK;
  for (int j : S.order) {
    hipStream_t s = st(S.lane[j]);
    // timing ablation (WRONG results: data races between lanes): 1 = no waits, 2 = no waits and no records.  Loud, so that a
    // leaked environment variable cannot silently corrupt a real run.
    static const int ablate = [] {
      const char* e = getenv("ACRMI_ABLATE_LANE_SYNC");
      const int v = e ? atoi(e) : 0;
      if (v) fprintf(stderr, "[acrmi] WARNING: ACRMI_ABLATE_LANE_SYNC=%d - cross-lane synchronisation is DISABLED, results are WRONG "
                             "(timing experiment only)\n", v);
      return v;
    }();
    if (!(ablate & 1)) for (int d : S.wait[j]) HIPCHK(c, hipStreamWaitEvent(s, c->op_ev[d], 0));
    r = run_op(c, c->ops[j], img, B, s, S.lane[j]);
    if (r) break;
    if (S.signal[j] && !(ablate & 2)) {
      if (!c->op_ev[j]) HIPCHK(c, hipEventCreateWithFlags(&c->op_ev[j], lane_event_flags()));
      HIPCHK(c, hipEventRecord(c->op_ev[j], s));
    }
  }
  for (int l = 1; l < S.n_lanes; ++l) {       // join also on the error path: nothing may outlive the call unordered
    HIPCHK(c, hipEventRecord(c->join_ev[l], c->lanes[l]));
    HIPCHK(c, hipStreamWaitEvent(user, c->join_ev[l], 0));
  }
  return r;
}

static int run_program(acrmi_ctx* c, const uint8_t* img, int B, void* stream, bool point) {
  if (!c || !img) return fail(c, ACRMI_EINVAL, "acrmi_backbone_heads: bad arguments");
  if (!c->have_program) return fail(c, ACRMI_ESTATE, "acrmi_backbone_heads: no program");
  if (B <= 0 || B > c->max_batch) return fail(c, ACRMI_EINVAL, "batch %d outside 1..%d", B, c->max_batch);
  ON_DEVICE(c);
  static const bool dbg_sync = getenv("ACRMI_DEBUG_SYNC") != nullptr;   // attribute a fault/hang to an op
  if (c->sched[point ? 1 : 0][B > AUTO_SMALL_BATCH ? 1 : 0].n_lanes > 1 && !dbg_sync)
    return run_program_lanes(c, img, B, (hipStream_t)stream, point);
  int i = 0;
  for (const acrmi_op& op : c->ops) {
    ++i;
    if (!op_active(op, point)) continue;
    if (dbg_sync) fprintf(stderr, "[acrmi] op %d kind %d B %d\n", i - 1, (int)op.kind, B), fflush(stderr);
    int r = run_op(c, op, img, B, (hipStream_t)stream);
    if (r) return r;
    if (dbg_sync) HIPCHK(c, hipStreamSynchronize((hipStream_t)stream));
  }
  return ACRMI_OK;
}

int acrmi_backbone_heads(acrmi_ctx* c, const uint8_t* img, int B, void* stream) {
  return run_program(c, img, B, stream, /*point=*/false);   // the dense maps are this call's result
}

int acrmi_point_heads(acrmi_ctx* c, int B, void* stream) {
  if (!c) return fail(c, ACRMI_EINVAL, "acrmi_point_heads: ctx is NULL");
  if (!c->have_program) return fail(c, ACRMI_ESTATE, "acrmi_point_heads: no program");
  if (B <= 0 || B > c->max_batch) return fail(c, ACRMI_EINVAL, "batch %d outside 1..%d", B, c->max_batch);
  ON_DEVICE(c);
  int n = 0;
  for (const acrmi_op& op : c->ops)
    if (op.kind == ACRMI_OP_POINTHEADS) {
      int r = run_op(c, op, nullptr, B, (hipStream_t)stream);
      if (r) return r;
      ++n;
    }
  if (!n) return fail(c, ACRMI_EINVAL, "acrmi_point_heads: the program has no point-heads ops");
  return ACRMI_OK;
}

int acrmi_set_option(acrmi_ctx* c, int option, int value) {
  if (!c) return fail(c, ACRMI_EINVAL, "acrmi_set_option: ctx is NULL");
  if (option == ACRMI_OPT_POINT_HEADS) {
    if (value && c->have_program) {
      bool any = false;
      for (const acrmi_op& op : c->ops) any |= op.kind == ACRMI_OP_POINTHEADS;
      if (!any) return fail(c, ACRMI_EINVAL, "acrmi_set_option: the program has no point-heads ops");
    }
    c->point_heads = value != 0;
    return ACRMI_OK;
  }
  if (option == ACRMI_OPT_LANES) {
    if (value < 0 || value > MAX_LANES) return fail(c, ACRMI_EINVAL, "acrmi_set_option: lanes %d outside 0..%d", value, MAX_LANES);
    c->want_lanes = value;
    if (c->have_program)
      for (int v = 0; v < 4; ++v) build_schedule(c, v & 1, v & 2);
    return ACRMI_OK;
  }
  if (option == ACRMI_OPT_LANE_PLAN) {
    c->lane_plan = value != 0;
    if (c->have_program)
      for (int v = 0; v < 4; ++v) build_schedule(c, v & 1, v & 2);
    return ACRMI_OK;
  }
  if (option == ACRMI_OPT_CENTER_IDX) {
    if (value < -1 || value > 20) return fail(c, ACRMI_EINVAL, "acrmi_set_option: center_idx %d outside -1..20", value);
    c->center_idx = value;
    return ACRMI_OK;
  }
  if (option == ACRMI_OPT_TEMPORAL) {
    c->temporal = value != 0;
    return ACRMI_OK;
  }
  if (option == ACRMI_OPT_MANO_FP16) {
    c->mano_f16 = value != 0;
    return ACRMI_OK;
  }
  return fail(c, ACRMI_EINVAL, "acrmi_set_option: unknown option %d", option);
}

int acrmi_set_option_f(acrmi_ctx* c, int option, float value) {
  if (!c) return fail(c, ACRMI_EINVAL, "acrmi_set_option_f: ctx is NULL");
  if (option == ACRMI_OPT_CONF_THRESH) {
    if (!(value == value)) return fail(c, ACRMI_EINVAL, "acrmi_set_option_f: threshold is NaN");
    c->conf_thresh = value;
    return ACRMI_OK;
  }
  if (option == ACRMI_OPT_SMOOTH_COEFF) {
    if (!(value > 0.f)) return fail(c, ACRMI_EINVAL, "acrmi_set_option_f: smooth_coeff must be > 0");
    c->smooth_coeff = value;
    return ACRMI_OK;
  }
  return fail(c, ACRMI_EINVAL, "acrmi_set_option_f: unknown option %d", option);
}

// ---- temporal smoothing (acr/main.py:69-83) -------------------------------------------------------------
constexpr size_t SMOOTH_STATE_BYTES = 2 * 3 * 64 * sizeof(float) + 2 * sizeof(int);

int acrmi_smooth_reset(acrmi_ctx* c, void* stream) {
  if (!c) return fail(c, ACRMI_EINVAL, "acrmi_smooth_reset: ctx is NULL");
  ON_DEVICE(c);
  if (!c->smooth_state) HIPCHK(c, hipMalloc(&c->smooth_state, SMOOTH_STATE_BYTES));
  HIPCHK(c, hipMemsetAsync(c->smooth_state, 0, SMOOTH_STATE_BYTES, (hipStream_t)stream));
  return ACRMI_OK;
}

int acrmi_smooth(acrmi_ctx* c, float* slots, int B, void* stream) {
  if (!c || !slots || B <= 0) return fail(c, ACRMI_EINVAL, "acrmi_smooth: bad arguments");
  ON_DEVICE(c);
  if (!c->smooth_state) {
    int r = acrmi_smooth_reset(c, stream);
    if (r) return r;
  }
  SmoothArgs a{};
  a.slots = slots; a.B = B;
  a.state = c->smooth_state;
  a.init = reinterpret_cast<int*>(c->smooth_state + 2 * 3 * 64);
  // create_OneEuroFilter (acr/utils.py:1472-1473): poses / global_orient (smooth_coeff, 0.7), betas (0.6, 0.7);
  // dcutoff 1.0, freq 30.  The derivative filter's alpha is a python double rounded once when it meets the tensor.
  a.mincutoff = c->smooth_coeff; a.mincutoff_betas = 0.6f; a.beta = 0.7f; a.freq = 30.f;
  const double te = 1.0 / 30.0, tau = 1.0 / (2 * M_PI * 1.0), alpha_d = 1.0 / (1.0 + tau / te);
  a.alpha_d = (float)alpha_d; a.one_minus_alpha_d = (float)(1.0 - alpha_d);
  a.two_pi = (float)(2 * M_PI); a.te = (float)te;
  HIPCHK(c, launch_smooth(a, (hipStream_t)stream));
  return ACRMI_OK;
}

int acrmi_profile_ops(acrmi_ctx* c, const uint8_t* img, int B, float* ms_out, int n_ms, void* stream) {
  if (!c || !img || !ms_out) return fail(c, ACRMI_EINVAL, "acrmi_profile_ops: bad arguments");
  if (!c->have_program) return fail(c, ACRMI_ESTATE, "no program");
  const int n = (int)c->ops.size();
  if (n_ms < n) return fail(c, ACRMI_EINVAL, "ms_out too small (%d < %d)", n_ms, n);
  ON_DEVICE(c);
  hipStream_t s = (hipStream_t)stream;
  const bool point = c->point_heads;
  int r = run_program(c, img, B, stream, point);
  if (r) return r;
  std::vector<hipEvent_t> ev(n + 1, nullptr);
  auto cleanup = [&]() {
    for (auto& e : ev)
      if (e) (void)hipEventDestroy(e);
  };
  hipError_t he = hipSuccess;
  for (auto& e : ev)
    if ((he = hipEventCreate(&e)) != hipSuccess) break;
  if (he == hipSuccess) he = hipEventRecord(ev[0], s);
  for (int i = 0; i < n && he == hipSuccess && r == ACRMI_OK; ++i) {
    if (op_active(c->ops[i], point)) r = run_op(c, c->ops[i], img, B, s);
    if (r == ACRMI_OK) he = hipEventRecord(ev[i + 1], s);
  }
  if (he == hipSuccess && r == ACRMI_OK) he = hipStreamSynchronize(s);
  for (int i = 0; i < n && he == hipSuccess && r == ACRMI_OK; ++i) he = hipEventElapsedTime(&ms_out[i], ev[i], ev[i + 1]);
  cleanup();
  if (r) return r;
  if (he != hipSuccess) return fail(c, ACRMI_EHIP, "acrmi_profile_ops: %s", hipGetErrorString(he));
  if (B <= AUTO_SMALL_BATCH) {      // the small-batch schedules are planned from these times (ACRMI_OPT_LANE_PLAN)
    c->op_ms[point ? 1 : 0].assign(ms_out, ms_out + n);
    build_schedule(c, point, false);
  }
  return n;
}

void* acrmi_buffer_ptr(acrmi_ctx* c, int buf, int* h, int* w, int* cs) {
  if (!c || !c->have_program || buf < 0 || buf >= (int)c->bufs.size()) return nullptr;
  if (h) *h = c->bufs[buf].h;
  if (w) *w = c->bufs[buf].w;
  if (cs) *cs = c->bufs[buf].cs;
  return c->buf_ptr[buf];
}

int acrmi_buffer_dtype(acrmi_ctx* c, int buf) {
  if (!c || !c->have_program || buf < 0 || buf >= (int)c->bufs.size()) return -1;
  return c->bufs[buf].dtype;
}

int acrmi_decode_maps_gated(const float* l_center, const float* r_center, int center_cs, const float* l_params,
                            const float* r_params, int params_cs, const float* l_prior, const float* r_prior, int prior_cs,
                            int B, float conf_thresh, const int32_t* prior_gate, float* slots, void* stream) {
  if (!l_center || !r_center || !l_params || !r_params || !l_prior || !r_prior || !slots || B <= 0 || params_cs < 109 ||
      prior_cs < 106 || center_cs < 1 || !(conf_thresh == conf_thresh))
    return fail(nullptr, ACRMI_EINVAL, "acrmi_decode_maps: bad arguments");
  DecodeArgs d{};
  d.center[0] = l_center; d.center[1] = r_center; d.center_cs = center_cs;
  d.params[0] = l_params; d.params[1] = r_params; d.params_cs = params_cs;
  d.prior[0] = l_prior; d.prior[1] = r_prior; d.prior_cs = prior_cs;
  d.B = B; d.slots = slots; d.thresh = conf_thresh;
  d.prior_gate = prior_gate;
  hipError_t e = launch_decode(d, (hipStream_t)stream);
  if (e != hipSuccess) return fail(nullptr, ACRMI_EHIP, "decode launch: %s", hipGetErrorString(e));
  return ACRMI_OK;
}

int acrmi_decode_maps(const float* l_center, const float* r_center, int center_cs, const float* l_params,
                      const float* r_params, int params_cs, const float* l_prior, const float* r_prior, int prior_cs,
                      int B, float conf_thresh, float* slots, void* stream) {
  return acrmi_decode_maps_gated(l_center, r_center, center_cs, l_params, r_params, params_cs, l_prior, r_prior, prior_cs, B,
                                 conf_thresh, nullptr, slots, stream);
}

int acrmi_decode(acrmi_ctx* c, int B, float* slots, void* stream) { return acrmi_decode_gated(c, B, nullptr, slots, stream); }

int acrmi_decode_gated(acrmi_ctx* c, int B, const int32_t* prior_gate, float* slots, void* stream) {
  if (!c || !slots) return fail(c, ACRMI_EINVAL, "acrmi_decode: bad arguments");
  if (!c->have_program) return fail(c, ACRMI_ESTATE, "acrmi_decode: no program");
  if (B <= 0 || B > c->max_batch) return fail(c, ACRMI_EINVAL, "batch %d outside 1..%d", B, c->max_batch);
  ON_DEVICE(c);
  const acrmi_head_layout& h = c->heads;
  int r = acrmi_decode_maps_gated(c->buf_ptr[h.center_buf[0]], c->buf_ptr[h.center_buf[1]], c->bufs[h.center_buf[0]].cs,
                                  c->buf_ptr[h.params_buf[0]], c->buf_ptr[h.params_buf[1]], c->bufs[h.params_buf[0]].cs,
                                  c->buf_ptr[h.prior_buf[0]], c->buf_ptr[h.prior_buf[1]], c->bufs[h.prior_buf[0]].cs, B,
                                  c->conf_thresh, prior_gate, slots, stream);
  if (r) c->err = g_err;
  return r;
}

int acrmi_mano(acrmi_ctx* c, const float* poses, int pose_stride, const float* betas, int beta_stride,
               const int32_t* side, int H, int center_idx, float* verts, float* joints, float* center,
               const float* cam, int cam_stride, const float* offsets, float* verts_camed, float* pj2d,
               float* pj2d_org, void* stream) {
  if (!c) return fail(c, ACRMI_EINVAL, "acrmi_mano: ctx is NULL");
  if (H == 0) return ACRMI_OK;    // ManoLayer accepts N == 0 (acr/mano_wrapper.py:43 comment)
  if (H < 0 || !poses || !betas || !verts || !joints || center_idx >= 21)
    return fail(c, ACRMI_EINVAL, "acrmi_mano: bad arguments");
  if (!c->have_mano[0] && !c->have_mano[1]) return fail(c, ACRMI_ESTATE, "acrmi_mano: MANO tables not loaded");
  ON_DEVICE(c);
  ManoArgs m{};
  // a context may hold one side only (a lone ManoLayer); rows must then all be of that side
  m.t[0] = c->have_mano[0] ? c->mano[0] : c->mano[1];
  m.t[1] = c->have_mano[1] ? c->mano[1] : c->mano[0];
  m.poses = poses; m.pose_stride = pose_stride; m.betas = betas; m.beta_stride = beta_stride;
  m.side = side; m.H = H; m.center_idx = center_idx;
  m.verts = verts; m.joints = joints; m.center = center;
  m.cam = cam; m.cam_stride = cam_stride; m.offsets = offsets; m.off_div = 1;
  m.verts_camed = verts_camed; m.pj2d = pj2d; m.pj2d_org = pj2d_org;
  m.lbs_f16 = c->mano_f16;
  HIPCHK(c, launch_mano(m, (hipStream_t)stream));
  return ACRMI_OK;
}

// decode + MANO of acrmi_forward on one stream
static int forward_tail(acrmi_ctx* c, int B, const float* offsets, float* slots, float* verts, float* joints,
                        float* verts_camed, float* pj2d, float* pj2d_org, hipStream_t stream) {
  int r = acrmi_decode(c, B, slots, stream);
  if (r) return r;
  if (c->temporal && (r = acrmi_smooth(c, slots, B, stream))) return r;   // acr/main.py:69-83, before MANO
  ManoArgs m{};
  m.t[0] = c->mano[0]; m.t[1] = c->mano[1];
  m.poses = slots + ACRMI_SLOT_POSES; m.pose_stride = ACRMI_SLOT;
  m.betas = slots + ACRMI_SLOT_BETAS; m.beta_stride = ACRMI_SLOT;
  m.side = nullptr; m.H = 2 * B; m.center_idx = c->center_idx;
  m.verts = verts; m.joints = joints; m.center = nullptr;
  const bool proj = verts_camed || pj2d || pj2d_org;
  m.cam = proj ? slots + ACRMI_SLOT_CAM : nullptr; m.cam_stride = ACRMI_SLOT;
  m.offsets = offsets; m.off_div = 2;
  m.verts_camed = verts_camed; m.pj2d = pj2d; m.pj2d_org = pj2d_org;
  m.lbs_f16 = c->mano_f16;
  HIPCHK(c, launch_mano(m, stream));
  return ACRMI_OK;
}

int acrmi_forward(acrmi_ctx* c, const uint8_t* img, int B, const float* offsets, float* slots, float* verts,
                  float* joints, float* verts_camed, float* pj2d, float* pj2d_org, void* stream) {
  if (!c || !slots || !verts || !joints) return fail(c, ACRMI_EINVAL, "acrmi_forward: bad arguments");
  if (!c->have_mano[0] || !c->have_mano[1]) return fail(c, ACRMI_ESTATE, "acrmi_forward: MANO tables not loaded");
  ON_DEVICE(c);
  int r = run_program(c, img, B, stream, c->point_heads);
  if (r) return r;
  return forward_tail(c, B, offsets, slots, verts, joints, verts_camed, pj2d, pj2d_org, (hipStream_t)stream);
}

// ---- stand-alone operators -------------------------------------------------------------------------
int acrmi_conv2d(const float* in, int B, int H, int W, int in_cs, int in_coff, int cin, const float* w_packed,
                 const float* bias, int bias_frame_stride, const float* res, int res_cs, int res_coff, float* out,
                 int out_cs, int out_coff, int cout, int ksize, int stride, int relu, int groups, int algo,
                 void* stream) {
  if (!in || !w_packed || !bias || !out || B <= 0 || H <= 0 || W <= 0 || cin <= 0 || cout <= 0 || groups <= 0)
    return fail(nullptr, ACRMI_EINVAL, "acrmi_conv2d: bad arguments");
  const int bias_map = algo >= 0 ? (algo & ACRMI_CONV_BIAS_MAP) : 0;      // res = ONE map [Ho][Wo][res_cs] for all frames
  if (algo >= 0) algo &= ~ACRMI_CONV_BIAS_MAP;
  if (bias_map && (!res || algo == 3))
    return fail(nullptr, ACRMI_EINVAL, "acrmi_conv2d: ACRMI_CONV_BIAS_MAP needs res (the map) and an algo other than 3");
  if (algo != 0 && !((algo >= 1 && algo <= 4) && ksize == 3 && stride == 1))
    return fail(nullptr, ACRMI_EINVAL, "acrmi_conv2d: algo %d needs a 3x3 stride-1 convolution", algo);
  if (algo == 3 && (groups != 1 || cin > 32 || cout != 32 || bias_frame_stride != 0 || H % 8 || W % 16 || out_cs % 4 ||
                    out_coff % 4 || (res && (res_cs % 4 || res_coff % 4))))
    return fail(nullptr, ACRMI_EINVAL, "acrmi_conv2d: algo 3 needs groups 1, Cin <= 32, Cout = 32, H %% 8 == 0, W %% 16 == 0");
  if (algo == 4 && (cin < 32 || (cin == 32 && (cout % 32 || H % 8 || W % 32))))
    return fail(nullptr, ACRMI_EINVAL, "acrmi_conv2d: algo 4 needs Cin > 32 (or Cin = 32, Cout %% 32 = 0 on a map of 8x32-pixel tiles)");
  if ((ksize != 1 && ksize != 3) || (stride != 1 && stride != 2))
    return fail(nullptr, ACRMI_EINVAL, "acrmi_conv2d: only 3x3 and 1x1 at stride 1 / 2 are implemented (got k%d s%d)", ksize, stride);
  if (in_cs % 4 || in_coff % 4 || (groups > 1 && cin % 4))
    return fail(nullptr, ACRMI_EINVAL, "acrmi_conv2d: input channel stride/offset must be multiples of 4");
  if (in_coff < 0 || out_coff < 0 || res_coff < 0 || in_coff + groups * cin > in_cs || out_coff + groups * cout > out_cs ||
      (res && res_coff + groups * cout > res_cs) || bias_frame_stride < 0 ||
      (bias_frame_stride > 0 && bias_frame_stride < groups * cout))
    return fail(nullptr, ACRMI_EINVAL, "acrmi_conv2d: channel slice outside its tensor's channel stride");
  ConvArgs a{};
  a.in = in; a.w = w_packed; a.bias = bias; a.res = res; a.out = out;
  a.B = B; a.H = H; a.W = W;
  const int pad = ksize / 2;
  a.Ho = (H + 2 * pad - ksize) / stride + 1; a.Wo = (W + 2 * pad - ksize) / stride + 1;
  a.in_cs = in_cs; a.in_coff = in_coff; a.Cin = cin;
  a.out_cs = out_cs; a.out_coff = out_coff; a.Cout = cout;
  a.res_cs = res_cs; a.res_coff = res_coff;
  a.ks = ksize; a.stride = stride; a.relu = relu; a.groups = groups;
  a.cin8 = (cin + 7) / 8;
  a.n_tiles = cout <= 32 ? 1 : ((cout + 63) / 64) * 2;
  a.bias_fstride = bias_frame_stride;
  a.algo = algo;
  a.res_bcast = bias_map ? 1 : 0;
  hipError_t e = launch_conv(a, (hipStream_t)stream);
  if (e != hipSuccess) return fail(nullptr, ACRMI_EHIP, "conv launch: %s", hipGetErrorString(e));
  return ACRMI_OK;
}

size_t acrmi_conv2d_splitk_workspace(int B, int H, int W, int cout, int splits) {
  if (B <= 0 || H <= 0 || W <= 0 || cout <= 0 || splits < 2) return 0;
  const size_t cnt = (conv_splitk_counters(B, H, W, cout) * sizeof(unsigned) + 255) / 256 * 256;
  return cnt + conv_splitk_ws_floats(B, H, W, cout, splits) * sizeof(float);
}

int acrmi_conv2d_splitk(const float* in, int B, int H, int W, int in_cs, int in_coff, int cin_slice, int splits,
                        const float* w_packed, const float* bias, const float* res, int res_cs, int res_coff, float* out,
                        int out_cs, int out_coff, int cout, int relu, void* workspace, size_t workspace_bytes, void* stream) {
  if (!in || !w_packed || !bias || !out || !workspace || B <= 0 || H <= 0 || W <= 0 || cout <= 0)
    return fail(nullptr, ACRMI_EINVAL, "acrmi_conv2d_splitk: bad arguments");
  if (splits < 2 || splits > 8 || cin_slice < 64 || cin_slice % 32 || cout == 33)
    return fail(nullptr, ACRMI_EINVAL, "acrmi_conv2d_splitk: 2..8 slices of >= 64 channels (a multiple of 32) each; Cout != 33");
  if (in_cs % 4 || in_coff % 4 || in_coff < 0 || out_coff < 0 || res_coff < 0 || in_coff + splits * cin_slice > in_cs ||
      out_coff + cout > out_cs || (res && res_coff + cout > res_cs))
    return fail(nullptr, ACRMI_EINVAL, "acrmi_conv2d_splitk: channel slice outside its tensor's channel stride");
  if (workspace_bytes < acrmi_conv2d_splitk_workspace(B, H, W, cout, splits) || ((uintptr_t)workspace & 15))
    return fail(nullptr, ACRMI_EINVAL, "acrmi_conv2d_splitk: workspace too small (acrmi_conv2d_splitk_workspace) or unaligned");
  ConvArgs a{};
  a.in = in; a.w = w_packed; a.bias = bias; a.res = res; a.out = out;
  a.B = B; a.H = H; a.W = W; a.Ho = H; a.Wo = W;
  a.in_cs = in_cs; a.in_coff = in_coff; a.Cin = cin_slice;
  a.out_cs = out_cs; a.out_coff = out_coff; a.Cout = cout;
  a.res_cs = res_cs; a.res_coff = res_coff;
  a.ks = 3; a.stride = 1; a.relu = relu; a.groups = splits;
  a.cin8 = cin_slice / 8;
  a.n_tiles = cout <= 32 ? 1 : ((cout + 63) / 64) * 2;
  a.algo = 2;
  a.splitk = 1;
  a.split_cnt = reinterpret_cast<unsigned*>(workspace);
  a.split_ws = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) +
                                        (conv_splitk_counters(B, H, W, cout) * sizeof(unsigned) + 255) / 256 * 256);
  hipError_t e = launch_conv(a, (hipStream_t)stream);
  if (e != hipSuccess) return fail(nullptr, ACRMI_EHIP, "conv launch: %s", hipGetErrorString(e));
  return ACRMI_OK;
}

int acrmi_conv2d_h16(const void* in, int B, int H, int W, int in_cs, int in_coff, int cin, const void* w_packed,
                     const float* bias, int bias_frame_stride, const void* res, int res_cs, int res_coff, void* out,
                     int out_cs, int out_coff, int cout, int ksize, int stride, int relu, int groups, int dtype,
                     int out_f32, void* stream) {
  if (!in || !w_packed || !bias || !out || B <= 0 || H <= 0 || W <= 0 || cin <= 0 || cout <= 0 || groups <= 0 ||
      (dtype != ACRMI_DT_F16 && dtype != ACRMI_DT_BF16))
    return fail(nullptr, ACRMI_EINVAL, "acrmi_conv2d_h16: bad arguments");
  if ((ksize != 1 && ksize != 3) || (stride != 1 && stride != 2) || (out_f32 && stride != 1))
    return fail(nullptr, ACRMI_EINVAL, "acrmi_conv2d_h16: 3x3 and 1x1 at stride 1 / 2 (fp32 output: stride 1 only); got k%d s%d", ksize, stride);
  const int oq = out_f32 ? 4 : 8;      // elements per 16-byte vector of the output / residual
  if (in_cs % 8 || in_coff % 8 || (groups > 1 && cin % 8) || out_cs % oq || (res && res_cs % oq))
    return fail(nullptr, ACRMI_EINVAL, "acrmi_conv2d_h16: channel strides must be multiples of 16 bytes");
  if (in_coff < 0 || out_coff < 0 || res_coff < 0 || in_coff + groups * cin > in_cs || out_coff + groups * cout > out_cs ||
      (res && res_coff + groups * cout > res_cs) || bias_frame_stride < 0 ||
      (bias_frame_stride > 0 && bias_frame_stride < groups * cout))
    return fail(nullptr, ACRMI_EINVAL, "acrmi_conv2d_h16: channel slice outside its tensor's channel stride");
  ConvArgs a{};
  a.in = reinterpret_cast<const float*>(in); a.w = reinterpret_cast<const float*>(w_packed); a.bias = bias;
  a.res = reinterpret_cast<const float*>(res); a.out = reinterpret_cast<float*>(out);
  a.B = B; a.H = H; a.W = W;
  const int pad = ksize / 2;
  a.Ho = (H + 2 * pad - ksize) / stride + 1; a.Wo = (W + 2 * pad - ksize) / stride + 1;
  a.in_cs = in_cs; a.in_coff = in_coff; a.Cin = cin;
  a.out_cs = out_cs; a.out_coff = out_coff; a.Cout = cout;
  a.res_cs = res_cs; a.res_coff = res_coff;
  a.ks = ksize; a.stride = stride; a.relu = relu; a.groups = groups;
  a.cin8 = (cin + 15) / 16;
  a.n_tiles = cout <= 32 ? 1 : ((cout + 63) / 64) * 2;
  a.bias_fstride = bias_frame_stride;
  a.algo = 0; a.dtype = dtype; a.out_f32 = out_f32 ? 1 : 0;
  hipError_t e = launch_conv(a, (hipStream_t)stream);
  if (e != hipSuccess) return fail(nullptr, ACRMI_EHIP, "conv launch: %s", hipGetErrorString(e));
  return ACRMI_OK;
}

int acrmi_preprocess(const uint8_t* bgr_dev, int n, int H, int W, uint8_t* out_rgb_dev, float* offsets_host,
                     void* stream) {
  if (!bgr_dev || !out_rgb_dev || n <= 0 || H <= 0 || W <= 0)
    return fail(nullptr, ACRMI_EINVAL, "acrmi_preprocess: bad arguments");
  // imgaug compute_paddings_to_reach_aspect_ratio(shape, 1.0): pad the shorter side, extra pixel bottom/right
  const int S = H > W ? H : W;
  int top = 0, right = 0, bottom = 0, left = 0;
  if (W < H) { const int d = H - W; right = (d + 1) / 2; left = d / 2; }
  else if (H < W) { const int d = W - H; top = d / 2; bottom = (d + 1) / 2; }
  if (offsets_host) {
    for (int i = 0; i < n; ++i) {
      float* o = offsets_host + (size_t)i * 10;
      o[0] = (float)S; o[1] = (float)S; o[2] = o[3] = o[4] = o[5] = 0.f;
      o[6] = (float)top; o[7] = (float)right; o[8] = (float)bottom; o[9] = (float)left;
    }
  }
  hipError_t e = launch_preprocess(bgr_dev, n, H, W, S, top, left, 512, out_rgb_dev, (hipStream_t)stream);
  return e == hipSuccess ? ACRMI_OK : fail(nullptr, ACRMI_EHIP, "preprocess: %s", hipGetErrorString(e));
}

int acrmi_u8norm(const uint8_t* img, int n_pixels, float* out, void* stream) {
  if (!img || !out || n_pixels <= 0) return fail(nullptr, ACRMI_EINVAL, "acrmi_u8norm: bad arguments");
  hipError_t e = launch_u8norm(img, n_pixels, out, (hipStream_t)stream);
  return e == hipSuccess ? ACRMI_OK : fail(nullptr, ACRMI_EHIP, "u8norm: %s", hipGetErrorString(e));
}

int acrmi_bilinear2x(const float* in, int B, int H, int W, int in_cs, int C, float* out, int out_cs, void* stream) {
  if (!in || !out || B <= 0 || H < 2 || W < 2 || C % 4 || in_cs % 4 || out_cs % 4)
    return fail(nullptr, ACRMI_EINVAL, "acrmi_bilinear2x: bad arguments");
  hipError_t e = launch_bilinear2x(in, B, H, W, in_cs, 0, C, out, out_cs, 0, (hipStream_t)stream);
  return e == hipSuccess ? ACRMI_OK : fail(nullptr, ACRMI_EHIP, "bilinear2x: %s", hipGetErrorString(e));
}

int acrmi_fuse_sum(int nterms, const float* const* terms, const int* term_cs, const int* term_shift, int B, int H,
                   int W, int C, float* out, int out_cs, int relu, void* stream) {
  if (nterms < 1 || nterms > 4 || !terms || !term_cs || !term_shift || !out || C % 4)
    return fail(nullptr, ACRMI_EINVAL, "acrmi_fuse_sum: bad arguments");
  FuseArgs f{};
  f.nterms = nterms; f.B = B; f.H = H; f.W = W; f.C = C; f.out = out; f.out_cs = out_cs; f.relu = relu;
  for (int t = 0; t < nterms; ++t) { f.term[t] = terms[t]; f.cs[t] = term_cs[t]; f.shift[t] = term_shift[t]; }
  hipError_t e = launch_fuse_sum(f, (hipStream_t)stream);
  return e == hipSuccess ? ACRMI_OK : fail(nullptr, ACRMI_EHIP, "fuse_sum: %s", hipGetErrorString(e));
}

int acrmi_stem_conv(const uint8_t* img, int B, int H, int W, const float* w_packed, const float* bias, float* out,
                    int out_cs, int out_coff, int relu, void* stream) {
  if (!img || !w_packed || !bias || !out || B <= 0 || !stem_shape_ok(H, W, out_cs, out_coff))
    return fail(nullptr, ACRMI_EINVAL, "acrmi_stem_conv: bad arguments (H %% 16, W %% 128, 64 channels inside out_cs)");
  hipError_t e = launch_stem(img, B, H, W, w_packed, bias, out, out_cs, out_coff, relu, (hipStream_t)stream);
  return e == hipSuccess ? ACRMI_OK : fail(nullptr, ACRMI_EHIP, "stem: %s", hipGetErrorString(e));
}

size_t acrmi_attpool_ws_floats(int B, int C) { return B > 0 && C > 0 ? attpool_ws_floats(B, C) : 0; }

int acrmi_attpool(const float* segm, int segm_cs, const float* feat, int feat_cs, int C, int B, float* ws,
                  float* pooled, void* stream) {
  if (!segm || !feat || !ws || !pooled || B <= 0) return fail(nullptr, ACRMI_EINVAL, "acrmi_attpool: bad arguments");
  hipError_t e = launch_attpool(segm, segm_cs, feat, feat_cs, C, B, 128, 128, ws, pooled, (hipStream_t)stream);
  return e == hipSuccess ? ACRMI_OK : fail(nullptr, ACRMI_EHIP, "attpool: %s", hipGetErrorString(e));
}

int acrmi_parebias(const float* pooled, int C, int part0, const float* lc_w, const float* lin_w, const float* lin_b,
                   const float* mix_wp, const float* mix_b, int B, float* out, int out_stride, void* stream) {
  if (!pooled || !lc_w || !lin_w || !lin_b || !mix_wp || !mix_b || !out || B <= 0 || (C != 256 && C != 320) ||
      (part0 != 0 && part0 != 16) || out_stride < 109 || out_stride > 256)
    return fail(nullptr, ACRMI_EINVAL, "acrmi_parebias: bad arguments");
  PareArgs a{};
  a.pooled = pooled; a.lc_w = lc_w; a.lin_w = lin_w; a.lin_b = lin_b; a.mix_wp = mix_wp; a.mix_b = mix_b;
  a.out = out; a.B = B; a.C = C; a.part0 = part0; a.out_stride = out_stride;
  hipError_t e = launch_parebias(a, (hipStream_t)stream);
  return e == hipSuccess ? ACRMI_OK : fail(nullptr, ACRMI_EHIP, "parebias: %s", hipGetErrorString(e));
}

int acrmi_cam_trans(const float* joints_dev, const float* pj2d_dev, int n, float focal_length, float img_size,
                    float* trans_dev, void* stream) {
  if (n < 0 || (n > 0 && (!joints_dev || !pj2d_dev || !trans_dev)) || !(focal_length > 0.f) || !(img_size > 0.f))
    return fail(nullptr, ACRMI_EINVAL, "acrmi_cam_trans: bad arguments");
  hipError_t e = launch_cam_trans(joints_dev, pj2d_dev, n, focal_length, img_size, trans_dev, (hipStream_t)stream);
  return e == hipSuccess ? ACRMI_OK : fail(nullptr, ACRMI_EHIP, "cam_trans: %s", hipGetErrorString(e));
}

// ---- multi-GPU: RCCL all-gather of the result slots -----------------------------------
// libacrmi.so has no link-time dependency on RCCL: the library is resolved at the first call - the copy the process
// already holds (PyTorch ships its own librccl.so) or the ROCm one.
namespace {
struct RcclId { char internal[128]; };   // ncclUniqueId (rccl.h:43)
struct Rccl {
  void* lib = nullptr;
  int (*GetUniqueId)(RcclId*) = nullptr;
  int (*CommInitRank)(void**, int, RcclId, int) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  bool tried = false;
};
Rccl g_rccl;
constexpr int RCCL_FLOAT32 = 7;          // ncclFloat32 (rccl.h:466)

bool rccl_load() {
  if (g_rccl.tried) return g_rccl.lib != nullptr;
  g_rccl.tried = true;
  const char* env = getenv("ACRMI_RCCL_LIB");
  const char* names[] = {env, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void* h = nullptr;
  for (const char* n : names)      // already mapped into the process?
    if (n && (h = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
  for (const char* n : names) {
    if (h) break;
    if (n) h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
  }
  if (!h) return false;
  g_rccl.GetUniqueId = reinterpret_cast<decltype(g_rccl.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
  g_rccl.CommInitRank = reinterpret_cast<decltype(g_rccl.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
  g_rccl.CommDestroy = reinterpret_cast<decltype(g_rccl.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
  g_rccl.AllGather = reinterpret_cast<decltype(g_rccl.AllGather)>(dlsym(h, "ncclAllGather"));
  g_rccl.GetErrorString = reinterpret_cast<decltype(g_rccl.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
  if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.CommDestroy || !g_rccl.AllGather || !g_rccl.GetErrorString)
    return false;
  g_rccl.lib = h;
  return true;
}
const char* rccl_err(int r) { return g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?"; }
}  // namespace

static void comm_destroy(acrmi_ctx* c) {
  if (c->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
  c->comm = nullptr;
  c->comm_ranks = 0;
}

int acrmi_comm_unique_id(void* id128) {
  if (!id128) return fail(nullptr, ACRMI_EINVAL, "acrmi_comm_unique_id: id is NULL");
  if (!rccl_load()) return fail(nullptr, ACRMI_ESTATE, "RCCL (librccl.so) could not be loaded: %s", dlerror());
  RcclId id;
  const int r = g_rccl.GetUniqueId(&id);
  if (r) return fail(nullptr, ACRMI_EHIP, "ncclGetUniqueId: %s", rccl_err(r));
  memcpy(id128, &id, sizeof id);
  return ACRMI_OK;
}

int acrmi_comm_init(acrmi_ctx* c, int n_ranks, int rank, const void* id128) {
  if (!c || !id128 || n_ranks <= 0 || rank < 0 || rank >= n_ranks) return fail(c, ACRMI_EINVAL, "acrmi_comm_init: bad arguments");
  if (!rccl_load()) return fail(c, ACRMI_ESTATE, "RCCL (librccl.so) could not be loaded: %s", dlerror());
  ON_DEVICE(c);
  comm_destroy(c);
  RcclId id;
  memcpy(&id, id128, sizeof id);
  const int r = g_rccl.CommInitRank(&c->comm, n_ranks, id, rank);
  if (r) {
    c->comm = nullptr;
    return fail(c, ACRMI_EHIP, "ncclCommInitRank(%d of %d): %s", rank, n_ranks, rccl_err(r));
  }
  c->comm_ranks = n_ranks;
  return ACRMI_OK;
}

int acrmi_comm_destroy(acrmi_ctx* c) {
  if (!c) return fail(c, ACRMI_EINVAL, "acrmi_comm_destroy: ctx is NULL");
  ON_DEVICE(c);
  comm_destroy(c);
  return ACRMI_OK;
}

int acrmi_allgather(acrmi_ctx* c, void* nccl_comm, const float* send_dev, float* recv_dev, size_t n_floats, void* stream) {
  if (!c || !send_dev || !recv_dev || n_floats == 0) return fail(c, ACRMI_EINVAL, "acrmi_allgather: bad arguments");
  void* comm = nccl_comm ? nccl_comm : c->comm;
  if (!comm) return fail(c, ACRMI_ESTATE, "acrmi_allgather: no communicator (acrmi_comm_init, or pass an ncclComm_t)");
  if (!rccl_load()) return fail(c, ACRMI_ESTATE, "RCCL (librccl.so) could not be loaded");
  ON_DEVICE(c);
  const int r = g_rccl.AllGather(send_dev, recv_dev, n_floats, RCCL_FLOAT32, comm, (hipStream_t)stream);
  if (r) return fail(c, ACRMI_EHIP, "ncclAllGather: %s", rccl_err(r));
  return ACRMI_OK;
}

// Plain HIP streams for hosts that have no stream API of their own at hand (Python: torch.cuda.Stream() instantiates
// torch's whole pool of 32 streams per priority, and with that many streams alive the few in use share hardware
// queues); engine.EnginePool runs its contexts on these.
int acrmi_stream_create(int device, void** stream) {
  if (!stream) return fail(nullptr, ACRMI_EINVAL, "acrmi_stream_create: null argument");
  int prev = 0;
  if (hipGetDevice(&prev) != hipSuccess || hipSetDevice(device) != hipSuccess)
    return fail(nullptr, ACRMI_EHIP, "acrmi_stream_create: cannot select device %d", device);
  hipStream_t st = nullptr;
  const hipError_t e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  (void)hipSetDevice(prev);
  if (e != hipSuccess) return fail(nullptr, ACRMI_EHIP, "hipStreamCreate: %s", hipGetErrorString(e));
  *stream = st;
  return ACRMI_OK;
}

int acrmi_stream_destroy(void* stream) {
  if (!stream) return ACRMI_OK;
  const hipError_t e = hipStreamDestroy((hipStream_t)stream);
  return e == hipSuccess ? ACRMI_OK : fail(nullptr, ACRMI_EHIP, "hipStreamDestroy: %s", hipGetErrorString(e));
}

int acrmi_tune(int key, int value) {
  if (key == 0) { conv_force_cfg(value); return ACRMI_OK; }
  if (key == 3) { conv_set_phase_delay(value); return ACRMI_OK; }
  if (key == 4) { conv_set_xcd_swizzle(value); return ACRMI_OK; }   // XCD-banded item order on/off (default on)
  static long long* dbg = nullptr;
  if (key == 1) {   // enable (value != 0) / disable the conv kernel's cycle stamps (workgroup 0, wave 0)
    if (value && !dbg) { if (hipMalloc(&dbg, 128 * sizeof(long long)) != hipSuccess) return ACRMI_EHIP; }
    if (dbg) (void)hipMemset(dbg, 0, 128 * sizeof(long long));
    conv_set_debug(value ? dbg : nullptr);
    return ACRMI_OK;
  }
  if (key == 2) {   // print the stamps of the last conv launch as deltas (device is synchronised first)
    if (!dbg) return ACRMI_OK;
    long long h[128];
    if (hipDeviceSynchronize() != hipSuccess) return ACRMI_EHIP;
    if (hipMemcpy(h, dbg, sizeof h, hipMemcpyDeviceToHost) != hipSuccess) return ACRMI_EHIP;
    const int n = (int)h[63];
    printf("conv stamps (%d):", n);
    for (int i = 1; i < n && i < 60; ++i) printf(" %lld", h[i] - h[i - 1]);
    printf("\n");
    const int nl = (int)h[127];   // loader wave 0 (conv_wino2_kernel only)
    if (nl > 0) {
      printf("loader stamps (%d):", nl);
      for (int i = 1; i < nl && i < 60; ++i) printf(" %lld", h[64 + i] - h[64 + i - 1]);
      printf("\n");
    }
    return ACRMI_OK;
  }
  return fail(nullptr, ACRMI_EINVAL, "acrmi_tune: unknown key %d", key);
}

}  // extern "C"
