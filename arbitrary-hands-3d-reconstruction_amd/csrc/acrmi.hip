// libacrmi.so: context life cycle, options and the fused entry points of the C ABI declared in include/acrmi.h
// (program replay: acrmi_program.hip; stand-alone operators: acrmi_ops.hip; RCCL: acrmi_comm.hip).
#include "acrmi_ctx.h"

std::string g_err;

int fail(acrmi_ctx* c, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (c) c->err = buf; else g_err = buf;
  return code;
}

// the blob is freed by the last context that holds it (contexts are single-threaded per the ABI; sharing contexts of one
// pool are driven by one host thread)
void release_weights(acrmi_ctx* c) {
  // (the use count is atomic: contexts of one pool may be destroyed / re-programmed from different host threads)
  if (c->weights_ref && c->weights_ref->fetch_sub(1) == 1) {
    (void)hipFree(c->weights);
    delete c->weights_ref;
  }
  c->weights = nullptr;
  c->weights_ref = nullptr;
  c->n_weights = 0;
}

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
  release_weights(c);
  if (c->smooth_state) (void)hipFree(c->smooth_state);
  for (auto& side : c->mano_allocs)
    for (float* p : side)
      if (p) (void)hipFree(p);
  delete c;
}

int acrmi_load_weights(acrmi_ctx* c, const float* blob, size_t n) {
  if (!c || !blob || n == 0) return fail(c, ACRMI_EINVAL, "acrmi_load_weights: bad arguments");
  ON_DEVICE(c);
  release_weights(c);
  // the context takes the blob only once it is complete on the device: a failed copy leaves it without weights, not with a
  // blob that has no use count
  float* w = nullptr;
  HIPCHK(c, hipMalloc(&w, n * sizeof(float)));
  hipError_t e = hipMemcpy(w, blob, n * sizeof(float), hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    (void)hipFree(w);
    return fail(c, ACRMI_EHIP, "acrmi_load_weights: hipMemcpy of %zu floats: %s", n, hipGetErrorString(e));
  }
  c->weights = w;
  c->n_weights = n;
  c->weights_ref = new std::atomic<int>(1);
  return ACRMI_OK;
}

int acrmi_share_weights(acrmi_ctx* c, acrmi_ctx* donor) {
  if (!c || !donor || c == donor) return fail(c, ACRMI_EINVAL, "acrmi_share_weights: bad arguments");
  if (!donor->weights || !donor->weights_ref) return fail(c, ACRMI_ESTATE, "acrmi_share_weights: the donor holds no weights");
  if (c->device != donor->device) return fail(c, ACRMI_EINVAL, "acrmi_share_weights: contexts on different devices (%d, %d)", c->device, donor->device);
  ON_DEVICE(c);
  release_weights(c);
  c->weights = donor->weights;
  c->n_weights = donor->n_weights;
  c->weights_ref = donor->weights_ref;
  c->weights_ref->fetch_add(1);
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
  // the residuals of the two small tables (value = hi + lo): what brings the f16 MANO stage from 6.4e-5 m to the 1e-5 class
  auto residual = [](const float* h, size_t n) {
    std::vector<float> lo(n);
    for (size_t i = 0; i < n; ++i) lo[i] = h[i] - (float)(_Float16)h[i];
    return lo;
  };
  {
    const std::vector<float> lo_s = residual(sd_t.data(), sd_t.size()), lo_w = residual(weights, 778 * 16);
    if ((r = up16(lo_s.data(), lo_s.size(), &t.shapedirs_l))) return r;
    if ((r = up16(lo_w.data(), lo_w.size(), &t.weights_l))) return r;
  }
  c->have_mano[side] = true;
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
  if (option == ACRMI_OPT_BATCH_PRIOR) {
    c->batch_prior = value != 0;
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

int acrmi_decode(acrmi_ctx* c, int B, float* slots, void* stream) { return acrmi_decode_gated(c, B, nullptr, slots, stream); }

int acrmi_decode_gated(acrmi_ctx* c, int B, const int32_t* prior_gate, float* slots, void* stream) {
  if (!c || !slots) return fail(c, ACRMI_EINVAL, "acrmi_decode: bad arguments");
  if (!c->have_program) return fail(c, ACRMI_ESTATE, "acrmi_decode: no program");
  if (B <= 0 || B > c->max_batch) return fail(c, ACRMI_EINVAL, "batch %d outside 1..%d", B, c->max_batch);
  ON_DEVICE(c);
  const acrmi_head_layout& h = c->heads;
  int r = decode_maps_impl(c->buf_ptr[h.center_buf[0]], c->buf_ptr[h.center_buf[1]], c->bufs[h.center_buf[0]].cs,
                           c->buf_ptr[h.params_buf[0]], c->buf_ptr[h.params_buf[1]], c->bufs[h.params_buf[0]].cs,
                           c->buf_ptr[h.prior_buf[0]], c->buf_ptr[h.prior_buf[1]], c->bufs[h.prior_buf[0]].cs, B,
                           c->conf_thresh, prior_gate, c->range_flag, slots, stream);
  if (r) c->err = g_err;
  return r;
}

int acrmi_prior_gate(acrmi_ctx* c, const float* slots, int B, int32_t* gate, void* stream) {
  if (!slots || !gate || B <= 0) return fail(c, ACRMI_EINVAL, "acrmi_prior_gate: bad arguments");
  if (c) {
    ON_DEVICE(c);
    HIPCHK(c, launch_prior_gate(slots, B, gate, (hipStream_t)stream));
    return ACRMI_OK;
  }
  HIPCHK(c, launch_prior_gate(slots, B, gate, (hipStream_t)stream));      // stand-alone (like acrmi_decode_maps): current device
  return ACRMI_OK;
}

int acrmi_check_range(acrmi_ctx* c, void* stream) {
  if (!c) return fail(c, ACRMI_EINVAL, "acrmi_check_range: ctx is NULL");
  if (!c->range_flag) return ACRMI_OK;      // no split-f16 convolution in the program: nothing can overflow
  ON_DEVICE(c);
  unsigned v = 0;
  HIPCHK(c, hipStreamSynchronize((hipStream_t)stream));
  HIPCHK(c, hipMemcpy(&v, c->range_flag, sizeof(v), hipMemcpyDeviceToHost));
  if (!v) return ACRMI_OK;
  // cleared ON the caller's stream (the streams of acrmi_stream_create are non-blocking: a null-stream memset is not ordered
  // against work the caller queues next and could wipe a later launch's flag), then waited for
  HIPCHK(c, hipMemsetAsync(c->range_flag, 0, sizeof(v), (hipStream_t)stream));
  HIPCHK(c, hipStreamSynchronize((hipStream_t)stream));
  return fail(c, ACRMI_ERANGE, "an activation of the 'fp16x3' program left the f16 range (|x| > 65504): its split halves are "
                               "inf / -inf and the results since the last check are invalid (slots and meshes were written as "
                               "NaN); use precision 'bf16x3' or 'fp32' for this checkpoint");
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

int acrmi_mano_rotmat(acrmi_ctx* c, const float* rotmats, const float* betas, int beta_stride, const int32_t* side, int H,
                      int center_idx, float* verts, float* joints, float* center, void* stream) {
  if (!c) return fail(c, ACRMI_EINVAL, "acrmi_mano_rotmat: ctx is NULL");
  if (H == 0) return ACRMI_OK;
  if (H < 0 || !rotmats || !betas || !verts || !joints || center_idx >= 21)
    return fail(c, ACRMI_EINVAL, "acrmi_mano_rotmat: bad arguments");
  if (!c->have_mano[0] && !c->have_mano[1]) return fail(c, ACRMI_ESTATE, "acrmi_mano_rotmat: MANO tables not loaded");
  ON_DEVICE(c);
  ManoArgs m{};
  m.t[0] = c->have_mano[0] ? c->mano[0] : c->mano[1];
  m.t[1] = c->have_mano[1] ? c->mano[1] : c->mano[0];
  m.poses = rotmats; m.pose_stride = 144; m.pose_rotmat = 1; m.betas = betas; m.beta_stride = beta_stride;
  m.side = side; m.H = H; m.center_idx = center_idx;
  m.verts = verts; m.joints = joints; m.center = center;
  m.off_div = 1;
  m.lbs_f16 = c->mano_f16;
  HIPCHK(c, launch_mano(m, (hipStream_t)stream));
  return ACRMI_OK;
}

// The last op that writes the backbone map (heads.backbone_buf) and the channels it leaves there: an op whose out_buf is the
// map - or, in large-batch fp32 programs, the ACRMI_CONV_DUAL convolution whose SECOND output (aux_buf: the full-resolution HR
// fuse sum of the last stage, written at channel 0) is the map.  first = the op behind it (-1: none), c0 = channels.
static void backbone_writer(const acrmi_ctx* c, int* first, int* c0) {
  const int bb = c->heads.backbone_buf;
  *first = -1;
  *c0 = 0;
  for (int i = 0; i < (int)c->ops.size(); ++i) {
    const acrmi_op& op = c->ops[i];
    if (op.kind == ACRMI_OP_COORDFILL) continue;
    if (op.out_buf == bb) {
      *first = i + 1;
      *c0 = op.out_coff + op.cout * (op.kind == ACRMI_OP_CONV ? op.groups : 1);
    } else if (op.kind == ACRMI_OP_CONV && (op.flags & ACRMI_CONV_DUAL) && op.aux_buf == bb) {
      *first = i + 1;
      *c0 = op.cout;
    }
  }
}

int acrmi_heads(acrmi_ctx* c, const float* feat_nchw, int B, void* stream) {
  if (!c || !feat_nchw) return fail(c, ACRMI_EINVAL, "acrmi_heads: bad arguments");
  if (!c->have_program) return fail(c, ACRMI_ESTATE, "acrmi_heads: no program");
  if (B <= 0 || B > c->max_batch) return fail(c, ACRMI_EINVAL, "batch %d outside 1..%d", B, c->max_batch);
  const int bb = c->heads.backbone_buf;
  const acrmi_buffer_desc& d = c->bufs[bb];
  if (d.dtype != ACRMI_DT_F32)
    return fail(c, ACRMI_EINVAL, "acrmi_heads: the program stores its backbone output in 16 bits; features are taken by "
                                 "fp32-storage programs (fp32, fp16x3, bf16x3)");
  // the head ops = everything behind the last op that writes the backbone map (acr/model.py:47: head_forward starts there)
  int first = -1, c0 = 0;
  backbone_writer(c, &first, &c0);
  if (first < 0 || c0 <= 0 || c0 > d.cs) return fail(c, ACRMI_ESTATE, "acrmi_heads: the program has no backbone output op");
  ON_DEVICE(c);
  HIPCHK(c, launch_nchw_to_nhwc(feat_nchw, B, c0, d.h, d.w, c->buf_ptr[bb], d.cs, 0, (hipStream_t)stream));
  return run_program(c, nullptr, B, stream, /*point=*/false, first);
}

int acrmi_backbone_channels(acrmi_ctx* c) {
  if (!c || !c->have_program) return fail(c, ACRMI_ESTATE, "acrmi_backbone_channels: no program");
  int first = -1, c0 = 0;
  backbone_writer(c, &first, &c0);
  return c0;
}

// decode + MANO of acrmi_forward on one stream
static int forward_tail(acrmi_ctx* c, int B, const float* offsets, float* slots, float* verts, float* joints,
                        float* verts_camed, float* pj2d, float* pj2d_org, hipStream_t stream) {
  int r = acrmi_decode(c, B, slots, stream);
  if (r) return r;
  if (c->batch_prior && B > 1) {
    // the reference's batch-wide prior rules (acr/result_parser.py:42-47, 102-145), all on the device: the first decode's flags
    // and centers -> one decision per frame -> a second decode that applies it (the decode is ~40 us per 64 frames)
    HIPCHK(c, launch_prior_gate(slots, B, c->gate_buf, stream));
    if ((r = acrmi_decode_gated(c, B, c->gate_buf, slots, stream))) return r;
  }
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
