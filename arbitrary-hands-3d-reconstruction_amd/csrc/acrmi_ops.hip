// libacrmi.so: the stand-alone operators of the C ABI (unit tests / callers with their own tensors; no context).
#include "acrmi_ctx.h"

// (poison: acrmi_decode_gated's range flag of an 'fp16x3' program; null for the stand-alone operator)
int decode_maps_impl(const float* l_center, const float* r_center, int center_cs, const float* l_params,
                     const float* r_params, int params_cs, const float* l_prior, const float* r_prior, int prior_cs,
                     int B, float conf_thresh, const int32_t* prior_gate, const unsigned* poison, float* slots, void* stream) {
  if (!l_center || !r_center || !l_params || !r_params || !l_prior || !r_prior || !slots || B <= 0 || params_cs < 109 ||
      prior_cs < 106 || center_cs < 1 || !(conf_thresh == conf_thresh))
    return fail(nullptr, ACRMI_EINVAL, "acrmi_decode_maps: bad arguments");
  DecodeArgs d{};
  d.center[0] = l_center; d.center[1] = r_center; d.center_cs = center_cs;
  d.params[0] = l_params; d.params[1] = r_params; d.params_cs = params_cs;
  d.prior[0] = l_prior; d.prior[1] = r_prior; d.prior_cs = prior_cs;
  d.B = B; d.slots = slots; d.thresh = conf_thresh;
  d.prior_gate = prior_gate;
  d.poison = poison;
  hipError_t e = launch_decode(d, (hipStream_t)stream);
  if (e != hipSuccess) return fail(nullptr, ACRMI_EHIP, "decode launch: %s", hipGetErrorString(e));
  return ACRMI_OK;
}

extern "C" {

int acrmi_decode_maps_gated(const float* l_center, const float* r_center, int center_cs, const float* l_params,
                            const float* r_params, int params_cs, const float* l_prior, const float* r_prior, int prior_cs,
                            int B, float conf_thresh, const int32_t* prior_gate, float* slots, void* stream) {
  return decode_maps_impl(l_center, r_center, center_cs, l_params, r_params, params_cs, l_prior, r_prior, prior_cs, B, conf_thresh,
                          prior_gate, nullptr, slots, stream);
}

int acrmi_decode_maps(const float* l_center, const float* r_center, int center_cs, const float* l_params,
                      const float* r_params, int params_cs, const float* l_prior, const float* r_prior, int prior_cs,
                      int B, float conf_thresh, float* slots, void* stream) {
  return acrmi_decode_maps_gated(l_center, r_center, center_cs, l_params, r_params, params_cs, l_prior, r_prior, prior_cs, B,
                                 conf_thresh, nullptr, slots, stream);
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
  const bool x3s2 = (algo == 6 || algo == 7) && ksize == 3 && stride == 2;      // conv_x3s2.inc
  if (algo != 0 && !x3s2 && !((algo >= 1 && algo <= 7) && (ksize == 3 || (algo >= 6 && ksize == 1)) && stride == (algo == 5 ? 2 : 1)))
    return fail(nullptr, ACRMI_EINVAL, "acrmi_conv2d: algo %d needs a 3x3 stride-%d convolution", algo, algo == 5 ? 2 : 1);
  if (x3s2 && (cin % 32 || cout % 32 || H % 16 || W % 64))
    return fail(nullptr, ACRMI_EINVAL, "acrmi_conv2d: algo 6 / 7 at stride 2 needs Cin %% 32 = 0, Cout %% 32 = 0, H %% 16 == 0, W %% 64 == 0");
  if ((algo == 6 || algo == 7) && !x3s2 && (cin % 32 || cout % 32 || (ksize == 3 ? ((H % 8 || W % 32) && (H % 16 || W % 16)) : ((H * W) % 256 != 0))))
    return fail(nullptr, ACRMI_EINVAL, "acrmi_conv2d: algo 6 / 7 needs Cin %% 32 = 0, Cout %% 32 = 0, H %% 8 == 0, W %% 32 == 0");
  if (algo == 5 && (cin % 16 || cout % 32 || H % 16 || W % 32))
    return fail(nullptr, ACRMI_EINVAL, "acrmi_conv2d: algo 5 needs Cin %% 16 = 0, Cout %% 32 = 0, H %% 16 == 0, W %% 32 == 0");
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

int acrmi_preprocess_frames(const acrmi_frame* frames_host, int n, uint8_t* out_rgb_dev, float* offsets_host, void* stream) {
  if (!frames_host || !out_rgb_dev || n <= 0) return fail(nullptr, ACRMI_EINVAL, "acrmi_preprocess_frames: bad arguments");
  for (int i = 0; i < n; ++i)
    if (!frames_host[i].bgr_dev || frames_host[i].H <= 0 || frames_host[i].W <= 0)
      return fail(nullptr, ACRMI_EINVAL, "acrmi_preprocess_frames: frame %d: null pointer or empty size (%d x %d)", i,
                  frames_host[i].H, frames_host[i].W);
  for (int i0 = 0; i0 < n; i0 += PRE_FRAMES_PER_LAUNCH) {
    const int m = n - i0 < PRE_FRAMES_PER_LAUNCH ? n - i0 : PRE_FRAMES_PER_LAUNCH;
    PreBatch pb{};
    for (int i = 0; i < m; ++i) {
      const acrmi_frame& fr = frames_host[i0 + i];
      pb.f[i].bgr = fr.bgr_dev; pb.f[i].H = fr.H; pb.f[i].W = fr.W;
      if (offsets_host) {     // the reference's `offsets` row of this image (acr/utils.py:1276-1313)
        const int H = fr.H, W = fr.W, S = H > W ? H : W;
        int top = 0, right = 0, bottom = 0, left = 0;
        if (W < H) { const int d = H - W; right = (d + 1) / 2; left = d / 2; }
        else if (H < W) { const int d = W - H; top = d / 2; bottom = (d + 1) / 2; }
        float* o = offsets_host + (size_t)(i0 + i) * 10;
        o[0] = (float)S; o[1] = (float)S; o[2] = o[3] = o[4] = o[5] = 0.f;
        o[6] = (float)top; o[7] = (float)right; o[8] = (float)bottom; o[9] = (float)left;
      }
    }
    hipError_t e = launch_preprocess_frames(pb, m, 512, out_rgb_dev + (size_t)i0 * 512 * 512 * 3, (hipStream_t)stream);
    if (e != hipSuccess) return fail(nullptr, ACRMI_EHIP, "preprocess_frames: %s", hipGetErrorString(e));
  }
  return ACRMI_OK;
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

}  // extern "C"
