// Internal launch interface between the C ABI (acrmi.hip) and the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <mutex>

#include <stdlib.h>

namespace acrmi {

// Environment switches of timing experiments / A-B runs (lane-sync ablation, event flags, loader and kernel-frame A/B) exist
// only in a library built with -DACRMI_EXPERIMENTS (`python -m <package>.build --experiments`); the production library does
// not read them.  Diagnostics (ACRMI_DEBUG*) and configuration (ACRMI_RCCL_LIB) are plain getenv.
inline const char* experiment_env(const char* name) {
#ifdef ACRMI_EXPERIMENTS
  return getenv(name);
#else
  (void)name;
  return nullptr;
#endif
}

struct ConvArgs {
  const float* in;
  const float* w;      // packed [group][tap][cin8][ntile][lane64][4]
  const float* bias;   // [group][coutP] or per frame
  const float* res;    // may be null
  float* out;
  int B, H, W, Ho, Wo;
  int in_cs, in_coff, Cin;
  int out_cs, out_coff, Cout;
  int res_cs, res_coff;
  int ks, stride, relu, groups;
  int n_tiles;          // 32-cout tiles per group in the packed weights (CoutP/32)
  int cin8;             // ceil(Cin/8)
  int bias_fstride;     // floats between frames' bias rows (0 = shared)
  int algo;             // 0 direct; 1 Winograd F(2,3) along x (3x3 stride 1, weights packed with 3x4 taps);
                        // 2 Winograd F(2x2,3x3) (3x3 stride 1, weights packed with 4x4 taps)
  long long* dbg;       // optional device buffer for cycle stamps (tuning only)
  int phase_delay;      // tuning switches of the loader waves (acrmi_tune key 3): 8 = idle loader (timing ablation, wrong
                        // results), 9 = loader at priority 0; 0 = off
  int xcd_swizzle;      // 1: work items are dealt to the XCDs in contiguous bands (see virtual_block, conv_mfma.hip)
  const float* zeros;   // >= 16 bytes of device zeros (source of halo / pad-channel lanes of conv_wino3's LDS-DMA loader)
  // Storage type of the activations (ACRMI_DT_*).  F16 / BF16: in, w (and out / res unless out_f32) point at 16-bit
  // data, channel strides / offsets / counts are in elements of their tensor, cin8 = ceil(Cin / 16), the weights are
  // packer.pack_conv_h16 fragments and algo must be 0 (conv_h16.hip).
  int dtype;
  int out_f32;          // 16-bit input, fp32 output and fp32 residual (the head exits)
  int res_bcast;        // 1: res is ONE [Ho][Wo][res_cs] map added to every frame (not conv_wino3 / the 16-bit kernels)
  // Split-K (algo 2, small batches; conv_wino2.inc SPLIT): groups = K-slices of one convolution (slice s reads input
  // channels [s*Cin, (s+1)*Cin), all slices produce the same Cout channels).  split_ws: conv_splitk_ws_floats() floats,
  // split_cnt: conv_splitk_counters() zeroed unsigned, both private to the launches of one stream.
  int in_sub;           // 2: the loader reads every other input row / column (a 1x1 stride-2 convolution runs as the 1x1
                        // stride-1 kernel on the subsampled map; H, W stay the input's, Ho, Wo the output's); 0 / 1: dense
  int splitk;
  float* split_ws;
  unsigned* split_cnt;
  // Extra residual terms (the HR-module fuse, acr/model.py:672-686, folded into the epilogue of the downsampling chain's
  // last convolution): out = [relu]((conv + bias + res) + up(xt[0]) + up(xt[1]) + up(xt[2])), summed in this order; term t
  // is a [B][Ho >> shift][Wo >> shift][xt_cs] map read at pixel (y >> shift, x >> shift) = nearest-neighbour upsampling,
  // channels [xt_coff, xt_coff + groups * Cout).  3x3 stride-2 convolutions only (conv_pp2_kernel<1, true> / <2, true>, the 32-cout
  // conv_ws2_kernel), Cout % 32 == 0, 16-byte aligned channel slices.
  int nxt;
  const float* xt[3];
  int xt_cs[3], xt_coff[3], xt_shift[3];
  // ACRMI_CONV_DUAL (conv_wino3_kernel with store waves only): `out` stays the convolution's own output (bias, residual, ReLU);
  // the xt terms are NOT added to it but to a SECOND output, out2 = relu(((out + up(xt[0])) + up(xt[1])) + up(xt[2])) - the
  // full-resolution HR fuse sum, whose first term (branch 0's last conv2) is also read by the downsampling chains
  float* out2;
  int out2_cs, out2_coff;
  unsigned* range_flag;   // conv_x3 / conv_x3p with f16 halves: set to 1 when an activation beyond the f16 range was split
                          // (null: not tracked - the stand-alone operator calls)
};
// workspace of a split-K launch: (8x16-pixel tiles) x (32-cout blocks) groups of `splits` partial tiles of 4096 floats
inline size_t conv_splitk_groups(int B, int Ho, int Wo, int cout) {
  return (size_t)B * ((Ho + 7) / 8) * ((Wo + 15) / 16) * (cout <= 32 ? 1 : ((cout + 63) / 64) * 2);
}
inline size_t conv_splitk_ws_floats(int B, int Ho, int Wo, int cout, int splits) { return conv_splitk_groups(B, Ho, Wo, cout) * splits * 4096; }
inline size_t conv_splitk_counters(int B, int Ho, int Wo, int cout) { return conv_splitk_groups(B, Ho, Wo, cout) * 4; }

// one-time per-DEVICE kernel setup (dynamic LDS attribute): true the first time `flags` (one static array per kernel
// instantiation) is asked about the current device
constexpr int MAX_DEVICES = 64;
bool first_use_on_device(unsigned char* flags);
// held by a launcher from the first_use_on_device() test until the attribute is set (and by launch_conv throughout)
std::recursive_mutex& launch_mutex();

// returns hipSuccess or the launch error; cout tiles etc. derived inside
hipError_t launch_conv(ConvArgs a, hipStream_t s);
hipError_t launch_conv_h16(ConvArgs a, hipStream_t s);   // called by launch_conv when a.dtype != ACRMI_DT_F32
const char* conv_kernel_name(const ConvArgs& a);
void conv_force_cfg(int cfg);
void conv_set_debug(long long* dbg);
void conv_set_phase_delay(int cycles);
void conv_set_xcd_swizzle(int on);

hipError_t launch_u8norm(const uint8_t* img, long n_pixels, float* out, hipStream_t s);
// stem.hip: uint8 RGB [B,H,W,3] -> relu(conv3x3 s2 (x/255*2-1) + b) [B,H/2,W/2,64 of out_cs]; wpk = packer.pack_stem
bool stem_shape_ok(int H, int W, int out_cs, int out_coff);
hipError_t launch_stem(const uint8_t* img, int B, int H, int W, const float* wpk, const float* bias, float* out,
                       int out_cs, int out_coff, int relu, hipStream_t s);
// the same with the output rounded to f16 / bf16 (out_cs / out_coff in elements, multiples of 8)
// layer1's conv3 (64 -> 256, + residual, ReLU) chained with the next block's conv1 (256 -> 64, ReLU) in one kernel
// (csrc/pair1x1.hip); wpk = packer.pack_pair1x1 (33088 floats); npix = B * H * W
hipError_t launch_pair1x1(const float* t2, int t2_cs, int t2_coff, const float* x, int x_cs, int x_coff, float* y, int y_cs,
                          int y_coff, float* t, int t_cs, int t_coff, const float* wpk, long npix, hipStream_t s);
constexpr long PAIR1X1_FLOATS = 256 * 64 + 64 * 256 + 256 + 64;
// ResNet stem (7x7 stride 2 pad 3, csrc/stem7.hip): wpk = packer.pack_stem7 fragments [74][2][64]
bool stem7_shape_ok(int H, int W, int out_cs, int out_coff);
hipError_t launch_stem7(const uint8_t* img, int B, int H, int W, const float* wpk, const float* bias, float* out,
                        int out_cs, int out_coff, int relu, hipStream_t s);
hipError_t launch_stem7_h16(const uint8_t* img, int B, int H, int W, const float* wpk, const float* bias, void* out,
                            int out_cs, int out_coff, int relu, int dtype, hipStream_t s);
hipError_t launch_stem_h16(const uint8_t* img, int B, int H, int W, const float* wpk, const float* bias, void* out,
                           int out_cs, int out_coff, int relu, int dtype, hipStream_t s);
hipError_t launch_bilinear2x(const float* in, int B, int H, int W, int in_cs, int in_coff, int C, float* out,
                             int out_cs, int out_coff, hipStream_t s);
struct FuseArgs {
  const float* term[4];
  int cs[4], shift[4];
  int nterms, B, H, W, C, out_cs, relu;
  float* out;
};
hipError_t launch_fuse_sum(const FuseArgs& a, hipStream_t s);
hipError_t launch_preprocess(const uint8_t* bgr, int n, int H, int W, int S, int pad_top, int pad_left, int out_size,
                             uint8_t* out, hipStream_t s);
// frames of DIFFERENT sizes in one launch (acrmi_preprocess_frames): geometry by value, PRE_FRAMES_PER_LAUNCH per launch
constexpr int PRE_FRAMES_PER_LAUNCH = 128;
struct PreFrame {
  const uint8_t* bgr;
  int H, W;
};
struct PreBatch {
  PreFrame f[PRE_FRAMES_PER_LAUNCH];
};
hipError_t launch_preprocess_frames(const PreBatch& pb, int n, int out_size, uint8_t* out, hipStream_t s);
hipError_t launch_pow11(float* buf, long n_pixels, int cs, int ch, hipStream_t s);
// fp32 NCHW [B,C,H,W] -> channels [coff, coff + C) of an NHWC buffer with channel stride cs (acrmi_heads: backbone features a
// caller hands to head_forward, acr/model.py:47-53)
hipError_t launch_nchw_to_nhwc(const float* in, int B, int C, int H, int W, float* out, int cs, int coff, hipStream_t s);
hipError_t launch_coordfill(float* buf, int B, int H, int W, int cs, int coff, hipStream_t s);
// 16-bit storage variants (dtype = ACRMI_DT_F16 / ACRMI_DT_BF16; strides and offsets in elements, C % 8 == 0)
hipError_t launch_maxpool3s2(const float* in, int B, int H, int W, int in_cs, int in_coff, int C, float* out, int out_cs,
                             int out_coff, hipStream_t s);
hipError_t launch_maxpool3s2_h16(const void* in, int B, int H, int W, int in_cs, int in_coff, int C, void* out, int out_cs,
                                 int out_coff, int dtype, hipStream_t s);
hipError_t launch_bilinear2x_h16(const void* in, int B, int H, int W, int in_cs, int in_coff, int C, void* out, int out_cs,
                                 int out_coff, int dtype, hipStream_t s);
hipError_t launch_fuse_sum_h16(const FuseArgs& a, int dtype, hipStream_t s);   // a.term / a.out carry 16-bit pointers
hipError_t launch_pow11_h16(void* buf, long n_pixels, int cs, int ch, int dtype, hipStream_t s);
hipError_t launch_coordfill_h16(void* buf, int B, int H, int W, int cs, int coff, int dtype, hipStream_t s);

// attention pooling: segm [B,2H,2W,segm_cs] logits (channels 1..32 at even pixels), feat [B,H,W,feat_cs] (C ch)
// stats_ws: [B,32,2] (max, 1/sumexp); pooled [B,32,C]
hipError_t launch_attpool(const float* segm, int segm_cs, const float* feat, int feat_cs, int C, int B, int H, int W,
                          float* ws, float* pooled, hipStream_t s, int feat_dtype = 0);   // feat may be f16 / bf16 (ACRMI_DT_*)
size_t attpool_ws_floats(int B, int C);

// pare bias: pooled [B,32,320] -> per-frame bias row [B, biasP] for the 109->109 mix conv
struct PareArgs {
  const float* pooled;   // [B,32,C] (C = 320: 256 contact + 64 shape; C = 256: contact only, shape conv folded into lin_w)
  const float* lc_w;     // LocallyConnected2d weight [6][256][16]
  const float* lin_w;    // [10][(C == 320 ? 64 : 256) * 16]
  const float* lin_b;    // [10]
  const float* mix_wp;   // [109][106] columns of the mix conv that multiply pare
  const float* mix_b;    // [109]
  float* out;            // [B, out_stride]
  int B, C, part0, out_stride;
};
hipError_t launch_parebias(const PareArgs& a, hipStream_t s);

struct DecodeArgs {
  const float* center[2];
  const float* params[2];
  const float* prior[2];
  int center_cs, params_cs, prior_cs;
  int B;
  float* slots;
  float thresh;          // centermap_conf_thresh (acr/result_parser.py:241), strict >
  const unsigned* poison; // non-null and != 0 on the device: the program's results are invalid (fp16x3 range overflow) - every
                         // slot is written as NaN, so that MANO's meshes are NaN too instead of plausible numbers
  const int* prior_gate; // null: the cross-hand prior is decided per frame (both hands found, centers <= 32 px apart);
                         // else [B]: < 0 = per frame, 0 = no prior, 1 = prior whenever the frame has both hands (the caller
                         // has applied the reference's batch-wide rules, acr/result_parser.py:42-47,131)
};
hipError_t launch_decode(const DecodeArgs& a, hipStream_t s);
// The reference's BATCH-WIDE prior decision (acr/result_parser.py:42-47, 102-145) from a first decode's slots [B,2,ACRMI_SLOT]:
// gate[b] = 1 when frame b has both hands AND the batch has a left and a right detection AND the left center of the FIRST
// left-detected frame and the right center of the FIRST right-detected frame are <= 32 map pixels apart; else 0.  One workgroup.
hipError_t launch_prior_gate(const float* slots, int B, int* gate, hipStream_t s);

// One-Euro smoothing of the decoded (poses, betas) of one video stream, frames in order (acr/utils.py:1466-1527)
struct SmoothArgs {
  float* slots;          // [B,2,ACRMI_SLOT], smoothed in place where the flag is set
  int B;
  float* state;          // [2 hands][x_raw | x_filt | dx_filt][64]
  int* init;             // [2]
  float mincutoff, mincutoff_betas, beta, freq;
  float alpha_d, one_minus_alpha_d;   // derivative filter: compute_alpha(dcutoff) in double, rounded once
  float two_pi, te;
};
hipError_t launch_smooth(const SmoothArgs& a, hipStream_t s);

// Point heads: the six non-center head towers evaluated only at the pixels the decode reads (heads.hip).
// Per tower (floats): entry W [9 taps][9 cin/4][64 cout][4] + b[64]; 4 x (conv W [9][16][64][4] + b[64]);
// exit W [64][112] + b[112].  Three towers per side in the order params (k=1), cam (k=3), prior (k=4).
constexpr int TP_ENTRY_W = 9 * 9 * 64 * 4;
constexpr int TP_CONV_W = 9 * 16 * 64 * 4;
constexpr int TP_EXIT_N = 112;
constexpr int TP_TOWER_FLOATS = TP_ENTRY_W + 64 + 4 * (TP_CONV_W + 64) + 64 * TP_EXIT_N + TP_EXIT_N;
struct PointArgs {
  const float* x34; int x_cs;              // [B,128,128,x_cs] backbone features + coord maps (34 channels)
  const float* center[2]; int center_cs;   // [B,64,64,center_cs] channel 0
  const float* w;                          // 3 x TP_TOWER_FLOATS (this side)
  const float* mix_w;                      // [109 cin][TP_EXIT_N] mix conv (cam3 columns merged)
  const float* bias; int bias_stride;      // per-frame pare bias rows
  float* p109; int p109_cs;                // [B,64,64,cs] cam(3) | params(106) before the mix
  float* prior; int prior_cs;              // [B,64,64,cs] 106-ch prior map
  float* final_; int final_cs;             // [B,64,64,cs] 109-ch params map after the mix
  int* picks;                              // [B,4] workspace: flat_l, flat_r, prior gate
  int side, B;
  float thresh;                            // centermap_conf_thresh
  int prior_when_both;                     // ACRMI_OPT_BATCH_PRIOR: evaluate the prior point whenever the frame has BOTH hands - the
                                           // batch-wide rule (acrmi_prior_gate) may open the gate for a frame whose own centers are
                                           // more than 32 px apart; the gated decode then decides whether the value is added
};
hipError_t launch_point_heads(const PointArgs& a, hipStream_t s);

struct ManoTables {   // device pointers
  const float* v_template;   // [778*3]
  const float* shapedirs_t;  // [10][2334]   (transposed for coalesced reads)
  const float* posedirs_t;   // [135][2334]
  const float* jreg;         // [16][778]
  const float* weights;      // [778][16]
  const float* hands_mean;   // [45]
  // f16 copies for ACRMI_OPT_MANO_FP16 (same layouts)
  const unsigned short* shapedirs_h;
  const unsigned short* posedirs_h;
  const unsigned short* weights_h;
  // round 6: the two SMALL tables as f16 pairs - lo = f16(x - float(hi)); value = float(hi) + float(lo), ~22 bits - the
  // skinning weights (a plain-f16 weight near 0.5 is off by 2.4e-4: x a ~0.1 m joint offset = the 6e-5 m of r5's configs[4]) and
  // the shape blend shapes.  The pose blend table (1.26 of the 1.46 MB per side: what "fp16 MANO LBS" halves) stays plain f16.
  const unsigned short* shapedirs_l;
  const unsigned short* weights_l;
};
struct ManoArgs {
  ManoTables t[2];
  const float* poses; int pose_stride;
  const float* betas; int beta_stride;
  const int32_t* side;
  int H, center_idx;
  float *verts, *joints, *center;
  const float* cam; int cam_stride;
  const float* offsets; int off_div;   // offsets row = hand row / off_div
  float *verts_camed, *pj2d, *pj2d_org;
  int lbs_f16;                         // blend-shape tables and skinning weights from their f16 copies
  int slices;                          // workgroups per hand (launch_mano sets it): slice s computes pose blend, skinning and
                                       // outputs of vertices [s * ceil(778 / slices), ...) - everything before the pose blend
                                       // (rotations, shape blend, joint regression, chain) is cheap and repeated per slice
  int pose_rotmat;                     // 1: `poses` rows are 16 row-major 3x3 rotation matrices (joint_rot_mode='rotmat',
                                       // mano/manolayer.py:151-162): no Rodrigues, no hands_mean
};
hipError_t launch_mano(const ManoArgs& a, hipStream_t s);
hipError_t launch_cam_trans(const float* joints, const float* pj2d, int n, float focal, float img, float* out, hipStream_t s);

}  // namespace acrmi
