/*
 * acrmi.h - C ABI of libacrmi.so: the MI355X (gfx950) ACR inference hot path.
 *
 * uint8 frame batch -> HRNet-W32 backbone -> ACR heads -> center decode -> MANO (L+R)
 * -> 778x3 vertices + 21x3 joints per hand.  Plain pointers and sizes only: no torch,
 * no C++ types.  All pointers marked "dev" are device (HBM) pointers owned by the caller
 * unless stated otherwise; every call is asynchronous on the given hipStream_t (passed as
 * void*; NULL = the null stream) and performs no hidden synchronisation.  Return value:
 * 0 on success, a negative ACRMI_E* code otherwise (acrmi_last_error() has the text).
 * One context per device, not thread-safe (the reference is single-threaded:
 * /root/reference/acr/main.py:126-141).
 *
 * The reference has no FFI layer; each entry point names the reference Python
 * interface it replaces (paths relative to the reference tree).
 */
#ifndef ACRMI_H
#define ACRMI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ACRMI_VERSION 303

#define ACRMI_OK 0
#define ACRMI_EINVAL (-1)  /* bad argument / unsupported shape  (reference: ValueError / assert) */
#define ACRMI_EHIP (-2)    /* HIP runtime error                                              */
#define ACRMI_ESTATE (-3)  /* call order violated (weights/program/MANO tables not loaded)   */
#define ACRMI_ENOMEM (-4)
#define ACRMI_ERANGE (-5)  /* an 'fp16x3' program met an activation outside the f16 range (acrmi_check_range) */

typedef struct acrmi_ctx acrmi_ctx;

/* ---------------------------------------------------------------------------------------
 * Program description.  The Python host (packer.py) folds BatchNorm into the conv weights,
 * packs them in the MFMA fragment order, and lowers the network topology
 * (acr/model.py:785-865 backbone, :47-166 heads) into a flat op list over numbered
 * activation buffers (NHWC, channel stride = cs elements; fp32, or f16 / bf16 between the layers of
 * a 16-bit program).  The library owns the buffers and replays the list; it knows nothing about HRNet.
 * ------------------------------------------------------------------------------------- */
/* Storage type of an activation buffer.  An fp32 program (the reference's configs/demo.yml, model_precision fp32) uses
 * ACRMI_DT_F32 everywhere.  A 16-bit program (the reference's autocast branch, acr/model.py:33-37, --model_precision
 * fp16; bf16 is the same program with the other 16-bit type) stores the activations BETWEEN layers as f16 / bf16, keeps
 * fp32 accumulation and fp32 bias / residual / ReLU arithmetic inside every kernel, and keeps the head outputs (center,
 * params, prior, segm maps), the pooled part features and everything behind them in fp32 (acr/model.py:56-62 .float()). */
enum { ACRMI_DT_F32 = 0, ACRMI_DT_F16 = 1, ACRMI_DT_BF16 = 2 };

typedef struct {
  int32_t h, w, cs;      /* per-frame height, width, channel stride in ELEMENTS (a multiple of 16 bytes) */
  int32_t persistent;    /* 1: never aliased with another buffer (holds init-time constants) */
  int32_t dtype;         /* ACRMI_DT_*; all 16-bit buffers of a program share one type */
} acrmi_buffer_desc;

enum {
  ACRMI_OP_U8NORM = 1,   /* uint8 NHWC image -> fp32 (x/255*2-1), pad channel zeroed (acr/model.py:832) */
  ACRMI_OP_CONV = 2,     /* KxK conv (K in {1,3}, stride in {1,2}) + bias [+ residual] [+ ReLU] */
  ACRMI_OP_FUSESUM = 3,  /* out = [relu](sum_t nearest_up(term_t, 2^shift_t))  (acr/model.py:677-684) */
  ACRMI_OP_BILINEAR2X = 4, /* bilinear x2, align_corners=True (acr/model.py:432)              */
  ACRMI_OP_POW11 = 5,    /* ch0 := 1.1 ** ch0 (acr/model.py:95-96)                            */
  ACRMI_OP_ATTPOOL = 6,  /* softmax-over-pixels weighted feature pooling (acr/model.py:103-113) */
  ACRMI_OP_PAREBIAS = 7, /* LocallyConnected2d + Linear + mix-conv pare bias (acr/model.py:145-164) */
  ACRMI_OP_COORDFILL = 8, /* init-time: write coord maps into 2 channels (acr/model.py:340-369) */
  ACRMI_OP_POINTHEADS = 9, /* params/cam/prior head towers + 109x109 mix at the decoded centers only (one op per
                             side = flags; in = backbone+coord buffer, res = pre-mix 109-ch map, out = params
                             map, aux = per-frame bias; w_off = 3 packed towers, w_off2 = mix weights);
                             acr/model.py:71-99,160-164 restricted to the pixels acr/result_parser.py:49-57,
                             141-145 samples */
  ACRMI_OP_PAIR1X1 = 12,  /* two chained 1x1 convolutions of layer1 (acr/model.py:519-539) in one kernel: out = relu(W3 in +
                             b3 + res) (64 -> 256 channels; in_buf, res_buf, out_buf) and aux = relu(W1 out + b1) (256 -> 64;
                             aux_buf = the next block's conv1 output); w_off = packer.pack_pair1x1 (both matrices + biases in
                             the kernel's LDS order, 33088 floats); fp32 programs */
  ACRMI_OP_MAXPOOL = 11,  /* max pooling 3x3 stride 2 pad 1 of cin channels (ResNet stem; in -> out [B,(H-1)/2+1,(W-1)/2+1]) */
  ACRMI_OP_STEM = 10      /* uint8 image -> relu(conv3x3 stride 2 (x/255*2-1) + b), 3 -> 64 channels: U8NORM + the
                             first CONV in one kernel (acr/model.py:832,589-603; in = the image, out = [B,H/2,W/2,>=64]);
                             w_off = packer.pack_stem fragments [14][2][64], b_off = 64 biases.  ksize = 7: the ResNet stem
                             (7x7 stride 2 pad 3; BASELINE.json configs[1]'s backbone), w_off = packer.pack_stem7
                             fragments [74][2][64] */
};

/* acrmi_op.flags of a CONV, bit 3: a position-bias map - [Ho][Wo][round4(groups*Cout)] fp32 at blob offset w_off2 -
 * is added to EVERY frame's output before the ReLU, in the place of a residual (res_buf must be -1; fp32 programs; not
 * algo 3).  It stands for input channels that are a fixed function of the pixel position: the two coordinate channels
 * of the reference's head convs (acr/model.py:52,340-369) leave the contraction (34 -> 32 input channels) and come back
 * as conv(coord maps, their filter columns), computed once at lowering time (packer.coord_bias_map). */
#define ACRMI_CONV_BIAS_MAP 8
/* acrmi_op.flags of a CONV, bit 4: split-K (algo 2, fp32, small-batch programs).  `groups` K-slices of ONE convolution:
 * slice s reads input channels [in_coff + s*cin, in_coff + (s+1)*cin) with its own packed filters (w_off: `groups`
 * consecutive pack_conv(winograd2d_weights(w[:, s*cin:(s+1)*cin])), b_off: `groups` bias rows of which the first is
 * used), all slices produce the same cout channels.  The slices are separate work items on separate CUs; their partial
 * tiles meet in a workspace of the context and are summed in slice order by whichever item arrives last (bit-reproducible
 * results).  cin % 32 == 0, cin >= 64, 2 <= groups <= 8.  What it is for: a 256 -> 256 layer on a 16x16 map is 16 work
 * items of 8 chunks for 256 CUs at batch 1 - 64 items of 2 chunks with 4 slices. */
#define ACRMI_CONV_SPLITK 16
/* acrmi_op.flags of a CONV, bit 5: second output (algo 3 only: 3x3 stride 1, Cin <= 32, Cout = 32).  out_buf receives the
 * convolution as usual (bias, residual, ReLU); the op's nterms extra maps are NOT added to it but to a second map:
 * aux_buf = relu(((out + up(term 0)) + up(term 1)) + up(term 2)), each term read at (y >> shift, x >> shift).  This is the
 * HR-module fuse sum of the FULL resolution (acr/model.py:672-686, i = 0): its first term, the output of branch 0's last
 * conv2, is also the input of the downsampling chains, so both maps are needed.  Bit-equal to the ACRMI_OP_FUSESUM launch. */
#define ACRMI_CONV_DUAL 32

/* acrmi_op.mode: which variant of the head program an op belongs to. */
enum {
  ACRMI_MODE_BOTH = 0,
  ACRMI_MODE_DENSE = 1,  /* only when the full head maps are computed (default)            */
  ACRMI_MODE_POINT = 2   /* only with ACRMI_OPT_POINT_HEADS (acrmi_forward)                */
};

typedef struct {
  int32_t kind;
  int32_t in_buf, out_buf, res_buf;       /* buffer ids, -1 = none (U8NORM: in = the image)  */
  int32_t in_coff, out_coff, res_coff;    /* channel offsets inside the buffers              */
  int32_t cin, cout;                      /* logical channels (per group)                    */
  int32_t ksize, stride, relu, groups;
  int64_t w_off, b_off;                   /* float offsets into the weight blob              */
  int32_t bias_per_frame;                 /* 1: bias comes from aux buffer, row per frame    */
  int32_t aux_buf;                        /* PAREBIAS/ATTPOOL output or per-frame bias input */
  int32_t nterms;                         /* FUSESUM: 1..4 terms.  CONV (fp32 3x3 stride 2, Cout % 32 = 0): 0..3 EXTRA residual terms */
  int32_t term_buf[4], term_coff[4], term_shift[4];   /* added behind res_buf and before the ReLU, in this order, term t read
                                             at pixel (y >> shift, x >> shift) of a map 1 / 2^shift the output's size (nearest
                                             upsampling): the HR-module fuse sum (acr/model.py:672-686) in the epilogue of the
                                             x0 downsampling chain's last convolution */
  int64_t w_off2, b_off2, w_off3;         /* PAREBIAS: linear weights/bias, mix-conv pare columns */
  int32_t flags;                          /* CONV: algo 0..7 (bits 0-2) | ACRMI_CONV_BIAS_MAP; PAREBIAS: part slice start
                                             (0 right / 16 left); POINTHEADS: side (0 left / 1 right) */
  int32_t mode;                           /* ACRMI_MODE_*                                    */
} acrmi_op;

/* Where the head outputs live (buffer ids of the program), needed by acrmi_decode. */
typedef struct {
  int32_t center_buf[2];   /* [left,right] center maps  [B,64,64,cs], channel 0   */
  int32_t params_buf[2];   /* final 109-ch params maps  (acr/model.py:163-164)    */
  int32_t prior_buf[2];    /* 106-ch prior maps                                   */
  int32_t segm_buf;        /* 33-ch part-segmentation logits [B,256,256,cs]       */
  int32_t backbone_buf;    /* 32(+2 coord)-ch backbone output [B,128,128,cs]      */
} acrmi_head_layout;

/* Fixed per-(frame,hand) result slot written by acrmi_decode: ACRMI_SLOT floats.
 * hand 0 = left, 1 = right (acr/result_parser.py:166-168 ordering is rebuilt on the host). */
#define ACRMI_SLOT 176
#define ACRMI_SLOT_FLAG 0      /* 1.0 if center score > ACRMI_OPT_CONF_THRESH (strict)      */
#define ACRMI_SLOT_FLATIND 1   /* y*64+x of the center (0 when not detected)               */
#define ACRMI_SLOT_SCORE 2
#define ACRMI_SLOT_CAM 3       /* 3  */
#define ACRMI_SLOT_POSES 6     /* 48: global orient (3) + 15 joints axis-angle              */
#define ACRMI_SLOT_BETAS 54    /* 10 */
#define ACRMI_SLOT_PARAMS 64   /* 109: sampled params (+ cross-hand prior)                  */

int acrmi_version(void);
const char* acrmi_last_error(const acrmi_ctx* ctx); /* ctx may be NULL: last error of acrmi_create */

/* acr/main.py:57-63 (_build_model_): one context per device. */
int acrmi_create(acrmi_ctx** out, int device);
void acrmi_destroy(acrmi_ctx* ctx);

/* acr/utils.py:1153-1168 (load_model/copy_state_dict): host blob of packed fp32 weights
 * (BN folded, MFMA fragment order) copied once into HBM; static residency replaces
 * nn.DataParallel's per-call replicate (acr/main.py:61). */
int acrmi_load_weights(acrmi_ctx* ctx, const float* blob_host, size_t n_floats);
/* The same blob for several contexts of one device (a pool of contexts that take batches in turn): `ctx` uses
 * the device copy `donor` holds instead of uploading its own - 330 MB per extra context at HRNet-W32 fp32.  The copy is freed
 * when the last context holding it is destroyed or loads other weights.  Both contexts need the same program.  Thread safety:
 * contexts sharing a blob may be destroyed from different threads; this call itself must not run concurrently with a destroy or
 * a weight reload of `donor` (serialize them in the host). */
int acrmi_share_weights(acrmi_ctx* ctx, acrmi_ctx* donor);

/* Lowered topology of acr/model.py:785-865 + :47-166; allocates activation buffers for
 * up to max_batch frames (aliasing non-overlapping lifetimes) and runs the init-time ops. */
int acrmi_set_program(acrmi_ctx* ctx, const acrmi_buffer_desc* bufs, int n_bufs, const acrmi_op* ops,
                      int n_ops, const acrmi_head_layout* heads, int max_batch);

/* mano/manolayer.py:13-102 (ManoLayer.__init__ buffers) for side 0 = left, 1 = right.
 * Host pointers, row-major as the reference registers them: v_template[778*3],
 * shapedirs[778*3*10], posedirs[778*3*135], J_regressor[16*778], weights[778*16],
 * hands_mean[45].  The left-hand shapedirs x-flip (acr/mano_wrapper.py:35) is the caller's job. */
int acrmi_load_mano(acrmi_ctx* ctx, int side, const float* v_template, const float* shapedirs,
                    const float* posedirs, const float* J_regressor, const float* weights,
                    const float* hands_mean);

/* acr/model.py:32-44 minus the parser: backbone + head_forward on B frames.
 * img_dev: uint8 [B,512,512,3] RGB NHWC (meta_data['image']).  Results stay in the
 * program's head buffers (see acrmi_buffer_ptr). */
int acrmi_backbone_heads(acrmi_ctx* ctx, const uint8_t* img_dev, int B, void* stream);
/* acr/model.py:47-65 (ACR.head_forward(x)): the heads alone on backbone features the caller holds - feat_nchw_dev is fp32
 * [B, C0, 128, 128] (C0 = acrmi_backbone_channels: 32 for HRNet-W32), e.g. what acr.model.ACR.backbone returned.  The
 * features are copied into the program's backbone map (coordinate channels are already there) and the ops behind the backbone
 * run in program order on `stream`; results as after acrmi_backbone_heads (acrmi_buffer_ptr / acrmi_decode).  fp32-storage
 * programs only (ACRMI_EINVAL for the 16-bit storage programs). */
int acrmi_heads(acrmi_ctx* ctx, const float* feat_nchw_dev, int B, void* stream);
int acrmi_backbone_channels(acrmi_ctx* ctx); /* channels of the backbone output (without the 2 coordinate maps); < 0: error */

/* Device pointer / geometry of program buffer `buf` (valid until the next set_program); cs counts elements of
 * acrmi_buffer_dtype(ctx, buf) (ACRMI_DT_*, -1 for an unknown buffer). */
void* acrmi_buffer_ptr(acrmi_ctx* ctx, int buf, int* h, int* w, int* cs);
int acrmi_buffer_dtype(acrmi_ctx* ctx, int buf);

/* acr/result_parser.py:21-40,85-190 (ResultParser.parse/parse_maps) + acr/utils.py:334-382
 * (6D -> axis-angle), per-frame semantics: slots_dev [B,2,ACRMI_SLOT]. */
int acrmi_decode(acrmi_ctx* ctx, int B, float* slots_dev, void* stream);
/* The same with the cross-hand prior decided by the CALLER per frame: prior_gate_dev [B] int32 on the device (NULL = as
 * acrmi_decode): < 0 = this frame by the per-frame rule, 0 = no prior, 1 = prior when the frame has both hands.  This is how
 * the host reproduces the reference's batch > 1 behaviour (acr/result_parser.py:131: the prior only when EVERY flag of the
 * batch is set; :42-47: determine_coeff reads row 0 of each side's list) - acr/result_parser.py ResultParser(batch_semantics=
 * 'reference') decodes once, applies those batch-wide rules to the flags / centers, and decodes again with the gate. */
int acrmi_decode_gated(acrmi_ctx* ctx, int B, const int32_t* prior_gate_dev, float* slots_dev, void* stream);
/* acr/result_parser.py:42-47 (determine_coeff) + :102-145 (parse_maps at batch > 1), the rules that look ACROSS frames, on the
 * device: slots_dev [B,2,ACRMI_SLOT] of a first decode -> gate_dev [B] int32 for acrmi_decode_gated: 1 in the frames that have
 * both hands, provided the batch holds a left and a right detection at all (:131) and the left center of the first
 * left-detected frame is <= 32 map pixels from the right center of the first right-detected frame (:42-47, row 0 of each
 * list); else 0.  acrmi_forward does decode -> this -> gated decode by itself when ACRMI_OPT_BATCH_PRIOR is set.  ctx may be
 * NULL (stand-alone use next to acrmi_decode_maps_gated: the current device). */
int acrmi_prior_gate(acrmi_ctx* ctx, const float* slots_dev, int B, int32_t* gate_dev, void* stream);
/* 'fp16x3' programs (fp32 tensors, operands split into two f16 numbers - conv algo 6): an activation with |x| > 65504 cannot
 * be split (hi = inf).  The split kernels track it per launch at one v_max3 per two values; while the flag is set acrmi_decode /
 * acrmi_forward write every slot as NaN (so the meshes are NaN too: nothing plausible-looking leaves the library).  This call
 * waits for `stream`, returns ACRMI_ERANGE if the flag was set since the last check and clears it; ACRMI_OK otherwise and for
 * programs without split-f16 convolutions (fp32, bf16x3: fp32's exponent range). */
int acrmi_check_range(acrmi_ctx* ctx, void* stream);

/* Same decode on caller-supplied NHWC maps (unit tests / callers with their own maps):
 * center [B,64,64,center_cs] (ch 0), params [B,64,64,params_cs] (109 ch), prior (106 ch). */
int acrmi_decode_maps(const float* l_center, const float* r_center, int center_cs,
                      const float* l_params, const float* r_params, int params_cs,
                      const float* l_prior, const float* r_prior, int prior_cs, int B,
                      float conf_thresh /* CenterMap.conf_thresh = args().centermap_conf_thresh, 0.35 */,
                      float* slots_dev, void* stream);
int acrmi_decode_maps_gated(const float* l_center, const float* r_center, int center_cs,
                            const float* l_params, const float* r_params, int params_cs,
                            const float* l_prior, const float* r_prior, int prior_cs, int B, float conf_thresh,
                            const int32_t* prior_gate_dev /* see acrmi_decode_gated */, float* slots_dev, void* stream);

/* mano/manolayer.py:104-276 (ManoLayer.forward, use_pca=False, flat_hand_mean=False,
 * center_idx as given; <0 = no root alignment) fused with acr/utils.py:384-412
 * (batch_orth_proj / convert_kp2d_from_input_to_orgimg).
 * Row r reads poses[r*pose_stride..+48], betas[r*beta_stride..+10]; side[r] (0 left / 1 right)
 * or, when side == NULL, side = r & 1 (slot order).  cam/offsets/verts_camed/pj2d/pj2d_org may
 * be NULL (projection skipped).  cam row stride = cam_stride, offsets [H,10] row per hand. */
int acrmi_mano(acrmi_ctx* ctx, const float* poses, int pose_stride, const float* betas, int beta_stride,
               const int32_t* side, int H, int center_idx, float* verts, float* joints, float* center,
               const float* cam, int cam_stride, const float* offsets, float* verts_camed, float* pj2d,
               float* pj2d_org, void* stream);

/* mano/manolayer.py:151-162 (joint_rot_mode='rotmat', use_pca=False): the pose of row r is rotmats[r*144..+144] = 16 row-major
 * 3x3 rotation matrices, root first, ALREADY orthonormal (the reference projects them with a CPU SVD in batch_rotprojs,
 * :436-453; the Python ManoLayer of this package does the same on the host).  No Rodrigues, th_hands_mean is not applied.
 * Everything behind the rotations - blend shapes, joint regression, chain, skinning, tips, root alignment - is acrmi_mano's. */
int acrmi_mano_rotmat(acrmi_ctx* ctx, const float* rotmats, const float* betas, int beta_stride, const int32_t* side,
                      int H, int center_idx, float* verts, float* joints, float* center, void* stream);

/* acr/utils.py:430-472 + :474-519 (estimate_translation_np / estimate_translation, the reference's closed-form
 * least-squares branch, unit joint confidences): joints [n,21,3] and pj2d [n,21,2] (in [-1,1]; the 2-D targets are
 * (pj2d+1)*img_size/2) on the device -> cam_trans [n,3].  Solved in fp64 like the reference. */
int acrmi_cam_trans(const float* joints_dev, const float* pj2d_dev, int n, float focal_length, float img_size,
                    float* trans_dev, void* stream);

/* acr/main.py:126-141 + :85 in one call: frames -> slots [B,2,ACRMI_SLOT], verts [B,2,778,3],
 * joints [B,2,21,3] (root-aligned on joint ACRMI_OPT_CENTER_IDX = 9, metres; threshold ACRMI_OPT_CONF_THRESH;
 * smoothed when ACRMI_OPT_TEMPORAL).  offsets_dev [B,10] may be NULL. */
int acrmi_forward(acrmi_ctx* ctx, const uint8_t* img_dev, int B, const float* offsets_dev, float* slots_dev,
                  float* verts_dev, float* joints_dev, float* verts_camed_dev, float* pj2d_dev,
                  float* pj2d_org_dev, void* stream);

/* Stand-alone operators (same kernels the program uses; for parity tests and embedding).
 * w_packed/bias come from the Python packer (pack_conv).  algo: 0 = direct convolution;
 * 1 = Winograd F(2,3) along x (3x3 stride 1 only; w_packed = pack_conv(winograd_weights(w)));
 * 2 = Winograd F(2x2,3x3) (3x3 stride 1 only; w_packed = pack_conv(winograd2d_weights(w)));
 * 3 = Winograd F(2x2,3x3) with the layer's taps resident in LDS (groups 1, Cin <= 32, Cout = 32, H % 8 == 0,
 *     W % 16 == 0; w_packed = pack_wino3(w), 64 KiB);
 * 4 = Winograd F(2x4,3x3): F(2,3) along y, F(4,3) along x (3x3 stride 1, Cin > 32; w_packed =
 *     pack_conv(winograd24_weights(w)), 4x6 taps);
 * 5 = 3x3 STRIDE 2 in polyphase form with F(2,2) on the two-tap phases (Cin % 16 == 0, Cout % 32 == 0, H % 16 == 0,
 *     W % 32 == 0; w_packed = pack_conv(polyphase2_weights(w)), 4 waves x 7 taps; csrc/conv_pp2.inc);
 * 6 = 3x3 stride 1 (or 1x1 stride 1 with H W % 256 == 0: csrc/conv_x3p.inc) with SPLIT operands on the 16-bit matrix pipe: fp32 tensors, every operand split into f16 hi + lo
 *     in registers, three products per MAC on v_mfma_f32_32x32x16_f16, fp32 accumulation (Cin % 32 == 0, Cout % 32 == 0,
 *     H % 8 == 0, W % 32 == 0, |x| < 65504; w_packed = pack_conv_x3([(w, b)]): split f16 fragments of the filters scaled by
 *     a power of two + one trailing float holding the inverse scale; csrc/conv_x3.inc).  The 'fp16x3' programs;
 * 7 = the same with bf16 halves (v_mfma_f32_32x32x16_bf16; 16-bit operands, fp32's exponent range: no |x| limit;
 *     w_packed = pack_conv_x3([(w, b)], DT_BF16)).  The 'bf16x3' programs.
 * algo | ACRMI_CONV_BIAS_MAP (not with 3): res is ONE map [Ho][Wo][res_cs] added to every frame (see acrmi_op.flags). */
int acrmi_conv2d(const float* in, int B, int H, int W, int in_cs, int in_coff, int cin, const float* w_packed,
                 const float* bias, int bias_frame_stride, const float* res, int res_cs, int res_coff,
                 float* out, int out_cs, int out_coff, int cout, int ksize, int stride, int relu, int groups,
                 int algo, void* stream);
/* The convolution of a 16-bit program (acr/model.py:33-37, the autocast branch): in / res / out point at f16 (dtype =
 * ACRMI_DT_F16) or bf16 (ACRMI_DT_BF16) NHWC tensors, strides / offsets / channel counts in elements (strides multiples
 * of 8), w_packed = packer.pack_conv_h16(w, b, dtype) (BN-folded filters rounded once to the storage type, A fragments
 * of v_mfma_f32_32x32x16_{f16,bf16}), bias fp32.  fp32 accumulation; bias, residual and ReLU in fp32; ONE rounding of
 * the result.  out_f32 != 0: out AND res are fp32 tensors (strides multiples of 4; stride-1 shapes only) - the head
 * exits, whose maps the reference converts with .float() (acr/model.py:56-62). */
/* Split-K form of acrmi_conv2d with algo 2 (see ACRMI_CONV_SPLITK): cin_slice channels per slice, `splits` slices,
 * w_packed / bias = the slices' pack_conv outputs one after the other.  workspace: acrmi_conv2d_splitk_workspace() bytes
 * of device memory, 16-byte aligned, ZEROED ONCE by the caller (every launch leaves its counters zero again) and not
 * shared by launches that may overlap. */
size_t acrmi_conv2d_splitk_workspace(int B, int H, int W, int cout, int splits);
int acrmi_conv2d_splitk(const float* in, int B, int H, int W, int in_cs, int in_coff, int cin_slice, int splits,
                        const float* w_packed, const float* bias, const float* res, int res_cs, int res_coff, float* out,
                        int out_cs, int out_coff, int cout, int relu, void* workspace, size_t workspace_bytes, void* stream);

int acrmi_conv2d_h16(const void* in, int B, int H, int W, int in_cs, int in_coff, int cin, const void* w_packed,
                     const float* bias, int bias_frame_stride, const void* res, int res_cs, int res_coff, void* out,
                     int out_cs, int out_coff, int cout, int ksize, int stride, int relu, int groups, int dtype,
                     int out_f32, void* stream);
/* acr/utils.py:1276-1337 (img_preprocess / image_pad_white_bg / cv2.resize INTER_CUBIC): n BGR uint8 frames
 * [n,H,W,3] on the device -> RGB uint8 [n,512,512,3]: white pad to square (imgaug 0.4.0
 * compute_paddings_to_reach_aspect_ratio + Pad), then OpenCV's uint8 INTER_CUBIC restated bit for bit (a = -0.75,
 * 11-bit fixed-point coefficients, clamped border, (v + 2^21) >> 22; see oracle/preprocess.py).  offsets_host
 * [n,10] (may be NULL) receives the reference's `offsets` rows (padded h, padded w, crop trbl = 0, pad trbl). */
int acrmi_preprocess(const uint8_t* bgr_dev, int n, int H, int W, uint8_t* out_rgb_dev, float* offsets_host,
                     void* stream);
/* The same for frames of DIFFERENT sizes in one call (img_preprocess is per image, acr/utils.py:1315-1337; folder mode,
 * acr/main.py:144-205, mixes sizes): frames_host [n] = where each BGR uint8 frame [H,W,3] lives on the device and its size -
 * frames need not share an allocation.  Same arithmetic, bit for bit; out_rgb_dev [n,512,512,3]; offsets_host [n,10] (may be
 * NULL) = each image's own `offsets` row.  The geometry travels in the kernel arguments (128 frames per launch): nothing is
 * allocated or uploaded, the host array may be freed when the call returns. */
typedef struct acrmi_frame {
  const uint8_t* bgr_dev;
  int32_t H, W;
} acrmi_frame;
int acrmi_preprocess_frames(const acrmi_frame* frames_host, int n, uint8_t* out_rgb_dev, float* offsets_host, void* stream);
int acrmi_u8norm(const uint8_t* img, int n_pixels, float* out, void* stream);
/* ACRMI_OP_STEM stand-alone: img uint8 RGB [B,H,W,3] (H % 16 == 0, W % 128 == 0) -> [relu](conv3x3 stride 2 pad 1 of
 * (x/255*2-1) + bias) into channels out_coff..out_coff+63 of out [B,H/2,W/2,out_cs]; w_packed = packer.pack_stem(w
 * [64,3,3,3]) (acr/model.py:832,589-603). */
int acrmi_stem_conv(const uint8_t* img, int B, int H, int W, const float* w_packed, const float* bias, float* out,
                    int out_cs, int out_coff, int relu, void* stream);
int acrmi_bilinear2x(const float* in, int B, int H, int W, int in_cs, int C, float* out, int out_cs, void* stream);
int acrmi_fuse_sum(int nterms, const float* const* terms, const int* term_cs, const int* term_shift, int B, int H,
                   int W, int C, float* out, int out_cs, int relu, void* stream);
/* acr/model.py:103-113,126-136: pooled[b][part][c] = sum over the 128x128 pixels of softmax_pix(segm logit of the
 * part at the even pixels of the 256x256 map) * feat[b][pix][c].  ws: device workspace of at least
 * acrmi_attpool_ws_floats(B, C) floats. */
size_t acrmi_attpool_ws_floats(int B, int C);
int acrmi_attpool(const float* segm, int segm_cs, const float* feat, int feat_cs, int C, int B, float* ws,
                  float* pooled, void* stream);
/* acr/model.py:141-164 per frame and hand: LocallyConnected2d (:559-569) on the 16 pooled part features of this side
 * (parts part0..part0+15 of pooled [B,32,C]; C = 320: 256 contact + 64 shape channels as the reference pools them),
 * the shape Linear, and the mix conv's pare columns: out[b][co] = mix_b[co] + sum_k mix_wp[co][k] * pare[k],
 * pare = [offsets 96 | shape 10].  lc_w [6][256][16], lin_w [10][(C==320 ? 64 : 256)*16], mix_wp [109][106]. */
int acrmi_parebias(const float* pooled_dev, int C, int part0, const float* lc_w_dev, const float* lin_w_dev,
                   const float* lin_b_dev, const float* mix_wp_dev, const float* mix_b_dev, int B, float* out_dev,
                   int out_stride, void* stream);

/* Options.  ACRMI_OPT_POINT_HEADS (0/1, default 0): acrmi_forward evaluates the params/cam/prior head towers and
 * the mix conv only at the pixels the decode samples (same slots/vertices within fp32 round-off; the dense
 * l/r_params_maps and l/r_prior_maps are then NOT produced - acrmi_backbone_heads always computes them). */
#define ACRMI_OPT_POINT_HEADS 1
/* ACRMI_OPT_LANES (0..8, default 0): the program's independent chains (HRNet branches, head towers, segm and part
 * heads; found from the ops' buffer reads/writes) run on that many HIP streams: lane 0 is the caller's stream, the
 * others fork from it at the start of a call and join it before the call's decode, so the caller still sees one
 * stream-ordered operation.  0 = chosen by batch size (4 lanes up to 32 frames, 2 above), 1 = single stream.
 * Same kernels, bit-identical results. */
#define ACRMI_OPT_LANES 2
/* ACRMI_OPT_CENTER_IDX (-1..20, default 9): the joint acrmi_forward's MANO stage aligns the mesh root on
 * (args().align_idx when args().mano_mesh_root_align, acr/mano_wrapper.py:19-33); -1 = no alignment. */
#define ACRMI_OPT_CENTER_IDX 3
/* ACRMI_OPT_TEMPORAL (0/1, default 0): acrmi_forward smooths the decoded poses/betas (acrmi_smooth) before MANO -
 * the reference's -t / temporal_optimization (acr/main.py:69-83).  The frames of a call are then ONE video stream
 * in order. */
#define ACRMI_OPT_TEMPORAL 4
/* ACRMI_OPT_MANO_FP16 (0/1, default 0; BASELINE.json configs[4] "fp16 MANO LBS"): the MANO stage (acrmi_mano and
 * acrmi_forward) reads its blend-shape tables (shapedirs, posedirs) and skinning weights from f16 copies made at
 * acrmi_load_mano - half the table traffic per hand; all products and sums stay fp32, v_template / J_regressor / the
 * kinematic chain are untouched.  Measured deviation from the fp32 tables: see tests/test_gpu_h16.py. */
#define ACRMI_OPT_MANO_FP16 7
/* ACRMI_OPT_LANE_PLAN (0/1, default 0 - planning is explicit): once acrmi_profile_ops has run at a small batch (<= 32 frames;
 * it stores the op times of the head mode that was active, dense or point, measured on one stream), the lanes of the
 * small-batch schedules are assigned by list scheduling over the MEASURED per-op times (every op goes to the lane where
 * it can start first; a cross-stream dependency is charged what it costs on MI355X, ~16 us over an in-stream one)
 * instead of by the structure of the graph alone.  Results do not depend on the assignment.  0 = structural only. */
#define ACRMI_OPT_LANE_PLAN 8
/* ACRMI_OPT_BATCH_PRIOR (0/1, default 0): 0 = every frame decides its cross-hand prior for itself - how the reference treats a
 * batch of ONE frame, the only way acr/main.py:126-141 ever calls it; 1 = acrmi_forward applies the reference's BATCH-WIDE
 * rules at B > 1 (acr/result_parser.py:42-47, 102-145: see acrmi_prior_gate) - decode, acrmi_prior_gate, gated decode, all on
 * the stream, no host round trip.  What ResultParser(batch_semantics='reference') / forward_batch(batch_semantics=...) set. */
#define ACRMI_OPT_BATCH_PRIOR 9
int acrmi_set_option(acrmi_ctx* ctx, int option, int value);
/* ACRMI_OPT_CONF_THRESH (default 0.35): center score threshold, strict > (args().centermap_conf_thresh,
 * acr/result_parser.py:198-205,241).  ACRMI_OPT_SMOOTH_COEFF (default 4.0): One-Euro mincutoff of the pose filters
 * (args().smooth_coeff, acr/main.py:45-47). */
#define ACRMI_OPT_CONF_THRESH 5
#define ACRMI_OPT_SMOOTH_COEFF 6
int acrmi_set_option_f(acrmi_ctx* ctx, int option, float value);

/* acr/main.py:69-83 + acr/utils.py:1466-1527 (smooth_results / OneEuroFilter / LowPassFilter): One-Euro smoothing
 * of slots_dev [B,2,ACRMI_SLOT] in place - poses[3:48] and betas directly, the global orientation as a rotation
 * matrix (acr/utils.py:1466-1470) - for the hands whose flag is set, frames in order, one filter set per hand type.
 * The filter state of ONE video stream lives in the context; acrmi_smooth_reset starts a new stream. */
int acrmi_smooth(acrmi_ctx* ctx, float* slots_dev, int B, void* stream);
int acrmi_smooth_reset(acrmi_ctx* ctx, void* stream);

/* Multi-GPU (replaces nn.DataParallel's scatter/gather, acr/main.py:61): frames are sharded by the caller, one
 * process per GPU; the only collective is ONE all-gather per batch of each rank's flat result buffer
 * [slots | verts | joints] over RCCL/xGMI.  recv_dev holds n_ranks * n_floats floats, rank-major.
 * acrmi_comm_unique_id: 128-byte ncclUniqueId, created on rank 0 and handed to every rank by the host's own means;
 * acrmi_comm_init: ncclCommInitRank on the context's device (collective: every rank calls it);
 * acrmi_allgather: ncclAllGather on `stream`; nccl_comm = NULL uses the context's communicator, otherwise a
 * caller-owned ncclComm_t.  RCCL is resolved at run time (dlopen); without it these calls return ACRMI_ESTATE.
 * ONE RCCL communicator per process: a process that also holds another one (e.g. torch.distributed's NCCL backend) pays ~11 ms
 * per batch for a side-stream all-gather next to a running batch (measured at world size 1: 46.8 vs 35.4 ms; +0.2 ms with this
 * library's communicator alone).  The library cannot see other libraries' communicators; the Python host refuses the
 * combination (Engine.comm_init / parallel.ShardedRunner(transport='c') next to a torch NCCL group raise unless overridden) -
 * another host either passes ITS communicator as nccl_comm, or runs its control plane over something else (bench.py: gloo). */
int acrmi_comm_unique_id(void* id128_out);
int acrmi_comm_init(acrmi_ctx* ctx, int n_ranks, int rank, const void* id128);
int acrmi_comm_destroy(acrmi_ctx* ctx);
int acrmi_allgather(acrmi_ctx* ctx, void* nccl_comm, const float* send_dev, float* recv_dev, size_t n_floats,
                    void* stream);

/* Runs only the point-heads ops on the resident buffers of the last acrmi_backbone_heads / acrmi_forward call:
 * params/cam/prior towers + mix (acr/model.py:71-99,160-164) at the centers of the CURRENT center maps, written
 * into the pixels of the params/prior maps acrmi_decode samples. */
int acrmi_point_heads(acrmi_ctx* ctx, int B, void* stream);

/* Profiling aid for bench.py: time every op of the program with hipEvents on `stream`
 * (one untimed warm-up pass first; ops outside the active head mode report 0).  ms_out[n_ops]; returns n_ops or <0.
 * At a small batch (<= 32 frames) the times are also kept in the context (per head mode) as the input of
 * ACRMI_OPT_LANE_PLAN; they change how later calls are scheduled only after that option has been switched on. */
int acrmi_profile_ops(acrmi_ctx* ctx, const uint8_t* img_dev, int B, float* ms_out, int n_ms, void* stream);

/* A non-blocking HIP stream on `device` / its release - for hosts without a stream API of their own (the Python
 * package runs the contexts of an EnginePool on these instead of torch's pooled streams). */
int acrmi_stream_create(int device, void** stream);
int acrmi_stream_destroy(void* stream);

/* Tuning hook for kernel experiments (tools/conv_bench.py, tools/ab_cfg.py); process-wide, not part of the
 * reference-facing surface.  key 0: force a conv kernel variant (-1 = automatic selection; 8xx ids are listed next to
 * the launchers in csrc/conv_mfma.hip, conv_wino2.inc, conv_wino3.inc, conv_ws2.inc - e.g. 806 large-batch item shapes
 * at any batch, 837 the 16x32 wave tile of conv_wino3b_kernel, 839 no store waves); key 1: cycle stamps of
 * workgroup 0 on/off; key 2: print them; key 3: loader-wave switches (8 idle loader - wrong results, 9 priority 0);
 * key 4: XCD-banded item order on/off. */
int acrmi_tune(int key, int value);

#ifdef __cplusplus
}
#endif
#endif /* ACRMI_H */
