"""Stand-alone operator wrappers over the C ABI (NHWC fp32 device tensors in, device tensors out).
Same kernels the resident program uses; handy for parity tests and for embedding single ops."""
import ctypes as C

import numpy as np
import torch

from . import _lib
from .packer import (DT_BF16, DT_F16, PRECISIONS, pack_conv, pack_conv_h16, pack_conv_x3, pack_stem, pack_wino3, n_tiles_for,
                     winograd_weights, winograd2d_weights, winograd24_weights, polyphase2_weights)


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _s(t):
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.AcrmiError('device tensors required: libacrmi has no CPU path')


def to_nhwc(x_nchw, cs=None, device='cuda'):
    """[B,C,H,W] (any device) -> contiguous NHWC on `device` with channel stride cs (zero padded)."""
    B, Cc, H, W = x_nchw.shape
    cs = cs or (Cc + 3) // 4 * 4
    out = torch.zeros(B, H, W, cs, dtype=torch.float32, device=device)
    out[..., :Cc] = x_nchw.permute(0, 2, 3, 1).to(device)
    return out


def conv2d(x, weight, bias=None, stride=1, relu=False, residual=None, groups=1, cin=None, in_coff=0, out=None,
           out_coff=0, out_cs=None, frame_bias=None, algo='direct'):
    """x: NHWC [B,H,W,cs] device fp32; weight: [Cout_total, Cin/groups, k, k] (torch/numpy, host or device);
    padding = k//2 (the only padding the ACR network uses).  Returns NHWC [B,Ho,Wo,out_cs].
    residual: [B,Ho,Wo,rcs] added before the ReLU, or [1,Ho,Wo,rcs] = one map added to EVERY frame
    (ACRMI_CONV_BIAS_MAP: the position-bias map of the head convs; not with 'winograd2d_lds')."""
    _need_cuda(x, residual, out, frame_bias)
    w = weight.detach().cpu().numpy() if hasattr(weight, 'detach') else np.asarray(weight)
    cout_t, cin_g, k, _ = w.shape
    cout = cout_t // groups
    b = np.zeros(cout_t, np.float32) if bias is None else (
        bias.detach().cpu().numpy() if hasattr(bias, 'detach') else np.asarray(bias))
    if algo in ('split16', 'split_bf16'):
        if k not in (1, 3) or stride not in (1, 2) or (k == 1 and stride != 1):
            raise ValueError('the split-operand kernels are for 3x3 (stride 1 or 2) and 1x1 stride-1 convolutions')
        tr = None
    elif algo == 'polyphase2':
        if k != 3 or stride != 2:
            raise ValueError('the polyphase kernel is for 3x3 stride-2 convolutions')
        tr = polyphase2_weights
    elif algo in ('winograd', 'winograd2d', 'winograd2d_lds', 'winograd24'):
        if k != 3 or stride != 1:
            raise ValueError('winograd needs a 3x3 stride-1 convolution')
        tr = {'winograd': winograd_weights, 'winograd24': winograd24_weights}.get(algo, winograd2d_weights)
    elif algo == 'direct':
        tr = lambda t: t
    else:
        raise ValueError('algo must be "direct", "winograd", "winograd2d", "winograd2d_lds", "winograd24", "polyphase2", "split16" or "split_bf16"')
    algo_id = {'direct': 0, 'winograd': 1, 'winograd2d': 2, 'winograd2d_lds': 3, 'winograd24': 4, 'polyphase2': 5, 'split16': 6, 'split_bf16': 7}[algo]
    if algo_id == 3:
        if groups != 1 or cout != 32 or cin_g > 32:
            raise ValueError('winograd2d_lds needs groups = 1, Cout = 32, Cin <= 32')
        packed = [pack_wino3(w.astype(np.float64), b)]
    elif algo_id in (6, 7):
        packed = [pack_conv_x3([(w[g * cout:(g + 1) * cout].astype(np.float64), b[g * cout:(g + 1) * cout]) for g in range(groups)],
                               DT_BF16 if algo_id == 7 else DT_F16)]
    else:
        packed = [pack_conv(tr(w[g * cout:(g + 1) * cout].astype(np.float64)), b[g * cout:(g + 1) * cout])
                  for g in range(groups)]
    wp = torch.from_numpy(np.concatenate([p[0] for p in packed])).to(x.device)
    bp = torch.from_numpy(np.concatenate([p[1] for p in packed])).to(x.device)
    B, H, W, cs = x.shape
    cin = cin_g if cin is None else cin
    pad = k // 2
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    if out is None:
        out_cs = out_cs or (cout_t + 3) // 4 * 4
        out = torch.zeros(B, Ho, Wo, out_cs, dtype=torch.float32, device=x.device)
    bias_t, fstride = bp, 0
    if frame_bias is not None:
        bias_t, fstride = frame_bias.contiguous(), frame_bias.shape[-1]
    if residual is not None and residual.shape[0] == 1 and B > 1:
        algo_id |= _lib.CONV_BIAS_MAP
    L = _lib.lib()
    _lib.check(L.acrmi_conv2d(_p(x), B, H, W, cs, in_coff, cin, _p(wp), _p(bias_t), fstride, _p(residual),
                              residual.shape[-1] if residual is not None else 0, 0, _p(out), out.shape[-1], out_coff,
                              cout, k, stride, int(relu), groups, algo_id, _s(x)))
    return out


def conv2d_splitk(x, weight, bias=None, splits=2, relu=False, residual=None, in_coff=0, out=None, out_coff=0, out_cs=None,
                  workspace=None):
    """3x3 stride-1 convolution as Winograd F(2x2,3x3) with the input channels split into `splits` slices that run as
    separate work items and meet in `workspace` (acrmi_conv2d_splitk; the small-batch form of the low-resolution HRNet
    branches).  x NHWC fp32 [B,H,W,cs]; weight [Cout, Cin, 3, 3] with Cin = splits * (a multiple of 32, >= 64).
    workspace: zeroed uint8 device tensor of conv2d_splitk_workspace() bytes (allocated here when None)."""
    _need_cuda(x, residual, out, workspace)
    w = weight.detach().cpu().numpy() if hasattr(weight, 'detach') else np.asarray(weight)
    cout, cin, k, _ = w.shape
    if k != 3 or cin % splits:
        raise ValueError('split-K needs a 3x3 convolution whose Cin is a multiple of the slice count')
    ks = cin // splits
    b = np.zeros(cout, np.float32) if bias is None else (
        bias.detach().cpu().numpy() if hasattr(bias, 'detach') else np.asarray(bias))
    packed = [pack_conv(winograd2d_weights(w[:, s * ks:(s + 1) * ks].astype(np.float64)), b) for s in range(splits)]
    wp = torch.from_numpy(np.concatenate([p[0] for p in packed])).to(x.device)
    bp = torch.from_numpy(np.concatenate([p[1] for p in packed])).to(x.device)
    B, H, W, cs = x.shape
    if out is None:
        out_cs = out_cs or (cout + 3) // 4 * 4
        out = torch.zeros(B, H, W, out_cs, dtype=torch.float32, device=x.device)
    L = _lib.lib()
    need = int(L.acrmi_conv2d_splitk_workspace(B, H, W, cout, splits))
    if workspace is None:
        workspace = torch.zeros(need, dtype=torch.uint8, device=x.device)
    _lib.check(L.acrmi_conv2d_splitk(_p(x), B, H, W, cs, in_coff, ks, splits, _p(wp), _p(bp), _p(residual),
                                     residual.shape[-1] if residual is not None else 0, 0, _p(out), out.shape[-1], out_coff,
                                     cout, int(relu), _p(workspace), workspace.numel(), _s(x)))
    return out


def to_nhwc16(x_nchw, precision='fp16', cs=None, device='cuda'):
    """[B,C,H,W] float (any device) -> contiguous NHWC float16 / bfloat16 on `device`, channel stride cs (a multiple
    of 8, zero padded): the activation layout of a 16-bit program."""
    B, Cc, H, W = x_nchw.shape
    cs = cs or (Cc + 7) // 8 * 8
    out = torch.zeros(B, H, W, cs, dtype=torch.float16 if precision == 'fp16' else torch.bfloat16, device=device)
    out[..., :Cc] = x_nchw.permute(0, 2, 3, 1).to(device).to(out.dtype)
    return out


def conv2d_h16(x, weight, bias=None, stride=1, relu=False, residual=None, groups=1, cin=None, in_coff=0, out=None,
               out_coff=0, out_cs=None, frame_bias=None, out_f32=False):
    """The convolution of a 16-bit program (acrmi_conv2d_h16): x NHWC float16 / bfloat16 [B,H,W,cs] on the device,
    weight [Cout_total, Cin/groups, k, k] and bias as float arrays (rounded once to x's type by the packer / kept fp32),
    fp32 accumulation, one rounding of the output.  out_f32: fp32 output and residual (the head exits)."""
    _need_cuda(x, residual, out, frame_bias)
    if x.dtype not in (torch.float16, torch.bfloat16):
        raise ValueError('x must be float16 or bfloat16')
    dt = DT_F16 if x.dtype == torch.float16 else DT_BF16
    w = weight.detach().cpu().numpy() if hasattr(weight, 'detach') else np.asarray(weight)
    cout_t, cin_g, k, _ = w.shape
    cout = cout_t // groups
    b = np.zeros(cout_t, np.float32) if bias is None else (
        bias.detach().cpu().numpy() if hasattr(bias, 'detach') else np.asarray(bias))
    packed = [pack_conv_h16(w[g * cout:(g + 1) * cout].astype(np.float64), b[g * cout:(g + 1) * cout], dt) for g in range(groups)]
    wp = torch.from_numpy(np.concatenate([p[0] for p in packed]).view(np.int16)).to(x.device)
    bp = torch.from_numpy(np.concatenate([p[1] for p in packed])).to(x.device)
    B, H, W, cs = x.shape
    cin = cin_g if cin is None else cin
    pad = k // 2
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    odt = torch.float32 if out_f32 else x.dtype
    if out is None:
        q = 4 if out_f32 else 8
        out_cs = out_cs or (cout_t + q - 1) // q * q
        out = torch.zeros(B, Ho, Wo, out_cs, dtype=odt, device=x.device)
    if out.dtype != odt or (residual is not None and residual.dtype != odt):
        raise ValueError('out / residual must be %s' % odt)
    bias_t, fstride = bp, 0
    if frame_bias is not None:
        bias_t, fstride = frame_bias.contiguous().float(), frame_bias.shape[-1]
    _lib.check(_lib.lib().acrmi_conv2d_h16(_p(x), B, H, W, cs, in_coff, cin, _p(wp), _p(bias_t), fstride, _p(residual),
                                           residual.shape[-1] if residual is not None else 0, 0, _p(out), out.shape[-1],
                                           out_coff, cout, k, stride, int(relu), groups, dt, int(bool(out_f32)), _s(x)))
    return out


def preprocess(bgr_frames):
    """uint8 BGR device frames [n,H,W,3] (all the same size) -> (uint8 RGB [n,512,512,3] device, offsets [n,10] host).
    The device counterpart of acr.utils.img_preprocess (reference acr/utils.py:1315-1337)."""
    _need_cuda(bgr_frames)
    if bgr_frames.dtype != torch.uint8 or bgr_frames.dim() != 4 or bgr_frames.shape[-1] != 3:
        raise ValueError('frames must be uint8 [n,H,W,3] BGR')
    n, H, W, _ = bgr_frames.shape
    out = torch.empty(n, 512, 512, 3, dtype=torch.uint8, device=bgr_frames.device)
    offsets = np.zeros((n, 10), np.float32)
    src = bgr_frames.contiguous()          # bound to a local until the call has been queued
    _lib.check(_lib.lib().acrmi_preprocess(_p(src), n, H, W, _p(out), offsets.ctypes.data_as(C.c_void_p), _s(src)))
    return out, torch.from_numpy(offsets)


def preprocess_frames(bgr_frames):
    """A list of uint8 BGR device frames [H_i,W_i,3] of ANY sizes -> (uint8 RGB [n,512,512,3] device, offsets [n,10] host) in
    one call (acrmi_preprocess_frames: per-frame geometry in the kernel arguments).  img_preprocess is per image on the
    reference (acr/utils.py:1315-1337); folder mode mixes sizes (acr/main.py:144-205)."""
    frames = list(bgr_frames)
    if not frames:
        raise ValueError('no frames')
    _need_cuda(*frames)
    keep = []
    arr = (_lib.Frame * len(frames))()
    for i, f in enumerate(frames):
        if f.dtype != torch.uint8 or f.dim() != 3 or f.shape[-1] != 3 or f.device != frames[0].device:
            raise ValueError('frames must be uint8 [H,W,3] BGR tensors on one device')
        f = f.contiguous()
        keep.append(f)                     # bound until the call has been queued
        arr[i].bgr_dev, arr[i].H, arr[i].W = f.data_ptr(), f.shape[0], f.shape[1]
    n = len(frames)
    out = torch.empty(n, 512, 512, 3, dtype=torch.uint8, device=frames[0].device)
    offsets = np.zeros((n, 10), np.float32)
    _lib.check(_lib.lib().acrmi_preprocess_frames(arr, n, _p(out), offsets.ctypes.data_as(C.c_void_p), _s(out)))
    return out, torch.from_numpy(offsets)


def cam_trans(joints, pj2d, focal_length=600.0, img_size=512.0):
    """joints [n,21,3], pj2d [n,21,2] (device fp32) -> cam_trans [n,3]: the reference's closed-form least squares
    (acr/utils.py:430-472, unit confidences) on the device."""
    _need_cuda(joints, pj2d)
    n = joints.shape[0]
    if tuple(joints.shape[1:]) != (21, 3) or tuple(pj2d.shape) != (n, 21, 2):
        raise ValueError('joints must be [n,21,3] and pj2d [n,21,2]')
    out = torch.empty(n, 3, dtype=torch.float32, device=joints.device)
    j, p = joints.contiguous().float(), pj2d.contiguous().float()
    _lib.check(_lib.lib().acrmi_cam_trans(_p(j), _p(p), n, float(focal_length), float(img_size), _p(out), _s(joints)))
    return out


def u8norm(img):
    _need_cuda(img)
    B, H, W, _ = img.shape
    out = torch.empty(B, H, W, 4, dtype=torch.float32, device=img.device)
    src = img.contiguous()                 # bound to a local until the call has been queued
    _lib.check(_lib.lib().acrmi_u8norm(_p(src), B * H * W, _p(out), _s(src)))
    return out


def stem_conv(img, w, b, relu=True):
    """uint8 RGB [B,H,W,3] on the device + conv1 filters [64,3,3,3] / bias [64] (host arrays, BN folded) ->
    relu(conv3x3 stride 2 pad 1 of (img/255*2-1)) NHWC [B,H/2,W/2,64] (acr/model.py:832,589-603) in one kernel."""
    _need_cuda(img)
    B, H, W, _ = img.shape
    wp, bp = pack_stem(np.asarray(w, np.float64), np.asarray(b, np.float64))
    wd, bd = torch.from_numpy(wp).to(img.device), torch.from_numpy(bp).to(img.device)
    out = torch.empty(B, H // 2, W // 2, 64, dtype=torch.float32, device=img.device)
    src = img.contiguous()
    _lib.check(_lib.lib().acrmi_stem_conv(_p(src), B, H, W, _p(wd), _p(bd), _p(out), 64, 0, int(relu), _s(src)))
    return out


def bilinear2x(x, channels=None):
    _need_cuda(x)
    B, H, W, cs = x.shape
    Cc = channels or cs
    out = torch.zeros(B, 2 * H, 2 * W, Cc, dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().acrmi_bilinear2x(_p(x), B, H, W, cs, Cc, _p(out), Cc, _s(x)))
    return out


def fuse_sum(terms, shifts, relu=True):
    _need_cuda(*terms)
    B, H, W, Cc = terms[0].shape
    n = len(terms)
    out = torch.empty(B, H, W, Cc, dtype=torch.float32, device=terms[0].device)
    ptrs = (C.c_void_p * n)(*[t.data_ptr() for t in terms])
    cs = (C.c_int * n)(*[t.shape[-1] for t in terms])
    sh = (C.c_int * n)(*shifts)
    _lib.check(_lib.lib().acrmi_fuse_sum(n, ptrs, cs, sh, B, H, W, Cc, _p(out), Cc, int(relu), _s(out)))
    return out


def attpool(segm, feat, channels):
    """segm NHWC [B,256,256,cs] logits, feat NHWC [B,128,128,cs]; -> pooled [B,32,channels]."""
    _need_cuda(segm, feat)
    B = segm.shape[0]
    ws = torch.empty(int(_lib.lib().acrmi_attpool_ws_floats(B, channels)), dtype=torch.float32, device=segm.device)
    pooled = torch.empty(B, 32, channels, dtype=torch.float32, device=segm.device)
    _lib.check(_lib.lib().acrmi_attpool(_p(segm), segm.shape[-1], _p(feat), feat.shape[-1], channels, B, _p(ws),
                                        _p(pooled), _s(segm)))
    return pooled


def parebias(pooled, lc_w, lin_w, lin_b, mix_wp, mix_b, part0):
    """pooled [B,32,C] device (C = 320: contact 256 | shape 64) + the reference's raw weights (host arrays:
    contact_layers.{2,3}.weight -> [6,256,16], cam_shape_layers.{2,3} -> [10,1024] / [10], the pare columns of
    contact_layers.{4,5} -> [109,106] / bias [109]) -> per-frame mix-conv bias [B,112] (acr/model.py:141-164)."""
    _need_cuda(pooled)
    B, _, Cc = pooled.shape
    dev = pooled.device
    w = [torch.as_tensor(np.ascontiguousarray(np.asarray(a, np.float32))).to(dev) for a in (lc_w, lin_w, lin_b, mix_wp, mix_b)]
    out = torch.zeros(B, 112, dtype=torch.float32, device=dev)
    src = pooled.contiguous().float()
    _lib.check(_lib.lib().acrmi_parebias(_p(src), Cc, int(part0), _p(w[0]), _p(w[1]), _p(w[2]), _p(w[3]), _p(w[4]), B,
                                         _p(out), 112, _s(src)))
    return out


def decode_maps(l_center, r_center, l_params, r_params, l_prior, r_prior, conf_thresh=0.35, prior_gate=None):
    """NHWC device maps -> slots [B,2,176].  conf_thresh = args().centermap_conf_thresh (strict >).
    prior_gate: int32 device tensor [B] (acrmi_decode_maps_gated) or None = the per-frame prior rule."""
    _need_cuda(l_center, r_center, l_params, r_params, l_prior, r_prior, prior_gate)
    B = l_center.shape[0]
    slots = torch.empty(B, 2, _lib.SLOT, dtype=torch.float32, device=l_center.device)
    gate = None if prior_gate is None else prior_gate.to(torch.int32).contiguous()
    if gate is not None and gate.numel() != B:
        raise ValueError('prior_gate must hold one int per frame')
    _lib.check(_lib.lib().acrmi_decode_maps_gated(_p(l_center), _p(r_center), l_center.shape[-1], _p(l_params), _p(r_params),
                                                  l_params.shape[-1], _p(l_prior), _p(r_prior), l_prior.shape[-1], B,
                                                  float(conf_thresh), _p(gate), _p(slots), _s(slots)))
    return slots


def prior_gate(slots):
    """acrmi_prior_gate (stand-alone form): slots [B,2,176] of a first decode -> int32 [B] for decode_maps(prior_gate=...):
    the reference's batch-wide prior decision (acr/result_parser.py:42-47,102-145), computed on the device."""
    _need_cuda(slots)
    if slots.dtype != torch.float32 or not slots.is_contiguous():
        raise ValueError('slots must be a contiguous float32 device tensor')
    gate = torch.empty(slots.shape[0], dtype=torch.int32, device=slots.device)
    _lib.check(_lib.lib().acrmi_prior_gate(None, _p(slots), slots.shape[0], _p(gate), _s(slots)))
    return gate
