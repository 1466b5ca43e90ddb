"""ORACLE (test infrastructure, not product): CPU interpreter of a lowered op program (packer.lower).

What it is for: the 16-bit programs (the reference's autocast branch, acr/model.py:33-37, re-stated for gfx950 in
packer.lower) and the HRNet-W48 programs (BASELINE.json configs[4]) have NO reference oracle - autocast is CUDA-only
 and the reference hard-wires HRNet-W32.  This module restates the semantics the C library implements,
op by op, in torch-CPU arithmetic on the un-packed folded filters the packer keeps with keep_weights=True:

  * a 16-bit buffer holds values of its storage type (f16 / bf16); every op computes in float64 / float32 on those
    values and rounds its result ONCE (nearest even) when it writes a 16-bit buffer;
  * CONV: filters rounded to the storage type (fp32 programs: to fp32), exact products, float64 accumulation
    (the kernels accumulate in fp32: the difference is ~1e-7 relative, far below one 16-bit ulp, but it can move a
    value across a rounding boundary - comparisons allow a few ulps of the storage type), + bias (fp32, shared or per
    frame) + residual, ReLU;
  * the element-wise ops follow the kernels' fp32 expressions term by term;
  * the lowering's own constructions are read back into the convolutions they stand for: a CONV with ACRMI_CONV_SPLITK is
    ONE convolution over its concatenated K-slices, ACRMI_CONV_BIAS_MAP adds the position-bias map of the blob to every
    frame, PAIR1X1 is two 1x1 convolutions (64 -> 256 + residual + ReLU, then 256 -> 64 + ReLU), MAXPOOL and the 7x7 stem
    belong to the build-defined ResNet-50 (oracle/acr_net.resnet50_backbone); a CONV with nterms > 0 adds that many extra
    residual maps (nearest-upsampled by 2^shift) before its ReLU: the HR-module fuse sum (acr/model.py:672-686) folded into
    the last convolution of the x0 downsampling chain.

For fp32 W32 programs the same interpreter is cross-checked against oracle/acr_net.py (pinned to the reference), which
pins the interpreter's reading of the op list; tests/test_program_oracle.py.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import numpy as np
import torch
import torch.nn.functional as F

OP_U8NORM, OP_CONV, OP_FUSESUM, OP_BILINEAR2X, OP_POW11, OP_ATTPOOL, OP_PAREBIAS, OP_COORDFILL, OP_POINTHEADS, OP_STEM = range(1, 11)
OP_MAXPOOL, OP_PAIR1X1 = 11, 12
MODE_POINT = 2
CONV_BIAS_MAP = 8      # acrmi_op.flags of a CONV (include/acrmi.h)
CONV_SPLITK = 16
CONV_DUAL = 32
DT_F32, DT_F16, DT_BF16 = 0, 1, 2


def rnd(x, dt):
    """float tensor -> values representable in storage type dt, as float32 (round to nearest even)."""
    x = x.to(torch.float32)
    if dt == DT_F16:
        return x.to(torch.float16).to(torch.float32)
    if dt == DT_BF16:
        return x.to(torch.bfloat16).to(torch.float32)
    return x


class Interp(object):
    def __init__(self, prog, B):
        self.prog, self.B = prog, B
        self.bufs = [torch.zeros(B, h, w, cs, dtype=torch.float32) for (h, w, cs, _p, _dt) in prog['bufs']]
        self.dts = [b[4] for b in prog['bufs']]
        self.blob = torch.from_numpy(np.asarray(prog['blob']))
        self.prog_dt = max(self.dts)

    def w(self, off, n):
        return self.blob[off:off + n].to(torch.float64)

    # ---- ops -------------------------------------------------------------------------------------------
    def conv(self, op, info):
        x = self.bufs[op.in_buf][..., op.in_coff:op.in_coff + op.groups * op.cin]
        wdt = self.dts[op.in_buf]                      # filters are stored in the input's type (fp32 programs: fp32)
        ws, bs = [], []
        for (w, b) in info['wb']:
            ws.append(rnd(torch.from_numpy(np.ascontiguousarray(w)), wdt).to(torch.float64))
            bs.append(torch.from_numpy(np.asarray(b, np.float32)).to(torch.float64))    # bias: fp32 in the blob
        splitk = bool(op.flags & CONV_SPLITK)          # the "groups" are K-slices of one convolution: same Cout
        if splitk:
            w, bs = torch.cat(ws, 1), bs[:1]
            y = F.conv2d(x.permute(0, 3, 1, 2).to(torch.float64), w, None, op.stride, op.ksize // 2, 1, 1)
        else:
            w = torch.cat(ws, 0)
            y = F.conv2d(x.permute(0, 3, 1, 2).to(torch.float64), w, None, op.stride, op.ksize // 2, 1, op.groups)
        y = y.permute(0, 2, 3, 1)
        if op.bias_per_frame:
            y = y + self.bufs[op.aux_buf][:, 0, 0, :y.shape[-1]].to(torch.float64)[:, None, None, :]
        else:
            y = y + torch.cat(bs)[None, None, None, :]
        y = y.to(torch.float32)                        # the kernels add bias / residual in fp32
        n = op.cout if splitk else op.groups * op.cout
        if op.flags & CONV_BIAS_MAP:                   # position-bias map in the blob: one map for every frame
            ho, wo, cs = y.shape[1], y.shape[2], (n + 3) // 4 * 4
            y = y + self.blob[op.w_off2:op.w_off2 + ho * wo * cs].view(1, ho, wo, cs)[..., :n]
        if op.res_buf >= 0:
            y = y + self.bufs[op.res_buf][..., op.res_coff:op.res_coff + n]
        dual = bool(op.flags & CONV_DUAL)              # the terms go into a SECOND output (aux_buf), not into this one
        if dual:
            if op.relu:
                y = torch.relu(y)
            self.bufs[op.out_buf][..., op.out_coff:op.out_coff + n] = rnd(y, self.dts[op.out_buf])
        for t in range(op.nterms):                     # extra residual terms: the HR fuse sum in this conv's epilogue
            v = self.bufs[op.term_buf[t]][..., op.term_coff[t]:op.term_coff[t] + n]
            sh = op.term_shift[t]
            if sh:
                v = v.repeat_interleave(1 << sh, 1).repeat_interleave(1 << sh, 2)    # nearest up (acr/model.py:639)
            y = y + v
        if dual:
            self.bufs[op.aux_buf][..., :n] = rnd(torch.relu(y), self.dts[op.aux_buf])
            return
        if op.relu:
            y = torch.relu(y)
        self.bufs[op.out_buf][..., op.out_coff:op.out_coff + n] = rnd(y, self.dts[op.out_buf])

    def stem(self, op, info, img):
        (w, b), = info['wb']
        x = (img.to(torch.float32) / 255.0) * 2.0 - 1.0                     # stem_kernel's table expression
        w = torch.from_numpy(np.asarray(w, np.float32)).to(torch.float64)   # pack_stem keeps fp32 filters
        y = F.conv2d(x.permute(0, 3, 1, 2).to(torch.float64), w, None, 2, op.ksize // 2).permute(0, 2, 3, 1)   # 3x3 / 7x7 (ResNet)
        y = (y + torch.from_numpy(np.asarray(b, np.float32)).to(torch.float64)).to(torch.float32)
        if op.relu:
            y = torch.relu(y)
        self.bufs[op.out_buf][..., op.out_coff:op.out_coff + 64] = rnd(y, self.dts[op.out_buf])

    def fuse_sum(self, op):
        acc = None
        for t in range(op.nterms):
            v = self.bufs[op.term_buf[t]][..., op.term_coff[t]:op.term_coff[t] + op.cout]
            sh = op.term_shift[t]
            if sh:
                v = v.repeat_interleave(1 << sh, 1).repeat_interleave(1 << sh, 2)    # nearest up (acr/model.py:639)
            acc = v.clone() if acc is None else acc + v
        if op.relu:
            acc = torch.relu(acc)
        self.bufs[op.out_buf][..., op.out_coff:op.out_coff + op.cout] = rnd(acc, self.dts[op.out_buf])

    def bilinear2x(self, op):
        x = self.bufs[op.in_buf][..., op.in_coff:op.in_coff + op.cin]
        B, H, W, C = x.shape
        Ho, Wo = 2 * H, 2 * W
        f32 = torch.float32
        sh = (torch.tensor(H - 1, dtype=f32) / torch.tensor(Ho - 1, dtype=f32))
        sw = (torch.tensor(W - 1, dtype=f32) / torch.tensor(Wo - 1, dtype=f32))
        fy = sh * torch.arange(Ho, dtype=f32)
        fx = sw * torch.arange(Wo, dtype=f32)
        y0, x0 = fy.to(torch.int64), fx.to(torch.int64)
        y1 = y0 + (y0 < H - 1).to(torch.int64)
        x1 = x0 + (x0 < W - 1).to(torch.int64)
        ly, lx = (fy - y0.to(f32)), (fx - x0.to(f32))
        hy, hx = 1.0 - ly, 1.0 - lx
        v00, v01 = x[:, y0][:, :, x0], x[:, y0][:, :, x1]
        v10, v11 = x[:, y1][:, :, x0], x[:, y1][:, :, x1]
        hx_, lx_ = hx[None, None, :, None], lx[None, None, :, None]
        hy_, ly_ = hy[None, :, None, None], ly[None, :, None, None]
        r = hy_ * (hx_ * v00 + lx_ * v01) + ly_ * (hx_ * v10 + lx_ * v11)
        self.bufs[op.out_buf][..., op.out_coff:op.out_coff + op.cin] = rnd(r, self.dts[op.out_buf])

    def pair1x1(self, op, info):
        """out = relu(W3 in + b3 + res), aux = relu(W1 out + b1) (csrc/pair1x1.hip; fp32 programs)"""
        (w3, b3), (w1, b1) = [(torch.from_numpy(np.asarray(w, np.float64)), torch.from_numpy(np.asarray(b, np.float64)))
                              for (w, b) in info['wb']]
        t2 = self.bufs[op.in_buf][..., op.in_coff:op.in_coff + 64].to(torch.float64)
        y = torch.einsum('bhwc,oc->bhwo', t2, w3.reshape(256, 64)) + b3
        y = torch.relu(y.to(torch.float32) + self.bufs[op.res_buf][..., op.res_coff:op.res_coff + 256])
        self.bufs[op.out_buf][..., op.out_coff:op.out_coff + 256] = y
        t = torch.einsum('bhwc,oc->bhwo', y.to(torch.float64), w1.reshape(64, 256)) + b1
        self.bufs[op.aux_buf][..., :64] = torch.relu(t.to(torch.float32))

    def maxpool(self, op):
        x = self.bufs[op.in_buf][..., op.in_coff:op.in_coff + op.cin]
        y = F.max_pool2d(x.permute(0, 3, 1, 2), 3, 2, 1).permute(0, 2, 3, 1)      # exact in every storage type
        self.bufs[op.out_buf][..., op.out_coff:op.out_coff + op.cin] = y

    def pow11(self, op):
        b = self.bufs[op.out_buf]
        b[..., op.out_coff] = rnd(torch.pow(torch.tensor(1.1, dtype=torch.float32), b[..., op.out_coff]), self.dts[op.out_buf])

    def coordfill(self, op):
        b = self.bufs[op.out_buf]
        _, H, W, _ = b.shape
        f32 = torch.float32
        xs = (torch.arange(W, dtype=f32) / torch.tensor(W - 1, dtype=f32)) * 2.0 - 1.0
        ys = (torch.arange(H, dtype=f32) / torch.tensor(H - 1, dtype=f32)) * 2.0 - 1.0
        b[..., op.out_coff] = rnd(xs[None, None, :].expand(b.shape[0], H, W), self.dts[op.out_buf])
        b[..., op.out_coff + 1] = rnd(ys[None, :, None].expand(b.shape[0], H, W), self.dts[op.out_buf])

    def attpool(self, op):
        segm = self.bufs[op.in_buf]
        feat = self.bufs[op.res_buf][..., op.res_coff:op.res_coff + op.cin]
        B = segm.shape[0]
        logits = segm[:, ::2, ::2, 1:33].reshape(B, -1, 32).to(torch.float64)        # [B, pix, part]
        wts = torch.softmax(logits, 1)
        pooled = torch.einsum('bpj,bpc->bjc', wts, feat.reshape(B, -1, op.cin).to(torch.float64))
        self.bufs[op.out_buf].view(B, -1)[:, :32 * op.cin] = pooled.reshape(B, -1).to(torch.float32)

    def parebias(self, op):
        C, part0 = op.cin, op.flags
        B = self.B
        pooled = self.bufs[op.in_buf].view(B, -1)[:, :32 * C].reshape(B, 32, C).to(torch.float64)[:, part0:part0 + 16]   # [B,16,C]
        lc = self.w(op.w_off, 6 * 256 * 16).view(6, 256, 16)
        nsh = (64 if C == 320 else 256)
        lin_w = self.w(op.w_off2, 10 * nsh * 16).view(10, nsh, 16)
        lin_b = self.w(op.b_off2, 10)
        mix_wp = self.w(op.w_off3, 109 * 106).view(109, 106)
        mix_b = self.w(op.b_off, 109)
        off = torch.einsum('ocj,bjc->bjo', lc, pooled[:, :, :256]).reshape(B, 96)
        shp = lin_b + torch.einsum('kcj,bjc->bk', lin_w, pooled[:, :, C - nsh:] if C == 320 else pooled)
        pare = torch.cat([off, shp], 1)
        out = mix_b + pare @ mix_wp.t()
        self.bufs[op.out_buf][:, 0, 0, :109] = out.to(torch.float32)

    # ---- driver ----------------------------------------------------------------------------------------
    def run(self, img_u8):
        """img_u8: torch uint8 [B,512,512,3].  Runs the dense variant of the program; returns self."""
        for op, info in zip(self.prog['ops'], self.prog['op_info']):
            if op.mode == MODE_POINT:
                continue
            k = op.kind
            if k == OP_CONV:
                self.conv(op, info)
            elif k == OP_STEM:
                self.stem(op, info, img_u8)
            elif k == OP_FUSESUM:
                self.fuse_sum(op)
            elif k == OP_BILINEAR2X:
                self.bilinear2x(op)
            elif k == OP_MAXPOOL:
                self.maxpool(op)
            elif k == OP_PAIR1X1:
                self.pair1x1(op, info)
            elif k == OP_POW11:
                self.pow11(op)
            elif k == OP_COORDFILL:
                self.coordfill(op)
            elif k == OP_ATTPOOL:
                self.attpool(op)
            elif k == OP_PAREBIAS:
                self.parebias(op)
            else:
                raise ValueError('op kind %d is not part of the dense program' % k)
        return self

    def head_maps(self):
        """The reference's H11 dict (NCHW float32) from the interpreted head buffers."""
        hl = self.prog['heads']
        out = {}
        for si, side in enumerate('lr'):
            out[side + '_params_maps'] = self.bufs[hl.params_buf[si]][..., :109]
            out[side + '_center_map'] = self.bufs[hl.center_buf[si]][..., :1]
            out[side + '_prior_maps'] = self.bufs[hl.prior_buf[si]][..., :106]
        out['segms'] = self.bufs[hl.segm_buf][..., :33]
        return {k: v.permute(0, 3, 1, 2).contiguous() for k, v in out.items()}


@torch.no_grad()
def run_program(prog, img_u8):
    """prog = packer.lower(sd, keep_weights=True, ...); img_u8 uint8 [B,512,512,3] -> Interp (buffers + head_maps())."""
    if 'wb' not in next(i for i in prog['op_info'] if i['kind'] == OP_CONV):
        raise ValueError('lower the checkpoint with keep_weights=True')
    return Interp(prog, img_u8.shape[0]).run(img_u8)
