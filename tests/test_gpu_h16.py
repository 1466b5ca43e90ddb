"""GPU parity of the 16-bit path (the reference's --model_precision fp16 branch, acr/model.py:18-19,33-37, and its bf16
twin) and of the HRNet-W48 variant (BASELINE.json configs[4]).

NO REFERENCE ORACLE exists for either: autocast is CUDA-only and the reference hard-wires HRNet-W32.
What is checked instead:
  * kernel level: every 16-bit convolution against an exact fp64 convolution of the 16-bit inputs - the result must be
    the correctly rounded value up to the fp32 accumulation error (ONE rounding per layer);
  * program level: the resident 16-bit program against oracle/program.py (the op-list interpreter, itself pinned on
    fp32 programs to the reference-pinned oracle/acr_net.py), a few 16-bit ulps of drift allowed;
  * against the fp32 reference fixtures: the 16-bit error is REPORTED (vertices / joints / maps), with loose bounds.
Run on the MI355X box with `pytest -m gpu`."""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import cases
from conftest import ROOT, golden, pkg
from oracle import acr_net, decode as odec, mano as omano, program as oprog

pytestmark = pytest.mark.gpu

TD = {'fp16': torch.float16, 'bf16': torch.bfloat16}
MANT = {'fp16': 10, 'bf16': 7}


def ulp(v, precision):
    """Spacing of the storage type at |v| (float64 tensor)."""
    e = torch.floor(torch.log2(v.abs().clamp_min(1e-30)))
    if precision == 'fp16':
        e = e.clamp_min(-14.0)             # f16 subnormals: fixed spacing 2^-24
    return torch.pow(2.0, e - MANT[precision])


@pytest.fixture(scope='module')
def ops():
    assert torch.cuda.is_available(), 'gpu tests need a GPU'
    return pkg('ops')


H16_CASES = [
    # (B, Cin, Cout, H, W, k, stride, groups, relu, residual, out_f32)
    (2, 32, 32, 128, 128, 3, 1, 1, True, True, False),      # branch-0 BasicBlock conv (one chunk, 16x16 tiles)
    (2, 64, 64, 64, 64, 3, 1, 1, True, True, False),        # branch-1 (two n-tiles per wave)
    (3, 128, 128, 32, 32, 3, 1, 1, True, False, False),     # branch-2 (chunks + n-blocks)
    (3, 256, 256, 16, 16, 3, 1, 1, False, True, False),     # branch-3 (small map)
    (2, 48, 48, 64, 64, 3, 1, 1, True, True, False),        # HRNet-W48 branch 0: ragged second n-tile (octets)
    (2, 96, 192, 32, 32, 3, 2, 1, True, False, False),      # W48 downsample
    (2, 64, 64, 64, 64, 3, 2, 1, True, False, False),       # stem conv2 / fuse downsample
    (2, 32, 128, 64, 64, 3, 2, 1, False, False, False),
    (2, 34, 256, 32, 32, 3, 1, 1, True, False, False),      # contact_layers.1.0 (Cin = 34: ragged element pair)
    (2, 34, 512, 64, 64, 3, 2, 1, True, False, False),      # towers entry
    (1, 33, 33, 48, 48, 3, 1, 1, False, False, True),       # last segm conv: fp32 logits, ragged both
    (1, 16, 64, 32, 32, 3, 1, 1, True, False, False),
    (1, 32, 32, 256, 256, 3, 1, 1, True, False, False),     # segm head first conv (padded 16 -> 32)
    (2, 64, 256, 32, 32, 1, 1, 1, True, True, False),       # bottleneck conv3 (+ residual), n-blocks from one patch
    (2, 256, 64, 32, 32, 1, 1, 1, True, False, False),
    (2, 128, 32, 16, 16, 1, 1, 1, False, False, False),     # fuse 1x1
    (2, 64, 128, 64, 64, 1, 1, 1, False, False, True),      # tower exit (params x mix): fp32 out
    (2, 64, 32, 64, 64, 1, 1, 1, False, False, True),       # center exit: fp32 out
    (2, 3, 128, 64, 64, 1, 1, 1, False, True, True),        # cam mix: Cin = 3, fp32 residual accumulated in place
    (2, 512, 512, 32, 32, 3, 1, 8, True, True, False),      # 8 grouped head towers
    (1, 32, 32, 19, 23, 3, 1, 1, True, False, False),       # ragged spatial size (tile bounds)
    (1, 40, 72, 21, 17, 3, 2, 1, False, False, False),      # ragged + stride 2
    (1, 24, 48, 9, 31, 1, 1, 1, True, False, False),
]


@pytest.mark.parametrize('precision', ['fp16', 'bf16'])
@pytest.mark.parametrize('case', H16_CASES, ids=lambda c: 'B%d_%dto%d_%dx%d_k%ds%dg%d%s' % (c[:8] + ('_f32' if c[10] else '',)))
def test_conv2d_h16_is_the_correctly_rounded_convolution(ops, case, precision):
    B, cin, cout, H, W, k, stride, groups, relu, use_res, out_f32 = case
    g = torch.Generator().manual_seed((hash(case) + len(precision)) % (2 ** 31))
    td = TD[precision]
    x = torch.randn(B, cin, H, W, generator=g).to(td)
    w = (torch.randn(cout, cin // groups, k, k, generator=g) / np.sqrt(cin // groups * k * k)).to(td)
    b = torch.randn(cout, generator=g) * 0.1
    ref = F.conv2d(x.double(), w.double(), None, stride, k // 2, 1, groups)
    mag = F.conv2d(x.double().abs(), w.double().abs(), None, stride, k // 2, 1, groups)     # sum of |terms|
    ref = (ref + b.double()[None, :, None, None]).float().double()                          # fp32 bias add
    res = None
    if use_res:
        res = torch.randn(ref.shape, generator=g).to(torch.float32 if out_f32 else td)
        ref = ref + res.double()
    if relu:
        ref = F.relu(ref)
    xd = ops.to_nhwc16(x, precision)
    rd = None
    if res is not None:
        rd = ops.to_nhwc(res) if out_f32 else ops.to_nhwc16(res, precision)
    out = ops.conv2d_h16(xd, w.float(), b, stride=stride, relu=relu, groups=groups, cin=cin // groups, residual=rd,
                         out_f32=out_f32)
    torch.cuda.synchronize()
    got = out[..., :cout].permute(0, 3, 1, 2).cpu().double()
    err = (got - ref).abs()
    acc = 4e-7 * (mag + b.abs().double()[None, :, None, None] + 1.0)      # fp32 accumulation over K <= 2304 terms
    if out_f32:
        bound = acc + 2e-7 * ref.abs()
    else:
        bound = 0.5 * ulp(ref, precision) + acc            # ONE rounding to the storage type
    worst = (err - bound).max().item()
    assert worst <= 0, (worst, err.max().item())
    if out.shape[-1] > cout:
        assert out[..., cout:].float().abs().max().item() == 0.0     # pad channels untouched


def test_conv2d_h16_per_frame_bias(ops):
    """The mix conv's per-frame bias row (acr/model.py:160-164 pare columns) on the fp32-output kernel."""
    g = torch.Generator().manual_seed(5)
    x = torch.randn(3, 3, 64, 64, generator=g).to(torch.float16)
    w = torch.randn(109, 3, 1, 1, generator=g).to(torch.float16)
    fb = torch.randn(3, 128, generator=g)
    res = torch.randn(3, 109, 64, 64, generator=g)
    ref = F.conv2d(x.double(), w.double()) + fb[:, :109].double()[:, :, None, None] + res.double()
    wp = torch.zeros(128, 3, 1, 1)
    wp[:109] = w.float()
    out = ops.conv2d_h16(ops.to_nhwc16(x), wp, None, frame_bias=fb.cuda(), residual=ops.to_nhwc(res, cs=128), out_f32=True)
    torch.cuda.synchronize()
    assert (out[..., :109].permute(0, 3, 1, 2).cpu().double() - ref).abs().max().item() < 2e-5
