"""GPU parity: every HIP kernel (through the C ABI) vs the CPU oracle / torch fp32 reference op.
Run on the MI355X box with `pytest -m gpu`."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import cases
from conftest import golden, pkg
from oracle import acr_net, decode as odec, mano as omano

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ops():
    assert torch.cuda.is_available(), 'gpu tests need a GPU'
    return pkg('ops')


def _nchw(x_nhwc, c):
    return x_nhwc[..., :c].permute(0, 3, 1, 2).cpu()


CONV_CASES = [
    # (B, Cin, Cout, H, W, k, stride, groups, relu, residual)
    (2, 32, 32, 128, 128, 3, 1, 1, True, True),      # branch-0 BasicBlock conv
    (2, 64, 64, 64, 64, 3, 1, 1, True, True),        # branch-1
    (3, 128, 128, 32, 32, 3, 1, 1, True, False),     # branch-2
    (3, 256, 256, 16, 16, 3, 1, 1, False, True),     # branch-3 (small-frame tile)
    (1, 3, 64, 64, 64, 3, 2, 1, True, False),        # stem conv1 (padded 4-channel input)
    (2, 64, 64, 64, 64, 3, 2, 1, True, False),       # stem conv2 / fuse downsample
    (2, 32, 128, 64, 64, 3, 2, 1, False, False),
    (2, 34, 256, 32, 32, 3, 1, 1, True, False),      # contact_layers.1.0 (odd Cin)
    (1, 33, 33, 48, 48, 3, 1, 1, False, False),      # last segm conv (odd both)
    (1, 16, 64, 32, 32, 3, 1, 1, True, False),
    (2, 64, 256, 32, 32, 1, 1, 1, False, True),      # bottleneck conv3
    (2, 256, 64, 32, 32, 1, 1, 1, True, False),
    (2, 128, 32, 16, 16, 1, 1, 1, False, False),     # fuse 1x1
    (2, 64, 106, 16, 16, 1, 1, 1, False, False),     # tower exit
    (2, 512, 512, 32, 32, 3, 1, 8, True, True),      # 8 grouped head towers
    (1, 32, 32, 19, 23, 3, 1, 1, True, False),       # ragged spatial size (tile bounds)
    (1, 40, 70, 21, 17, 3, 2, 1, False, False),      # ragged + stride 2
    (1, 24, 48, 9, 31, 1, 1, 1, True, False),
    (2, 256, 512, 32, 32, 1, 2, 1, False, False),    # ResNet-50 projection shortcut: 1x1 stride 2
    (2, 64, 32, 16, 32, 1, 2, 1, True, False),       # ... one 32-cout tile
    (1, 40, 70, 21, 17, 1, 2, 1, False, True),       # ... ragged, with a residual
]


@pytest.mark.parametrize('case', CONV_CASES, ids=lambda c: 'B%d_%dto%d_%dx%d_k%ds%dg%d' % c[:8])
def test_conv2d_matches_torch(ops, case):
    B, cin, cout, H, W, k, stride, groups, relu, use_res = case
    g = torch.Generator().manual_seed(hash(case) % (2 ** 31))
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin // groups, k, k, generator=g) / np.sqrt(cin // groups * k * k)
    b = torch.randn(cout, generator=g) * 0.1
    ref = F.conv2d(x.double(), w.double(), b.double(), stride, k // 2, 1, groups)
    res = None
    if use_res:
        res = torch.randn(ref.shape, generator=g)
        ref = ref + res.double()
    if relu:
        ref = F.relu(ref)
    out = ops.conv2d(ops.to_nhwc(x), w, b, stride=stride, relu=relu, groups=groups, cin=cin // groups,
                     residual=None if res is None else ops.to_nhwc(res))
    torch.cuda.synchronize()
    got = _nchw(out, cout)
    err = (got.double() - ref).abs().max().item()
    assert err < 2e-5, err          # fp32 accumulate over K <= 2304, values O(1)
    if out.shape[-1] > cout:
        assert out[..., cout:].abs().max().item() == 0.0     # pad channels untouched


WINO_CASES = [c for c in CONV_CASES if c[5] == 3 and c[6] == 1] + [
    (1, 32, 32, 16, 18, 3, 1, 1, False, True),       # Wo not a multiple of the tile, even
    (1, 8, 40, 7, 9, 3, 1, 1, True, False),          # odd width: the last pair has one valid pixel
]


@pytest.mark.parametrize('case', WINO_CASES, ids=lambda c: 'wino_B%d_%dto%d_%dx%d_g%d' % (c[0], c[1], c[2], c[3], c[4], c[7]))
def test_conv2d_winograd_matches_torch(ops, case):
    """3x3 stride-1 convolutions through the Winograd F(2,3)-along-x kernel vs an fp64 direct convolution."""
    B, cin, cout, H, W, k, stride, groups, relu, use_res = case
    g = torch.Generator().manual_seed(hash(case) % (2 ** 31) + 1)
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin // groups, k, k, generator=g) / np.sqrt(cin // groups * k * k)
    b = torch.randn(cout, generator=g) * 0.1
    ref = F.conv2d(x.double(), w.double(), b.double(), 1, 1, 1, groups)
    res = None
    if use_res:
        res = torch.randn(ref.shape, generator=g)
        ref = ref + res.double()
    if relu:
        ref = F.relu(ref)
    out = ops.conv2d(ops.to_nhwc(x), w, b, relu=relu, groups=groups, cin=cin // groups, algo='winograd',
                     residual=None if res is None else ops.to_nhwc(res))
    torch.cuda.synchronize()
    err = (_nchw(out, cout).double() - ref).abs().max().item()
    assert err < 3e-5, err          # Winograd F(2,3) in fp32: ~2x the direct kernel's round-off
    if out.shape[-1] > cout:
        assert out[..., cout:].abs().max().item() == 0.0


WINO2D_CASES = WINO_CASES + [
    (1, 24, 32, 10, 36, 3, 1, 1, True, True),        # ragged tile rows/cols, one short chunk
    (2, 48, 96, 16, 16, 3, 1, 1, False, True),       # n_tiles = 4 with 96 couts (last n-tile empty), 1.5 chunks
    (1, 17, 5, 7, 9, 3, 1, 1, True, False),          # odd everything
    (3, 40, 40, 24, 32, 3, 1, 1, True, True),        # 2 chunks (32 + 8), several items per workgroup? no: 36 items
    (1, 96, 64, 8, 16, 3, 1, 1, False, False),       # one tile, three full chunks
    (2, 8, 24, 16, 32, 3, 1, 1, True, True),         # single one-step chunk per item, double-buffered exchange
    (2, 64, 33, 32, 32, 3, 1, 1, True, True),        # 33rd channel on 4x4x1 MFMAs (segm_net conv 0), with residual
    (1, 40, 33, 19, 23, 3, 1, 1, False, True),       # ... ragged tiles and a short last chunk
    (3, 96, 33, 8, 16, 3, 1, 1, True, False),        # ... three chunks, one tile per frame
]


@pytest.mark.parametrize('case', WINO2D_CASES, ids=lambda c: 'wino2d_B%d_%dto%d_%dx%d_g%d' % (c[0], c[1], c[2], c[3], c[4], c[7]))
def test_conv2d_winograd2d_matches_torch(ops, case):
    """3x3 stride-1 convolutions through the Winograd F(2x2,3x3) kernel vs an fp64 direct convolution."""
    B, cin, cout, H, W, k, stride, groups, relu, use_res = case
    g = torch.Generator().manual_seed(hash(case) % (2 ** 31) + 2)
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin // groups, k, k, generator=g) / np.sqrt(cin // groups * k * k)
    b = torch.randn(cout, generator=g) * 0.1
    ref = F.conv2d(x.double(), w.double(), b.double(), 1, 1, 1, groups)
    res = None
    if use_res:
        res = torch.randn(ref.shape, generator=g)
        ref = ref + res.double()
    if relu:
        ref = F.relu(ref)
    out = ops.conv2d(ops.to_nhwc(x), w, b, relu=relu, groups=groups, cin=cin // groups, algo='winograd2d',
                     residual=None if res is None else ops.to_nhwc(res))
    torch.cuda.synchronize()
    err = (_nchw(out, cout).double() - ref).abs().max().item()
    assert err < 5e-5, err          # Winograd F(2x2,3x3) in fp32: ~4x the direct kernel's round-off
    if out.shape[-1] > cout:
        assert out[..., cout:].abs().max().item() == 0.0


WINO24_CASES = [c for c in WINO2D_CASES if c[1] // c[7] > 32 and c[2] != 33] + [
    (2, 64, 64, 64, 64, 3, 1, 1, True, True),        # branch 1: 4 chunks of 16 channels, 2 n-tiles
    (2, 128, 128, 32, 32, 3, 1, 1, True, False),     # branch 2: one 32-pixel tile column, 8 chunks
    (1, 64, 64, 8, 32, 3, 1, 1, False, False),       # exactly two chunks, one tile
    (1, 40, 64, 16, 64, 3, 1, 1, True, True),        # 1.25 chunks: the last chunk is a single (tail) step
    (1, 72, 16, 11, 45, 3, 1, 1, True, True),        # 2.25 chunks, ragged tile, Cout < 32
    (3, 64, 96, 24, 96, 3, 1, 1, True, True),        # several items per workgroup with a tiny grid? 27 tiles x 3 n-tiles
    (2, 128, 128, 32, 32, 3, 1, 2, True, True),      # groups
]


@pytest.mark.parametrize('case', WINO24_CASES, ids=lambda c: 'wino24_B%d_%dto%d_%dx%d_g%d' % (c[0], c[1], c[2], c[3], c[4], c[7]))
def test_conv2d_winograd24_matches_torch(ops, case):
    """3x3 stride-1 convolutions through the Winograd F(2x4,3x3) kernel (F(2,3) along y, F(4,3) along x) vs an fp64
    direct convolution.  F(4,3) carries ~10x the round-off of F(2,3) in fp32."""
    B, cin, cout, H, W, k, stride, groups, relu, use_res = case
    g = torch.Generator().manual_seed(hash(case) % (2 ** 31) + 24)
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin // groups, k, k, generator=g) / np.sqrt(cin // groups * k * k)
    b = torch.randn(cout, generator=g) * 0.1
    ref = F.conv2d(x.double(), w.double(), b.double(), 1, 1, 1, groups)
    res = None
    if use_res:
        res = torch.randn(ref.shape, generator=g)
        ref = ref + res.double()
    if relu:
        ref = F.relu(ref)
    out = ops.conv2d(ops.to_nhwc(x), w, b, relu=relu, groups=groups, cin=cin // groups, algo='winograd24',
                     residual=None if res is None else ops.to_nhwc(res))
    torch.cuda.synchronize()
    err = (_nchw(out, cout).double() - ref).abs().max().item()
    assert err < 2e-4, err
    if out.shape[-1] > cout:
        assert out[..., cout:].abs().max().item() == 0.0


WINO24B_CASES = [
    # (B, Cin, Cout, H, W, groups, relu, residual kind: 0 none / 1 per frame / 2 one map for every frame, frame_bias)
    (2, 64, 64, 64, 64, 1, True, 1, False),        # HRNet branch 1: two chunks, all four border kinds
    (20, 64, 64, 64, 64, 1, True, 1, False),       # 320 items: more than one item per workgroup
    (3, 128, 128, 32, 32, 1, True, 0, False),      # branch 2: four chunks, two n-blocks, every tile touches left AND right
    (1, 512, 512, 64, 64, 8, True, 1, False),      # the eight head towers as groups
    (2, 96, 128, 8, 32, 1, False, 2, False),       # three chunks, one tile per frame (all four borders), no ReLU, map residual
    (2, 64, 64, 16, 96, 1, False, 0, True),        # interior tile columns, per-frame bias rows
    (1, 64, 192, 24, 64, 1, True, 1, True),        # three n-blocks
    (3, 256, 256, 16, 16, 1, True, 1, False),      # HRNet branch 3: the 16x16-pixel items, one per frame and n-block, 8 chunks
    (2, 64, 128, 32, 16, 1, True, 0, False),       # ... two tiles per frame (top / bottom borders differ)
    (2, 64, 64, 16, 48, 1, False, 2, True),        # ... three tile columns: an interior one; map residual + frame bias
    (3, 32, 256, 32, 64, 1, True, 2, False),       # single-chunk items (Cin = 32), two n-tiles per wave: the contact conv + its bias map
    (2, 32, 128, 16, 32, 1, False, 1, True),       # ... residual per frame, frame bias, no ReLU
    (2, 256, 32, 32, 64, 1, True, 1, False),       # one n-tile per wave (Cout = 32), 8 chunks
    (2, 64, 96, 16, 32, 1, True, 0, False),        # one n-tile per wave, three n-blocks
    (3, 32, 32, 24, 64, 1, True, 1, False),        # single chunk AND one n-tile (the shape class conv_wino3 takes in the program)
]


@pytest.mark.parametrize('case', WINO24B_CASES, ids=lambda c: 'wino24b_B%d_%dto%d_%dx%d_g%d_r%d_fb%d' % (c[:6] + (c[7], int(c[8]))))
def test_conv2d_winograd24_four_wave_frame(ops, case):
    """conv_wino24b_kernel (F(2x4,3x3) on the four-wave / 512-register frame: two n-tiles per wave, LDS-DMA patches
    issued by the compute waves, early chunk barrier) vs an fp64 direct convolution AND bit for bit vs conv_wino24_kernel
    (acrmi_tune cfg 840 keeps the old frame: same products in the same order).  Input, residual and output live in
    channel slices of wider buffers; the neighbour channels must stay untouched."""
    B, cin, cout, H, W, groups, relu, res_kind, use_fb = case
    L = pkg('_lib').lib()
    g = torch.Generator().manual_seed(hash(case) % (2 ** 31) + 240)
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin // groups, 3, 3, generator=g) / np.sqrt(cin // groups * 9)
    b = torch.randn(cout, generator=g) * 0.1
    fb = torch.randn(B, cout, generator=g) if use_fb else None
    ref = F.conv2d(x.double(), w.double(), None if use_fb else b.double(), 1, 1, 1, groups)
    if use_fb:
        ref = ref + fb[:, :, None, None].double()
    res = None
    if res_kind:
        res = torch.randn((B if res_kind == 1 else 1, cout, H, W), generator=g)
        ref = ref + res.double()
    if relu:
        ref = F.relu(ref)
    xin = torch.full((B, H, W, cin + 8), 3.0, device='cuda')            # input in channels 4.. of a wider buffer
    xin[..., 4:4 + cin] = x.permute(0, 2, 3, 1).cuda()
    outs = []
    # (conv_wino24_kernel needs two chunks per item: the single-chunk shapes are checked against the fp64 convolution only)
    for cfg in ((-1, 840) if cin // groups > 32 else (-1,)):
        dst = torch.full((B, H, W, cout + 16), 7.0, device='cuda')     # output into channels 8..
        L.acrmi_tune(0, cfg)
        try:
            ops.conv2d(xin, w, None if use_fb else b, relu=relu, groups=groups, cin=cin // groups, in_coff=4, algo='winograd24',
                       out=dst, out_coff=8, residual=None if res is None else ops.to_nhwc(res),
                       frame_bias=None if fb is None else fb.cuda())
            torch.cuda.synchronize()
        finally:
            L.acrmi_tune(0, -1)
        got = dst[..., 8:8 + cout].permute(0, 3, 1, 2).cpu()
        err = (got.double() - ref).abs().max().item()
        assert err < 2e-4, (cfg, err)
        assert (dst[..., :8] == 7).all() and (dst[..., 8 + cout:] == 7).all(), cfg
        outs.append(got)
    if len(outs) == 2:
        assert torch.equal(outs[0], outs[1]), (outs[0] - outs[1]).abs().max().item()


WINO24C_CASES = [c for c in WINO24B_CASES if c[1] // c[5] >= 64 and (c[2] // c[5]) % 64 == 0 and c[3] % 8 == 0 and c[4] % 32 == 0]


@pytest.mark.parametrize('case', WINO24C_CASES, ids=lambda c: 'wino24c_B%d_%dto%d_%dx%d_g%d_r%d_fb%d' % (c[:6] + (c[7], int(c[8]))))
def test_conv2d_winograd24_all_positions_frame(ops, case):
    """conv_wino24c_kernel (round 6; acrmi_tune cfg 842 - NOT the default, it measured 7 % slower): all 24 Winograd positions
    of a tile in one wave on v_mfma_f32_16x16x4_f32, transformed fragments exchanged through LDS once per 8-channel step, the
    whole output transform in registers.  Same products as conv_wino24b_kernel in another summation order: vs an fp64 direct
    convolution, and within fp32 round-off of the default kernel; channel slices of wider buffers, neighbours untouched."""
    B, cin, cout, H, W, groups, relu, res_kind, use_fb = case
    L = pkg('_lib').lib()
    g = torch.Generator().manual_seed(hash(case) % (2 ** 31) + 241)
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin // groups, 3, 3, generator=g) / np.sqrt(cin // groups * 9)
    b = torch.randn(cout, generator=g) * 0.1
    fb = torch.randn(B, cout, generator=g) if use_fb else None
    ref = F.conv2d(x.double(), w.double(), None if use_fb else b.double(), 1, 1, 1, groups)
    if use_fb:
        ref = ref + fb[:, :, None, None].double()
    res = None
    if res_kind:
        res = torch.randn((B if res_kind == 1 else 1, cout, H, W), generator=g)
        ref = ref + res.double()
    if relu:
        ref = F.relu(ref)
    xin = torch.full((B, H, W, cin + 8), 3.0, device='cuda')
    xin[..., 4:4 + cin] = x.permute(0, 2, 3, 1).cuda()
    outs = []
    for cfg in (842, -1):
        dst = torch.full((B, H, W, cout + 16), 7.0, device='cuda')
        L.acrmi_tune(0, cfg)
        try:
            ops.conv2d(xin, w, None if use_fb else b, relu=relu, groups=groups, cin=cin // groups, in_coff=4, algo='winograd24',
                       out=dst, out_coff=8, residual=None if res is None else ops.to_nhwc(res),
                       frame_bias=None if fb is None else fb.cuda())
            torch.cuda.synchronize()
        finally:
            L.acrmi_tune(0, -1)
        got = dst[..., 8:8 + cout].permute(0, 3, 1, 2).cpu()
        assert (got.double() - ref).abs().max().item() < 2e-4, cfg
        assert (dst[..., :8] == 7).all() and (dst[..., 8 + cout:] == 7).all(), cfg
        outs.append(got)
    assert (outs[0] - outs[1]).abs().max().item() < 2e-5          # another summation order, the same arithmetic


PP2_CASES = [
    # (B, Cin, Cout, H, W (INPUT map), groups, relu, residual kind: 0 none / 1 per frame / 2 one map for every frame, frame_bias)
    (2, 32, 64, 32, 64, 1, False, 0, False),       # HRNet fuse chain 32 -> 64: two chunks, 2x2 tiles (all border kinds of a stride-2 map)
    (20, 64, 64, 64, 64, 1, True, 0, False),       # more items than workgroups can hold at once
    (3, 32, 32, 48, 96, 1, True, 1, False),        # one n-tile per wave; interior tiles; residual per frame
    (2, 16, 64, 16, 32, 1, True, 0, False),        # single-chunk items (Cin = 16), one tile per frame
    (2, 48, 128, 16, 64, 1, False, 2, True),       # three chunks, two n-blocks, map residual + per-frame bias rows, no ReLU
    (1, 256, 64, 32, 32, 1, True, 1, False),       # 16 chunks (layer1 -> transition1 branch 1)
    (2, 64, 192, 32, 32, 2, True, 0, False),       # two groups of 32 -> 96 (one n-tile per wave, three n-blocks each)
    (1, 128, 256, 32, 64, 1, False, 0, False),     # transition3: 128 -> 256
]


@pytest.mark.parametrize('case', PP2_CASES, ids=lambda c: 'pp2_B%d_%dto%d_%dx%d_g%d_r%d_fb%d' % (c[:6] + (c[7], int(c[8]))))
def test_conv2d_stride2_polyphase(ops, case):
    """conv_pp2_kernel (3x3 stride 2 in polyphase form with F(2,2) on the two-tap phases, four-wave frame: 25 products per
    2x2 output block instead of 36) vs an fp64 direct convolution and vs the direct stride-2 kernel.  Input, residual and
    output live in channel slices of wider buffers; the neighbour channels must stay untouched."""
    B, cin, cout, H, W, groups, relu, res_kind, use_fb = case
    g = torch.Generator().manual_seed(hash(case) % (2 ** 31) + 250)
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin // groups, 3, 3, generator=g) / np.sqrt(cin // groups * 9)
    b = torch.randn(cout, generator=g) * 0.1
    fb = torch.randn(B, cout, generator=g) if use_fb else None
    ref = F.conv2d(x.double(), w.double(), None if use_fb else b.double(), 2, 1, 1, groups)
    if use_fb:
        ref = ref + fb[:, :, None, None].double()
    res = None
    if res_kind:
        res = torch.randn((B if res_kind == 1 else 1, cout, H // 2, W // 2), generator=g)
        ref = ref + res.double()
    if relu:
        ref = F.relu(ref)
    xin = torch.full((B, H, W, cin + 8), 3.0, device='cuda')            # input in channels 4.. of a wider buffer
    xin[..., 4:4 + cin] = x.permute(0, 2, 3, 1).cuda()
    errs = {}
    for algo in ('polyphase2', 'direct'):
        dst = torch.full((B, H // 2, W // 2, cout + 16), 7.0, device='cuda')     # output into channels 8..
        ops.conv2d(xin, w, None if use_fb else b, stride=2, relu=relu, groups=groups, cin=cin // groups, in_coff=4, algo=algo,
                   out=dst, out_coff=8, residual=None if res is None else ops.to_nhwc(res),
                   frame_bias=None if fb is None else fb.cuda())
        torch.cuda.synchronize()
        got = dst[..., 8:8 + cout].permute(0, 3, 1, 2).cpu()
        errs[algo] = (got.double() - ref).abs().max().item()
        assert (dst[..., :8] == 7).all() and (dst[..., 8 + cout:] == 7).all(), algo
    assert errs['polyphase2'] < 1e-4, errs
    assert errs['polyphase2'] < 4 * errs['direct'] + 1e-6, errs      # round-off of the same order as the direct kernel's


X3_CASES = [
    # (B, Cin, Cout, H, W, groups, relu, residual kind: 0 none / 1 per frame / 2 one map for every frame, frame_bias)
    (2, 64, 64, 64, 64, 1, True, 1, False),        # HRNet branch 1: two chunks, all four border kinds
    (20, 64, 64, 64, 64, 1, True, 1, False),       # 320 items: more than one item per workgroup
    (3, 128, 128, 32, 32, 1, True, 0, False),      # branch 2: four chunks, two n-blocks, every tile touches left AND right
    (1, 512, 512, 64, 64, 8, True, 1, False),      # the eight head towers as groups
    (2, 96, 128, 8, 32, 1, False, 2, False),       # three chunks, one tile per frame (all four borders), no ReLU, map residual
    (2, 64, 64, 16, 96, 1, False, 0, True),        # interior tile columns, per-frame bias rows
    (3, 32, 256, 32, 64, 1, True, 2, False),       # single-chunk items (Cin = 32): the contact conv + its bias map
    (2, 256, 32, 32, 64, 1, True, 1, False),       # one n-tile per wave (Cout = 32), 8 chunks
    (3, 32, 32, 24, 64, 1, True, 1, False),        # single chunk AND one n-tile (HRNet branch 0)
    (3, 256, 256, 16, 16, 1, True, 1, False),      # HRNet branch 3: the 16x16-pixel items, one per frame and n-block, 8 chunks
    (2, 64, 128, 32, 16, 1, True, 0, False),       # ... two tiles per frame (top / bottom borders differ)
    (2, 64, 32, 16, 48, 1, False, 2, True),        # ... three tile columns (an interior one), one n-tile per wave, map residual + frame bias
]


@pytest.mark.parametrize('case', X3_CASES, ids=lambda c: 'x3_B%d_%dto%d_%dx%d_g%d_r%d_fb%d' % (c[:6] + (c[7], int(c[8]))))
def test_conv2d_split_f16_operands(ops, case):
    """conv_x3_kernel (fp32 storage; every operand split into hi + lo f16, three products per MAC on
    v_mfma_f32_32x32x16_f16, fp32 accumulation) vs an fp64 direct convolution and vs the fp32 Winograd kernel on the same
    data: the round-off must be of the fp32 kernels' order.  Channel slices of wider buffers; neighbours untouched."""
    B, cin, cout, H, W, groups, relu, res_kind, use_fb = case
    g = torch.Generator().manual_seed(hash(case) % (2 ** 31) + 260)
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin // groups, 3, 3, generator=g) / np.sqrt(cin // groups * 9)
    b = torch.randn(cout, generator=g) * 0.1
    fb = torch.randn(B, cout, generator=g) if use_fb else None
    ref = F.conv2d(x.double(), w.double(), None if use_fb else b.double(), 1, 1, 1, groups)
    if use_fb:
        ref = ref + fb[:, :, None, None].double()
    res = None
    if res_kind:
        res = torch.randn((B if res_kind == 1 else 1, cout, H, W), generator=g)
        ref = ref + res.double()
    if relu:
        ref = F.relu(ref)
    xin = torch.full((B, H, W, cin + 8), 3.0, device='cuda')            # input in channels 4.. of a wider buffer
    xin[..., 4:4 + cin] = x.permute(0, 2, 3, 1).cuda()
    errs = {}
    for algo in ('split16', 'split_bf16', 'winograd2d'):
        dst = torch.full((B, H, W, cout + 16), 7.0, device='cuda')     # output into channels 8..
        ops.conv2d(xin, w, None if use_fb else b, relu=relu, groups=groups, cin=cin // groups, in_coff=4, algo=algo,
                   out=dst, out_coff=8, residual=None if res is None else ops.to_nhwc(res),
                   frame_bias=None if fb is None else fb.cuda())
        torch.cuda.synchronize()
        got = dst[..., 8:8 + cout].permute(0, 3, 1, 2).cpu()
        errs[algo] = (got.double() - ref).abs().max().item()
        assert (dst[..., :8] == 7).all() and (dst[..., 8 + cout:] == 7).all(), algo
    assert errs['split16'] < 2e-5, errs
    assert errs['split16'] < 4 * errs['winograd2d'] + 1e-6, errs
    assert errs['split_bf16'] < 5e-4, errs      # 16-bit operands: 2^-16 relative per product


X3S2_CASES = [
    # (B, Cin, Cout, H, W (input), groups, relu, residual kind: 0 none / 1 per frame / 2 one map for every frame, frame_bias)
    (2, 64, 64, 64, 128, 1, True, 0, False),       # stem conv2 shape: two chunks, two tile columns, four tile rows
    (20, 32, 64, 32, 64, 1, False, 0, False),      # fuse chain 32 -> 64: single-chunk items, 40 of them x 1 tile
    (3, 32, 32, 16, 64, 1, False, 0, False),       # fuse chain 32 -> 32: one n-tile per wave, one tile per frame (all borders)
    (2, 128, 256, 16, 64, 1, True, 1, False),      # transition3-like: four chunks, four n-blocks, per-frame residual
    (2, 32, 512, 32, 128, 1, True, 2, False),      # tower entry: eight n-blocks, the position-bias map as a residual for every frame
    (2, 64, 128, 48, 192, 1, False, 0, True),      # interior tile columns and rows, per-frame bias rows
    (1, 128, 192, 16, 64, 2, True, 1, False),      # two groups of 64 -> 96 (one n-tile per wave, three n-blocks each)
    (2, 256, 32, 16, 64, 1, True, 0, False),       # one n-tile per wave, eight chunks (the instantiation whose first f16 build - 584 bytes of scratch - summed wrongly)
    (3, 64, 32, 32, 128, 1, False, 1, False),      # ... two chunks, several tiles per frame, per-frame residual
    (5, 96, 64, 16, 64, 1, True, 0, False),        # three chunks (an odd count: a chunk = two 16-channel steps), two n-tiles per wave
]


@pytest.mark.parametrize('case', X3S2_CASES, ids=lambda c: 'x3s2_B%d_%dto%d_%dx%d_g%d_r%d_fb%d' % (c[:6] + (c[7], int(c[8]))))
def test_conv2d_stride2_split_f16_operands(ops, case):
    """conv_x3s2_kernel (3x3 stride 2, fp32 storage, operands split into hi + lo f16 / bf16, de-interleaved patch columns, one
    16-channel step per plane buffer) vs an fp64 direct convolution and vs the fp32 polyphase kernel on the same data.
    Channel slices of wider buffers; neighbours untouched."""
    B, cin, cout, H, W, groups, relu, res_kind, use_fb = case
    g = torch.Generator().manual_seed(hash(case) % (2 ** 31) + 270)
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin // groups, 3, 3, generator=g) / np.sqrt(cin // groups * 9)
    b = torch.randn(cout, generator=g) * 0.1
    fb = torch.randn(B, cout, generator=g) if use_fb else None
    ref = F.conv2d(x.double(), w.double(), None if use_fb else b.double(), 2, 1, 1, groups)
    if use_fb:
        ref = ref + fb[:, :, None, None].double()
    res = None
    if res_kind:
        res = torch.randn((B if res_kind == 1 else 1, cout, H // 2, W // 2), generator=g)
        ref = ref + res.double()
    if relu:
        ref = F.relu(ref)
    xin = torch.full((B, H, W, cin + 8), 3.0, device='cuda')            # input in channels 4.. of a wider buffer
    xin[..., 4:4 + cin] = x.permute(0, 2, 3, 1).cuda()
    errs = {}
    for algo in ('split16', 'split_bf16', 'polyphase2'):
        dst = torch.full((B, H // 2, W // 2, cout + 16), 7.0, device='cuda')     # output into channels 8..
        ops.conv2d(xin, w, None if use_fb else b, stride=2, relu=relu, groups=groups, cin=cin // groups, in_coff=4, algo=algo,
                   out=dst, out_coff=8, residual=None if res is None else ops.to_nhwc(res),
                   frame_bias=None if fb is None else fb.cuda())
        torch.cuda.synchronize()
        got = dst[..., 8:8 + cout].permute(0, 3, 1, 2).cpu()
        errs[algo] = (got.double() - ref).abs().max().item()
        assert (dst[..., :8] == 7).all() and (dst[..., 8 + cout:] == 7).all(), algo
    assert errs['split16'] < 2e-5, errs
    assert errs['split16'] < 4 * errs['polyphase2'] + 1e-6, errs
    assert errs['split_bf16'] < 5e-4, errs      # 16-bit operands: 2^-16 relative per product


X3P_CASES = [
    # (B, Cin, Cout, H, W, groups, relu, residual kind: 0 none / 1 per frame, frame_bias)
    (2, 64, 256, 16, 16, 1, True, 1, False),       # one item per frame, four n-blocks (layer1's conv3: residual + ReLU)
    (20, 256, 64, 32, 32, 1, True, 0, False),      # 8 chunks; more items than workgroups can hold at once
    (3, 32, 32, 16, 32, 1, False, 0, True),        # single-chunk items, one n-tile per wave, per-frame bias rows
    (2, 96, 128, 16, 16, 1, True, 1, False),       # three chunks
    (1, 128, 192, 32, 64, 2, False, 0, False),     # two groups of 64 -> 96 (one n-tile per wave, three n-blocks each)
]


@pytest.mark.parametrize('case', X3P_CASES, ids=lambda c: 'x3p_B%d_%dto%d_%dx%d_g%d_r%d_fb%d' % (c[:6] + (c[7], int(c[8]))))
def test_conv1x1_split_operands(ops, case):
    """conv_x3p_kernel (1x1, fp32 storage, operands split into f16 / bf16 hi + lo, three products per MAC on the 16-bit matrix
    pipe) vs an fp64 convolution and vs the fp32 direct kernel; channel slices of wider buffers, neighbours untouched."""
    B, cin, cout, H, W, groups, relu, res_kind, use_fb = case
    g = torch.Generator().manual_seed(hash(case) % (2 ** 31) + 270)
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin // groups, 1, 1, generator=g) / np.sqrt(cin // groups)
    b = torch.randn(cout, generator=g) * 0.1
    fb = torch.randn(B, cout, generator=g) if use_fb else None
    ref = F.conv2d(x.double(), w.double(), None if use_fb else b.double(), 1, 0, 1, groups)
    if use_fb:
        ref = ref + fb[:, :, None, None].double()
    res = None
    if res_kind:
        res = torch.randn((B, cout, H, W), generator=g)
        ref = ref + res.double()
    if relu:
        ref = F.relu(ref)
    xin = torch.full((B, H, W, cin + 8), 3.0, device='cuda')            # input in channels 4.. of a wider buffer
    xin[..., 4:4 + cin] = x.permute(0, 2, 3, 1).cuda()
    errs = {}
    for algo in ('split16', 'split_bf16', 'direct'):
        dst = torch.full((B, H, W, cout + 16), 7.0, device='cuda')     # output into channels 8..
        ops.conv2d(xin, w, None if use_fb else b, relu=relu, groups=groups, cin=cin // groups, in_coff=4, algo=algo,
                   out=dst, out_coff=8, residual=None if res is None else ops.to_nhwc(res),
                   frame_bias=None if fb is None else fb.cuda())
        torch.cuda.synchronize()
        got = dst[..., 8:8 + cout].permute(0, 3, 1, 2).cpu()
        errs[algo] = (got.double() - ref).abs().max().item()
        assert (dst[..., :8] == 7).all() and (dst[..., 8 + cout:] == 7).all(), algo
    assert errs['split16'] < 2e-5 and errs['split16'] < 4 * errs['direct'] + 2e-6, errs
    assert errs['split_bf16'] < 5e-4, errs


P1_CASES = [
    # (B, Cin, Cout, H, W, groups, relu, residual kind, frame_bias) - enough 256-pixel x 64-cout items to fill 256 CUs
    (8, 64, 256, 64, 64, 1, True, 1, False),       # layer1's conv3: two chunks, four n-blocks, residual + ReLU
    (16, 256, 64, 64, 64, 1, True, 0, False),      # eight chunks
    (40, 32, 32, 32, 64, 1, False, 0, True),       # single-chunk items, one n-tile per wave, per-frame bias rows
    (12, 96, 128, 32, 32, 1, True, 1, False),      # three chunks
    (10, 128, 192, 32, 64, 2, False, 0, False),    # two groups of 64 -> 96 (one n-tile per wave)
]


@pytest.mark.parametrize('case', P1_CASES, ids=lambda c: 'p1_B%d_%dto%d_%dx%d_g%d_r%d_fb%d' % (c[:6] + (c[7], int(c[8]))))
def test_conv1x1_streaming_frame(ops, case):
    """conv_p1_kernel (fp32 1x1 on the four-wave streaming frame: register-staged loads one chunk ahead, fp32 planes in LDS, one
    barrier per chunk) vs an fp64 convolution and vs the eight-wave direct kernel (acrmi_tune cfg 809 keeps it)."""
    B, cin, cout, H, W, groups, relu, res_kind, use_fb = case
    L = pkg('_lib').lib()
    g = torch.Generator().manual_seed(hash(case) % (2 ** 31) + 280)
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin // groups, 1, 1, generator=g) / np.sqrt(cin // groups)
    b = torch.randn(cout, generator=g) * 0.1
    fb = torch.randn(B, cout, generator=g) if use_fb else None
    ref = F.conv2d(x.double(), w.double(), None if use_fb else b.double(), 1, 0, 1, groups)
    if use_fb:
        ref = ref + fb[:, :, None, None].double()
    res = None
    if res_kind:
        res = torch.randn((B, cout, H, W), generator=g)
        ref = ref + res.double()
    if relu:
        ref = F.relu(ref)
    xin = torch.full((B, H, W, cin + 8), 3.0, device='cuda')            # input in channels 4.. of a wider buffer
    xin[..., 4:4 + cin] = x.permute(0, 2, 3, 1).cuda()
    outs = {}
    for cfg in (-1, 809):
        dst = torch.full((B, H, W, cout + 16), 7.0, device='cuda')     # output into channels 8..
        L.acrmi_tune(0, cfg)
        try:
            ops.conv2d(xin, w, None if use_fb else b, relu=relu, groups=groups, cin=cin // groups, in_coff=4, algo='direct',
                       out=dst, out_coff=8, residual=None if res is None else ops.to_nhwc(res),
                       frame_bias=None if fb is None else fb.cuda())
            torch.cuda.synchronize()
        finally:
            L.acrmi_tune(0, -1)
        got = dst[..., 8:8 + cout].permute(0, 3, 1, 2).cpu()
        err = (got.double() - ref).abs().max().item()
        assert err < 1e-4, (cfg, err)
        assert (dst[..., :8] == 7).all() and (dst[..., 8 + cout:] == 7).all(), cfg
        outs[cfg] = got
    assert (outs[-1] - outs[809]).abs().max().item() < 2e-5      # (different accumulation orders of the same fp32 products)


WINO3_CASES = [
    # B, Cin, H, W, relu, residual
    (2, 32, 16, 32, True, True),         # HRNet branch 0 shape class: BasicBlock conv2 (residual + ReLU)
    (1, 32, 8, 16, False, False),        # a single 8x16 tile
    (3, 32, 24, 48, True, False),        # 27 tiles: several items per workgroup only with a tiny grid (see cfg loop)
    (2, 16, 16, 16, True, True),         # Cin = 16: the second 16-channel step multiplies zero-padded taps
    (1, 20, 32, 32, False, True),        # ragged Cin inside the second step
    (1, 3, 16, 16, True, False),         # Cin = 3
    (4, 32, 64, 64, True, True),         # 128 / 64 items
]


@pytest.mark.parametrize('case', WINO3_CASES, ids=lambda c: 'wino3_B%d_%dto32_%dx%d' % c[:4])
@pytest.mark.parametrize('cfg', [-1, 839, 833, 832, 836, 837], ids=['default', 'no_store_waves', 'register_loader', 'tile16x16_8waves', 'lds_dma_1wave', 'wave_tile_16x32'])
def test_conv2d_winograd_lds_matches_torch(ops, case, cfg):
    """conv_wino3_kernel (F(2x2,3x3), taps resident in LDS, 16 positions per wave on v_mfma_f32_16x16x4_f32) vs an
    fp64 direct convolution, in both workgroup shapes; input / residual / output in channel slices of wider buffers."""
    B, cin, H, W, relu, use_res = case
    if cfg in (832, 837) and H % 16:
        pytest.skip('16x16 tiles need H % 16 == 0')
    L = pkg('_lib').lib()
    g = torch.Generator().manual_seed(hash(case) % (2 ** 31) + 3)
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(32, cin, 3, 3, generator=g) / np.sqrt(cin * 9)
    b = torch.randn(32, generator=g) * 0.1
    ref = F.conv2d(x.double(), w.double(), b.double(), 1, 1)
    res = None
    if use_res:
        res = torch.randn(ref.shape, generator=g)
        ref = ref + res.double()
    if relu:
        ref = F.relu(ref)
    xin = torch.full((B, H, W, 40), 3.0, device='cuda')                 # input in channels 4.. of a wider buffer
    xin[..., 4:4 + cin] = x.permute(0, 2, 3, 1).cuda()
    dst = torch.full((B, H, W, 48), 7.0, device='cuda')                 # output into channels 8..39
    L.acrmi_tune(0, cfg)
    try:
        ops.conv2d(xin, w, b, relu=relu, cin=cin, in_coff=4, algo='winograd2d_lds', out=dst, out_coff=8,
                   residual=None if res is None else ops.to_nhwc(res))
        torch.cuda.synchronize()
    finally:
        L.acrmi_tune(0, -1)
    got = dst[..., 8:40].permute(0, 3, 1, 2).cpu().double()
    err = (got - ref).abs().max().item()
    assert err < 5e-5, err          # Winograd F(2x2,3x3) in fp32: ~4x the direct kernel's round-off
    assert (dst[..., :8] == 7).all() and (dst[..., 40:] == 7).all()     # neighbour channels untouched


def test_conv2d_winograd_lds_rejects_what_it_cannot_do(ops):
    x = torch.zeros(1, 16, 32, 32, device='cuda')
    for kw in (dict(weight=torch.zeros(64, 32, 3, 3)), dict(weight=torch.zeros(32, 32, 3, 3), groups=1, stride=2),
               dict(weight=torch.zeros(32, 16, 3, 3), groups=2)):
        with pytest.raises(ValueError):
            ops.conv2d(x, algo='winograd2d_lds', **kw)
    with pytest.raises(ValueError):                                      # 12 rows: not a whole number of 8-row tiles
        ops.conv2d(torch.zeros(1, 12, 32, 32, device='cuda'), torch.zeros(32, 32, 3, 3), algo='winograd2d_lds')


def test_conv2d_channel_slices_and_frame_bias(ops):
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 109, 16, 16, generator=g)
    w = torch.randn(109, 109, 1, 1, generator=g) * 0.1
    fb = torch.randn(2, 112, generator=g)
    ref = F.conv2d(x, w) + fb[:, :109, None, None]
    out = ops.conv2d(ops.to_nhwc(x), w, None, frame_bias=fb.cuda())
    assert (_nchw(out, 109) - ref).abs().max().item() < 2e-5
    # write into a channel slice of a wider buffer, read from a channel slice
    xw = torch.randn(1, 96, 32, 32, generator=g)
    w2 = torch.randn(3, 32, 1, 1, generator=g)
    dst = torch.full((1, 32, 32, 112), 7.0, device='cuda')
    ops.conv2d(ops.to_nhwc(xw), w2, None, cin=32, in_coff=64, out=dst, out_coff=5)
    ref2 = F.conv2d(xw[:, 64:96], w2)
    assert (dst[..., 5:8].permute(0, 3, 1, 2).cpu() - ref2).abs().max().item() < 2e-5
    assert (dst[..., :5] == 7).all() and (dst[..., 8:] == 7).all()


@pytest.mark.parametrize('algo', ['winograd2d', 'direct'])
def test_conv3x3_channel_slices_frame_bias_residual(ops, algo):
    """The 3x3 kernels on the buffer conventions the program uses: input read from a channel slice of a wider buffer,
    output written into an aligned / an unaligned channel slice (neighbour channels untouched), per-frame bias,
    residual whose last rows end exactly at the end of its tensor (bounded residual descriptor), ragged Cout."""
    g = torch.Generator().manual_seed(11)
    B, H, W = 3, 16, 32
    xw = torch.randn(B, 80, H, W, generator=g)
    w = torch.randn(40, 48, 3, 3, generator=g) / np.sqrt(48 * 9)
    fb = torch.randn(B, 64, generator=g)
    res = torch.randn(B, 40, H, W, generator=g)
    ref = F.relu(F.conv2d(xw[:, 32:80].double(), w.double(), None, 1, 1) + fb[:, :40, None, None].double() + res.double())
    for coff in (8, 5):
        dst = torch.full((B, H, W, 60), 7.0, device='cuda')
        ops.conv2d(ops.to_nhwc(xw), w, None, relu=True, cin=48, in_coff=32, out=dst, out_coff=coff,
                   frame_bias=fb.cuda(), residual=ops.to_nhwc(res), algo=algo)
        got = dst[..., coff:coff + 40].permute(0, 3, 1, 2).cpu().double()
        assert (got - ref).abs().max().item() < 5e-5, (algo, coff)
        assert (dst[..., :coff] == 7).all() and (dst[..., coff + 40:] == 7).all()


BIAS_MAP_CASES = [
    # (B, Cin, Cout, H, W, k, stride, groups, algo)
    (3, 32, 512, 32, 32, 3, 2, 1, 'direct'),         # tower entry: 3x3 stride 2 (conv_ws2)
    (3, 32, 256, 16, 32, 3, 1, 1, 'winograd2d'),     # contact conv: one 32-channel chunk (conv_wino2)
    (2, 48, 256, 16, 32, 3, 1, 1, 'winograd24'),     # HRNet-W48 contact conv (conv_wino24)
    (3, 48, 96, 20, 24, 3, 2, 1, 'direct'),          # ragged tiles, stride 2
    (2, 24, 40, 9, 13, 3, 1, 1, 'winograd2d'),       # ragged everything: the element-wise epilogue
    (2, 24, 40, 9, 13, 3, 1, 1, 'direct'),
    (2, 64, 96, 16, 16, 1, 1, 1, 'direct'),          # 1x1
    (2, 16, 24, 12, 16, 3, 1, 2, 'winograd'),        # groups: the map covers groups * Cout channels
]


@pytest.mark.parametrize('case', BIAS_MAP_CASES, ids=lambda c: 'map_B%d_%dto%d_%dx%d_k%ds%dg%d_%s' % c)
def test_conv2d_position_bias_map(ops, case):
    """ACRMI_CONV_BIAS_MAP: ONE [Ho,Wo,C] map added to every frame before the ReLU (the coordinate channels of the head
    convs folded into a per-pixel bias, packer.coord_bias_map) - every conv kernel that takes a residual."""
    B, cin, cout, H, W, k, stride, groups, algo = case
    g = torch.Generator().manual_seed(31)
    x = torch.randn(B, cin * groups, H, W, generator=g)
    w = torch.randn(cout * groups, cin, k, k, generator=g) / np.sqrt(cin * k * k)
    b = torch.randn(cout * groups, generator=g) * 0.1
    ref0 = F.conv2d(x.double(), w.double(), b.double(), stride, k // 2, 1, groups)
    m = torch.randn(1, cout * groups, ref0.shape[2], ref0.shape[3], generator=g)
    ref = F.relu(ref0 + m.double())
    out = ops.conv2d(ops.to_nhwc(x), w, b, stride=stride, relu=True, groups=groups, cin=cin, residual=ops.to_nhwc(m), algo=algo)
    assert (_nchw(out, cout * groups).double() - ref).abs().max().item() < 5e-5
    # (and the same call with a per-frame residual still means a per-frame residual)
    r = torch.randn(B, cout * groups, ref0.shape[2], ref0.shape[3], generator=g)
    out = ops.conv2d(ops.to_nhwc(x), w, b, stride=stride, relu=True, groups=groups, cin=cin, residual=ops.to_nhwc(r), algo=algo)
    assert (_nchw(out, cout * groups).double() - F.relu(ref0 + r.double())).abs().max().item() < 5e-5


def test_conv2d_position_bias_map_rejected_by_the_lds_tap_kernel(ops):
    x = torch.zeros(2, 16, 32, 32, device='cuda')
    with pytest.raises(Exception):
        ops.conv2d(x, torch.zeros(32, 32, 3, 3), None, residual=torch.zeros(1, 16, 32, 32, device='cuda'), algo='winograd2d_lds')


SPLITK_CASES = [
    # (B, Cin, Cout, H, W, splits, residual)
    (1, 256, 256, 16, 16, 4, True),      # stage-4 branch 3, one frame: 16 items of 8 chunks -> 64 of 2
    (1, 128, 128, 32, 32, 2, True),      # branch 2
    (2, 256, 256, 16, 16, 2, False),
    (1, 192, 96, 12, 20, 3, True),       # ragged tiles, ragged Cout (HRNet-W48's third branch), three slices
    (3, 512, 64, 8, 16, 8, False),       # eight slices
    (8, 256, 256, 16, 16, 4, True),      # more items than CUs: a workgroup takes several (tile, slice) items in turn
]


@pytest.mark.parametrize('case', SPLITK_CASES, ids=lambda c: 'splitk_B%d_%dto%d_%dx%d_s%d_res%d' % c)
def test_conv2d_splitk_matches_torch_and_is_reproducible(ops, case):
    """ACRMI_CONV_SPLITK / acrmi_conv2d_splitk: the K-slices of one convolution as separate work items whose partial
    tiles the last arriver sums in slice order.  Against an fp64 convolution; BIT-equal run after run (the sum order does
    not depend on who arrives last) with the workspace re-used as the kernel left it; and equal to the unsplit Winograd
    kernel within fp32 round-off."""
    B, cin, cout, H, W, splits, with_res = case
    g = torch.Generator().manual_seed(41)
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) / np.sqrt(cin * 9)
    b = torch.randn(cout, generator=g) * 0.1
    res = torch.randn(B, cout, H, W, generator=g) if with_res else None
    ref = F.conv2d(x.double(), w.double(), b.double(), 1, 1)
    if with_res:
        ref = ref + res.double()
    ref = F.relu(ref)
    xd, rd = ops.to_nhwc(x), (ops.to_nhwc(res) if with_res else None)
    L = pkg('_lib').lib()
    ws = torch.zeros(int(L.acrmi_conv2d_splitk_workspace(B, H, W, cout, splits)), dtype=torch.uint8, device='cuda')
    outs = []
    for rep in range(4):
        out = ops.conv2d_splitk(xd, w, b, splits=splits, relu=True, residual=rd, workspace=ws)
        torch.cuda.synchronize()
        outs.append(out.clone())
    assert (_nchw(outs[0], cout).double() - ref).abs().max().item() < 5e-5
    for o in outs[1:]:
        assert torch.equal(o, outs[0])
    plain = ops.conv2d(xd, w, b, relu=True, residual=rd, algo='winograd2d')
    assert (plain - outs[0]).abs().max().item() < 2e-5
    # the counters are back to zero: the workspace holds nothing a later launch depends on
    ncnt = B * ((H + 7) // 8) * ((W + 15) // 16) * (1 if cout <= 32 else (cout + 63) // 64 * 2) * 4
    assert int(ws[:4 * ncnt].view(torch.int32).abs().sum()) == 0


def test_conv2d_splitk_arrival_order_stress(ops):
    """60 back-to-back launches of the case with more (tile, slice) items than CUs, while a second stream keeps the GPU
    busy with other work (arrival order of the slices varies): every launch bit-equal to the first."""
    g = torch.Generator().manual_seed(43)
    B, cin, cout, H, W, splits = 8, 256, 256, 16, 16, 4
    x = ops.to_nhwc(torch.randn(B, cin, H, W, generator=g))
    w = torch.randn(cout, cin, 3, 3, generator=g) / np.sqrt(cin * 9)
    L = pkg('_lib').lib()
    ws = torch.zeros(int(L.acrmi_conv2d_splitk_workspace(B, H, W, cout, splits)), dtype=torch.uint8, device='cuda')
    first = ops.conv2d_splitk(x, w, None, splits=splits, workspace=ws).clone()
    noise = torch.randn(4096, 4096, device='cuda')
    side = torch.cuda.Stream()
    outs = []
    for rep in range(60):
        if rep % 3 == 0:
            with torch.cuda.stream(side):
                noise = torch.sin(noise)
        outs.append(ops.conv2d_splitk(x, w, None, splits=splits, workspace=ws).clone())
    torch.cuda.synchronize()
    for rep, o in enumerate(outs):
        assert torch.equal(o, first), rep


def test_conv2d_splitk_rejects_what_it_cannot_do(ops):
    x = torch.zeros(1, 16, 16, 128, device='cuda')
    with pytest.raises(Exception):
        ops.conv2d_splitk(x, torch.zeros(64, 128, 3, 3), splits=4)          # 32-channel slices: one chunk per item
    with pytest.raises(Exception):
        ops.conv2d_splitk(x, torch.zeros(33, 128, 3, 3), splits=2)          # the 33-channel kernel is not split
    with pytest.raises(Exception):
        ops.conv2d_splitk(x, torch.zeros(64, 128, 3, 3), splits=2, workspace=torch.zeros(64, dtype=torch.uint8, device='cuda'))


def test_conv_stride2_groups_and_slices(ops):
    g = torch.Generator().manual_seed(12)
    x = torch.randn(2, 64, 20, 24, generator=g)
    w = torch.randn(96, 32, 3, 3, generator=g) / np.sqrt(32 * 9)     # 2 groups of 32 -> 48
    b = torch.randn(96, generator=g) * 0.1
    ref = F.conv2d(x.double(), w.double(), b.double(), 2, 1, 1, 2)
    out = ops.conv2d(ops.to_nhwc(x), w, b, stride=2, groups=2, cin=32)
    assert (_nchw(out, 96).double() - ref).abs().max().item() < 2e-5


def _random_conv_case(rs):
    k, stride = [(3, 1), (3, 1), (3, 2), (1, 1)][rs.randint(4)]
    groups = [1, 1, 1, 2, 4][rs.randint(5)]
    cin_g = int(rs.choice([4, 8, 12, 16, 20, 32, 36, 48, 64, 72])) if groups > 1 else int(rs.randint(1, 97))
    cout_g = int(rs.randint(1, 100))
    algo = 'direct' if (k, stride) != (3, 1) else ['direct', 'winograd', 'winograd2d', 'winograd2d'][rs.randint(4)]
    return dict(B=int(rs.randint(1, 4)), H=int(rs.randint(1, 41)), W=int(rs.randint(1, 41)), k=k, stride=stride,
                groups=groups, cin_g=cin_g, cout_g=cout_g, algo=algo, relu=bool(rs.randint(2)), res=bool(rs.randint(2)),
                frame_bias=bool(rs.randint(3) == 0) and groups == 1, in_coff=4 * int(rs.randint(0, 3)),
                out_coff=int(rs.randint(0, 7)), in_tail=4 * int(rs.randint(0, 3)), out_tail=int(rs.randint(0, 6)))


@pytest.mark.parametrize('seed', range(6))
def test_conv2d_random_sweep(ops, seed):
    """Seeded sweep over what the hand-picked cases leave out: random sizes down to 1x1 frames, ragged channel counts,
    groups, channel slices on both sides, per-frame bias rows, residuals, all three algorithms (12 configurations per
    seed).  Reference: fp64 torch convolution."""
    rs = np.random.RandomState(1000 + seed)
    for it in range(12):
        c = _random_conv_case(rs)
        g = torch.Generator().manual_seed(int(rs.randint(1 << 30)))
        cin, cout = c['cin_g'] * c['groups'], c['cout_g'] * c['groups']
        in_cs = (c['in_coff'] + cin + c['in_tail'] + 3) // 4 * 4
        out_cs = c['out_coff'] + cout + c['out_tail']
        xw = torch.randn(c['B'], in_cs, c['H'], c['W'], generator=g)
        w = torch.randn(cout, c['cin_g'], c['k'], c['k'], generator=g) / np.sqrt(c['cin_g'] * c['k'] ** 2)
        b = torch.randn(cout, generator=g) * 0.1
        x = xw[:, c['in_coff']:c['in_coff'] + cin]
        ref = F.conv2d(x.double(), w.double(), None, c['stride'], c['k'] // 2, 1, c['groups'])
        fb = None
        if c['frame_bias']:
            fb = torch.randn(c['B'], cout + int(rs.randint(0, 5)), generator=g)
            ref = ref + fb[:, :cout, None, None].double()
        else:
            ref = ref + b.double()[None, :, None, None]
        res = None
        if c['res']:
            res = torch.randn(ref.shape, generator=g)
            ref = ref + res.double()
        if c['relu']:
            ref = F.relu(ref)
        Ho, Wo = ref.shape[2:]
        dst = torch.full((c['B'], Ho, Wo, out_cs), 7.0, device='cuda')
        ops.conv2d(ops.to_nhwc(xw), w, None if c['frame_bias'] else b, stride=c['stride'], relu=c['relu'],
                   groups=c['groups'], cin=c['cin_g'], in_coff=c['in_coff'], out=dst, out_coff=c['out_coff'],
                   frame_bias=None if fb is None else fb.cuda(), residual=None if res is None else ops.to_nhwc(res),
                   algo=c['algo'])
        torch.cuda.synchronize()
        got = dst[..., c['out_coff']:c['out_coff'] + cout].permute(0, 3, 1, 2).cpu().double()
        err = (got - ref).abs().max().item()
        assert err < 6e-5, (seed, it, c, err)
        assert (dst[..., :c['out_coff']] == 7).all() and (dst[..., c['out_coff'] + cout:] == 7).all(), (seed, it, c)


def test_conv2d_rejects_unsupported(ops):
    x = torch.zeros(1, 8, 8, 8, device='cuda')
    with pytest.raises(ValueError):
        ops.conv2d(x, torch.zeros(8, 8, 5, 5))
    with pytest.raises(ValueError):
        ops.conv2d(x, torch.zeros(8, 8, 1, 1), stride=3)
    with pytest.raises(ValueError):
        ops.conv2d(x, torch.zeros(8, 8, 3, 3), stride=2, algo='winograd2d')


def test_u8norm_bit_exact(ops):
    img = torch.arange(256, dtype=torch.uint8).repeat(3 * 16).view(1, 16, 256, 3).contiguous()
    out = ops.u8norm(img.cuda()).cpu()
    ref = (img.float() / 255.) * 2.0 - 1.0
    assert torch.equal(out[..., :3], ref) and (out[..., 3] == 0).all()


@pytest.mark.parametrize('B,H,W', [(2, 32, 128), (1, 16, 256), (3, 48, 128)])
def test_stem_conv_reads_uint8_and_matches_fp64_conv(ops, B, H, W):
    """csrc/stem.hip: relu(conv1(x/255*2-1)) straight from the uint8 frame (acr/model.py:832,589-603) against an fp64
    convolution of u8norm's fp32 values - every border row/column included (the conv pads the NORMALISED map), extreme
    pixel values at the corners."""
    g = torch.Generator().manual_seed(B * 1000 + W)
    img = torch.randint(0, 256, (B, H, W, 3), dtype=torch.uint8, generator=g)
    img[:, 0, 0] = 255
    img[:, -1, -1] = 0
    w = torch.randn(64, 3, 3, 3, generator=g, dtype=torch.float64) * 0.3
    b = torch.randn(64, generator=g, dtype=torch.float64)
    xn = ((img.float() / 255.) * 2.0 - 1.0).double().permute(0, 3, 1, 2)
    for relu in (True, False):
        ref = F.conv2d(xn, w, b, stride=2, padding=1)
        ref = F.relu(ref) if relu else ref
        out = ops.stem_conv(img.cuda(), w.numpy(), b.numpy(), relu=relu)
        assert out.shape == (B, H // 2, W // 2, 64)
        assert (_nchw(out, 64).double() - ref).abs().max().item() < 2e-5


def test_stem_conv_rejects_shapes_outside_its_tiling(ops):
    w, b = np.zeros((64, 3, 3, 3)), np.zeros(64)
    with pytest.raises(ValueError):
        ops.stem_conv(torch.zeros(1, 24, 128, 3, dtype=torch.uint8, device='cuda'), w, b)     # H % 16
    with pytest.raises(ValueError):
        ops.stem_conv(torch.zeros(1, 32, 64, 3, dtype=torch.uint8, device='cuda'), w, b)      # W % 128


def test_bilinear2x_matches_torch(ops):
    x = torch.randn(2, 32, 24, 40)
    ref = F.interpolate(x, scale_factor=(2, 2), mode='bilinear', align_corners=True)
    out = ops.bilinear2x(ops.to_nhwc(x))
    assert (_nchw(out, 32) - ref).abs().max().item() < 2e-6


def test_fuse_sum_matches_reference_order(ops):
    t0, t1, t2 = torch.randn(2, 32, 32, 32), torch.randn(2, 32, 16, 16), torch.randn(2, 32, 4, 4)
    ref = F.relu(t0 + F.interpolate(t1, scale_factor=2, mode='nearest') + F.interpolate(t2, scale_factor=8, mode='nearest'))
    out = ops.fuse_sum([ops.to_nhwc(t0), ops.to_nhwc(t1), ops.to_nhwc(t2)], [0, 1, 3])
    assert torch.equal(_nchw(out, 32), ref)          # same order of fp32 adds -> bit exact


def test_attpool_matches_oracle(ops):
    g = torch.Generator().manual_seed(3)
    segm = torch.randn(2, 33, 256, 256, generator=g) * 2
    feat = torch.randn(2, 320, 128, 128, generator=g)
    part = F.interpolate(segm, scale_factor=(0.5, 0.5), mode='nearest')[:, 1:]
    ref = acr_net.hadamard(feat.double(), part.double())          # [B,320,32]
    got = ops.attpool(ops.to_nhwc(segm), ops.to_nhwc(feat), 320).cpu()     # [B,32,320]
    assert (got.permute(0, 2, 1).double() - ref).abs().max().item() < 1e-5


@pytest.mark.parametrize('name', list(cases.DECODE_CASES))
def test_decode_cases_match_reference_golden(ops, name):
    g = golden('decode_cases.npz')
    maps = {k: ops.to_nhwc(torch.from_numpy(v)) for k, v in cases.decode_maps(name).items()}
    slots = ops.decode_maps(maps['l_center_map'], maps['r_center_map'], maps['l_params_maps'], maps['r_params_maps'],
                            maps['l_prior_maps'], maps['r_prior_maps']).cpu().numpy()
    L = pkg('_lib')
    flag = slots[0, :, L.SLOT_FLAG] > 0.5
    np.testing.assert_array_equal(flag, g[name + '_detection_flag'].astype(bool))
    np.testing.assert_allclose(slots[0, :, L.SLOT_PARAMS:L.SLOT_PARAMS + 109], g[name + '_params_pred'], 1e-6, 1e-6)
    np.testing.assert_allclose(slots[0, :, L.SLOT_CAM:L.SLOT_CAM + 3], g[name + '_cam'], 1e-6, 1e-6)
    np.testing.assert_allclose(slots[0, :, L.SLOT_BETAS:L.SLOT_BETAS + 10], g[name + '_betas'], 1e-6, 1e-6)
    np.testing.assert_allclose(slots[0, :, L.SLOT_POSES:L.SLOT_POSES + 48], g[name + '_poses'], 2e-5, 2e-5)
    lc, rc = g[name + '_l_centers_pred'][0], g[name + '_r_centers_pred'][0]
    assert slots[0, 0, L.SLOT_FLATIND] == lc[1] * 64 + lc[0] and slots[0, 1, L.SLOT_FLATIND] == rc[1] * 64 + rc[0]


def test_decode_batch_matches_oracle(ops):
    """A batch of mixed cases == the oracle's per-frame decode (N x batch-1 semantics)."""
    names = list(cases.DECODE_CASES)
    maps = {k: torch.cat([torch.from_numpy(cases.decode_maps(n)[k]) for n in names]) for k in cases.decode_maps(names[0])}
    want = odec.decode(maps)
    m = {k: ops.to_nhwc(v) for k, v in maps.items()}
    slots = ops.decode_maps(m['l_center_map'], m['r_center_map'], m['l_params_maps'], m['r_params_maps'],
                            m['l_prior_maps'], m['r_prior_maps']).cpu().numpy()
    L = pkg('_lib')
    np.testing.assert_array_equal(slots[:, :, L.SLOT_FLAG] > 0.5, want['flag'])
    np.testing.assert_array_equal(slots[:, :, L.SLOT_FLATIND].astype(np.int64), want['flat_ind'])
    np.testing.assert_allclose(slots[:, :, L.SLOT_PARAMS:L.SLOT_PARAMS + 109], want['params_pred'], 1e-6, 1e-6)
    np.testing.assert_allclose(slots[:, :, L.SLOT_POSES:L.SLOT_POSES + 48], want['poses'], 2e-5, 2e-5)


def test_rot6d_kat_on_device(ops):
    """6D -> axis-angle known answers (reference golden) through the decode kernel's gather path."""
    g = golden('rot6d_kat.npz')
    x6, aa = g['x6'], g['aa']
    n = x6.shape[0]
    B = (n + 15) // 16
    params = np.zeros((B, 109, 64, 64), np.float32)
    for i in range(n):
        params[i // 16, 3 + 6 * (i % 16):9 + 6 * (i % 16), 0, 0] = x6[i]
    center = np.full((B, 1, 64, 64), -1.0, np.float32)
    center[:, 0, 0, 0] = 1.0                              # detection at pixel 0, left hand
    zc = np.full((B, 1, 64, 64), -1.0, np.float32)
    t = lambda a: ops.to_nhwc(torch.from_numpy(a))
    pr = np.zeros((B, 106, 64, 64), np.float32)
    slots = ops.decode_maps(t(center), t(zc), t(params), t(params), t(pr), t(pr)).cpu().numpy()
    L = pkg('_lib')
    got = slots[:, 0, L.SLOT_POSES:L.SLOT_POSES + 48].reshape(-1, 3)[:n]
    np.testing.assert_allclose(got, aa, rtol=2e-5, atol=2e-6)


@pytest.fixture(scope='module')
def engine(mano_tables):
    eng = pkg('engine').Engine(0)
    t = {k: dict(v) for k, v in mano_tables.items()}
    t['left']['shapedirs'] = t['left']['shapedirs'].copy()
    t['left']['shapedirs'][:, 0, :] *= -1
    eng.load_mano(t)
    eng._flipped_tables = t
    return eng


@pytest.mark.parametrize('n,seed', [(1, 1), (2, 2), (16, 3)])
def test_mano_matches_reference_golden(engine, n, seed):
    g = golden('mano_cases.npz')
    poses, betas = cases.mano_inputs(n, seed)
    for side, sid in (('l', 0), ('r', 1)):
        v, j, c, _ = engine.mano(torch.from_numpy(poses), torch.from_numpy(betas), torch.full((n,), sid))
        key = 'n%d_%s' % (n, side)
        assert np.abs(v.cpu().numpy() - g[key + '_verts']).max() < 2e-6       # metres
        assert np.abs(j.cpu().numpy() - g[key + '_joints']).max() < 2e-6
        assert np.abs(c.cpu().numpy() - g[key + '_center']).max() < 2e-6


def test_mano_mixed_sides_projection_and_empty(engine):
    g = golden('mano_cases.npz')
    poses, betas = cases.mano_inputs(4, 9)
    cam, offsets = cases.proj_inputs(4, 9)
    v, j, c, extra = engine.mano(torch.from_numpy(poses), torch.from_numpy(betas), torch.ones(4),
                                 cam=torch.from_numpy(cam), offsets=torch.from_numpy(offsets))
    np.testing.assert_allclose(extra['verts_camed'].cpu().numpy(), g['proj_verts_camed'], 1e-5, 2e-6)
    np.testing.assert_allclose(extra['pj2d'].cpu().numpy(), g['proj_pj2d'], 1e-5, 2e-6)
    np.testing.assert_allclose(extra['pj2d_org'].cpu().numpy(), g['proj_pj2d_org'], 1e-5, 2e-3)
    # interleaved sides in one launch == oracle per side
    side = torch.tensor([0, 1, 1, 0])
    v, j, _, _ = engine.mano(torch.from_numpy(poses), torch.from_numpy(betas), side)
    for r in range(4):
        name = 'left' if side[r] == 0 else 'right'
        ov, oj, _ = omano.mano_forward(engine._flipped_tables[name], name, poses[r:r + 1], betas[r:r + 1])
        assert np.abs(v[r].cpu().numpy() - ov[0]).max() < 2e-6 and np.abs(j[r].cpu().numpy() - oj[0]).max() < 2e-6
    v0, j0, _, _ = engine.mano(torch.zeros(0, 48), torch.zeros(0, 10), torch.zeros(0))
    assert v0.shape == (0, 778, 3) and j0.shape == (0, 21, 3)              # N == 0 is legal (mano_wrapper.py:43)
