"""CPU checks of the round-3 lowering paths (no GPU): the op-list interpreter (oracle/program.py) is pinned on the fp32
HRNet-W32 program to the reference-pinned oracle; the 16-bit / HRNet-W48 lowerings are structurally sound; the 16-bit
weight packing and rounding are what conv_h16_kernel expects."""
import numpy as np
import pytest
import torch

from conftest import pkg
from oracle import acr_net, program as oprog


@pytest.fixture(scope='module')
def frame():
    return torch.from_numpy(pkg('synth').make_frames(1, seed=0))


def test_interpreter_reads_the_fp32_program_like_the_pinned_oracle(synth_sd, frame):
    """oracle/program.py on the fp32 W32 program == oracle/acr_net.py (itself pinned to the real reference by
    tests/test_oracle_pinned.py) on every head map: this pins the interpreter's reading of the op list - strides, channel
    slices, groups, residuals, per-frame bias, the composed exits - before it serves as own-oracle of the 16-bit programs."""
    torch.set_num_threads(8)
    prog = pkg('packer').lower(synth_sd, keep_weights=True, point_heads=False)
    got = oprog.run_program(prog, frame).head_maps()
    with torch.no_grad():
        ref = acr_net.network(synth_sd, frame)
    for k in ref:
        err, scale = float((got[k] - ref[k]).abs().max()), float(ref[k].abs().max())
        assert err < 2e-5 * max(1.0, scale), (k, err, scale)


def test_16bit_interpreter_stays_within_quantisation_noise(synth_sd, frame):
    """The fp16 program, interpreted: rounding once per layer costs ~1e-3 relative per layer; through the network the head
    maps stay within a few 1e-3 of the fp32 oracle's (the bound the GPU test compares the kernels against)."""
    torch.set_num_threads(8)
    prog = pkg('packer').lower(synth_sd, keep_weights=True, precision='fp16')
    assert prog['precision'] == 'fp16' and prog['width'] == 32
    got = oprog.run_program(prog, frame).head_maps()
    with torch.no_grad():
        ref = acr_net.network(synth_sd, frame)
    for k in ref:
        err, scale = float((got[k] - ref[k]).abs().max()), float(ref[k].abs().max())
        assert 1e-5 < err < 1e-2 * max(1.0, scale), (k, err, scale)


@pytest.mark.parametrize('precision,width', [('fp16', 32), ('bf16', 32), ('fp32', 48), ('fp16', 48)])
def test_lowering_of_16bit_and_w48_programs_is_consistent(precision, width):
    packer, synth, L = pkg('packer'), pkg('synth'), pkg('_lib')
    sd = synth.make_state_dict(seed=0, width=width)
    prog = packer.lower(sd, precision=precision)
    dt = packer.PRECISIONS[precision]
    bufs, ops = prog['bufs'], prog['ops']
    assert prog['width'] == width and {b[4] for b in bufs} <= {0, dt}
    hl = prog['heads']
    for b in (hl.center_buf[0], hl.center_buf[1], hl.params_buf[0], hl.params_buf[1], hl.prior_buf[0], hl.prior_buf[1], hl.segm_buf):
        assert bufs[b][4] == 0                                         # head maps stay fp32 (acr/model.py:56-62 .float())
    assert bufs[hl.backbone_buf][4] == dt and bufs[hl.backbone_buf][2] >= width + 2
    for b in bufs:
        assert b[2] % (8 if b[4] else 4) == 0                          # 16-byte vectors
    n_conv = 0
    for o in ops:
        if o.kind == L.OP_CONV:
            n_conv += 1
            idt, odt = bufs[o.in_buf][4], bufs[o.out_buf][4]
            assert idt == dt and odt in (dt, 0)
            assert o.res_buf < 0 or bufs[o.res_buf][4] == odt         # a residual has the type of the output
            if dt:
                assert (o.flags & 7) == 0                              # 16-bit programs: direct kernel only
        assert o.kind != L.OP_POINTHEADS or (dt == 0 and width == 32)
    assert n_conv >= 300
    if width == 48:
        assert sd['backbone.stage4.0.branches.3.0.conv1.weight'].shape == (384, 384, 3, 3)
        assert sd['l_final_layers.1.0.0.weight'].shape[1] == 50


def test_pack_conv_h16_layout_and_rounding():
    packer = pkg('packer')
    rng = np.random.default_rng(0)
    w = rng.normal(size=(40, 20, 3, 3))
    for name, dt, td in (('fp16', packer.DT_F16, torch.float16), ('bf16', packer.DT_BF16, torch.bfloat16)):
        bits, bp = packer.pack_conv_h16(w, np.arange(40, dtype=np.float32), dt)
        assert bits.dtype == np.uint16 and bits.size == 9 * 2 * 2 * 64 * 8 and bp.size == 64      # 2 steps of 16, 2 n-tiles
        vals = torch.from_numpy(bits.view(np.int16).copy()).view(td).float().numpy().reshape(3, 3, 2, 2, 64, 8)
        want = torch.from_numpy(w.astype(np.float32)).to(td).float().numpy()                       # torch's nearest-even rounding
        # [ky][kx][s][nt][lane][e]: cout = nt*32 + (lane & 31), ci = 16 s + 8 (lane >> 5) + e
        for (ky, kx, s, nt, lane, e) in ((0, 0, 0, 0, 0, 0), (1, 2, 1, 1, 37, 3), (2, 1, 0, 1, 7, 7), (1, 1, 1, 0, 63, 1)):
            co, ci = nt * 32 + (lane & 31), 16 * s + 8 * (lane >> 5) + e
            exp = want[co, ci, ky, kx] if co < 40 and ci < 20 else 0.0
            assert vals[ky, kx, s, nt, lane, e] == exp, (name, ky, kx, s, nt, lane, e)
        assert np.array_equal(packer.round_to(w, dt), want)


def test_planted_center_peaks_are_where_they_were_planted(frame):
    """synth.plant_center_peaks: the oracle's center maps peak exactly at the planted interior pixels (the fixture
    e2e_interior.npz is the real reference's view of the same checkpoints)."""
    import cases
    from oracle import decode as odec
    torch.set_num_threads(8)
    sd = cases.interior_state_dict(pkg('synth'), 'mid')
    with torch.no_grad():
        maps = acr_net.network(sd, frame)
    s = odec.decode(maps)
    _, lp, rp = cases.INTERIOR_CASES['mid']
    assert s['flag'].tolist() == [[True, True]]
    assert s['flat_ind'].tolist() == [[lp[0] * 64 + lp[1], rp[0] * 64 + rp[1]]]


@pytest.mark.parametrize('stride', [1, 2])
def test_coord_bias_map_is_the_convolution_of_the_coordinate_channels(stride):
    """packer.coord_bias_map == conv3x3(coordinate maps of the reference's get_coord_maps, acr/model.py:340-369, restated
    in oracle/acr_net.coord_maps) with the coordinate channels' filter columns, zero padding included; and the lowered
    fp32 program really drops those channels from the two head convs (Cin 34 -> 32) while the 16-bit one keeps them."""
    import torch.nn.functional as F
    packer = pkg('packer')
    rs = np.random.RandomState(5)
    wc = rs.randn(24, 2, 3, 3)
    got = packer.coord_bias_map(wc, 128, stride)
    cm = acr_net.coord_maps(128).double()                                  # [1,2,128,128]: x, y
    ref = F.conv2d(cm, torch.from_numpy(wc), None, stride, 1)[0].permute(1, 2, 0).numpy()
    assert got.shape == (ref.shape[0], ref.shape[1], 24)
    assert np.abs(got - ref).max() < 1e-12


def test_fp32_program_takes_the_coordinate_channels_as_a_bias_map(synth_sd):
    packer, L = pkg('packer'), pkg('_lib')
    for precision, want_cin, want_flag in (('fp32', 32, L.CONV_BIAS_MAP), ('fp16', 34, 0)):
        prog = packer.lower(synth_sd, precision=precision, point_heads=False)
        ops = {i['name']: o for o, i in zip(prog['ops'], prog['op_info'])}
        for name in ('towers.entry', 'contact_layers.1.0'):
            assert ops[name].cin == want_cin and (ops[name].flags & L.CONV_BIAS_MAP) == want_flag, (precision, name)
        # the FLOP figure of the line keeps the reference's 34 input channels
        info = {i['name']: i for i in prog['op_info']}
        assert abs(info['contact_layers.1.0']['flops'] - 2.0 * 128 * 128 * 256 * 34 * 9) < 1.0


def test_small_batch_program_splits_the_low_resolution_layers(synth_sd, frame):
    """packer.lower(splitk=True) (what Engine.load_state_dict asks for below 16 frames): the 3x3 layers of HRNet branches
    2 (128 channels at 32x32) and 3 (256 at 16x16) become ACRMI_CONV_SPLITK ops - K-slices of 64 channels as the op's
    "groups", same output channels - and the interpreter, which reads such an op as ONE convolution over the concatenated
    slices, still agrees with the reference-pinned oracle on every head map."""
    torch.set_num_threads(8)
    packer, L = pkg('packer'), pkg('_lib')
    prog = packer.lower(synth_sd, keep_weights=True, point_heads=False, wino24=False, splitk=True)
    split = [(o, i) for o, i in zip(prog['ops'], prog['op_info']) if o.kind == L.OP_CONV and o.flags & L.CONV_SPLITK]
    assert len(split) == 7 * 8 + 3 * 8                       # branch 2 in 7 modules, branch 3 in 3, 8 convs each
    for o, i in split:
        assert o.cin == 64 and o.groups in (2, 4) and o.groups * o.cin == o.cout and (o.flags & 7) == 2, i['name']
        h, w, cs = prog['bufs'][o.out_buf][:3]
        assert cs == o.cout and (h, w) in ((32, 32), (16, 16))
        assert abs(i['flops'] - 2.0 * h * w * o.cout * o.cout * 9) < 1.0
    assert not any(o.flags & L.CONV_SPLITK for o in packer.lower(synth_sd, point_heads=False)['ops'] if o.kind == L.OP_CONV)
    got = oprog.run_program(prog, frame).head_maps()
    with torch.no_grad():
        ref = acr_net.network(synth_sd, frame)
    for k in ref:
        err, scale = float((got[k] - ref[k]).abs().max()), float(ref[k].abs().max())
        assert err < 2e-5 * max(1.0, scale), (k, err, scale)


def test_resnet50_program_matches_its_oracle(frame):
    """BASELINE.json configs[1]'s backbone (build-defined: schema._resnet50_backbone - the reference's `--backbone resnet50`
    is a dead flag, acr/config.py:95): the lowered fp32 program, read by the op-list interpreter, against the functional
    restatement oracle/acr_net.resnet50_backbone + the reference-pinned heads; the op list holds the three operators only
    this backbone uses (7x7 stem, max-pool, 1x1 stride-2 projections); the bf16 lowering (configs[1]'s dtype) is
    structurally sound.  NO REFERENCE ORACLE."""
    torch.set_num_threads(8)
    packer, synth, L, schema = pkg('packer'), pkg('synth'), pkg('_lib'), pkg('schema')
    sd = synth.make_state_dict(seed=0, width='resnet50')
    assert schema.width_of(packer.strip_prefix(sd)) == 'resnet50' and len(sd) == len(schema.state_dict_schema('resnet50'))
    prog = packer.lower(sd, keep_weights=True, point_heads=False)
    assert prog['width'] == 'resnet50'
    ops = prog['ops']
    stem = [o for o in ops if o.kind == L.OP_STEM]
    assert len(stem) == 1 and stem[0].ksize == 7 and stem[0].stride == 2
    assert sum(1 for o in ops if o.kind == L.OP_MAXPOOL) == 1
    assert sum(1 for o in ops if o.kind == L.OP_CONV and o.ksize == 1 and o.stride == 2) == 3      # layer2/3/4.0.downsample
    assert sum(1 for o in ops if o.kind == L.OP_BILINEAR2X) == 4                                    # 3 upsampling stages + segm head
    hl = prog['heads']
    assert prog['bufs'][hl.backbone_buf][:3] == (128, 128, 68)                                      # 64 + 2 channels, stride 68
    got = oprog.run_program(prog, frame).head_maps()
    with torch.no_grad():
        ref = acr_net.network(sd, frame)
    for k in ref:
        err, scale = float((got[k] - ref[k]).abs().max()), float(ref[k].abs().max())
        assert err < 2e-5 * max(1.0, scale), (k, err, scale)
    p16 = packer.lower(sd, precision='bf16', point_heads=False)
    assert all(o.flags & 7 == 0 for o in p16['ops'] if o.kind == L.OP_CONV) and len(p16['ops']) == len(ops)


def test_pack_stem7_layout():
    packer = pkg('packer')
    rs = np.random.RandomState(3)
    w, b = rs.randn(64, 3, 7, 7), rs.randn(64)
    frag, bias = packer.pack_stem7(w, b)
    frag = frag.reshape(74, 2, 2, 32)                       # [step][n-tile][k parity][cout row]
    for s, n, lh, li in ((0, 0, 0, 0), (10, 1, 1, 5), (73, 0, 0, 31), (36, 1, 0, 17)):
        k = 2 * s + lh
        ky, kx, c = k // 21, (k % 21) // 3, k % 3
        assert frag[s, n, lh, li] == np.float32(w[32 * n + li, c, ky, kx])
    assert (frag[73, :, 1, :] == 0).all() and np.array_equal(bias, b.astype(np.float32))


def test_layer1_pairs_are_lowered_only_where_they_pay(synth_sd):
    """OP_PAIR1X1 (block i's conv3 + residual + ReLU chained with block i + 1's conv1): three pair ops in the large-batch fp32
    program - whose interpreter reading is pinned to the oracle by the first test of this file - none in small-batch
    (split-K) or 16-bit programs; with pairs, layer1.0 keeps its projection shortcut as a separate 1x1 conv, without them it
    is folded into the block's last conv (128 -> 256)."""
    packer, L = pkg('packer'), pkg('_lib')
    big = packer.lower(synth_sd, point_heads=False)
    names = [i['name'] for i in big['op_info']]
    pairs = [o for o in big['ops'] if o.kind == L.OP_PAIR1X1]
    assert len(pairs) == 3 and all(o.cin == 64 and o.cout == 256 and o.res_buf >= 0 and o.aux_buf >= 0 for o in pairs)
    assert 'backbone.layer1.0.downsample.0' in names and 'backbone.layer1.1.conv1' not in names
    assert 'backbone.layer1.3.conv3' in names                     # the last block ends in a plain conv
    for kw in (dict(splitk=True, wino24=False), dict(precision='fp16'), dict(pairs=False)):
        prog = packer.lower(synth_sd, point_heads=False, **kw)
        assert not any(o.kind == L.OP_PAIR1X1 for o in prog['ops']), kw
        assert 'backbone.layer1.0.conv3+downsample' in [i['name'] for i in prog['op_info']], kw
    # the packed LDS image: both matrices in the kernel's k orders
    rs = np.random.RandomState(2)
    w3, b3, w1, b1 = rs.randn(256, 64), rs.randn(256), rs.randn(64, 256), rs.randn(64)
    img = packer.pack_pair1x1(w3, b3, w1, b1)
    a1 = img[:16384].reshape(8, 8, 64, 4)
    a2 = img[16384:32768].reshape(8, 2, 4, 64, 4)
    for c, s4, lane, e in ((0, 0, 0, 0), (3, 5, 40, 2), (7, 7, 63, 3)):
        m, h = lane % 32, lane // 32
        assert a1[c, s4, lane, e] == np.float32(w3[32 * c + m, 32 * h + 4 * s4 + e])
    for c, nt, g, lane, e in ((0, 0, 0, 0, 0), (5, 1, 2, 37, 1), (7, 1, 3, 63, 3)):
        m, h = lane % 32, lane // 32
        assert a2[c, nt, g, lane, e] == np.float32(w1[32 * nt + m, 32 * c + 8 * g + 4 * h + e])
    assert np.array_equal(img[32768:32768 + 256], b3.astype(np.float32)) and np.array_equal(img[-64:], b1.astype(np.float32))
