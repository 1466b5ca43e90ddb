"""CPU tests of the host logic that mirrors the reference's Python surface: slot -> row re-packing,
config validation, packer invariants, pre-processing, smoothing, camera translation."""
import numpy as np
import pytest
import torch

import cases
from conftest import golden, pkg
from oracle import decode as odec


def _slots_tensor(s):
    """oracle slots dict -> [B,2,176] tensor in the library's slot layout."""
    L = pkg('_lib')
    B = s['flag'].shape[0]
    t = torch.zeros(B, 2, L.SLOT)
    t[:, :, L.SLOT_FLAG] = torch.from_numpy(s['flag'].astype(np.float32))
    t[:, :, L.SLOT_FLATIND] = torch.from_numpy(s['flat_ind'].astype(np.float32))
    t[:, :, L.SLOT_SCORE] = torch.from_numpy(s['score'])
    t[:, :, L.SLOT_CAM:L.SLOT_CAM + 3] = torch.from_numpy(s['cam'])
    t[:, :, L.SLOT_POSES:L.SLOT_POSES + 48] = torch.from_numpy(s['poses'])
    t[:, :, L.SLOT_BETAS:L.SLOT_BETAS + 10] = torch.from_numpy(s['betas'])
    t[:, :, L.SLOT_PARAMS:L.SLOT_PARAMS + 109] = torch.from_numpy(s['params_pred'])
    return t


@pytest.mark.parametrize('name', list(cases.DECODE_CASES))
def test_rows_from_slots_reproduces_reference_dict(name):
    """Row order (left rows then right rows), placeholder rows, flags, centers (x,y), hand types and the
    params_dict split all equal what the reference's ResultParser.parse returned for the same maps."""
    g = golden('decode_cases.npz')
    maps = {k: torch.from_numpy(v) for k, v in cases.decode_maps(name).items()}
    meta = {'batch_ids': torch.arange(1), 'offsets': torch.zeros(1, 10), 'imgpath': ['a']}
    out = pkg('acr.result_parser').rows_from_slots(_slots_tensor(odec.decode(maps)), meta)
    np.testing.assert_array_equal(out['detection_flag'].numpy(), g[name + '_detection_flag'])
    np.testing.assert_allclose(out['params_pred'].numpy(), g[name + '_params_pred'], 1e-6, 1e-6)
    np.testing.assert_allclose(out['params_dict']['poses'].numpy(), g[name + '_poses'], 1e-5, 1e-5)
    np.testing.assert_allclose(out['params_dict']['betas'].numpy(), g[name + '_betas'], 1e-6, 1e-6)
    np.testing.assert_allclose(out['params_dict']['cam'].numpy(), g[name + '_cam'], 1e-6, 1e-6)
    np.testing.assert_array_equal(out['l_centers_pred'].numpy(), g[name + '_l_centers_pred'])
    np.testing.assert_array_equal(out['r_centers_pred'].numpy(), g[name + '_r_centers_pred'])
    np.testing.assert_array_equal(out['output_hand_type'].numpy(), g[name + '_hand_type'])
    assert out['output_hand_type'].dtype == torch.int32
    assert int(out['left_hand_num']) == 1 and int(out['right_hand_num']) == 1
    assert out['reorganize_idx'].tolist() == [0, 0] and meta['offsets'].shape == (2, 10)


def test_rows_from_slots_batch_ordering():
    names = ['both_near', 'left_only', 'none', 'right_only', 'both_far']
    maps = {k: torch.cat([torch.from_numpy(cases.decode_maps(n)[k]) for n in names]) for k in cases.decode_maps(names[0])}
    meta = {'batch_ids': torch.arange(10, 15)}
    out = pkg('acr.result_parser').rows_from_slots(_slots_tensor(odec.decode(maps)), meta)
    assert int(out['left_hand_num']) == 3 and int(out['right_hand_num']) == 3
    assert out['reorganize_idx'].tolist() == [10, 11, 14, 10, 13, 14]      # left rows ascending, then right rows
    assert out['output_hand_type'].tolist() == [0, 0, 0, 1, 1, 1]
    assert out['detection_flag'].tolist() == [1.0] * 6


def test_config_rejects_what_the_reference_rejects():
    cfg = pkg('config')
    ns = cfg.parse_args(['--configs_yml', '/nonexistent.yml'])
    assert ns.centermap_conf_thresh == 0.35 and ns.align_idx == 9 and ns.kernel_sizes == [5]
    for bad in (['--backbone', 'resnet'], ['--prior_mode', 'merge'], ['--Rot_type', 'aa'], ['--centermap_size', '32'],
                ['--model_precision', 'int8'], ['--attention_mode', 'none']):
        with pytest.raises(ValueError):
            cfg.parse_args(['--configs_yml', '/nonexistent.yml'] + bad)
    assert cfg.parse_args(['--configs_yml', '/nonexistent.yml', '-t']).temporal_optimization is True
    # the reference's autocast branch (acr/config.py:96, acr/model.py:33-37) and its bf16 twin are lowered as 16-bit programs
    for prec in ('fp32', 'fp16', 'bf16'):
        assert cfg.parse_args(['--configs_yml', '/nonexistent.yml', '--model_precision', prec]).model_precision == prec


def test_w48_checkpoint_reshapes_the_module_without_gpu():
    """BASELINE.json configs[4]: an HRNet-W48 checkpoint (same key names, wider tensors) re-shapes acr.model.ACR's state
    dict; a W32 checkpoint loaded afterwards shapes it back; a tensor of a third width is refused."""
    synth, schema = pkg('synth'), pkg('schema')
    m = pkg('acr.model').ACR()
    sd48 = synth.make_state_dict(seed=0, width=48)
    assert list(sd48.keys()) == list(schema.state_dict_schema(48).keys()) == list(schema.state_dict_schema(32).keys())
    missing, unexpected = m.load_state_dict(sd48)
    assert missing == [] and unexpected == [] and m._width == 48
    assert tuple(m.state_dict()['backbone.stage4.0.branches.3.0.conv1.weight'].shape) == (384, 384, 3, 3)
    assert tuple(m.state_dict()['contact_layers.1.0.weight'].shape) == (256, 50, 3, 3)
    m.load_state_dict(synth.make_state_dict(seed=0))
    assert m._width == 32 and tuple(m.state_dict()['contact_layers.1.0.weight'].shape) == (256, 34, 3, 3)
    with pytest.raises(ValueError):
        schema.stage_cfg(40)
    assert schema.schema_digest(48)['n_params'] > 2 * schema.schema_digest(32)['n_params']


def test_pack_conv_layout_and_bn_folding(synth_sd):
    packer = pkg('packer')
    w = np.arange(40 * 10 * 9, dtype=np.float32).reshape(40, 10, 3, 3)
    wp, bp = packer.pack_conv(w, np.arange(40, dtype=np.float32))
    assert wp.size == 9 * 2 * 2 * 64 * 4 and bp.size == 64           # cin8 = 2, n_tiles = 2
    wp = wp.reshape(3, 3, 2, 2, 64, 4)
    for (ky, kx, s, nt, lane, e) in [(0, 0, 0, 0, 0, 0), (2, 1, 1, 1, 37, 1), (1, 2, 0, 1, 7, 3), (1, 1, 1, 0, 63, 3)]:
        co, ci = nt * 32 + (lane & 31), 8 * s + 4 * (lane >> 5) + e
        want = w[co, ci, ky, kx] if (co < 40 and ci < 10) else 0.0
        assert wp[ky, kx, s, nt, lane, e] == want
    P = packer.Program(synth_sd)
    wf, bf = P.folded('backbone.conv1', 'backbone.bn1')
    x = torch.randn(1, 3, 8, 8, dtype=torch.float64)
    ref = torch.nn.functional.batch_norm(
        torch.nn.functional.conv2d(x, synth_sd['backbone.conv1.weight'].double(), None, 2, 1),
        synth_sd['backbone.bn1.running_mean'].double(), synth_sd['backbone.bn1.running_var'].double(),
        synth_sd['backbone.bn1.weight'].double(), synth_sd['backbone.bn1.bias'].double(), False, 0.0, 1e-5)
    got = torch.nn.functional.conv2d(x, torch.from_numpy(wf), torch.from_numpy(bf), 2, 1)
    assert (ref - got).abs().max() < 1e-12


def test_lowering_accounts_for_every_flop_and_rejects_bad_checkpoints(synth_sd):
    packer = pkg('packer')
    prog = packer.lower({'module.' + k: v for k, v in synth_sd.items()})       # wild.pkl style prefix
    L = pkg('_lib')
    dense = [o for o in prog['op_info'] if o['mode'] != L.MODE_POINT]
    point = [o for o in prog['op_info'] if o['mode'] != L.MODE_DENSE]
    assert abs(sum(o['flops'] for o in dense) / 1e9 - 102.1) < 0.1
    # point-heads variant: six of the eight head towers + the mix conv leave the dense op list
    assert 8.0 < (sum(o['flops'] for o in dense) - sum(o['flops'] for o in point)) / 1e9 < 10.5
    assert [o['name'] for o in point if o['kind'] == L.OP_POINTHEADS] == ['l.point_heads', 'r.point_heads']
    assert all(op.mode == info['mode'] for op, info in zip(prog['ops'], prog['op_info']))
    assert len(packer.lower(synth_sd, point_heads=False)['ops']) == len(dense)
    assert len(prog['ops']) < 400 and prog['blob'].dtype == np.float32
    bad = dict(synth_sd)
    del bad['backbone.stage3.1.branches.2.3.bn2.running_var']
    with pytest.raises(ValueError):
        packer.lower(bad)
    bad = dict(synth_sd)
    bad['contact_layers.4.weight'] = torch.zeros(109, 100, 1, 1)
    with pytest.raises(ValueError):
        packer.lower(bad)


def test_pad_geometry_matches_imgaug_rule():
    """The host-side geometry helper (acr.utils) == the oracle's imgaug 0.4.0 restatement; the pixels themselves
    only ever come from the HIP kernel (img_preprocess refuses to run without a GPU)."""
    from oracle import preprocess as opre
    u = pkg('acr.utils')
    for shape in ((1080, 1920, 3), (480, 640, 3), (700, 301, 3), (301, 700, 3), (512, 512, 3), (333, 1000, 3)):
        assert u.compute_paddings_to_reach_aspect_ratio(shape) == opre.compute_paddings_to_reach_aspect_ratio(shape)
    if not torch.cuda.is_available():
        with pytest.raises(pkg('_lib').AcrmiError):
            u.img_preprocess(np.zeros((8, 8, 3), np.uint8), 'x.jpg', single_img_input=True)
    with pytest.raises(ValueError):
        u.img_preprocess(np.zeros((8, 8, 3), np.float32))


def test_engine_and_pool_refuse_to_run_without_a_gpu():
    """No CPU fallback: without a GPU the context objects raise (the product path never routes through oracle/)."""
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    engine = pkg('engine')
    with pytest.raises(pkg('_lib').AcrmiError):
        engine.Engine(0)
    with pytest.raises(pkg('_lib').AcrmiError):
        engine.EnginePool(0, n=2)
    with pytest.raises(ValueError):
        engine.EnginePool(0, n=0)


def test_config_plumbed_and_rejected_options():
    """flags the kernels honour are accepted (centermap_conf_thresh, align_idx, mano_mesh_root_align,
    smooth_coeff); flags they cannot honour raise instead of being silently ignored."""
    cfg = pkg('config')
    base = ['--configs_yml', '/nonexistent.yml']
    ns = cfg.parse_args(base + ['--centermap_conf_thresh', '0.2', '--align_idx', '0', '--mano_mesh_root_align', 'false',
                                '--smooth_coeff', '3.0'])
    assert ns.centermap_conf_thresh == 0.2 and ns.align_idx == 0 and ns.mano_mesh_root_align is False
    for bad in (['--kernel_sizes', '3'], ['--kernel_sizes', '5', '7'], ['--align_idx', '21']):
        with pytest.raises(ValueError):
            cfg.parse_args(base + bad)
    # the reference's own default max_hand=4 (acr/config.py:161) is only read under train_flag
    # (acr/result_parser.py:221-224): carrying it over must not be an error
    assert cfg.parse_args(base + ['--max_hand', '4']).max_hand == 4


def test_state_dict_surface_without_gpu(synth_sd):
    """acr.model.ACR keeps the reference's key names; load_state_dict accepts 'module.'-prefixed checkpoints."""
    m = pkg('acr.model').ACR()
    assert list(m.state_dict().keys()) == list(pkg('schema').state_dict_schema().keys())
    missing, unexpected = m.load_state_dict({'module.' + k: v for k, v in synth_sd.items()})
    assert missing == [] and unexpected == []
    assert torch.equal(m.state_dict()['l_final_layers.2.2.bias'], synth_sd['l_final_layers.2.2.bias'])
    with pytest.raises(ValueError):
        m.load_state_dict({'backbone.conv1.weight': torch.zeros(3, 3)})


def test_splitk_slices_are_whole_32_channel_chunks():
    """split-K slices must satisfy the kernel's Cin % 32 == 0 per slice - 576 / 640 input channels used to become
    8 slices of 72 / 80 channels, a program acrmi_set_program rejects."""
    packer = pkg('packer')
    for cin in (128, 256, 384, 512, 576, 640, 1024):
        s = packer.splitk_slices(3, 1, cin, 256, 1, 16, 16, False)
        assert 1 <= s <= 8 and cin % s == 0 and (s == 1 or (cin // s) % 32 == 0), (cin, s)
    assert packer.splitk_slices(3, 1, 256, 256, 1, 16, 16, False) == 4          # HRNet branch 3 unchanged
    assert packer.splitk_slices(3, 1, 128, 128, 1, 32, 32, False) == 2          # branch 2 unchanged
    assert packer.splitk_slices(3, 1, 576, 256, 1, 16, 16, False) == 6
    assert packer.splitk_slices(3, 1, 640, 256, 1, 16, 16, False) == 5


@pytest.mark.parametrize('name', list(cases.DECODE_BATCHES))
def test_reference_batch_semantics_host_logic(name):
    """ResultParser(batch_semantics='reference'): result_parser.reference_prior_gate decides from a first decode's flags /
    centers what the reference decides batch-wide (acr/result_parser.py:42-47,131), rows_from_slots re-packs the second
    decode - both checked here on CPU against the real reference's rows (decode_batches.npz), with the oracle standing in
    for the decode kernel."""
    rp = pkg('acr.result_parser')
    S = pkg('_lib')
    g = golden('decode_batches.npz')
    maps = {k: torch.from_numpy(v) for k, v in cases.decode_batch_maps(name).items()}

    def as_slots(d):
        B = d['flag'].shape[0]
        sl = torch.zeros(B, 2, S.SLOT)
        sl[:, :, S.SLOT_FLAG] = torch.from_numpy(d['flag'].astype(np.float32))
        sl[:, :, S.SLOT_FLATIND] = torch.from_numpy(d['flat_ind'].astype(np.float32))
        sl[:, :, S.SLOT_SCORE] = torch.from_numpy(d['score'])
        sl[:, :, S.SLOT_CAM:S.SLOT_CAM + 3] = torch.from_numpy(d['cam'])
        sl[:, :, S.SLOT_POSES:S.SLOT_POSES + 48] = torch.from_numpy(d['poses'])
        sl[:, :, S.SLOT_BETAS:S.SLOT_BETAS + 10] = torch.from_numpy(d['betas'])
        sl[:, :, S.SLOT_PARAMS:S.SLOT_PARAMS + 109] = torch.from_numpy(d['params_pred'])
        return sl
    first = as_slots(odec.decode(maps))                                   # pass 1: per-frame rule (what acrmi_decode does)
    gate = rp.reference_prior_gate(first)
    want = odec.reference_gate(torch.from_numpy(odec.decode(maps)['flag']), torch.from_numpy(odec.decode(maps)['flat_ind']))
    np.testing.assert_array_equal(gate.numpy().astype(bool), want.numpy())
    second = as_slots(odec.decode(maps, batch_semantics='reference'))     # pass 2: the gated decode
    B = second.shape[0]
    meta = {'batch_ids': torch.arange(B), 'imgpath': ['f%d' % b for b in range(B)]}
    out = rp.rows_from_slots(second, meta)
    np.testing.assert_array_equal(out['detection_flag'].numpy(), g[name + '_detection_flag'])
    np.testing.assert_array_equal(out['reorganize_idx'].numpy(), g[name + '_reorganize_idx'])
    np.testing.assert_array_equal(out['output_hand_type'].numpy(), g[name + '_hand_type'])
    assert [int(out['left_hand_num']), int(out['right_hand_num'])] == g[name + '_hand_nums'].tolist()
    np.testing.assert_allclose(out['params_pred'].numpy(), g[name + '_params_pred'], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(out['params_dict']['poses'].numpy(), g[name + '_poses'], rtol=1e-5, atol=1e-5)
    np.testing.assert_array_equal(out['l_centers_pred'].numpy(), g[name + '_l_centers_pred'])
    np.testing.assert_array_equal(out['r_centers_pred'].numpy(), g[name + '_r_centers_pred'])
    assert list(meta['imgpath']) == ['f%d' % b for b in g[name + '_reorganize_idx']]
    with pytest.raises(ValueError):
        rp.ResultParser(batch_semantics='whole-batch')


def test_polyphase2_weights_reproduce_the_stride2_convolution():
    """packer.polyphase2_weights + the accumulation scheme of csrc/conv_pp2.inc, emulated in numpy: per 2x2 output block
    every wave (oy, ox) forms its seven input combinations from a 3x3 subset of the block's 5x5 window (transposed for the
    waves with oy != ox), accumulates corner / edge / centre products, and output (oy, ox) is its own corner + the two edge
    tiles that carry its row / column + wave 0's centre - 25 products instead of 36.  Must equal a stride-2 convolution."""
    packer = pkg('packer')
    rng = np.random.default_rng(5)
    cin, cout, H, W = 5, 4, 8, 12
    x = rng.standard_normal((cin, H, W))
    w = rng.standard_normal((cout, cin, 3, 3))
    ref = torch.nn.functional.conv2d(torch.from_numpy(x)[None], torch.from_numpy(w), None, 2, 1)[0].numpy()
    U = packer.polyphase2_weights(w)                       # [cout, cin, 4 waves, 7]
    assert U.shape == (cout, cin, 4, 7)
    xp = np.zeros((cin, H + 1, W + 1))
    xp[:, 1:, 1:] = x                                     # the patch origin is (-1, -1); even H, W: no bottom / right ring
    out = np.zeros_like(ref)
    for R in range(H // 4):
        for C in range(W // 4):
            win = xp[:, 4 * R:4 * R + 5, 4 * C:4 * C + 5]   # the block's 5x5 window
            acc = {}
            for oy in range(2):
                for ox in range(2):
                    yi, xj = [4 * oy, 2, 1 + 2 * oy], [4 * ox, 2, 1 + 2 * ox]
                    tr = oy != ox
                    P = [[win[:, yi[v], xj[u]] if tr else win[:, yi[u], xj[v]] for v in range(3)] for u in range(3)]
                    D = [P[0][v] - P[1][v] for v in range(3)]
                    k = [D[0] - D[1], D[2], P[2][0] - P[2][1], P[2][2], D[1], P[2][1], P[1][1]]
                    wv = 2 * oy + ox
                    prod = [U[:, :, wv, i] @ k[i] for i in range(7)]
                    acc[wv] = (prod[0] + prod[1] + prod[2] + prod[3], prod[4] + prod[5], prod[6])
            for oy in range(2):
                for ox in range(2):
                    wv = 2 * oy + ox
                    out[:, 2 * R + oy, 2 * C + ox] = acc[wv][0] + acc[3 * oy][1] + acc[2 - ox][1] + acc[0][2]
    np.testing.assert_allclose(out, ref, rtol=1e-12, atol=1e-12)
    # (round 5: small-batch programs - wino24=False - take the polyphase kernel too, packer.POLYPHASE2_SMALL)
    assert packer.conv_algo(3, 2, 32, 64, 1, 64, 64) == 5 and packer.conv_algo(3, 2, 32, 64, 1, 64, 64, wino24=False) == (5 if packer.POLYPHASE2_SMALL else 0)
    assert packer.conv_algo(3, 2, 24, 64, 1, 64, 64) == 0 and packer.conv_algo(3, 2, 32, 64, 1, 12, 64) == 0


def test_split16_and_pack_conv_x3_layout():
    """packer.split16: hi + lo reproduces the value to 2^-22 |x| + 2^-25 (the second term: lo falls into the f16 subnormals
    for |x| < 2^-3 - an ABSOLUTE error of at most half their spacing, 3e-8); pack_conv_x3 puts the two halves where
    conv_x3_kernel reads them ([tap][cin / 16][n-tile][hi | lo][lane][8]); conv_algo routes only split16 programs to algo 6."""
    packer = pkg('packer')
    rng = np.random.default_rng(6)
    x = rng.standard_normal(4096) * np.exp(rng.uniform(-6, 6, 4096))
    hi, lo = packer.split16(x)
    assert np.all(hi == hi.astype(np.float16)) and np.all(lo == lo.astype(np.float16))
    assert np.all(np.abs(hi.astype(np.float64) + lo - x) <= 2.0 ** -22 * np.abs(x) + 2.0 ** -25)
    w = rng.standard_normal((40, 32, 3, 3))
    b = rng.standard_normal(40)
    w[3] *= 2.0 ** -9                                       # a small filter row
    packed, bp = packer.pack_conv_x3([(w, b)])
    nt = packer.n_tiles_for(40)
    shift = packer.x3_weight_shift([w])
    assert 2.0 ** 12 <= np.abs(w).max() * 2.0 ** shift < 2.0 ** 13 and packed[-1] == np.float32(2.0 ** -shift)
    assert packed.size == 9 * (32 // 8) * nt * 256 + 1      # the fp32 packing's size + the scale
    arr = packed[:-1].view(np.float16).astype(np.float64).reshape(3, 3, 2, nt, 2, 2, 32, 8) * 2.0 ** -shift     # [ky, kx, s, nt, hl, kg, j, e]
    for (ky, kx, s, n, kgp, j, e) in [(0, 0, 0, 0, 0, 0, 0), (2, 1, 1, 1, 1, 7, 5), (1, 2, 0, 0, 1, 31, 7), (1, 1, 1, 0, 0, 3, 2)]:
        co, ci = 32 * n + j, 16 * s + 8 * kgp + e
        want = w[co, ci, ky, kx] if co < 40 else 0.0
        assert abs(arr[ky, kx, s, n, 0, kgp, j, e] + arr[ky, kx, s, n, 1, kgp, j, e] - want) <= 2.0 ** -22 * abs(want)    # (scaled: no subnormal lo)
    np.testing.assert_array_equal(bp[:40], b.astype(np.float32))
    # the weight scale: zero / non-finite filters fall back to no scaling; grouped ops share ONE scale (the largest group decides)
    assert packer.x3_weight_shift([np.zeros((32, 32, 3, 3))]) == 0 and packer.x3_weight_shift([np.full((32, 32, 1, 1), np.inf)]) == 0
    # folded filters above the f16 range shift DOWN: no inf halves, the kernel's inverse scale stays exact
    big = np.random.RandomState(3).normal(0, 1, (32, 32, 3, 3)) * 3e6
    sb = packer.x3_weight_shift([big])
    assert sb < 0 and 2.0 ** 12 <= np.abs(big).max() * 2.0 ** sb < 2.0 ** 13
    pb, _ = packer.pack_conv_x3([(big, np.zeros(32))])
    halves = pb[:-1].view(np.uint16).view(np.float16)
    assert np.isfinite(halves.astype(np.float32)).all() and pb[-1] == np.float32(2.0 ** -sb)
    pg, bg = packer.pack_conv_x3([(w[:32], b[:32]), (w[:32] * 2.0 ** -6, b[:32])], packer.DT_BF16)
    assert pg.size == 2 * 9 * (32 // 8) * 256 + 1 and pg[-1] == np.float32(2.0 ** -packer.x3_weight_shift([w[:32]])) and bg.size == 64
    assert packer.conv_algo(3, 1, 64, 64, 1, 64, 64, split16=True) == 6 and packer.conv_algo(3, 1, 64, 64, 1, 64, 64) == 4
    assert packer.conv_algo(3, 1, 64, 64, 1, 64, 64, split16='bf16') == 7
    assert packer.conv_algo(1, 1, 256, 64, 1, 128, 128, split16=True) == 6 and packer.conv_algo(1, 1, 256, 64, 1, 128, 128) == 0
    assert packer.conv_algo(1, 1, 256, 64, 1, 10, 10, split16=True) == 0 and packer.conv_algo(1, 2, 256, 64, 1, 64, 64, split16=True) == 0
    assert packer.conv_algo(3, 1, 64, 64, 1, 16, 16, split16=True) == 6 and packer.conv_algo(3, 1, 16, 64, 1, 64, 64, split16=True) == 2
    assert packer.conv_algo(3, 1, 64, 64, 1, 24, 16, split16=True) == 2
    # 3x3 stride 2 (round 6, conv_x3s2.inc): onto maps of 8x32-pixel tiles; the others keep the polyphase kernel
    assert packer.conv_algo(3, 2, 64, 64, 1, 128, 128, split16=True) == 6 and packer.conv_algo(3, 2, 64, 128, 1, 32, 32, split16='bf16') == 7
    assert packer.conv_algo(3, 2, 128, 256, 1, 16, 16, split16=True) == 5 and packer.conv_algo(3, 2, 16, 64, 1, 64, 64, split16=True) == 5


def test_fp16x3_program_is_the_fp32_program_with_other_kernels(synth_sd):
    """'fp16x3' lowers to the SAME op list, buffers and biases as 'fp32' - only the eligible 3x3 (stride 1, and stride 2 where the
    convolution is not a HR fuse host / the last link of an x0 chain) and 1x1 stride-1 convolutions change their kernel
    (algo 6) and weight packing."""
    packer, L = pkg('packer'), pkg('_lib')
    saved = packer.FUSE_FULLRES      # (the full-resolution fuse sum as conv_wino3's second output exists in the fp32 program only:
    packer.FUSE_FULLRES = False      #  branch 0's convolutions are conv_x3 launches in the split-operand program)
    try:
        p32 = packer.lower(synth_sd, precision='fp32', point_heads=False)
    finally:
        packer.FUSE_FULLRES = saved
    px3 = packer.lower(synth_sd, precision='fp16x3', point_heads=False)
    assert px3['bufs'] == p32['bufs'] and len(px3['ops']) == len(p32['ops'])
    n6 = n6s2 = 0
    for a, b in zip(p32['ops'], px3['ops']):
        assert (a.kind, a.in_buf, a.out_buf, a.res_buf, a.cin, a.cout, a.ksize, a.stride, a.groups) == (
            b.kind, b.in_buf, b.out_buf, b.res_buf, b.cin, b.cout, b.ksize, b.stride, b.groups)
        if b.kind == L.OP_CONV and (b.flags & 7) == 6:
            n6 += 1
            if b.stride == 2:
                n6s2 += 1
                assert b.ksize == 3 and (a.flags & 7) == 5 and b.nterms == 0 and a.flags - 5 == b.flags - 6      # (e.g. the bias-map flag)
            else:
                assert b.stride == 1 and ((b.ksize == 3 and (a.flags & 7) in (3, 4)) or (b.ksize == 1 and (a.flags & 7) == 0))
        else:
            assert a.flags == b.flags
    assert n6 >= 190 and n6s2 >= 20
    with pytest.raises(ValueError):
        packer.lower(synth_sd, precision='fp8')


@pytest.mark.parametrize('precision', ['fp16x3', 'bf16x3'])
def test_x3_weight_scale_sits_where_the_kernel_reads_it_for_every_op(synth_sd, precision):
    """conv_x3_kernel / conv_x3p_kernel read the op's power-of-two weight scale at
    a.w[groups * taps * ksteps * n_tiles * 512] - behind the fragments of ALL groups of the op AS LAUNCHED.  The point-heads
    variant used to launch 2-group clones of the 8-group tower packs and read garbage (4e+20) there.  Every algo-6/7 op of the
    program with point heads ON (the default of acr/main.py forward_batch) must find 2^-S of ITS OWN filters at that place."""
    packer, L = pkg('packer'), pkg('_lib')
    prog = packer.lower(synth_sd, precision=precision, point_heads=True, keep_weights=True)
    blob = prog['blob']
    seen, point = 0, 0
    for op, info in zip(prog['ops'], prog['op_info']):
        if op.kind != L.OP_CONV or (op.flags & 7) not in (6, 7):
            continue
        ksteps, n_tiles = op.cin // 16, (1 if op.cout <= 32 else (op.cout + 63) // 64 * 2)
        pos = op.w_off + op.groups * op.ksize * op.ksize * ksteps * n_tiles * 512
        ws = [w for (w, _) in info['wb']][:op.groups]
        assert len(ws) == op.groups, info['name']
        want = np.float32(2.0 ** -packer.x3_weight_shift(ws))
        assert blob[pos] == want, (info['name'], float(blob[pos]), float(want))
        seen += 1
        point += op.mode == L.MODE_POINT
    # the four center-tower convs of the point-heads variant carry their own packs; so does its 32 -> 128 stride-2 tower entry (a
    # split-operand launch since round 6, conv_x3s2.inc)
    assert seen >= 190 and point == 5


@pytest.mark.parametrize('lowering', ['large', 'small', 'fp16x3', 'fp16'])
def test_hr_fuse_sums_are_lowered_exactly_once_each(synth_sd, lowering):
    """acr/model.py:672-686: HRNet-W32 has 23 fuse sums (stage 2: 2, stage 3: 4 x 3, stage 4: 2 x 4 + 1).  Round 5 lowers them as
    (a) extra residual terms of the x0 downsampling chain's last stride-2 convolution (output resolutions i >= 1), (b) the second
    output of branch 0's last conv2 (i = 0, large-batch fp32 programs: ACRMI_CONV_DUAL) or (c) an OP_FUSESUM launch - every sum
    exactly once, with terms of the right geometry, in the reference's order (branch index ascending)."""
    packer, L = pkg('packer'), pkg('_lib')
    kw = {'large': {}, 'small': dict(wino24=False, splitk=True), 'fp16x3': dict(precision='fp16x3'), 'fp16': dict(precision='fp16')}[lowering]
    prog = packer.lower(synth_sd, point_heads=False, **kw)
    ops, info, bufs = prog['ops'], prog['op_info'], prog['bufs']
    hosted = [(o, i) for o, i in zip(ops, info) if o.kind == L.OP_CONV and o.nterms and not (o.flags & L.CONV_DUAL)]
    dual = [(o, i) for o, i in zip(ops, info) if o.kind == L.OP_CONV and (o.flags & L.CONV_DUAL)]
    sums = [(o, i) for o, i in zip(ops, info) if o.kind == L.OP_FUSESUM]
    want = {'large': (15, 8, 0), 'small': (15, 0, 8), 'fp16x3': (15, 0, 8), 'fp16': (0, 0, 23)}[lowering]
    assert (len(hosted), len(dual), len(sums)) == want
    for o, i in hosted:          # 3x3 stride 2, ReLU of the fuse, terms at the output's resolution >> shift, distinct buffers
        assert o.ksize == 3 and o.stride == 2 and o.relu == 1 and '+fuse' in i['name'] and 1 <= o.nterms <= 3
        ho, wo = bufs[o.out_buf][:2]
        seen = {o.out_buf, o.in_buf}
        for t in range(o.nterms):
            th, tw, tcs = bufs[o.term_buf[t]][:3]
            assert (th << o.term_shift[t], tw << o.term_shift[t]) == (ho, wo) and tcs >= o.cout and o.term_buf[t] not in seen
            seen.add(o.term_buf[t])
        # same-resolution terms (branches j <= i) come before the upsampled ones (j > i), shifts ascending: branch order
        shifts = [o.term_shift[t] for t in range(o.nterms)]
        assert shifts == sorted(shifts)
    for o, i in dual:            # branch 0's last conv2: residual block output + the upsampled projections of the lower branches
        assert (o.flags & 7) == 3 and o.res_buf >= 0 and o.relu == 1 and i['name'].endswith('conv2+fuse0')
        assert bufs[o.aux_buf][:2] == bufs[o.out_buf][:2] and o.aux_buf not in (o.out_buf, o.in_buf, o.res_buf)
        assert [o.term_shift[t] for t in range(o.nterms)] == list(range(1, o.nterms + 1))
    for o, i in sums:
        assert o.relu == 1 and 2 <= o.nterms <= 4


def test_rank_order_lane_planning_on_a_round_robin_program():
    """Round 6 lane planner (csrc/acrmi_program.hip build_schedule; tools/sched_sim.py is its cost model in Python): ops taken in
    UPWARD-RANK order - the longest remaining chain first - instead of program order.  On a program that walks four
    independent chains of unequal length round-robin (what HRNet's branches look like in program order) the rank order is a
    valid topological order, keeps the long chain on one lane and is never slower than the program-order plan in the model."""
    import importlib.util
    import os
    from conftest import ROOT
    spec = importlib.util.spec_from_file_location('sched_sim', os.path.join(ROOT, 'tools', 'sched_sim.py'))
    sim = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sim)
    rs = np.random.RandomState(0)
    lengths = [40, 12, 12, 6]                                  # links per chain; chain 0 is the critical one
    ms, deps, chain_of, last = [], [], [], [None] * 4
    for step in range(max(lengths)):                           # round-robin over the chains, like the lowered HR modules
        for c in range(4):
            if step < lengths[c]:
                deps.append([] if last[c] is None else [last[c]])
                last[c] = len(ms)
                ms.append(0.010 + 0.004 * rs.rand())
                chain_of.append(c)
    ms.append(0.02)                                            # a join (the heads)
    deps.append([l for l in last if l is not None])
    chain_of.append(-1)
    n, W = len(ms), 0.016
    ru, succ = sim.upward_rank(ms, deps)
    order = sorted(range(n), key=lambda j: (-ru[j], j))
    pos = {j: k for k, j in enumerate(order)}
    assert all(pos[d] < pos[j] for j in range(n) for d in deps[j])          # a valid enqueue order
    for L in (2, 4):
        greedy = sim.plan(list(range(n)), ms, deps, L, W)
        ranked = sim.plan(order, ms, deps, L, W)
        mg, _ = sim.simulate(list(range(n)), greedy, ms, deps, L, W)
        mr, _ = sim.simulate(order, ranked, ms, deps, L, W)
        assert mr <= mg + 1e-9, (L, mg, mr)
        crit = [j for j in range(n) if chain_of[j] == 0]
        assert len({ranked[j] for j in crit}) == 1                          # the critical chain never changes lanes
