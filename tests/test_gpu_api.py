"""GPU: the reference's Python surface (acr.model.ACR / MANOWrapper / ManoLayer / acr.main.ACR) on the HIP path
reproduces the dict the real reference produced (tests/golden/e2e_batch1.npz).  `pytest -m gpu`."""
import numpy as np
import pytest
import torch

import cases
from conftest import golden, pkg

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def model(synth_sd):
    m = pkg('acr.model').ACR(device=0, max_batch=2).eval()
    m.load_state_dict({'module.' + k: v for k, v in synth_sd.items()})      # checkpoint-style prefix
    return m.cuda()


@pytest.fixture(scope='module')
def wrapper(model, mano_tables):
    return pkg('acr.mano_wrapper').MANOWrapper(tables=mano_tables, engine=model.engine())


def _meta(frames, b):
    return {'image': torch.from_numpy(frames[b:b + 1]), 'offsets': torch.tensor([[512., 512, 0, 0, 0, 0, 0, 0, 0, 0]]),
            'batch_ids': torch.arange(1), 'imgpath': ['f%d' % b]}


def test_model_forward_and_mano_wrapper_match_reference_dict(model, wrapper, frames2):
    g = golden('e2e_batch1.npz')
    for b in range(2):
        out = model(_meta(frames2, b), mode='parsing', calc_loss=False)
        for k in ('l_params_maps', 'r_params_maps', 'l_center_map', 'r_center_map', 'l_prior_maps', 'r_prior_maps', 'segms',
                  'l_params_pred', 'r_params_pred', 'params_pred', 'detection_flag', 'detection_flag_cache',
                  'l_centers_pred', 'r_centers_pred', 'l_centers_conf', 'r_centers_conf', 'left_hand_num',
                  'right_hand_num', 'reorganize_idx', 'output_hand_type', 'params_dict', 'meta_data'):
            assert k in out, k                                             # SURVEY.md §8b key list
        assert tuple(out['segms'].shape) == (1, 33, 256, 256) and tuple(out['l_params_maps'].shape) == (1, 109, 64, 64)
        out = wrapper(out, out['meta_data'])
        np.testing.assert_array_equal(out['detection_flag'].cpu().numpy(), g['f%d_detection_flag' % b])
        np.testing.assert_array_equal(out['l_centers_pred'].cpu().numpy(), g['f%d_l_centers_pred' % b])
        np.testing.assert_array_equal(out['output_hand_type'].cpu().numpy(), g['f%d_output_hand_type' % b])
        np.testing.assert_allclose(out['params_pred'].cpu().numpy(), g['f%d_params_pred' % b], 2e-4, 2e-4)
        for k, tol in (('verts', 1e-4), ('j3d', 1e-4)):
            assert np.abs(out[k].cpu().numpy() - g['f%d_%s' % (b, k)]).max() < tol, k
        np.testing.assert_allclose(out['pj2d'].cpu().numpy(), g['f%d_pj2d' % b], 1e-3, 2e-4)
        np.testing.assert_allclose(out['pj2d_org'].cpu().numpy(), g['f%d_pj2d_org' % b], 1e-3, 5e-2)
        np.testing.assert_allclose(out['cam_trans'].cpu().numpy(), g['f%d_cam_trans' % b], 5e-3, 5e-3)


def test_result_parser_parse_on_maps(model, frames2):
    """ResultParser().parse(outputs, meta_data, cfg) on the head-map dict == the fused decode."""
    maps = model.head_forward(torch.from_numpy(frames2).cuda())
    meta = {'batch_ids': torch.arange(2)}
    rp = pkg('acr.result_parser').ResultParser()
    assert rp.params_num == 109
    out, meta = rp.parse(dict(maps), meta, {})
    fused = model({'image': torch.from_numpy(frames2), 'batch_ids': torch.arange(2)})
    assert torch.equal(out['params_pred'], fused['params_pred'])
    assert out['reorganize_idx'].tolist() == fused['reorganize_idx'].tolist() == [0, 1, 0, 1]


def test_manolayer_surface(mano_tables):
    g = golden('mano_cases.npz')
    ML = pkg('mano.manolayer').ManoLayer
    lay = ML(center_idx=9, flat_hand_mean=False, ncomps=45, side='right', use_pca=False, tables=mano_tables['right'])
    poses, betas = cases.mano_inputs(2, 2)
    v, j, c = lay(torch.from_numpy(poses), th_betas=torch.from_numpy(betas))
    assert np.abs(v.cpu().numpy() - g['n2_r_verts']).max() < 2e-6 and np.abs(c.cpu().numpy() - g['n2_r_center']).max() < 2e-6
    assert tuple(lay.th_faces.shape) == (1538, 3) and lay.th_faces.dtype == torch.int64
    v0, j0, _ = lay(torch.zeros(0, 48), th_betas=torch.zeros(0, 10))
    assert v0.shape == (0, 778, 3)
    with pytest.raises(ValueError):
        ML(side='right', use_pca=False, root_rot_mode='rotmat', tables=mano_tables['right'])     # (broken in the reference too)
    with pytest.raises(ValueError):
        ML(side='right', use_pca=False, joint_rot_mode='quat', tables=mano_tables['right'])      # ('axisang' | 'rotmat')
    assert ML(side='right', use_pca=False, joint_rot_mode='rotmat', tables=mano_tables['right']).joint_rot_mode == 'rotmat'
    with pytest.raises(FileNotFoundError):
        ML(side='left', use_pca=False, mano_root='/nonexistent/')


@pytest.mark.parametrize('name', list(cases.MANO_OPTION_CASES))
def test_manolayer_options_match_reference(name, mano_tables):
    """ManoLayer beyond acr/mano_wrapper.py's configuration - use_pca / ncomps, flat_hand_mean, root_palm, th_trans,
    share_betas (mano/manolayer.py:13-22,104-160,249-276) - against the REAL reference's ManoLayer on the same synthetic
    tables (tests/golden/mano_options.npz, make_golden_batch.py)."""
    g = golden('mano_options.npz')
    kw, opt, n, seed = cases.MANO_OPTION_CASES[name]
    lay = pkg('mano.manolayer').ManoLayer(tables=mano_tables[kw['side']], **kw)
    poses, betas, trans = cases.mano_option_inputs(name)
    args = dict(th_betas=torch.from_numpy(betas))
    if trans is not None:
        args['th_trans'] = torch.from_numpy(trans)
    if opt.get('root_palm'):
        args['root_palm'] = torch.Tensor([1])
    if opt.get('share_betas'):
        args['share_betas'] = torch.Tensor([1])
    v, j, c = lay(torch.from_numpy(poses), **args)
    assert np.abs(v.cpu().numpy() - g[name + '_verts']).max() < 5e-6
    assert np.abs(j.cpu().numpy() - g[name + '_joints']).max() < 5e-6
    if g[name + '_center'].size:
        assert np.abs(c.cpu().numpy() - g[name + '_center']).max() < 5e-6
    else:
        assert c is None


@pytest.mark.parametrize('name', list(cases.MANO_ROTMAT_CASES))
def test_manolayer_rotmat_mode_matches_reference(name, mano_tables):
    """ManoLayer(use_pca=False, joint_rot_mode='rotmat') - [N,16,3,3] pose matrices, projected on the
    host like the reference's batch_rotprojs (mano/manolayer.py:151-162, 436-453), then mano_kernel's rotation-matrix input
    mode (acrmi_mano_rotmat) - against the REAL reference's ManoLayer (tests/golden/mano_rotmat.npz)."""
    g = golden('mano_rotmat.npz')
    kw, n, seed, noise = cases.MANO_ROTMAT_CASES[name]
    lay = pkg('mano.manolayer').ManoLayer(tables=mano_tables[kw['side']], **kw)
    rot, betas = cases.mano_rotmat_inputs(name)
    v, j, c = lay(torch.from_numpy(rot), th_betas=torch.from_numpy(betas))
    assert np.abs(v.cpu().numpy() - g[name + '_verts']).max() < 5e-6
    assert np.abs(j.cpu().numpy() - g[name + '_joints']).max() < 5e-6
    if g[name + '_center'].size:
        assert np.abs(c.cpu().numpy() - g[name + '_center']).max() < 5e-6
    else:
        assert c is None
    with pytest.raises(ValueError):
        lay(torch.zeros(2, 48))          # axis-angle rows are not this mode's input


def test_head_forward_takes_backbone_features_like_the_reference(model, frames2):
    """VERDICT r4 item 8 / acr/model.py:47-53: `model.head_forward(model.backbone(img))` - valid on the reference - runs
    the head ops alone on the uploaded features (acrmi_heads).  Same kernels on the same values: every head map is
    BIT-EQUAL to the one-program result, and the center maps match the reference's fixture."""
    acr = model
    x = torch.from_numpy(frames2).cuda()
    whole = {k: v.clone() for k, v in acr.head_forward(x).items()}
    feats = acr.backbone(x)
    assert feats.shape == (2, 32, 128, 128) and feats.dtype == torch.float32
    hl = acr.engine(2).program['heads']
    for si in range(2):      # poison what the heads must rewrite
        acr.engine(2).buffer(hl.center_buf[si], 2).fill_(float('nan'))
        acr.engine(2).buffer(hl.params_buf[si], 2).fill_(float('nan'))
    acr.engine(2).buffer(hl.backbone_buf, 2)[..., :32].fill_(float('nan'))
    heads = acr.head_forward(feats)
    torch.cuda.synchronize()
    assert set(heads) == set(whole)
    for k in whole:
        assert torch.equal(heads[k], whole[k]), k
    g = golden('net_frame0.npz')
    for k in ('l_center_map', 'r_center_map'):
        np.testing.assert_allclose(heads[k][:1].cpu().numpy(), g[k], rtol=1e-4, atol=1e-4)
    # features of another frame order give that order's maps (nothing is cached from the image pass)
    swapped = acr.head_forward(feats.flip(0).contiguous())
    assert torch.equal(swapped['l_center_map'][0], whole['l_center_map'][1])
    with pytest.raises(ValueError):
        acr.head_forward(torch.zeros(2, 31, 128, 128).cuda())


def test_head_forward_on_the_large_batch_lowering(synth_sd, frames2):
    """contexts built for batches >= 16 write the backbone map as the SECOND output of an ACRMI_CONV_DUAL
    convolution (aux_buf), not through an op's out_buf - acrmi_heads / acrmi_backbone_channels must find that writer too.
    Same checks as the small-batch test on a max_batch = 16 engine (the lowering bench.py times)."""
    acr = pkg('acr.model').ACR(device=0, max_batch=16).eval()
    acr.load_state_dict(synth_sd)
    eng = acr.engine(2)
    L = pkg('_lib')
    info = eng.program['op_info']
    assert any(i.get('kernel') == 'conv_wino24b_kernel' for i in info), 'not the large-batch lowering'
    assert L.lib().acrmi_backbone_channels(eng.ctx) == 32
    x = torch.from_numpy(frames2).cuda()
    whole = {k: v.clone() for k, v in acr.head_forward(x).items()}
    feats = acr.backbone(x)
    assert feats.shape == (2, 32, 128, 128)
    hl = eng.program['heads']
    for si in range(2):
        eng.buffer(hl.center_buf[si], 2).fill_(float('nan'))
        eng.buffer(hl.params_buf[si], 2).fill_(float('nan'))
    eng.buffer(hl.backbone_buf, 2)[..., :32].fill_(float('nan'))
    heads = acr.head_forward(feats)
    torch.cuda.synchronize()
    for k in whole:
        assert torch.equal(heads[k], whole[k]), k


def test_main_acr_results_dict(synth_sd, mano_tables, frames2):
    """acr.main.ACR(...)(bgr_frame, path) -> {path: [float16 hand dicts]} (acr/main.py:92-123, acr/utils.py:1226-1271)."""
    g = golden('e2e_batch1.npz')
    acr = pkg('acr.main').ACR(state_dict=synth_sd, mano_tables=mano_tables, max_batch=2)
    res = acr(np.ascontiguousarray(frames2[0][:, :, ::-1]), 'a.jpg')
    hands = res['a.jpg']
    assert len(hands) == 2 and [int(h['hand_type']) for h in hands] == [0, 1]
    for i, h in enumerate(hands):
        for k in ('cam', 'cam_trans', 'poses', 'betas', 'j3d', 'verts', 'pj2d', 'pj2d_org'):
            assert h[k].dtype == np.float16, k
        assert np.abs(h['verts'].astype(np.float32) - g['f0_verts'][i]).max() < 2e-3       # fp16 packaging
        assert np.abs(h['poses'].astype(np.float32) - g['f0_poses'][i]).max() < 5e-3
    batch = acr.forward_batch(torch.from_numpy(frames2), ['a', 'b'])                    # point heads (default)
    assert np.abs(batch['b'][1]['verts'].astype(np.float32) - g['f1_verts'][1]).max() < 2e-3
    dense = acr.forward_batch(torch.from_numpy(frames2), ['a', 'b'], point_heads=False)
    for path in ('a', 'b'):
        for hp, hd in zip(batch[path], dense[path]):
            assert np.abs(hp['verts'].astype(np.float32) - hd['verts'].astype(np.float32)).max() < 1e-3   # 1 fp16 ulp
    # nothing detected -> {path: {}}
    sd = pkg('synth').make_state_dict(seed=0, center_bias=(-50.0, -50.0))
    acr2 = pkg('acr.main').ACR(state_dict=sd, mano_tables=mano_tables)
    assert acr2(np.ascontiguousarray(frames2[0][:, :, ::-1]), 'n.jpg') == {'n.jpg': {}}


def test_forward_batch_on_a_program_without_point_heads(synth_sd, mano_tables, frames2):
    """acr.main.ACR.forward_batch defaults to point_heads=True, but packer.lower only emits the point-heads ops
    for fp32 HRNet-W32 programs - a 16-bit program has none and acrmi_set_option(ACRMI_OPT_POINT_HEADS, 1) would fail.
    Engine.set_point_heads now falls back to the dense heads (same results) and says so."""
    a = pkg('config').parse_args(['--model_precision', 'fp16'])
    acr = pkg('acr.main').ACR(args_set=a, state_dict=synth_sd, mano_tables=mano_tables, max_batch=2)
    eng = acr.model.engine(2)
    assert not eng.has_point_heads and eng.set_point_heads(True) is False
    batch = acr.forward_batch(torch.from_numpy(frames2), ['a', 'b'])                      # default point_heads=True
    dense = acr.forward_batch(torch.from_numpy(frames2), ['a', 'b'], point_heads=False)
    g = golden('e2e_batch1.npz')
    assert len(batch['a']) == 2 and len(batch['b']) == 2
    for path in ('a', 'b'):
        for hp, hd in zip(batch[path], dense[path]):
            assert np.array_equal(hp['verts'], hd['verts'])
    # the 16-bit contract against the reference's fp32 frames: millimetres, not the fp32 bar
    assert np.abs(batch['a'][0]['verts'].astype(np.float32) - g['f0_verts'][0]).max() < 5e-3
    # the fp32 engine of the other tests does carry them
    acr32 = pkg('acr.main').ACR(state_dict=synth_sd, mano_tables=mano_tables, max_batch=2)
    assert acr32.model.engine(2).has_point_heads


@pytest.mark.parametrize('name', list(cases.DECODE_BATCHES))
def test_result_parser_reference_batch_semantics(name):
    """ResultParser(batch_semantics='reference').parse on device maps of a batch > 1 with mixed detection states returns
    the rows the real reference returns (tests/golden/decode_batches.npz, captured by make_golden_batch.py): prior gated on
    every flag of the batch, determine_coeff from row 0 of each side's list, whole-batch placeholder rows
    (acr/result_parser.py:42-47,102-131).  The default per-frame mode differs where the fixture says it must."""
    g = golden('decode_batches.npz')
    rp = pkg('acr.result_parser')
    maps = cases.decode_batch_maps(name)
    B = maps['l_center_map'].shape[0]
    res = {}
    for mode in ('reference', 'frame'):
        outputs = {k: torch.from_numpy(v).cuda() for k, v in maps.items()}
        meta = {'batch_ids': torch.arange(B), 'offsets': torch.zeros(B, 10), 'imgpath': ['f%d' % b for b in range(B)]}
        out, meta = rp.ResultParser(batch_semantics=mode).parse(outputs, meta, {})
        res[mode] = out['params_pred'].cpu().numpy()
        if mode == 'frame':
            continue
        np.testing.assert_array_equal(out['detection_flag'].cpu().numpy(), g[name + '_detection_flag'])
        np.testing.assert_array_equal(out['reorganize_idx'].cpu().numpy(), g[name + '_reorganize_idx'])
        np.testing.assert_array_equal(out['output_hand_type'].cpu().numpy(), g[name + '_hand_type'])
        np.testing.assert_allclose(res[mode], g[name + '_params_pred'], rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(out['params_dict']['cam'].cpu().numpy(), g[name + '_cam'], rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(out['params_dict']['poses'].cpu().numpy(), g[name + '_poses'], rtol=2e-5, atol=2e-6)
        np.testing.assert_allclose(out['params_dict']['betas'].cpu().numpy(), g[name + '_betas'], rtol=1e-6, atol=1e-6)
        np.testing.assert_array_equal(out['l_centers_pred'].cpu().numpy(), g[name + '_l_centers_pred'])
        np.testing.assert_array_equal(out['r_centers_pred'].cpu().numpy(), g[name + '_r_centers_pred'])
    differs = np.abs(res['frame'] - res['reference']).max() > 1e-3
    assert differs == (name in ('b2_far_near', 'b2_near_far', 'b2_leftonly_dist32', 'b3_rightonly_near_none', 'b4_mixed')), name


def test_model_forward_reference_batch_semantics(synth_sd, frames2):
    """acr.model.ACR(batch_semantics='reference').forward at batch 2 goes through the gated second decode: on the two
    synthetic frames (both hands in both frames) the batch-wide gate equals the per-frame one unless the frames' center
    distances fall on different sides of 32 px - either way the rows must equal a hand-gated decode of the same maps."""
    rp = pkg('acr.result_parser')
    m = pkg('acr.model').ACR(device=0, max_batch=2, batch_semantics='reference').eval()
    m.load_state_dict(synth_sd)
    meta = {'image': torch.from_numpy(frames2), 'offsets': torch.tensor([[512., 512, 0, 0, 0, 0, 0, 0, 0, 0]] * 2),
            'batch_ids': torch.arange(2), 'imgpath': ['a', 'b']}
    out = m.cuda()(meta, mode='parsing', calc_loss=False)
    eng = m.engine(2)
    first = eng.decode(2)
    gate = rp.reference_prior_gate(first)
    want = rp.rows_from_slots(eng.decode(2, prior_gate=gate))
    assert torch.equal(out['params_pred'], want['params_pred']) and out['params_pred'].shape[0] == 4
    forced_off = rp.rows_from_slots(eng.decode(2, prior_gate=torch.zeros(2, dtype=torch.int32)))
    if int(gate.sum()) > 0:
        assert (forced_off['params_pred'] - want['params_pred']).abs().max().item() > 1e-4     # the gate is live


@pytest.mark.parametrize('name', list(cases.E2E_BATCHES))
def test_model_forward_at_batch_gt_1_matches_the_reference(name, mano_tables):
    """acr.model.ACR(batch_semantics='reference').forward(meta_data) + MANOWrapper on a batch of 4 / 3 / 2 frames returns what
    the REAL reference's model.forward + MANOWrapper returned on the same batch (tests/golden/e2e_batches.npz): rows, their
    order (all left rows, then the right rows), the whole-batch placeholder row, reorganize_idx, meshes."""
    g = golden('e2e_batches.npz')
    seed, B = cases.E2E_BATCHES[name]
    m = pkg('acr.model').ACR(device=0, max_batch=B, batch_semantics='reference').eval()
    m.load_state_dict(pkg('synth').make_state_dict(seed=seed))
    m.cuda()
    wrapper = pkg('acr.mano_wrapper').MANOWrapper(tables=mano_tables, engine=m.engine())
    frames = torch.from_numpy(pkg('synth').make_frames(B, seed=cases.STATE_FRAME_SEED))
    meta = {'image': frames, 'offsets': torch.tensor([[512., 512, 0, 0, 0, 0, 0, 0, 0, 0]] * B), 'batch_ids': torch.arange(B),
            'imgpath': ['b%d' % b for b in range(B)]}
    out = m(meta, mode='parsing', calc_loss=False)
    np.testing.assert_array_equal(out['detection_flag'].cpu().numpy(), g[name + '_detection_flag'])
    np.testing.assert_array_equal(out['reorganize_idx'].cpu().numpy(), g[name + '_reorganize_idx'])
    np.testing.assert_array_equal(out['output_hand_type'].cpu().numpy(), g[name + '_output_hand_type'])
    np.testing.assert_array_equal(out['l_centers_pred'].cpu().numpy(), g[name + '_l_centers_pred'])
    np.testing.assert_array_equal(out['r_centers_pred'].cpu().numpy(), g[name + '_r_centers_pred'])
    assert [int(out['left_hand_num']), int(out['right_hand_num'])] == g[name + '_hand_nums'].tolist()
    np.testing.assert_allclose(out['params_pred'].cpu().numpy(), g[name + '_params_pred'], 2e-4, 2e-4)
    assert list(out['meta_data']['imgpath']) == ['b%d' % b for b in g[name + '_reorganize_idx']]
    if out['detection_flag'].sum() > 0:                   # acr/main.py:96
        out = wrapper(out, out['meta_data'])
        for k in ('verts', 'j3d'):
            assert np.abs(out[k].cpu().numpy() - g[name + '_' + k]).max() < 1e-4, k
        np.testing.assert_allclose(out['pj2d'].cpu().numpy(), g[name + '_pj2d'], 1e-3, 2e-4)
        np.testing.assert_allclose(out['cam_trans'].cpu().numpy(), g[name + '_cam_trans'], 5e-3, 5e-3)
    m.engine().close()


@pytest.mark.parametrize('B', [1, 2, 3, 64, 300])
def test_prior_gate_kernel_equals_the_host_statement_of_the_rule(B):
    """acrmi_prior_gate (the batch-wide prior decision of acr/result_parser.py:42-47,102-145 on the device)
    == result_parser.reference_prior_gate (the host statement pinned to the real reference's rows by
    tests/test_host_api.py::test_reference_batch_semantics_host_logic) on random detection states - including batches without
    any left / any right detection, first detections in different frames, and centers exactly 32 px apart."""
    rp, ops, L = pkg('acr.result_parser'), pkg('ops'), pkg('_lib')
    rs = np.random.RandomState(B)
    for trial in range(12):
        slots = rs.randn(B, 2, L.SLOT).astype(np.float32)
        p_on = (0.0, 0.15, 0.5, 0.9)[trial % 4]
        slots[:, :, L.SLOT_FLAG] = (rs.rand(B, 2) < p_on).astype(np.float32)
        if trial == 5:
            slots[:, 0, L.SLOT_FLAG] = 0           # no left hand in the whole batch
        if trial == 6:
            slots[:, 1, L.SLOT_FLAG] = 0
        flat = rs.randint(0, 4096, size=(B, 2))
        if trial >= 8:                             # distances around the 32 px bound (exactly 32: still a prior)
            flat[:, 0] = 10 * 64 + 5
            flat[:, 1] = 10 * 64 + 5 + (32 if trial % 2 == 0 else 33)
        slots[:, :, L.SLOT_FLATIND] = flat
        dev = torch.from_numpy(slots).cuda()
        got = ops.prior_gate(dev).cpu().numpy()
        want = rp.reference_prior_gate(dev).cpu().numpy()
        np.testing.assert_array_equal(got, want, err_msg='B %d trial %d' % (B, trial))


@pytest.mark.parametrize('name', list(cases.E2E_BATCHES))
def test_fused_forward_applies_reference_batch_semantics_in_one_call(name, mano_tables):
    """Engine.forward / acrmi_forward with ACRMI_OPT_BATCH_PRIOR = decode -> acrmi_prior_gate -> gated decode
    -> MANO inside the ONE fused call.  Slots are bit-equal to the three-call path (decode, host rule, gated decode); the
    meshes are those of the real reference's batch (tests/golden/e2e_batches.npz); 'frame' mode stays the plain decode."""
    g = golden('e2e_batches.npz')
    rp, L = pkg('acr.result_parser'), pkg('_lib')
    seed, B = cases.E2E_BATCHES[name]
    eng = pkg('engine').Engine(0)
    eng.load_state_dict(pkg('synth').make_state_dict(seed=seed), max_batch=B)
    tables = {k: dict(v) for k, v in mano_tables.items()}
    tables['left']['shapedirs'] = tables['left']['shapedirs'].copy()
    tables['left']['shapedirs'][:, 0, :] *= -1          # acr/mano_wrapper.py:35 (MANOWrapper does this itself)
    eng.load_mano(tables)
    frames = torch.from_numpy(pkg('synth').make_frames(B, seed=cases.STATE_FRAME_SEED)).cuda()
    eng.set_batch_semantics('reference')
    out = eng.forward(frames)
    torch.cuda.synchronize()
    first = eng.decode(B)
    want = eng.decode(B, prior_gate=rp.reference_prior_gate(first))
    assert torch.equal(out['slots'], want)
    # the real reference's rows: row i = frame reorganize_idx[i], hand output_hand_type[i]
    flags = g[name + '_detection_flag']
    for i, (b, h) in enumerate(zip(g[name + '_reorganize_idx'], g[name + '_output_hand_type'])):
        if flags[i]:
            assert out['slots'][b, h, L.SLOT_FLAG].item() > 0.5
            assert np.abs(out['verts'][b, h].cpu().numpy() - g[name + '_verts'][i]).max() < 1e-4, (name, i)
    eng.set_batch_semantics('frame')
    plain = eng.forward(frames)
    torch.cuda.synchronize()
    assert torch.equal(plain['slots'], first)
    with pytest.raises(ValueError):
        eng.set_batch_semantics('whole-batch')
    eng.close()


def test_cam_trans_kernel_matches_reference_least_squares():
    """§8f-3: acrmi_cam_trans == the reference's closed-form least squares (acr/utils.py:430-472; restated in numpy
    fp64 in oracle/smooth.py and pinned against the reference's cam_trans in test_oracle_pinned)."""
    from oracle import smooth as osm
    u, ops = pkg('acr.utils'), pkg('ops')
    rs = np.random.RandomState(3)
    n = 37
    j3 = (rs.randn(n, 21, 3) * 0.08).astype(np.float32)
    t_true = np.stack([rs.uniform(-0.3, 0.3, n), rs.uniform(-0.3, 0.3, n), rs.uniform(0.4, 3.0, n)], 1)
    f = 1265.0
    p = j3.astype(np.float64) + t_true[:, None]
    pj2d = ((f * p[..., :2] / p[..., 2:] + 256) / 256 - 1 + rs.randn(n, 21, 2) * 2e-3).astype(np.float32)
    want = np.stack([osm.estimate_translation_np(j3[i].astype(np.float64), (pj2d[i].astype(np.float64) + 1) * 256,
                                                 np.ones(21, np.float32), focal_length=f) for i in range(n)])
    got = ops.cam_trans(torch.from_numpy(j3).cuda(), torch.from_numpy(pj2d).cuda(), focal_length=f).cpu().numpy()
    np.testing.assert_allclose(got, want, rtol=2e-5, atol=2e-6)
    assert np.abs(got - t_true).max() < 0.05                       # and it recovers the translation
    assert ops.cam_trans(torch.zeros(0, 21, 3).cuda(), torch.zeros(0, 21, 2).cuda()).shape == (0, 3)
    # the product path (MANOWrapper -> estimate_translation) takes the device kernel for device tensors
    t2 = u.estimate_translation(torch.from_numpy(j3).cuda(), torch.from_numpy(pj2d).cuda(), focal_length=f)
    assert t2.is_cuda and np.allclose(t2.cpu().numpy(), got)


def _blocky(rs, H, W):
    base = rs.randint(0, 256, (H // 8 + 2, W // 8 + 2, 3)).astype(np.float32)
    frame = np.kron(base, np.ones((8, 8, 1), np.float32))[:H, :W] + rs.randint(-9, 10, (H, W, 3))
    return np.clip(frame, 0, 255).astype(np.uint8)


def test_gpu_preprocess_is_bit_exact_to_the_opencv_restatement():
    """§8f-1: acrmi_preprocess == oracle/preprocess.py (imgaug 0.4.0 pad rule + OpenCV's uint8 fixed-point
    INTER_CUBIC restated from its published source) bit for bit, on up- and down-scaling, tall and wide frames."""
    from oracle import preprocess as opre
    ops = pkg('ops')
    rs = np.random.RandomState(4)
    for H, W in ((1080, 1920), (480, 640), (700, 301), (512, 512), (333, 333), (64, 48)):
        frame = _blocky(rs, H, W)
        want_img, want_off = opre.img_preprocess(frame)
        dev_img, dev_off = ops.preprocess(torch.from_numpy(frame)[None].cuda())
        assert torch.equal(dev_img[0].cpu(), torch.from_numpy(want_img)), (H, W)
        assert torch.equal(dev_off[0], torch.from_numpy(want_off)), (H, W)
    noise = rs.randint(0, 256, (2, 720, 1280, 3)).astype(np.uint8)         # batched call, worst-case content
    dev_img, _ = ops.preprocess(torch.from_numpy(noise).cuda())
    for i in range(2):
        assert torch.equal(dev_img[i].cpu(), torch.from_numpy(opre.img_preprocess(noise[i])[0]))


def test_gpu_preprocess_of_mixed_size_frames_in_one_call():
    """acrmi_preprocess_frames - per-frame geometry, ONE call for a folder of images of different sizes
    (img_preprocess is per image, acr/utils.py:1315-1337; acr/main.py:144-205 mixes sizes) - bit-exact to oracle/preprocess.py
    frame by frame, offsets rows included; frames live in separate allocations; > 128 frames take several launches."""
    from oracle import preprocess as opre
    ops, u = pkg('ops'), pkg('acr.utils')
    rs = np.random.RandomState(11)
    sizes = ((480, 640), (1080, 1920), (333, 517), (512, 512), (700, 301), (64, 48), (301, 700), (2, 3))
    frames = [_blocky(rs, H, W) if min(H, W) > 8 else rs.randint(0, 256, (H, W, 3)).astype(np.uint8) for H, W in sizes]
    dev = [torch.from_numpy(f).cuda() for f in frames]
    img, off = ops.preprocess_frames(dev)
    assert img.shape == (len(sizes), 512, 512, 3) and off.shape == (len(sizes), 10)
    for i, f in enumerate(frames):
        want_img, want_off = opre.img_preprocess(f)
        assert torch.equal(img[i].cpu(), torch.from_numpy(want_img)), sizes[i]
        assert torch.equal(off[i], torch.from_numpy(want_off)), sizes[i]
    # the equal-size entry point computes the same bytes
    same, _ = ops.preprocess(dev[0][None])
    assert torch.equal(same[0], img[0])
    # 130 small frames: two launches (128 geometry records per launch), every frame its own size
    many = [rs.randint(0, 256, (20 + i % 7, 31 + i % 5, 3)).astype(np.uint8) for i in range(130)]
    img2, off2 = ops.preprocess_frames([torch.from_numpy(f).cuda() for f in many])
    for i in (0, 1, 127, 128, 129):
        want_img, want_off = opre.img_preprocess(many[i])
        assert torch.equal(img2[i].cpu(), torch.from_numpy(want_img)), i
        assert torch.equal(off2[i], torch.from_numpy(want_off)), i
    meta = u.img_preprocess_gpu(dev[:3], ['a', 'b', 'c'])                 # the acr.utils surface takes the list form
    assert torch.equal(meta['image'], img[:3]) and meta['imgpath'] == ['a', 'b', 'c']
    with pytest.raises(ValueError):
        ops.preprocess_frames([])
    with pytest.raises(ValueError):
        ops.preprocess_frames([torch.zeros(4, 4, 4, dtype=torch.uint8).cuda()])


def test_magic_jpg_end_to_end_matches_reference(synth_sd, mano_tables):
    """BASELINE.json configs[0]: the reference's demo image through acr.main.ACR - JPEG decoded with PIL, device
    pre-processing, backbone, heads, decode, MANO - against what the real reference's img_preprocess + ACR +
    MANOWrapper produced from the same pixels (tests/golden/magic_e2e.npz)."""
    import os
    from PIL import Image
    from conftest import GOLDEN
    g = golden('magic_e2e.npz')
    bgr = np.ascontiguousarray(np.asarray(Image.open(os.path.join(GOLDEN, 'magic.jpg')).convert('RGB'))[:, :, ::-1])
    u = pkg('acr.utils')
    meta = u.img_preprocess(bgr, 'demo/magic.jpg', single_img_input=True)
    assert meta['image'].is_cuda and meta['offsets'].tolist() == g['offsets'].tolist()
    np.testing.assert_array_equal(meta['image'][0].cpu().numpy()[::4, ::4], g['image_sub'])
    assert int(meta['image'].long().sum()) == int(g['image_sum'][0])
    acr = pkg('acr.main').ACR(state_dict=synth_sd, mano_tables=mano_tables)
    out = acr.single_image_forward(bgr, 'demo/magic.jpg')
    out, res = acr.process_results(out)
    np.testing.assert_array_equal(out['detection_flag_cache'].cpu().numpy(), g['detection_flag'].astype(bool))
    np.testing.assert_array_equal(out['l_centers_pred'].cpu().numpy(), g['l_centers_pred'])
    np.testing.assert_allclose(out['params_pred'].cpu().numpy(), g['params_pred'], 2e-4, 2e-4)
    assert np.abs(out['verts'].cpu().numpy() - g['verts']).max() < 1e-4
    assert np.abs(out['j3d'].cpu().numpy() - g['j3d']).max() < 1e-4
    np.testing.assert_allclose(out['pj2d_org'].cpu().numpy(), g['pj2d_org'], 1e-3, 0.2)      # pixels of a 1920 frame
    np.testing.assert_allclose(out['cam_trans'].cpu().numpy(), g['cam_trans'], 5e-3, 5e-3)
    assert len(res['demo/magic.jpg']) == int(g['detection_flag'].sum())


def test_device_smoothing_matches_reference_sequence():
    """§8f-3: acrmi_smooth (One-Euro state in the context) == the reference's smooth_results over a 14-frame
    sequence (late-appearing hand, two-frame drop-out, near-pi orientation) - frame by frame as acr/main.py calls
    it, and as one 14-frame call of a video shard."""
    L = pkg('_lib')
    g = golden('smooth_seq.npz')
    poses, betas, flags = cases.smooth_inputs()
    T = poses.shape[0]
    slots = torch.zeros(T, 2, L.SLOT)
    slots[:, :, L.SLOT_FLAG] = torch.from_numpy(flags.astype(np.float32))
    slots[:, :, L.SLOT_POSES:L.SLOT_POSES + 48] = torch.from_numpy(poses)
    slots[:, :, L.SLOT_BETAS:L.SLOT_BETAS + 10] = torch.from_numpy(betas)
    eng = pkg('engine').Engine(0)
    eng.set_temporal(False, smooth_coeff=float(g['smooth_coeff']))
    one = slots.clone().cuda()
    eng.smooth_reset()
    for t in range(T):
        eng.smooth(one[t:t + 1])
    allatonce = slots.clone().cuda()
    eng.smooth_reset()
    eng.smooth(allatonce)
    torch.cuda.synchronize()
    assert torch.equal(one, allatonce)
    got = one.cpu().numpy()
    assert np.abs(got[:, :, L.SLOT_POSES:L.SLOT_POSES + 48] - g['poses']).max() < 1e-5
    assert np.abs(got[:, :, L.SLOT_BETAS:L.SLOT_BETAS + 10] - g['betas']).max() < 1e-5
    untouched = np.ones(L.SLOT, bool)
    untouched[L.SLOT_POSES:L.SLOT_BETAS + 10] = False
    np.testing.assert_array_equal(got[:, :, untouched], slots.numpy()[:, :, untouched])
    eng.close()


def test_temporal_optimization_through_acr_main(synth_sd, mano_tables, frames2):
    """`-t`: acr.main.ACR smooths the decoded poses on the device before MANO (acr/main.py:69-83).  Three calls on
    alternating frames == the oracle's One-Euro filter run over the un-smoothed per-frame parameters."""
    from oracle import smooth as osm
    cfg = pkg('config')
    base = ['--configs_yml', '/nonexistent.yml']
    plain = pkg('acr.main').ACR(args_set=cfg.parse_args(base), state_dict=synth_sd, mano_tables=mano_tables)
    temp = pkg('acr.main').ACR(args_set=cfg.parse_args(base + ['-t', '--smooth_coeff', '3.0']), state_dict=synth_sd,
                               mano_tables=mano_tables)
    seq = [frames2[0], frames2[1], frames2[0]]
    filt = {0: osm.new_filters(3.0), 1: osm.new_filters(3.0)}
    for t, f in enumerate(seq):
        bgr = np.ascontiguousarray(f[:, :, ::-1])
        raw = plain.process_results(plain.single_image_forward(bgr, 'p'))[0]
        sm = temp.process_results(temp.single_image_forward(bgr, 'p'))[0]
        for sid in range(2):
            p, b = osm.smooth_results(filt[sid], raw['params_dict']['poses'][sid].cpu().numpy(),
                                      raw['params_dict']['betas'][sid].cpu().numpy())
            assert np.abs(sm['params_dict']['poses'][sid].cpu().numpy() - p).max() < 2e-5, (t, sid)
            assert np.abs(sm['params_dict']['betas'][sid].cpu().numpy() - b).max() < 2e-5, (t, sid)
        if t == 0:
            assert torch.allclose(sm['verts'], raw['verts'], atol=1e-6)         # first sample passes through
        else:
            assert (sm['verts'] - raw['verts']).abs().max() > 1e-5              # later ones are filtered


def test_options_conf_thresh_and_center_idx(synth_sd, mano_tables, frames2):
    """centermap_conf_thresh and align_idx / mano_mesh_root_align reach the kernels of BOTH entry points
    (ACR.forward and the fused forward_batch) instead of being hard-coded."""
    cfg = pkg('config')
    base = ['--configs_yml', '/nonexistent.yml']
    g = golden('net_frame0.npz')
    l_max = float(g['l_center_map'].max())                   # 0.95: a threshold above it drops the left hand only
    acr = pkg('acr.main').ACR(args_set=cfg.parse_args(base + ['--centermap_conf_thresh', '%.3f' % (l_max + 0.02)]),
                              state_dict=synth_sd, mano_tables=mano_tables, max_batch=2)
    res = acr(np.ascontiguousarray(frames2[0][:, :, ::-1]), 'a')['a']
    assert [int(h['hand_type']) for h in res] == [1]
    batch = acr.forward_batch(torch.from_numpy(frames2), ['a', 'b'])
    assert [int(h['hand_type']) for h in batch['a']] == [1] and [int(h['hand_type']) for h in batch['b']] == [1]
    # root alignment: joint 0 instead of 9, and none at all
    for extra, idx in ((['--align_idx', '0'], 0), (['--mano_mesh_root_align', 'false'], None)):
        acr = pkg('acr.main').ACR(args_set=cfg.parse_args(base + extra), state_dict=synth_sd, mano_tables=mano_tables,
                                  max_batch=2)
        one = acr(np.ascontiguousarray(frames2[0][:, :, ::-1]), 'a')['a']
        fused = acr.forward_batch(torch.from_numpy(frames2[:1]), ['a'], point_heads=False)['a']
        for h1, h2 in zip(one, fused):
            assert np.abs(h1['j3d'].astype(np.float32) - h2['j3d'].astype(np.float32)).max() < 2e-3
            if idx is not None:
                assert np.abs(h1['j3d'][idx].astype(np.float32)).max() < 1e-6      # the alignment joint sits at the origin
            else:
                assert np.abs(h1['j3d'][9].astype(np.float32)).max() > 1e-3
    eng = acr.model.engine()
    with pytest.raises(ValueError):
        eng.set_center_idx(21)


def test_checkpoint_reload_keeps_mano_and_wrapper_working(synth_sd, mano_tables, frames2):
    """ACR_v1.load_state_dict rebuilds the engine; the new context adopts the MANO tables and options and
    MANOWrapper follows the model's current engine instead of the retired one."""
    g = golden('e2e_batch1.npz')
    acr = pkg('acr.main').ACR(state_dict=pkg('synth').make_state_dict(seed=3), mano_tables=mano_tables)
    first = acr.model.engine()
    acr.model.load_state_dict({'module.' + k: v for k, v in synth_sd.items()})
    res = acr(np.ascontiguousarray(frames2[0][:, :, ::-1]), 'a')['a']
    assert acr.model.engine() is not first and acr.mano_regression.engine() is acr.model.engine()
    for i, h in enumerate(res):
        assert np.abs(h['verts'].astype(np.float32) - g['f0_verts'][i]).max() < 2e-3
    fused = acr.forward_batch(torch.from_numpy(frames2[:1]), ['a'])['a']          # needs both sides' tables in the ctx
    assert 'cam_trans' in fused[0] and fused[0]['cam_trans'].dtype == np.float16
    assert np.abs(fused[0]['cam_trans'].astype(np.float32) - res[0]['cam_trans'].astype(np.float32)).max() < 2e-2


def test_allgather_through_the_c_abi_world_size_1():
    """acrmi_comm_unique_id / acrmi_comm_init / acrmi_allgather (RCCL resolved at run time) with one
    rank: the gather is the identity, on the caller's stream and on a side stream."""
    import ctypes as C
    L = pkg('_lib')
    eng = pkg('engine').Engine(0)
    uid = (C.c_char * 128)()
    L.check(L.lib().acrmi_comm_unique_id(uid))
    assert any(bytes(uid))
    send = torch.arange(40000, dtype=torch.float32, device='cuda') * 0.5
    recv = torch.zeros_like(send)
    with pytest.raises(L.AcrmiError):
        eng.comm_ranks = 1
        eng.allgather(send, recv)                               # no communicator yet: ACRMI_ESTATE, not a crash
    eng.comm_init(1, 0, bytes(uid))
    eng.allgather(send, recv)
    side = torch.cuda.Stream()
    recv2 = torch.zeros_like(send)
    side.wait_stream(torch.cuda.current_stream())
    eng.allgather(send, recv2, stream=side)
    torch.cuda.synchronize()
    assert torch.equal(recv, send) and torch.equal(recv2, send)
    eng.close()


def test_sharded_runner_over_rccl_world_size_1(synth_sd, mano_tables, frames2):
    """parallel.ShardedRunner on the GPU with both transports (torch.distributed nccl = RCCL, and acrmi_allgather)
    at world size 1: pipelined submit/collect returns exactly Engine.forward's results."""
    import os
    import torch.distributed as dist
    parallel = pkg('parallel')
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29533', RANK='0', WORLD_SIZE='1')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    try:
        eng = pkg('engine').Engine(0)
        eng.load_state_dict(synth_sd, max_batch=2)
        t = {k: dict(v) for k, v in mano_tables.items()}
        t['left']['shapedirs'] = t['left']['shapedirs'].copy()
        t['left']['shapedirs'][:, 0, :] *= -1
        eng.load_mano(t)
        x = torch.from_numpy(frames2).cuda()
        want = {k: v.clone() for k, v in eng.forward(x).items()}
        for transport in ('torch', 'c'):
            if transport == 'c':
                # the two-communicator hazard is refused by default ...
                with pytest.raises(pkg('_lib').AcrmiError):
                    parallel.ShardedRunner(lambda f, views: eng.forward(f, out=views), eng.device, engine=eng, transport='c')
            # ... and overridable: results are unaffected, only the side-stream gather gets slower
            runner = parallel.ShardedRunner(lambda f, views: eng.forward(f, out=views), eng.device, engine=eng,
                                            transport=transport, allow_second_communicator=True)
            t0 = runner.submit(x)
            t1 = runner.submit(x.flip(0).contiguous())
            r0, r1 = runner.collect(t0), runner.collect(t1)
            torch.cuda.synchronize()
            for k in ('slots', 'verts', 'joints'):
                assert torch.equal(r0[k], want[k]), (transport, k)
                assert torch.equal(r1[k], want[k].flip(0)), (transport, k)
        # the batches computed by an EnginePool (two contexts in turn on their own streams): the gather waits for the
        # event the pool returns
        pool = pkg('engine').EnginePool(0, n=2)
        pool.load_state_dict(synth_sd, max_batch=2)
        pool.load_mano(t)
        runner = parallel.ShardedRunner(lambda f, views: pool.release(pool.submit(f, out=views)), pool.device, engine=eng)
        tickets = [runner.submit(x)]
        got = []
        for i in range(1, 5):
            tickets.append(runner.submit(x.flip(0).contiguous() if i & 1 else x))
            got.append(runner.collect(tickets[i - 1]))
        got.append(runner.collect(tickets[-1]))
        torch.cuda.synchronize()
        for i, r in enumerate(got):
            for k in ('slots', 'verts', 'joints'):
                assert torch.equal(r[k], want[k].flip(0) if i & 1 else want[k]), (i, k)
        pool.close()
        eng.close()
    finally:
        dist.destroy_process_group()


def test_engine_pool_batches_equal_single_context(synth_sd, mano_tables):
    """engine.EnginePool: contexts taking batches in turn on their own streams return, batch for batch, exactly what one
    Engine.forward call returns - three contexts, seven different batches, at most three tickets outstanding; a fourth
    submit without a collect is refused."""
    t = {k: dict(v) for k, v in mano_tables.items()}
    t['left']['shapedirs'] = t['left']['shapedirs'].copy()
    t['left']['shapedirs'][:, 0, :] *= -1
    eng = pkg('engine').Engine(0)
    eng.load_state_dict(synth_sd, max_batch=3)
    eng.load_mano(t)
    pool = pkg('engine').EnginePool(0, n=3)
    pool.load_state_dict(synth_sd, max_batch=3)
    pool.load_mano(t)
    batches = [torch.from_numpy(pkg('synth').make_frames(3, seed=40 + i, structured=True)).cuda() for i in range(7)]
    want = [{k: v.clone() for k, v in eng.forward(b).items()} for b in batches]
    pending, got = [], []
    offs = torch.tensor([[512., 512., 0, 0, 0, 0, 0, 0, 0, 0]] * 3)          # a HOST tensor: converted before the hand-over
    proj = pool.collect(pool.submit(batches[0], offsets=offs, project=True))
    ref = eng.forward(batches[0], offsets=offs, project=True)
    torch.cuda.synchronize()
    for k in ('verts_camed', 'pj2d', 'pj2d_org'):
        assert torch.equal(proj[k], ref[k]), k
    for b in batches:
        pending.append(pool.submit(b))
        if len(pending) == 3:
            with pytest.raises(RuntimeError):
                pool.submit(b)
            got.append({k: v.clone() for k, v in pool.collect(pending.pop(0)).items()})
    while pending:
        got.append({k: v.clone() for k, v in pool.collect(pending.pop(0)).items()})
    torch.cuda.synchronize()
    for w, g in zip(want, got):
        for k in ('slots', 'verts', 'joints'):
            assert torch.equal(w[k], g[k]), k
    # the pool's contexts share ONE device copy of the weight blob (acrmi_share_weights): closing the context that
    # uploaded it leaves the others working, and a sharing request across devices / without weights is refused
    L = pkg('_lib')
    assert L.lib().acrmi_share_weights(pool.engines[1].ctx, pool.engines[1].ctx) == -1      # ACRMI_EINVAL
    fresh = pkg('engine').Engine(0)
    assert L.lib().acrmi_share_weights(pool.engines[1].ctx, fresh.ctx) != 0      # the donor holds no weights
    fresh.close()
    pool.engines[0].close()
    again = pool.engines[2].forward(batches[3])
    torch.cuda.synchronize()
    assert torch.equal(again['verts'], want[3]['verts'])
    pool.engines = pool.engines[1:]
    pool.close()
    eng.close()


def test_engine_pool_keeps_host_offsets_and_temporaries_alive(synth_sd, mano_tables):
    """two submits with DIFFERENT host offsets and temporary frame tensors are pipelined before anything is
    collected, the tickets are released at once (as parallel.ShardedRunner does) and the caller's stream allocates and
    overwrites new tensors meanwhile: the device copies of `offsets` / `img` the queued batches read must stay alive
    until the batches have run (EnginePool keeps released tickets referenced until their event has completed)."""
    t = {k: dict(v) for k, v in mano_tables.items()}
    t['left']['shapedirs'] = t['left']['shapedirs'].copy()
    t['left']['shapedirs'][:, 0, :] *= -1
    eng = pkg('engine').Engine(0)
    eng.load_state_dict(synth_sd, max_batch=8)
    eng.load_mano(t)
    pool = pkg('engine').EnginePool(0, n=2)
    pool.load_state_dict(synth_sd, max_batch=8)
    pool.load_mano(t)
    frames = [pkg('synth').make_frames(8, seed=60 + i, structured=True) for i in range(2)]
    offs = [torch.tensor([[512., 512., 0, 0, 0, 0, 0, 0, 0, 0]] * 8),
            torch.tensor([[1920., 1920., 0, 0, 0, 0, 420, 0, 420, 0]] * 8)]
    want = [eng.forward(torch.from_numpy(f).cuda(), offsets=o, project=True)['pj2d_org'].clone() for f, o in zip(frames, offs)]
    torch.cuda.synchronize()
    outs, events = [], []
    for f, o in zip(frames, offs):
        tk = pool.submit(torch.from_numpy(f).cuda().flip(0).flip(0).contiguous(), offsets=o, project=True)   # temporaries
        outs.append(tk['out'])
        events.append(pool.release(tk))
        del tk
        for _ in range(4):        # the caller's stream reuses whatever the allocator believes is free
            junk = torch.full((8, 10), float('nan'), device='cuda')
            junk2 = torch.zeros(8, 512, 512, 3, dtype=torch.uint8, device='cuda')
            del junk, junk2
    for e in events:
        torch.cuda.current_stream().wait_event(e)
    torch.cuda.synchronize()
    for w, g in zip(want, outs):
        assert torch.equal(w, g['pj2d_org'])
    assert float((want[0] - want[1]).abs().max()) > 1.0          # the two offsets rows really differ in the result
    pool.close()
    eng.close()


def test_malformed_programs_are_rejected(synth_sd):
    """acrmi_set_program validates buffer ids (incl. FUSESUM terms and the head layout), weight offsets
    and channel slices - ACRMI_EINVAL instead of a device fault."""
    import ctypes as C
    L, packer = pkg('_lib'), pkg('packer')
    prog = packer.lower(synth_sd)
    eng = pkg('engine').Engine(0)
    blob = prog['blob']
    L.check(L.lib().acrmi_load_weights(eng.ctx, blob.ctypes.data_as(C.c_void_p), blob.size), eng.ctx)
    bufs = (L.BufferDesc * len(prog['bufs']))(*[L.BufferDesc(*b) for b in prog['bufs']])

    def try_program(mutate):
        ops = (L.Op * len(prog['ops']))(*[L.Op.from_buffer_copy(bytes(o)) for o in prog['ops']])
        heads = L.HeadLayout.from_buffer_copy(bytes(prog['heads']))
        mutate(ops, heads)
        return L.lib().acrmi_set_program(eng.ctx, bufs, len(bufs), ops, len(ops), C.byref(heads), 1)
    conv = next(i for i, o in enumerate(prog['ops']) if o.kind == L.OP_CONV and o.ksize == 3)
    # round 5: the HR fuse sums of the large-batch program are extra residual terms of convolutions (nterms / term_buf) and the
    # second output of branch 0's last conv2 (ACRMI_CONV_DUAL, aux_buf): the same checks apply to them
    hosted = next(i for i, o in enumerate(prog['ops']) if o.kind == L.OP_CONV and o.nterms and not (o.flags & L.CONV_DUAL))
    dual = next(i for i, o in enumerate(prog['ops']) if o.kind == L.OP_CONV and o.flags & L.CONV_DUAL)
    fuse = hosted

    def set_(i, field, val):
        return lambda ops, heads: setattr(ops[i], field, val)

    def term(ops, heads):
        ops[fuse].term_buf[0] = len(prog['bufs']) + 3
    stem = next(i for i, o in enumerate(prog['ops']) if o.kind == L.OP_STEM)
    def term_shift(ops, heads):
        ops[hosted].term_shift[0] = ops[hosted].term_shift[0] + 1      # the term's map no longer fits the output

    def dual_alias(ops, heads):
        ops[dual].aux_buf = ops[dual].out_buf

    def dual_on_other_kernel(ops, heads):
        ops[hosted].flags = ops[hosted].flags | L.CONV_DUAL              # a second output needs algo 3

    cases_ = [term, term_shift, dual_alias, dual_on_other_kernel, set_(dual, 'aux_buf', -1), set_(hosted, 'nterms', 4),
              set_(conv, 'nterms', 1), set_(conv, 'w_off', blob.size - 8), set_(conv, 'in_buf', -1), set_(conv, 'out_coff', 4096),
              set_(conv, 'cin', 4096), set_(conv, 'b_off', -5), set_(conv, 'mode', 7), set_(conv, 'ksize', 5),
              set_(stem, 'cout', 32), set_(stem, 'w_off', blob.size - 100), set_(stem, 'out_coff', 2),
              lambda ops, heads: heads.center_buf.__setitem__(0, 999),
              lambda ops, heads: heads.params_buf.__setitem__(1, heads.center_buf[0])]
    for m in cases_:
        assert try_program(m) == L.E_INVAL
        assert L.lib().acrmi_last_error(eng.ctx)
    assert try_program(lambda ops, heads: None) == 0           # the untouched program still loads
    eng.close()
    # ... and the small-batch program's OP_FUSESUM terms
    prog = packer.lower(synth_sd, wino24=False, splitk=True)
    eng = pkg('engine').Engine(0)
    blob = prog['blob']
    L.check(L.lib().acrmi_load_weights(eng.ctx, blob.ctypes.data_as(C.c_void_p), blob.size), eng.ctx)
    bufs = (L.BufferDesc * len(prog['bufs']))(*[L.BufferDesc(*b) for b in prog['bufs']])
    fuse = next(i for i, o in enumerate(prog['ops']) if o.kind == L.OP_FUSESUM)
    assert try_program(term) == L.E_INVAL and try_program(lambda ops, heads: None) == 0
    eng.close()


def test_raw_1080p_batch_end_to_end(synth_sd, mano_tables):
    """BASELINE.json config 4 shape: 1080p BGR frames in HBM -> GPU pre-processing -> fused path -> per-image hands;
    equals the one-frame-at-a-time host-preprocessed route."""
    acr = pkg('acr.main').ACR(state_dict=synth_sd, mano_tables=mano_tables, max_batch=2)
    rs = np.random.RandomState(7)
    frames = np.clip(np.kron(rs.randint(0, 256, (2, 135, 240, 3)).astype(np.float32), np.ones((1, 8, 8, 1), np.float32)), 0, 255).astype(np.uint8)
    batch = acr.forward_raw_batch(torch.from_numpy(frames).cuda(), ['v0', 'v1'])
    for i, name in enumerate(('v0', 'v1')):
        single = acr(frames[i], name)[name]
        assert len(batch[name]) == len(single)
        for hb, hs in zip(batch[name], single):
            assert int(hb['hand_type']) == int(hs['hand_type'])
            assert np.abs(hb['verts'].astype(np.float32) - hs['verts'].astype(np.float32)).max() < 5e-3
            assert np.abs(hb['pj2d_org'].astype(np.float32) - hs['pj2d_org'].astype(np.float32)).max() < 4.0   # px of 1920
