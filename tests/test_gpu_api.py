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
        ML(side='right', use_pca=True, tables=mano_tables['right'])
    with pytest.raises(FileNotFoundError):
        ML(side='left', use_pca=False, mano_root='/nonexistent/')


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


def test_cam_trans_kernel_matches_reference_least_squares():
    """§8f-3: acrmi_cam_trans == the reference's closed-form least squares (acr/utils.py:430-472; restated in numpy
    fp64 in acr.utils.estimate_translation_np and pinned against the reference's cam_trans in test_host_api)."""
    u, ops = pkg('acr.utils'), pkg('ops')
    rs = np.random.RandomState(3)
    n = 37
    j3 = (rs.randn(n, 21, 3) * 0.08).astype(np.float32)
    t_true = np.stack([rs.uniform(-0.3, 0.3, n), rs.uniform(-0.3, 0.3, n), rs.uniform(0.4, 3.0, n)], 1)
    f = 1265.0
    p = j3.astype(np.float64) + t_true[:, None]
    pj2d = ((f * p[..., :2] / p[..., 2:] + 256) / 256 - 1 + rs.randn(n, 21, 2) * 2e-3).astype(np.float32)
    want = np.stack([u.estimate_translation_np(j3[i].astype(np.float64), (pj2d[i].astype(np.float64) + 1) * 256,
                                               np.ones(21, np.float32), focal_length=f) for i in range(n)])
    got = ops.cam_trans(torch.from_numpy(j3).cuda(), torch.from_numpy(pj2d).cuda(), focal_length=f).cpu().numpy()
    np.testing.assert_allclose(got, want, rtol=2e-5, atol=2e-6)
    assert np.abs(got - t_true).max() < 0.05                       # and it recovers the translation
    assert ops.cam_trans(torch.zeros(0, 21, 3).cuda(), torch.zeros(0, 21, 2).cuda()).shape == (0, 3)
    # the product path (MANOWrapper -> estimate_translation) takes the device kernel for device tensors
    t2 = u.estimate_translation(torch.from_numpy(j3).cuda(), torch.from_numpy(pj2d).cuda(), focal_length=f)
    assert t2.is_cuda and np.allclose(t2.cpu().numpy(), got)


def test_gpu_preprocess_matches_host_preprocess():
    """§8f-1: the HIP pre-processing kernel == acr.utils.img_preprocess (white pad + bicubic a=-0.75) on the same
    frames; cv2 itself is not available here, so parity with OpenCV's fixed-point INTER_CUBIC stays unpinned."""
    u, ops = pkg('acr.utils'), pkg('ops')
    rs = np.random.RandomState(4)
    for H, W in ((1080, 1920), (480, 640), (700, 300), (512, 512), (333, 333)):
        base = rs.randint(0, 256, (H // 8 + 2, W // 8 + 2, 3)).astype(np.float32)
        frame = np.kron(base, np.ones((8, 8, 1), np.float32))[:H, :W] + rs.randint(-9, 10, (H, W, 3))
        frame = np.clip(frame, 0, 255).astype(np.uint8)
        host = u.img_preprocess(frame, None, single_img_input=True)
        dev_img, dev_off = ops.preprocess(torch.from_numpy(frame)[None].cuda())
        diff = (dev_img.cpu().int() - host['image'].int()).abs()
        assert diff.max().item() <= 1 and (diff > 0).float().mean().item() < 2e-3, (H, W, diff.max().item())
        assert torch.equal(dev_off, host['offsets'])


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
