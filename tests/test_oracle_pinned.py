"""Pin the CPU oracle (oracle/) against golden vectors captured from the real reference
(tests/golden/make_golden.py).  CPU only."""
import json
import os

import numpy as np
import pytest
import torch

import cases
from conftest import GOLDEN, golden, pkg
from oracle import acr_net, decode as odec, mano as omano


def test_schema_digest_matches_reference():
    with open(os.path.join(GOLDEN, 'schema_digest.json')) as f:
        want = json.load(f)
    assert pkg('schema').schema_digest() == want
    assert want['n_keys'] == 2067 and want['n_params'] == 30249120   # SURVEY.md §2.1 census


@pytest.fixture(scope='module')
def net_out(synth_sd, frames2):
    torch.set_num_threads(8)
    taps = {}
    with torch.no_grad():
        x = acr_net.backbone(synth_sd, torch.from_numpy(frames2[:1]), taps)
        taps['backbone'] = x
        heads = acr_net.head_forward(synth_sd, x, taps)
    return taps, heads


def _close(a, b, rtol=2e-5, atol=2e-5):
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


def test_backbone_taps_match_reference(net_out):
    taps, _ = net_out
    g = golden('net_frame0.npz')
    for k in ('stem', 'layer1', 'stage2', 'stage3', 'backbone'):
        s, cs = cases.sub(taps[k])
        _close(s, g['tap_' + k])
        np.testing.assert_allclose(cs, g['tap_%s_sum' % k], rtol=1e-5)


def test_head_maps_match_reference(net_out):
    _, heads = net_out
    g = golden('net_frame0.npz')
    for k in ('l_center_map', 'r_center_map'):
        _close(heads[k].numpy(), g[k])
    for k in ('l_params_maps', 'r_params_maps', 'l_prior_maps', 'r_prior_maps', 'segms'):
        s, cs = cases.sub(heads[k], 8192)
        _close(s, g[k], rtol=5e-5, atol=5e-5)
        np.testing.assert_allclose(cs, g[k + '_sum'], rtol=1e-4)


@pytest.mark.parametrize('name', list(cases.DECODE_CASES))
def test_decode_cases_match_reference(name):
    g = golden('decode_cases.npz')
    maps = {k: torch.from_numpy(v) for k, v in cases.decode_maps(name).items()}
    rows = odec.slots_to_rows(odec.decode(maps))
    np.testing.assert_array_equal(rows['detection_flag'], g[name + '_detection_flag'].astype(bool))
    _close(rows['params_pred'], g[name + '_params_pred'], 1e-6, 1e-6)
    _close(rows['cam'], g[name + '_cam'], 1e-6, 1e-6)
    _close(rows['betas'], g[name + '_betas'], 1e-6, 1e-6)
    _close(rows['poses'], g[name + '_poses'], 1e-5, 1e-5)
    lc = g[name + '_l_centers_pred'][0]
    rc = g[name + '_r_centers_pred'][0]
    assert rows['flat_ind'][0] == lc[1] * 64 + lc[0] and rows['flat_ind'][1] == rc[1] * 64 + rc[0]


@pytest.mark.parametrize('name', list(cases.DECODE_BATCHES))
def test_decode_batches_match_reference(name):
    """The reference's parse_maps on a batch > 1 with mixed detection states (tests/golden/make_golden_batch.py): the
    oracle's batch_semantics='reference' mode reproduces its rows; the per-frame mode differs exactly in the batches
    the fixture was built to separate."""
    g = golden('decode_batches.npz')
    maps = {k: torch.from_numpy(v) for k, v in cases.decode_batch_maps(name).items()}
    rows = odec.slots_to_rows_batch(odec.decode(maps, batch_semantics='reference'))
    np.testing.assert_array_equal(rows['detection_flag'], g[name + '_detection_flag'].astype(bool))
    np.testing.assert_array_equal(rows['frame'], g[name + '_reorganize_idx'])
    np.testing.assert_array_equal(rows['hand_type'], g[name + '_hand_type'])
    _close(rows['params_pred'], g[name + '_params_pred'], 1e-6, 1e-6)
    _close(rows['cam'], g[name + '_cam'], 1e-6, 1e-6)
    _close(rows['betas'], g[name + '_betas'], 1e-6, 1e-6)
    _close(rows['poses'], g[name + '_poses'], 1e-5, 1e-5)
    L = int(g[name + '_hand_nums'][0])
    centers = np.concatenate([g[name + '_l_centers_pred'], g[name + '_r_centers_pred']], 0)
    np.testing.assert_array_equal(rows['flat_ind'], centers[:, 1] * 64 + centers[:, 0])
    assert len(g[name + '_l_centers_pred']) == L
    per_frame = odec.slots_to_rows_batch(odec.decode(maps))
    differs = np.abs(per_frame['params_pred'] - g[name + '_params_pred']).max() > 1e-3
    assert differs == (name in ('b2_far_near', 'b2_near_far', 'b2_leftonly_dist32', 'b3_rightonly_near_none', 'b4_mixed')), name


def test_rot6d_kat_match_reference():
    g = golden('rot6d_kat.npz')
    x6 = torch.from_numpy(g['x6'])
    _close(odec.rot6d_to_rotmat(x6).numpy(), g['R'], 1e-6, 1e-6)
    _close(odec.rot6d_to_aa(x6).numpy(), g['aa'], 1e-5, 1e-6)
    np.testing.assert_array_equal(cases.rot6d_inputs(), g['x6'])


def _tables(mano_tables, side):
    t = dict(mano_tables['left' if side == 'l' else 'right'])
    if side == 'l':
        t['shapedirs'] = t['shapedirs'].copy()
        t['shapedirs'][:, 0, :] *= -1            # acr/mano_wrapper.py:35
    return t


@pytest.mark.parametrize('n,seed', [(0, 0), (1, 1), (2, 2), (16, 3)])
@pytest.mark.parametrize('side', ['l', 'r'])
def test_mano_matches_reference(mano_tables, n, seed, side):
    g = golden('mano_cases.npz')
    poses, betas = cases.mano_inputs(n, seed)
    v, j, c = omano.mano_forward(_tables(mano_tables, side), 'left' if side == 'l' else 'right', poses, betas)
    key = 'n%d_%s' % (n, side)
    assert v.shape == (n, 778, 3) and j.shape == (n, 21, 3)
    _close(v, g[key + '_verts'], 1e-5, 2e-7)
    _close(j, g[key + '_joints'], 1e-5, 2e-7)
    _close(c, g[key + '_center'], 1e-5, 2e-7)


@pytest.mark.parametrize('name', list(cases.MANO_ROTMAT_CASES))
def test_mano_rotmat_mode_matches_reference(name, mano_tables):
    """joint_rot_mode='rotmat' (mano/manolayer.py:151-162, batch_rotprojs :436-453): the oracle against the REAL reference's
    ManoLayer on seeded rotation matrices - exact, noisy (the SVD projection matters) and one reflection (det < 0)
    (tests/golden/mano_rotmat.npz, make_golden_rotmat.py)."""
    g = golden('mano_rotmat.npz')
    kw, n, seed, noise = cases.MANO_ROTMAT_CASES[name]
    rot, betas = cases.mano_rotmat_inputs(name)
    v, j, c = omano.mano_forward(mano_tables[kw['side']], kw['side'], rot, betas, center_idx=kw['center_idx'])
    _close(v, g[name + '_verts'], 1e-5, 5e-7)
    _close(j, g[name + '_joints'], 1e-5, 5e-7)
    if g[name + '_center'].size:
        _close(c, g[name + '_center'], 1e-5, 5e-7)
    else:
        assert c is None


def test_projection_matches_reference(mano_tables):
    g = golden('mano_cases.npz')
    poses, betas = cases.mano_inputs(4, 9)
    cam, offsets = cases.proj_inputs(4, 9)
    v, j, _ = omano.mano_forward(_tables(mano_tables, 'r'), 'right', poses, betas)
    vc, pj, org = omano.project(v, j, cam, offsets)
    _close(vc, g['proj_verts_camed'], 1e-5, 1e-6)
    _close(pj, g['proj_pj2d'], 1e-5, 1e-6)
    _close(org, g['proj_pj2d_org'], 1e-5, 1e-3)


def test_end_to_end_batch1_matches_reference(synth_sd, frames2, mano_tables):
    """uint8 frame -> verts/joints through the oracle pieces == reference acr.model.ACR.forward +
    MANOWrapper.forward at batch 1 (acr/main.py:126-141,85)."""
    g = golden('e2e_batch1.npz')
    torch.set_num_threads(8)
    for b in range(2):
        with torch.no_grad():
            heads = acr_net.network(synth_sd, torch.from_numpy(frames2[b:b + 1]))
        slots = odec.decode(heads)
        rows = odec.slots_to_rows(slots)
        np.testing.assert_array_equal(rows['detection_flag'], g['f%d_detection_flag' % b].astype(bool))
        _close(rows['params_pred'], g['f%d_params_pred' % b], 1e-4, 1e-4)
        _close(rows['poses'], g['f%d_poses' % b], 1e-4, 1e-4)
        vl, jl, _ = omano.mano_forward(_tables(mano_tables, 'l'), 'left', rows['poses'][:1], rows['betas'][:1])
        vr, jr, _ = omano.mano_forward(_tables(mano_tables, 'r'), 'right', rows['poses'][1:], rows['betas'][1:])
        verts, joints = np.concatenate([vl, vr]), np.concatenate([jl, jr])
        assert np.abs(verts - g['f%d_verts' % b]).max() < 1e-5     # metres
        assert np.abs(joints - g['f%d_j3d' % b]).max() < 1e-5
        offsets = np.tile(np.array([[512., 512, 0, 0, 0, 0, 0, 0, 0, 0]], np.float32), (2, 1))
        vc, pj, org = omano.project(verts, joints, rows['cam'], offsets)
        _close(vc, g['f%d_verts_camed' % b], 1e-4, 1e-4)
        _close(pj, g['f%d_pj2d' % b], 1e-4, 1e-4)
        _close(org, g['f%d_pj2d_org' % b], 1e-4, 2e-2)


# ---- round 2: detection states, temporal smoothing, cam_trans, pre-processing ------------------------------
@pytest.mark.parametrize('name', list(cases.STATE_CHECKPOINTS))
def test_network_states_match_reference(name, mano_tables):
    """Left-only / right-only / none / both-with-prior through the WHOLE network (one checkpoint seed per state,
    tests/golden/cases.py) == what the real reference produced at batch 1 (e2e_states.npz)."""
    g = golden('e2e_states.npz')
    torch.set_num_threads(8)
    sd = pkg('synth').make_state_dict(seed=cases.STATE_CHECKPOINTS[name])
    frames = torch.from_numpy(pkg('synth').make_frames(2, seed=cases.STATE_FRAME_SEED))
    with torch.no_grad():
        heads = acr_net.network(sd, frames)
    slots = odec.decode(heads)
    for b in range(2):
        rows = odec.slots_to_rows({k: v[b:b + 1] for k, v in slots.items()})
        key = '%s_f%d_' % (name, b)
        np.testing.assert_array_equal(rows['detection_flag'], g[key + 'detection_flag'].astype(bool))
        lc, rc = g[key + 'l_centers_pred'][0], g[key + 'r_centers_pred'][0]
        assert rows['flat_ind'][0] == lc[1] * 64 + lc[0] and rows['flat_ind'][1] == rc[1] * 64 + rc[0]
        _close(rows['params_pred'], g[key + 'params_pred'], 2e-4, 2e-4)
        if name == 'none':
            assert key + 'verts' not in g.files        # acr/main.py:96: MANO is not run when nothing is detected
            continue
        vl, jl, _ = omano.mano_forward(_tables(mano_tables, 'l'), 'left', rows['poses'][:1], rows['betas'][:1])
        vr, jr, _ = omano.mano_forward(_tables(mano_tables, 'r'), 'right', rows['poses'][1:], rows['betas'][1:])
        assert np.abs(np.concatenate([vl, vr]) - g[key + 'verts']).max() < 2e-5
        assert np.abs(np.concatenate([jl, jr]) - g[key + 'j3d']).max() < 2e-5
    if name == 'both_near':                            # the cross-hand prior was applied (distance <= 32 px)
        lc, rc = g['both_near_f0_l_centers_pred'][0], g['both_near_f0_r_centers_pred'][0]
        assert np.hypot(*(lc - rc).astype(float)) <= 32


@pytest.mark.parametrize('name', list(cases.E2E_BATCHES))
def test_network_batches_match_reference(name, mano_tables):
    """The WHOLE reference at batch > 1 (model.forward on 4 / 3 / 2 frames + MANOWrapper, tests/golden/make_golden_batch.py ->
    e2e_batches.npz): the oracle with batch_semantics='reference' returns its rows in its order - all left rows of the
    batch, then the right rows, one placeholder row for a side without any hit in the batch."""
    g = golden('e2e_batches.npz')
    torch.set_num_threads(8)
    seed, B = cases.E2E_BATCHES[name]
    sd = pkg('synth').make_state_dict(seed=seed)
    frames = torch.from_numpy(pkg('synth').make_frames(B, seed=cases.STATE_FRAME_SEED))
    with torch.no_grad():
        heads = acr_net.network(sd, frames)
    rows = odec.slots_to_rows_batch(odec.decode(heads, batch_semantics='reference'))
    np.testing.assert_array_equal(rows['detection_flag'], g[name + '_detection_flag'].astype(bool))
    np.testing.assert_array_equal(rows['frame'], g[name + '_reorganize_idx'])
    np.testing.assert_array_equal(rows['hand_type'], g[name + '_output_hand_type'])
    centers = np.concatenate([g[name + '_l_centers_pred'], g[name + '_r_centers_pred']], 0)
    np.testing.assert_array_equal(rows['flat_ind'], centers[:, 1] * 64 + centers[:, 0])
    _close(rows['params_pred'], g[name + '_params_pred'], 2e-4, 2e-4)
    if name == 'none_b2':
        assert name + '_verts' not in g.files
        return
    L = int(g[name + '_hand_nums'][0])
    vl, jl, _ = omano.mano_forward(_tables(mano_tables, 'l'), 'left', rows['poses'][:L], rows['betas'][:L])
    vr, jr, _ = omano.mano_forward(_tables(mano_tables, 'r'), 'right', rows['poses'][L:], rows['betas'][L:])
    assert np.abs(np.concatenate([vl, vr]) - g[name + '_verts']).max() < 2e-5
    assert np.abs(np.concatenate([jl, jr]) - g[name + '_j3d']).max() < 2e-5


@pytest.mark.parametrize('name', list(cases.INTERIOR_CASES))
def test_interior_centers_match_reference(name, mano_tables):
    """centers >= 9 px from every border of the map (planted with synth.plant_center_peaks; every other
    whole-network fixture peaks on the border) through the WHOLE network == the real reference (e2e_interior.npz)."""
    g = golden('e2e_interior.npz')
    torch.set_num_threads(8)
    seed, lp, rp = cases.INTERIOR_CASES[name]
    sd = cases.interior_state_dict(pkg('synth'), name)
    frames = torch.from_numpy(pkg('synth').make_frames(2, seed=cases.STATE_FRAME_SEED))
    with torch.no_grad():
        heads = acr_net.network(sd, frames)
    slots = odec.decode(heads)
    for b in range(2):
        rows = odec.slots_to_rows({k: v[b:b + 1] for k, v in slots.items()})
        key = '%s_f%d_' % (name, b)
        assert g[key + 'detection_flag'].tolist() == [1.0, 1.0] and rows['detection_flag'].tolist() == [True, True]
        lc, rc = g[key + 'l_centers_pred'][0], g[key + 'r_centers_pred'][0]          # (x, y)
        assert (lc[1], lc[0]) == lp and (rc[1], rc[0]) == rp                          # the reference found the planted pixels
        assert all(5 <= v <= 58 for v in (*lp, *rp))
        assert rows['flat_ind'][0] == lp[0] * 64 + lp[1] and rows['flat_ind'][1] == rp[0] * 64 + rp[1]
        _close(rows['params_pred'], g[key + 'params_pred'], 2e-4, 2e-4)
        vl, jl, _ = omano.mano_forward(_tables(mano_tables, 'l'), 'left', rows['poses'][:1], rows['betas'][:1])
        vr, jr, _ = omano.mano_forward(_tables(mano_tables, 'r'), 'right', rows['poses'][1:], rows['betas'][1:])
        assert np.abs(np.concatenate([vl, vr]) - g[key + 'verts']).max() < 2e-5
        assert np.abs(np.concatenate([jl, jr]) - g[key + 'j3d']).max() < 2e-5


def test_hostile_checkpoint_is_the_same_function():
    """synth.make_state_dict(law='hostile') re-parametrises the benign checkpoint (raw conv rows over 2.5 decades with
    running_var over 1e-3..1e2, block-internal channel scales over two decades, backbone stream x50) without changing
    the function: the oracle gives the benign maps to fp32 round-off while the stream runs at ~50x the magnitude."""
    synth = pkg('synth')
    torch.set_num_threads(8)
    sd, hs = synth.make_state_dict(seed=0), synth.make_state_dict(seed=0, law='hostile')
    rv = torch.cat([v.flatten() for k, v in hs.items() if k.endswith('running_var')])
    assert float(rv.min()) < 2e-3 and float(rv.max()) > 50 and float(rv.min()) > 0
    frame = torch.from_numpy(synth.make_frames(1, seed=0))
    ta, tb = {}, {}
    with torch.no_grad():
        a, b = acr_net.network(sd, frame, ta), acr_net.network(hs, frame, tb)
    for k in a:
        assert float((a[k] - b[k]).abs().max()) < 5e-5 * max(1.0, float(a[k].abs().max())), k
    for k in ('stem', 'layer1', 'stage2', 'stage3'):
        assert float((ta[k] * 50 - tb[k]).abs().max()) < 1e-4 * float(tb[k].abs().max()), k
    assert float(tb['stage3'].abs().max()) > 50


def test_smoothing_oracle_matches_reference_sequence():
    """oracle.smooth (numpy f32) == the reference's smooth_results / OneEuroFilter over a 14-frame two-hand sequence
    with a late-appearing hand, a two-frame drop-out and a near-pi global orientation (smooth_seq.npz)."""
    from oracle import smooth as osm
    g = golden('smooth_seq.npz')
    poses, betas, flags = cases.smooth_inputs()
    filt = {0: osm.new_filters(float(g['smooth_coeff'])), 1: osm.new_filters(float(g['smooth_coeff']))}
    worst = 0.0
    for t in range(poses.shape[0]):
        for sid in range(2):
            want_p, want_b = g['poses'][t, sid], g['betas'][t, sid]
            if flags[t, sid]:
                p, b = osm.smooth_results(filt[sid], poses[t, sid], betas[t, sid])
            else:
                p, b = poses[t, sid], betas[t, sid]
            worst = max(worst, float(np.abs(p - want_p).max()), float(np.abs(b - want_b).max()))
    assert worst < 2e-6, worst
    assert np.abs(g['poses'][5] - poses[5]).max() > 1e-2       # the filter does something


def test_cam_trans_oracle_matches_reference_fallback():
    """oracle.smooth.estimate_translation == the reference's closed-form branch (acr/utils.py:430-472), which is what
    the reference itself ran when e2e_batch1.npz was captured (cv2 absent -> bare except, acr/utils.py:512-517)."""
    from oracle import smooth as osm
    for name, key in (('e2e_batch1.npz', 'f0_'), ('e2e_states.npz', 'both_near_f1_')):
        g = golden(name)
        t = osm.estimate_translation(g[key + 'j3d'], g[key + 'pj2d'], focal_length=1265)
        np.testing.assert_allclose(t, g[key + 'cam_trans'], rtol=1e-4, atol=1e-4)


def test_preprocess_oracle_magic_jpg():
    """BASELINE configs[0]: demo/magic.jpg -> img_preprocess.  The fixture was produced by the reference's own
    img_preprocess (acr/utils.py:1315-1337) with cv2.resize / imgaug served by oracle/preprocess.py."""
    from PIL import Image
    from oracle import preprocess as opre
    g = golden('magic_e2e.npz')
    bgr = np.ascontiguousarray(np.asarray(Image.open(os.path.join(GOLDEN, 'magic.jpg')).convert('RGB'))[:, :, ::-1])
    chk = np.array([bgr.astype(np.int64).sum(), (bgr.astype(np.int64) * (np.arange(bgr.size).reshape(bgr.shape) % 251)).sum()])
    np.testing.assert_array_equal(chk, g['bgr_sum'])           # same JPEG decode as in the authoring container
    img, offsets = opre.img_preprocess(bgr)
    np.testing.assert_array_equal(offsets[None], g['offsets'])
    assert offsets.tolist() == [1920, 1920, 0, 0, 0, 0, 420, 0, 420, 0]      # SURVEY.md 8d config 4 geometry
    np.testing.assert_array_equal(img[::4, ::4], g['image_sub'])
    assert int(img.astype(np.int64).sum()) == int(g['image_sum'][0])
    assert (img[:100] == 255).all()                            # white pad rows (420 px of 1920 -> 112 of 512)


def test_cubic_resize_oracle_properties():
    """OpenCV's fixed-point INTER_CUBIC as restated in oracle/preprocess.py: identity at scale 1, constants stay
    constant, coefficient rows sum to 2048 +- rounding, and the result stays within 1 LSB of a float bicubic
    (a = -0.75, half-pixel centres, clamped border) on smooth content."""
    from oracle import preprocess as opre
    rs = np.random.RandomState(0)
    img = rs.randint(0, 256, (64, 64, 3)).astype(np.uint8)
    np.testing.assert_array_equal(opre.resize_cubic_u8(img, 64, 64), img)
    flat = np.full((90, 70, 3), 137, np.uint8)
    assert (opre.resize_cubic_u8(flat, 32, 32) == 137).all()
    s, c = opre._taps(1920, 512)
    assert np.abs(c.sum(1) - 2048).max() <= 2 and s.min() == 1 and s.max() == 1917
    yy, xx = np.mgrid[0:300, 0:420].astype(np.float32)
    smooth = np.stack([127 + 100 * np.sin(xx / 23) * np.cos(yy / 31), 127 + 90 * np.cos(xx / 17 + yy / 29),
                       (xx + yy) / 3], -1)
    smooth = np.clip(smooth, 0, 255).astype(np.uint8)
    got = opre.resize_cubic_u8(smooth, 128, 128)
    x = torch.from_numpy(smooth).permute(2, 0, 1)[None].float()
    ref = torch.nn.functional.interpolate(x, size=(128, 128), mode='bicubic', align_corners=False)
    ref = ref.round().clamp(0, 255)[0].permute(1, 2, 0).numpy()
    assert np.abs(got.astype(np.float32) - ref).max() <= 1
    # imgaug 0.4.0 padding rule: the extra pixel goes to bottom / right
    assert opre.compute_paddings_to_reach_aspect_ratio((1080, 1920, 3)) == (420, 0, 420, 0)
    assert opre.compute_paddings_to_reach_aspect_ratio((701, 300, 3)) == (0, 201, 0, 200)
    assert opre.compute_paddings_to_reach_aspect_ratio((300, 701, 3)) == (200, 0, 201, 0)
    assert opre.compute_paddings_to_reach_aspect_ratio((512, 512, 3)) == (0, 0, 0, 0)
