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
