"""the real-asset loaders - `wild.pkl`-style checkpoints (acr/utils.py:1106-1168) and chumpy-pickled
MANO_{LEFT,RIGHT}.pkl (mano/manolayer.py:54-102,350-394) - exercised on files written here in the formats the
reference reads (the real files are licence gated).  CPU part: every branch of the loaders; GPU part: files ->
loaders -> engine -> the reference's golden vertices."""
import os
import pickle
import sys
import types

import numpy as np
import pytest
import scipy.sparse as sp
import torch

import cases
from conftest import golden, pkg


def write_mano_pkls(root, tables):
    """MANO_LEFT.pkl / MANO_RIGHT.pkl as the MPI release stores them: a py2-style pickle (protocol 2) of a dict whose
    blend shapes are chumpy `Ch` objects (array in attribute `x`), J_regressor a scipy.sparse.csc_matrix, the rest
    numpy arrays (mano/manolayer.py:59-102 lists the fields read)."""
    class Ch(object):                      # pickled as chumpy.ch.Ch; chumpy itself is not needed to READ the file
        def __init__(self, x):
            self.x = np.asarray(x, np.float64)
            self._cache = {'drs': {}}      # chumpy keeps bookkeeping fields next to the array
    Ch.__module__, Ch.__qualname__ = 'chumpy.ch', 'Ch'
    mod, parent = types.ModuleType('chumpy.ch'), types.ModuleType('chumpy')
    mod.Ch, parent.ch = Ch, mod
    saved = {k: sys.modules.get(k) for k in ('chumpy', 'chumpy.ch')}
    sys.modules.update({'chumpy': parent, 'chumpy.ch': mod})
    try:
        for side, fname in (('left', 'MANO_LEFT.pkl'), ('right', 'MANO_RIGHT.pkl')):
            t = tables[side]
            dd = {'v_template': t['v_template'].astype(np.float64), 'shapedirs': Ch(t['shapedirs']),
                  'posedirs': t['posedirs'].astype(np.float64), 'J_regressor': sp.csc_matrix(t['J_regressor'].astype(np.float64)),
                  'weights': t['weights'].astype(np.float64), 'hands_mean': t['hands_mean'].astype(np.float64),
                  'hands_components': t['hands_components'].astype(np.float64), 'f': t['faces'].astype(np.uint32),
                  'kintree_table': t['kintree_table'], 'bs_style': 'lbs', 'bs_type': 'lrotmin',
                  'J': np.zeros((16, 3)), 'hands_coeffs': np.zeros((10, 45))}
            with open(os.path.join(root, fname), 'wb') as f:
                pickle.dump(dd, f, protocol=2)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def write_wild_pkl(path, sd, wrapper='model_state_dict'):
    """checkpoints/wild.pkl as trained with nn.DataParallel: 'module.'-prefixed keys nested under 'model_state_dict'
    (acr/utils.py:1153-1168), next to training leftovers the loader must ignore."""
    body = {'module.' + k: v for k, v in sd.items()}
    body['module.some_training_only_head.weight'] = torch.zeros(3, 3)
    ckpt = {wrapper: body, 'optimizer_state_dict': {'state': {}}, 'epoch': 12} if wrapper else body
    torch.save(ckpt, path)


def test_load_mano_pkl_reads_chumpy_pickles_without_chumpy(tmp_path, mano_tables):
    write_mano_pkls(str(tmp_path), mano_tables)
    assert 'chumpy' not in sys.modules
    ml = pkg('mano.manolayer')
    for side, fname in (('left', 'MANO_LEFT.pkl'), ('right', 'MANO_RIGHT.pkl')):
        t = ml.load_mano_pkl(str(tmp_path / fname))
        for k in ('v_template', 'shapedirs', 'posedirs', 'J_regressor', 'weights', 'hands_mean', 'hands_components'):
            assert t[k].dtype == np.float32
            np.testing.assert_array_equal(t[k], mano_tables[side][k])
        assert t['J_regressor'].shape == (16, 778)                      # sparse -> dense
        np.testing.assert_array_equal(t['faces'], mano_tables[side]['faces'])
        assert t['faces'].dtype == np.int64 and t['kintree_table'].shape == (2, 16)
    with pytest.raises(FileNotFoundError):
        ml.load_mano_pkl(str(tmp_path / 'MANO_NONE.pkl'))
    # ManoLayer(mano_root=...) picks the file by side and registers the reference's buffers (host tensors)
    lay = ml.ManoLayer(center_idx=9, flat_hand_mean=False, ncomps=45, side='left', mano_root=str(tmp_path), use_pca=False)
    assert lay.mano_path.endswith('MANO_LEFT.pkl') and tuple(lay.th_shapedirs.shape) == (778, 3, 10)
    assert torch.equal(lay.th_faces, torch.from_numpy(mano_tables['left']['faces']))
    assert lay.kintree_parents[1:] == [0, 1, 2, 0, 4, 5, 0, 7, 8, 0, 10, 11, 0, 13, 14]
    flat = ml.ManoLayer(flat_hand_mean=True, side='right', mano_root=str(tmp_path), use_pca=False)
    assert float(flat.th_hands_mean.abs().sum()) == 0.0


@pytest.mark.parametrize('wrapper', ['model_state_dict', 'state_dict', None])
def test_load_model_reads_wild_pkl_layouts(tmp_path, synth_sd, wrapper):
    u = pkg('acr.utils')
    path = str(tmp_path / 'wild.pkl')
    write_wild_pkl(path, synth_sd, wrapper)
    m = pkg('acr.model').ACR()
    m = u.load_model(path, m, prefix='module.', drop_prefix='', fix_loaded=False)
    got = m.state_dict()
    for k, v in synth_sd.items():
        assert torch.equal(got[k], v), k
    with pytest.raises(ValueError):
        u.load_model(str(tmp_path / 'missing.pkl'), m)
    # a checkpoint with a mismatched tensor: that layer is skipped with a log line, the rest loads (copy_state_dict)
    bad = dict(synth_sd)
    bad['l_final_layers.2.2.bias'] = torch.zeros(5)
    write_wild_pkl(path, bad, wrapper)
    m2 = u.load_model(path, pkg('acr.model').ACR(), prefix='module.')
    assert torch.equal(m2.state_dict()['backbone.conv1.weight'], synth_sd['backbone.conv1.weight'])
    assert float(m2.state_dict()['l_final_layers.2.2.bias'].abs().sum()) == 0.0      # left at its initial value


@pytest.mark.gpu
def test_files_to_meshes_through_the_loaders(tmp_path, synth_sd, mano_tables, frames2):
    """wild.pkl + MANO_*.pkl on disk -> acr.main.ACR(args) exactly as the reference's CLI builds it
    (acr/main.py:57-63) -> the reference's golden results; and ManoLayer(mano_root=...) -> golden vertices."""
    write_mano_pkls(str(tmp_path), mano_tables)
    write_wild_pkl(str(tmp_path / 'wild.pkl'), synth_sd)
    g = golden('mano_cases.npz')
    ML = pkg('mano.manolayer').ManoLayer
    lay = ML(center_idx=9, flat_hand_mean=False, ncomps=45, side='right', mano_root=str(tmp_path), use_pca=False)
    poses, betas = cases.mano_inputs(2, 2)
    v, j, _ = lay(torch.from_numpy(poses), th_betas=torch.from_numpy(betas))
    assert np.abs(v.cpu().numpy() - g['n2_r_verts']).max() < 2e-6 and np.abs(j.cpu().numpy() - g['n2_r_joints']).max() < 2e-6
    cfg = pkg('config')
    a = cfg.parse_args(['--configs_yml', '/nonexistent.yml', '--model_path', str(tmp_path / 'wild.pkl'),
                        '--mano_root', str(tmp_path) + '/'])
    acr = pkg('acr.main').ACR(args_set=a)
    e = golden('e2e_batch1.npz')
    for b in range(2):
        res = acr(np.ascontiguousarray(frames2[b][:, :, ::-1]), 'f%d.jpg' % b)['f%d.jpg' % b]
        assert [int(h['hand_type']) for h in res] == [0, 1]
        for i, h in enumerate(res):
            assert np.abs(h['verts'].astype(np.float32) - e['f%d_verts' % b][i]).max() < 2e-3      # fp16 packaging
            assert np.abs(h['cam_trans'].astype(np.float32) - e['f%d_cam_trans' % b][i]).max() < 2e-2
