"""ManoLayer: the reference's constructor/forward signature (mano/manolayer.py:13-22,104-110,273-276)
over the fused HIP MANO kernel.  Buffers (th_faces, th_v_template, ...) are host tensors for callers
such as the visualiser (acr/visualization.py:80,120); the arithmetic runs in libacrmi.so.
"""
import io
import os
import pickle

import numpy as np
import torch

_SHARED = {}       # device index -> Engine used for MANO-only calls


def shared_engine(device=0):
    from ..engine import Engine
    if device not in _SHARED:
        _SHARED[device] = Engine(device)
    return _SHARED[device]


class _Stub(object):
    """Stands in for chumpy classes when unpickling MANO_*.pkl without chumpy installed."""

    def __init__(self, *a, **k):
        pass

    def __setstate__(self, state):
        self.__dict__.update(state if isinstance(state, dict) else {'state': state})

    @property
    def r(self):
        for key in ('x', 'a'):
            if key in self.__dict__:
                return np.asarray(self.__dict__[key])
        raise AttributeError('cannot recover array from chumpy stub')


class _Unpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if module.startswith('chumpy'):
            return _Stub
        return super().find_class(module, name)


def _arr(v):
    if isinstance(v, _Stub):
        v = v.r
    if hasattr(v, 'toarray'):
        v = v.toarray()
    return np.asarray(v)


def load_mano_pkl(path):
    """mano/manolayer.py:350-394 (ready_arguments) without chumpy: returns the float32 tables."""
    if not os.path.exists(path):
        raise FileNotFoundError('%s not found: the MANO model files are licence gated (reference README.md:36); '
                                'pass tables= for synthetic tables' % path)
    with open(path, 'rb') as f:
        dd = _Unpickler(io.BytesIO(f.read()), encoding='latin1').load()
    return {'v_template': _arr(dd['v_template']).astype(np.float32),
            'shapedirs': _arr(dd['shapedirs']).astype(np.float32),
            'posedirs': _arr(dd['posedirs']).astype(np.float32),
            'J_regressor': _arr(dd['J_regressor']).astype(np.float32),
            'weights': _arr(dd['weights']).astype(np.float32),
            'hands_mean': _arr(dd['hands_mean']).astype(np.float32),
            'hands_components': _arr(dd['hands_components']).astype(np.float32),
            'faces': _arr(dd['f']).astype(np.int64),
            'kintree_table': _arr(dd['kintree_table']).astype(np.int64)}


def batch_rotprojs(rotmats):
    """mano/manolayer.py:436-453: every 3x3 matrix of [N,J,3,3] -> the nearest rotation U V^T (CPU SVD, like the reference),
    the last COLUMN negated when the result is a reflection."""
    m = rotmats.detach().float().cpu()
    out = torch.empty_like(m)
    for b in range(m.shape[0]):
        for r in range(m.shape[1]):
            U, S, V = torch.svd(m[b, r])
            rot = torch.matmul(U, V.transpose(0, 1))
            if rot.det() < 0:
                rot[:, 2] = -1 * rot[:, 2]
            out[b, r] = rot
    return out


class ManoLayer(object):
    def __init__(self, center_idx=None, flat_hand_mean=True, ncomps=6, side='right', mano_root='model_data/mano/',
                 use_pca=True, root_rot_mode='axisang', joint_rot_mode='axisang', robust_rot=False, tables=None,
                 device=0):
        if root_rot_mode != 'axisang':
            # (the reference itself cannot run this mode: mano/manolayer.py:148-150 calls a module `rot6d` it never imports)
            raise ValueError("root_rot_mode=%r: only 'axisang' is implemented" % (root_rot_mode,))
        if not use_pca and joint_rot_mode not in ('axisang', 'rotmat'):
            raise ValueError("joint_rot_mode=%r: 'axisang' or 'rotmat'" % (joint_rot_mode,))
        if side not in ('left', 'right'):
            raise ValueError('side must be "left" or "right"')
        if use_pca and not 0 < ncomps <= 45:
            raise ValueError('ncomps must be 1..45')
        # use_pca: the pose coefficients are PCA coordinates, th_selected_comps = the first ncomps rows of hands_components
        # (mano/manolayer.py:47-50,89-93,124-129); ACR's own wrapper uses use_pca=False (acr/mano_wrapper.py:17-35)
        self.center_idx, self.side, self.use_pca, self.ncomps = center_idx, side, bool(use_pca), (ncomps if use_pca else 45)
        self.flat_hand_mean, self.rot, self.robust_rot = flat_hand_mean, 3, robust_rot
        self.root_rot_mode, self.joint_rot_mode = root_rot_mode, joint_rot_mode
        if tables is None:
            self.mano_path = os.path.join(mano_root, 'MANO_RIGHT.pkl' if side == 'right' else 'MANO_LEFT.pkl')
            tables = load_mano_pkl(self.mano_path)
        t = {k: np.asarray(v) for k, v in tables.items()}
        self.th_betas = torch.zeros(1, 10)
        self.th_shapedirs = torch.from_numpy(t['shapedirs'].astype(np.float32).copy())
        self.th_posedirs = torch.from_numpy(t['posedirs'].astype(np.float32).copy())
        self.th_v_template = torch.from_numpy(t['v_template'].astype(np.float32).copy()).unsqueeze(0)
        self.th_J_regressor = torch.from_numpy(t['J_regressor'].astype(np.float32).copy())
        self.th_weights = torch.from_numpy(t['weights'].astype(np.float32).copy())
        self.th_faces = torch.from_numpy(t['faces'].astype(np.int64).copy())
        hm = np.zeros(45, np.float32) if flat_hand_mean else t['hands_mean'].astype(np.float32)
        self.th_hands_mean = torch.from_numpy(hm.copy()).unsqueeze(0)
        if 'hands_components' in t:
            self.th_comps = torch.from_numpy(t['hands_components'].astype(np.float32).copy())
            self.th_selected_comps = self.th_comps[:self.ncomps].clone()
        elif use_pca:
            raise ValueError('use_pca=True needs the hands_components table')
        kt = t.get('kintree_table')
        self.kintree_parents = [-1, 0, 1, 2, 0, 4, 5, 0, 7, 8, 0, 10, 11, 0, 13, 14] if kt is None else list(kt[0].tolist())
        self._device = device
        self._engine = None
        self._dirty = True

    # the reference is an nn.Module; keep the calls user code makes on it
    def cuda(self, device=None):
        if device is not None:
            self._device = device if isinstance(device, int) else torch.device(device).index or 0
        return self

    def eval(self):
        return self

    def to(self, *a, **k):
        return self

    def sync(self, engine=None):
        """(Re-)upload this side's tables; call after editing a th_* buffer (acr/mano_wrapper.py:35)."""
        eng = engine or shared_engine(self._device)
        eng.load_mano_side(self.side, self.tables())
        self._engine = eng
        self._dirty = False

    def tables(self):
        return {'v_template': self.th_v_template[0].numpy(), 'shapedirs': self.th_shapedirs.numpy(),
                'posedirs': self.th_posedirs.numpy(), 'J_regressor': self.th_J_regressor.numpy(),
                'weights': self.th_weights.numpy(), 'hands_mean': self.th_hands_mean[0].numpy()}

    def forward(self, th_pose_coeffs, th_betas=torch.zeros(1), th_trans=torch.zeros(1), root_palm=torch.Tensor([0]),
                share_betas=torch.Tensor([0])):
        """-> (verts [N,778,3], joints [N,21,3], center_joint [N,1,3]) in metres (mano/manolayer.py:269-276)."""
        if self._dirty or self._engine is None:
            self.sync(self._engine)
        eng = self._engine
        N = th_pose_coeffs.shape[0]
        rotmat = not self.use_pca and self.joint_rot_mode == 'rotmat'
        if rotmat:
            # mano/manolayer.py:151-162: [N,16,3,3] matrices, projected onto SO(3) one by one with a CPU SVD exactly as the
            # reference's batch_rotprojs does (:436-453, U V^T with the last column flipped when det < 0); the kernel takes the
            # rotations as they are (acrmi_mano_rotmat)
            if th_pose_coeffs.dim() != 4 or tuple(th_pose_coeffs.shape[1:]) != (16, 3, 3):
                raise ValueError('joint_rot_mode="rotmat": th_pose_coeffs must be [N,16,3,3], got %s' % (tuple(th_pose_coeffs.shape),))
            th_pose_coeffs = batch_rotprojs(th_pose_coeffs.detach().float().cpu())
        if self.use_pca:      # PCA coordinates -> the 45 axis-angle values (the kernel adds th_hands_mean, :132-135)
            c = th_pose_coeffs.detach().float().cpu()
            th_pose_coeffs = torch.cat([c[:, :3], c[:, 3:3 + self.ncomps].mm(self.th_selected_comps)], 1)
        if th_betas is None or th_betas.numel() == 1:
            th_betas = self.th_betas.expand(N, 10)
        elif bool(share_betas):
            th_betas = th_betas.mean(0, keepdim=True).expand(N, 10)
        use_trans = not (th_trans is None or bool(torch.norm(th_trans.float()) == 0))
        side = torch.full((N,), 0 if self.side == 'left' else 1, dtype=torch.int32)
        if bool(root_palm):
            # mano/manolayer.py:249-251: the wrist joint is replaced by the palm = mean of vertices 95 and 22 BEFORE the
            # root alignment, so the kernel runs unaligned and the alignment (:258-266) happens here
            verts, joints, _, _ = eng.mano(th_pose_coeffs, th_betas.contiguous(), side, center_idx=None, rotmat=rotmat)
            joints = joints.clone()
            joints[:, 0] = (verts[:, 95] + verts[:, 22]) / 2
            if use_trans:
                t = th_trans.to(verts.device).float().unsqueeze(1)
                return verts + t, joints + t, t
            if self.center_idx is None:
                return verts, joints, None
            center = joints[:, self.center_idx].unsqueeze(1).clone()
            return verts - center, joints - center, center
        cidx = None if use_trans else self.center_idx
        verts, joints, center, _ = eng.mano(th_pose_coeffs, th_betas.contiguous(), side, center_idx=cidx, rotmat=rotmat)
        if use_trans:
            t = th_trans.to(verts.device).float().unsqueeze(1)
            return verts + t, joints + t, t
        if self.center_idx is None:
            return verts, joints, None
        return verts, joints, center

    __call__ = forward
