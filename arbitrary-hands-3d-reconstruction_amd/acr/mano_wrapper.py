"""MANOWrapper (acr/mano_wrapper.py:15-50): two ManoLayers (left shapedirs x-flipped), one fused HIP
launch for all rows, projection fused into the kernel epilogue; cam_trans from the device least-squares kernel."""
import torch

from ..config import args
from ..mano.manolayer import ManoLayer, shared_engine
from .utils import estimate_translation


class MANOWrapper(object):
    def __init__(self, mano_root=None, tables=None, device=0, engine=None):
        a = args()
        cidx = a.align_idx if a.mano_mesh_root_align else None
        root = mano_root or a.mano_root
        kw = dict(ncomps=45, center_idx=cidx, mano_root=root, use_pca=False, flat_hand_mean=False, device=device)
        self.mano_layer = {
            'r': ManoLayer(side='right', tables=None if tables is None else tables['right'], **kw),
            'l': ManoLayer(side='left', tables=None if tables is None else tables['left'], **kw)}
        self.mano_layer['l'].th_shapedirs[:, 0, :] *= -1          # acr/mano_wrapper.py:35
        self._model = None            # acr.model.ACR whose current engine is used (set by bind_model)
        self._engine = engine or shared_engine(device)
        for lay in self.mano_layer.values():
            lay.sync(self._engine)
        self.center_idx = cidx

    def bind_model(self, model):
        """Follow `model.engine()` instead of a fixed context: a checkpoint reload replaces the model's engine (the
        new one adopts these MANO tables), and this wrapper must not keep launching on the retired one."""
        self._model = model
        return self

    def engine(self):
        return self._model.engine() if self._model is not None else self._engine

    def cuda(self, device=None):
        return self

    def eval(self):
        return self

    @torch.no_grad()
    def forward(self, outputs, meta_data):
        """Adds verts, j3d, verts_camed, pj2d, pj2d_org, cam_trans, output_hand_type (acr/mano_wrapper.py:37-50)."""
        pd = outputs['params_dict']
        L, R = int(outputs['left_hand_num']), int(outputs['right_hand_num'])
        dev = pd['poses'].device
        side = torch.cat((torch.zeros(L), torch.ones(R))).to(torch.int32)
        outputs['output_hand_type'] = side.to(dev)
        offsets = meta_data.get('offsets') if meta_data is not None else None
        verts, joints, _, extra = self.engine().mano(pd['poses'][:L + R], pd['betas'][:L + R], side,
                                                   center_idx=self.center_idx, cam=pd['cam'][:L + R], offsets=offsets)
        outputs.update({'verts': verts, 'j3d': joints, 'verts_camed': extra['verts_camed'], 'pj2d': extra['pj2d']})
        if 'pj2d_org' in extra:
            outputs['pj2d_org'] = extra['pj2d_org']
        # per-hand camera translation: the reference's closed-form least squares (acr/utils.py:430-472) on the device
        outputs['cam_trans'] = estimate_translation(joints, extra['pj2d'], focal_length=args().focal_length).to(dev)
        return outputs

    __call__ = forward
