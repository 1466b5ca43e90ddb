"""acr.main.ACR: the demo-facing wrapper (acr/main.py:24-141) over the MI355X path.

    acr = ACR(args_set)                      # builds the model, loads the checkpoint + MANO tables
    results = acr(bgr_frame, path)           # {path: [per-hand dict of float16 arrays]}  or  {path: {}}
    results = acr.forward_batch(frames, paths)   # the batched form the reference never had

Rendering / video IO (acr/visualization.py, acr/renderer/*) are out of scope: results are returned,
nothing is drawn.
"""
import logging

import numpy as np
import torch

from ..config import ConfigContext, args, parse_args, validate
from .mano_wrapper import MANOWrapper
from .model import ACR as ACR_v1
from .utils import (create_OneEuroFilter, get_remove_keys, img_preprocess, justify_detection_state, load_model,
                    reorganize_results, save_results, smooth_results)


class ACR(object):
    def __init__(self, args_set=None, state_dict=None, mano_tables=None, device=0, max_batch=1):
        """args_set: namespace from config.parse_args (default: config.args()).  state_dict / mano_tables
        let callers inject in-memory assets (tests, synthetic runs) instead of model_path / mano_root files."""
        a = validate(args() if args_set is None else args_set)
        self.demo_cfg = {'mode': 'parsing', 'calc_loss': False}
        for k, v in vars(a).items():
            setattr(self, k, v)
        logging.basicConfig(level=logging.INFO)
        self._args = a
        self._device = device
        self._build_model_(state_dict, mano_tables, device, max_batch)
        if self.temporal_optimization:
            # acr/main.py:45-47: one filter set per hand type; the state lives in the engine's context
            self.filter_dict = create_OneEuroFilter(a.smooth_coeff, engine=self.model.engine)

    def _build_model_(self, state_dict, mano_tables, device, max_batch):
        """acr/main.py:57-63"""
        with ConfigContext(self._args):
            model = ACR_v1(device=device, max_batch=max_batch).eval()
            if state_dict is not None:
                model.load_state_dict(state_dict)
            else:
                model = load_model(self.model_path, model, prefix='module.', drop_prefix='', fix_loaded=False)
            self.model = model.cuda(device)
            self.mano_regression = MANOWrapper(mano_root=self.mano_root, tables=mano_tables, device=device,
                                               engine=self.model.engine()).bind_model(self.model)

    @torch.no_grad()
    def process_results(self, outputs):
        """acr/main.py:66-89"""
        if self.temporal_optimization:
            pd = outputs['params_dict']
            if len(pd['poses']) != 2:
                raise ValueError('temporal optimisation expects exactly one frame (2 rows), as acr/main.py:77 asserts')
            # rows of a one-frame batch are [left, right] = the slot order: smooth the slots on the device and
            # refresh the rows the MANO stage reads (acr/main.py:69-83)
            from .. import _lib as S
            slots = smooth_results(self.filter_dict, outputs['slots'])
            pd['poses'].copy_(slots[0, :, S.SLOT_POSES:S.SLOT_POSES + 48])
            pd['betas'].copy_(slots[0, :, S.SLOT_BETAS:S.SLOT_BETAS + 10])
            pd['global_orient'], pd['hand_pose'] = pd['poses'][:, :3].contiguous(), pd['poses'][:, 3:].contiguous()
        outputs = self.mano_regression(outputs, outputs['meta_data'])
        reorganize_idx = outputs['reorganize_idx'].cpu().numpy()
        results = reorganize_results(outputs, outputs['meta_data']['imgpath'], reorganize_idx)
        return outputs, results

    @torch.no_grad()
    def single_image_forward(self, bgr_frame, path):
        """acr/main.py:126-141"""
        meta = img_preprocess(bgr_frame, path, input_size=self.input_size, single_img_input=True, device=self._device)
        ds_org, imgpath_org = get_remove_keys(meta, keys=['data_set', 'imgpath'])
        meta['batch_ids'] = torch.arange(len(meta['image']))
        outputs = self.model(meta, **self.demo_cfg)
        outputs['detection_flag'], outputs['reorganize_idx'] = justify_detection_state(outputs['detection_flag'],
                                                                                       outputs['reorganize_idx'])
        meta.update({'imgpath': imgpath_org, 'data_set': ds_org})
        outputs['meta_data']['imgpath'] = [path] * len(outputs['params_pred'])
        return outputs

    @torch.no_grad()
    def forward(self, bgr_frame, path):
        """acr/main.py:92-123 without the drawing: {path: [hand dicts]} or {path: {}} when nothing is detected."""
        outputs = self.single_image_forward(bgr_frame, path)
        if outputs is not None and outputs['detection_flag']:
            outputs, results = self.process_results(outputs)
            if self.save_dict_results and self.output_dir:
                save_results(results, self.output_dir.rstrip('/') + '/results.pkl')
        else:
            print('no hand detected!')
            results = {path: {}}
        return results

    __call__ = forward

    @torch.no_grad()
    def forward_batch(self, rgb_u8_frames, paths, offsets=None, point_heads=True, batch_semantics=None):
        """Batched throughput path: uint8 [B,512,512,3] RGB (already pre-processed) -> per-image results.
        One fused call (backbone, heads, decode, MANO, projection) + one D2H of the packed results.
        The head maps are not part of these results, so by default the params/cam/prior towers run only at the
        decoded centers (Engine.set_point_heads; same results within fp32 round-off) - point_heads=False runs the
        dense heads as `forward` does.
        batch_semantics: 'frame' | 'reference' (None = the model's ResultParser setting, args().batch_semantics): with
        'reference' the fused call applies the reference's batch-wide prior rules (acr/result_parser.py:42-47,102-145) on
        the device - decode, acrmi_prior_gate, gated decode - still ONE call (ACRMI_OPT_BATCH_PRIOR)."""
        eng = self.model.engine(rgb_u8_frames.shape[0])
        semantics = batch_semantics or self.model._result_parser.batch_semantics
        B = rgb_u8_frames.shape[0]
        if offsets is None:
            offsets = torch.tensor([[512., 512, 0, 0, 0, 0, 0, 0, 0, 0]]).repeat(B, 1)
        eng.set_point_heads(point_heads)
        # frames of the batch = one video stream, in order (smooth_coeff travels with the call: a reloaded checkpoint
        # builds a new context)
        eng.set_temporal(bool(self.temporal_optimization), smooth_coeff=self._args.smooth_coeff)
        eng.set_batch_semantics(semantics)
        try:
            out = eng.forward(rgb_u8_frames, offsets=offsets, project=True)
        finally:
            eng.set_point_heads(False)
            eng.set_temporal(False)
            eng.set_batch_semantics('frame')
        eng.check_range()      # 'fp16x3' only: an activation outside the f16 range is an error, not an empty result
        # cam_trans for every slot (acr/utils.py:399-412): the device least-squares kernel on [B*2] hands
        from .. import ops
        out['cam_trans'] = ops.cam_trans(out['joints'].view(-1, 21, 3), out['pj2d'].view(-1, 21, 2),
                                         focal_length=self.focal_length).view(B, 2, 3)
        slots = out['slots'].cpu().numpy()
        host = {k: out[k].cpu().numpy() for k in ('verts', 'joints', 'pj2d', 'pj2d_org', 'cam_trans')}
        from .. import _lib as S
        results = {}
        for b, path in enumerate(paths):
            hands = []
            for h in (0, 1):
                if slots[b, h, S.SLOT_FLAG] > 0.5:
                    s = slots[b, h]
                    hands.append({'cam': s[S.SLOT_CAM:S.SLOT_CAM + 3].astype(np.float16),
                                  'cam_trans': host['cam_trans'][b, h].astype(np.float16),
                                  'poses': s[S.SLOT_POSES:S.SLOT_POSES + 48].astype(np.float16),
                                  'betas': s[S.SLOT_BETAS:S.SLOT_BETAS + 10].astype(np.float16),
                                  'j3d': host['joints'][b, h].astype(np.float16),
                                  'verts': host['verts'][b, h].astype(np.float16),
                                  'pj2d': host['pj2d'][b, h].astype(np.float16),
                                  'pj2d_org': host['pj2d_org'][b, h].astype(np.float16),
                                  'hand_type': np.int32(h), 'detection_flag_cache': True})
            results[path] = hands if hands else {}
        return results


def _forward_raw_batch(self, bgr_frames_dev, paths):
    """BASELINE.json config 4: raw BGR uint8 frames [n,H,W,3] resident in HBM (e.g. 1080p video) - or a LIST of device frames
    [H_i,W_i,3] of different sizes (a folder of images, acr/main.py:144-205) - -> per-image results.  Pre-processing (white square pad + bicubic resize to 512) runs on the GPU (ops.preprocess),
    then the fused path; `offsets` carry the pad geometry so pj2d_org lands in original-frame pixels."""
    from .utils import img_preprocess_gpu
    meta = img_preprocess_gpu(bgr_frames_dev, paths)
    return self.forward_batch(meta['image'], paths, offsets=meta['offsets'])


ACR.forward_raw_batch = _forward_raw_batch


def main(argv=None):
    """python -m <package>.acr.main --demo_mode folder --inputs DIR : runs the path on .npy / image files it can
    read without cv2 (uint8 HxWx3 BGR arrays saved with numpy)."""
    import glob
    import os
    import sys
    a = parse_args(sys.argv[1:] if argv is None else argv)
    with ConfigContext(a):
        acr = ACR(args_set=a)
        files = sorted(glob.glob(os.path.join(a.inputs or '.', '*.npy')))
        for f in files:
            res = acr(np.load(f), f)
            print(f, {k: (len(v) if isinstance(v, list) else 0) for k, v in res.items()})


if __name__ == '__main__':
    main()
