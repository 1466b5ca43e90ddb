"""Host glue of the ACR path with the reference's function names (acr/utils.py): checkpoint loading, result
packaging, and thin wrappers that send pre-processing, temporal smoothing and camera translation to their HIP
kernels (the reference runs those three on the host; here they stay on the device next to the data).
"""
import logging
import os
import pickle

import numpy as np
import torch


# ---- checkpoint (acr/utils.py:1106-1168) -----------------------------------------------------------
def load_model(path, model, prefix='module.', drop_prefix='', optimizer=None, **kwargs):
    """torch.load + unwrap 'model_state_dict'/'state_dict' + copy by key with the 'module.' prefix.
    Missing/mismatched tensors are skipped with a log line, as copy_state_dict does; a missing file
    raises ValueError like the reference."""
    logging.info('using fine_tune model: %s', path)
    if not os.path.exists(path):
        logging.warning('model %s not exist!', path)
        raise ValueError('checkpoint %s does not exist' % path)
    pre = torch.load(path, map_location='cpu', weights_only=False)
    if isinstance(pre, dict):
        for k in ('model_state_dict', 'state_dict'):
            if k in pre:
                pre = pre[k]
    cur = model.state_dict()
    picked, failed = {}, []
    for k in cur:
        key = prefix + k.replace(drop_prefix, '')
        v = pre.get(key, pre.get(k))
        if v is None or tuple(v.shape) != tuple(cur[k].shape):
            failed.append(k)
            continue
        picked[k] = v
    logging.info('missing parameters of layers:%d, %s', len(failed), failed[:8])
    logging.info('success layers:%d/%d, pre_state_dict have %d', len(picked), len(cur), len(pre))
    model.load_state_dict(picked, strict=False)
    return model


# ---- pre-processing (acr/utils.py:1276-1337) ---------------------------------------------------------
def compute_paddings_to_reach_aspect_ratio(shape, ratio=1.0):
    """imgaug.augmenters.size.compute_paddings_to_reach_aspect_ratio (imgaug 0.4.0) restated:
    pad the shorter side symmetrically (extra pixel goes to bottom/right) -> (top, right, bottom, left).
    (Geometry only; the device kernel computes the same numbers in acrmi_preprocess.)"""
    h, w = shape[:2]
    top = right = bottom = left = 0
    if w / float(h) < ratio:                     # too tall -> pad width
        diff = int(np.ceil(ratio * h - w))
        right, left = int(np.ceil(diff / 2)), int(np.floor(diff / 2))
    elif w / float(h) > ratio:
        diff = int(np.ceil(w / ratio - h))
        top, bottom = int(np.floor(diff / 2)), int(np.ceil(diff / 2))
    return top, right, bottom, left


def img_preprocess(image, imgpath=None, input_size=512, single_img_input=False, bbox=None, device=0):
    """BGR frame (numpy HxWx3 uint8, or a uint8 tensor) -> {'image': uint8 [1,512,512,3] RGB, 'offsets': [1,10]}
    (acr/utils.py:1315-1337).  The white pad + cv2.resize(INTER_CUBIC) run in the HIP pre-processing kernel
    (acrmi_preprocess: OpenCV's uint8 fixed-point cubic restated bit for bit); the image stays in HBM."""
    if input_size != 512:
        raise ValueError('only input_size=512 is implemented')
    if bbox is not None:
        raise ValueError('bbox cropping is not used by the demo path (acr/main.py:130) and is not implemented')
    frame = image if isinstance(image, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(image))
    if frame.dtype != torch.uint8 or frame.dim() != 3 or frame.shape[-1] != 3:
        raise ValueError('frame must be uint8 HxWx3 BGR')
    if not frame.is_cuda:
        if not torch.cuda.is_available():
            from .. import _lib
            raise _lib.AcrmiError('no GPU visible: pre-processing runs in the HIP kernel (no CPU fallback)')
        frame = frame.to(torch.device('cuda', device))
    data = img_preprocess_gpu(frame[None], None)
    img, offsets = data['image'], data['offsets']
    if not single_img_input:
        img, offsets = img[0], offsets[0]
    data = {'image': img, 'offsets': offsets, 'data_set': 'internet'}
    if imgpath is not None:
        data.update({'imgpath': imgpath, 'name': os.path.basename(imgpath)})
    return data


def img_preprocess_gpu(bgr_frames, imgpaths=None):
    """Batched device pre-processing: uint8 BGR frames [n,H,W,3] already in HBM - or a list of frames
    [H_i,W_i,3] of different sizes (acrmi_preprocess_frames) ->
    {'image': uint8 RGB [n,512,512,3] (device), 'offsets': [n,10], 'batch_ids': [n]}: one HIP kernel, no host
    round trip."""
    from .. import ops
    if isinstance(bgr_frames, (list, tuple)):        # frames of different sizes (folder mode): per-frame geometry, one call
        img, offsets = ops.preprocess_frames(bgr_frames)
    else:
        img, offsets = ops.preprocess(bgr_frames)
    data = {'image': img, 'offsets': offsets, 'data_set': 'internet', 'batch_ids': torch.arange(img.shape[0])}
    if imgpaths is not None:
        data['imgpath'] = list(imgpaths)
    return data


# ---- result packaging (acr/utils.py:1098-1104, 1192-1271) ----------------------------------------------
def justify_detection_state(detection_flag, reorganize_idx):
    if detection_flag.sum() == 0:
        detection_flag = False
    else:
        reorganize_idx = reorganize_idx[detection_flag.bool()].long()
        detection_flag = True
    return detection_flag, reorganize_idx


def get_remove_keys(dt, keys=()):
    targets = [dt[k] for k in keys]
    for k in keys:
        del dt[k]
    return targets


def reorganize_results(outputs, img_paths, reorganize_idx):
    """Per-image list of per-hand dicts, everything cast to float16 numpy (acr/utils.py:1226-1271)."""
    det = outputs['detection_flag_cache'].detach().cpu().numpy().astype(np.bool_)

    def f16(t):
        return t.detach().cpu().numpy().astype(np.float16)[det]
    pd = outputs['params_dict']
    fields = {'cam': f16(pd['cam']), 'cam_trans': f16(outputs['cam_trans']), 'poses': f16(pd['poses']),
              'betas': f16(pd['betas']), 'j3d': f16(outputs['j3d']), 'verts': f16(outputs['verts']),
              'pj2d': f16(outputs['pj2d']), 'pj2d_org': f16(outputs['pj2d_org'])}
    hand_type = outputs['output_hand_type'].detach().cpu().numpy().astype(np.int32)[det]
    results = {}
    for vid in np.unique(reorganize_idx):
        rows = np.where(reorganize_idx == vid)[0]
        path = img_paths[rows[0]]
        results[path] = []
        for r in rows:
            d = {k: v[r] for k, v in fields.items()}
            d['hand_type'] = hand_type[r]
            d['detection_flag_cache'] = det[det][r]
            results[path].append(d)
    return results


def save_results(results, path):
    with open(path, 'wb') as f:
        pickle.dump(results, f)


# ---- camera translation (acr/utils.py:430-519) -----------------------------------------------------------
def estimate_translation(joints_3d, pj2d, focal_length=600, img_size=np.array([512., 512.])):
    """Per-hand cam_trans (acr/utils.py:399-412,474-519): 2D targets are (pj2d+1)*256.
    The reference tries cv2 EPnP+RANSAC first (non-deterministic, render-only); this build always takes the
    reference's deterministic least-squares branch (acr/utils.py:430-472), on the device (acrmi_cam_trans)."""
    from .. import ops
    return ops.cam_trans(joints_3d, pj2d, focal_length=focal_length, img_size=float(np.asarray(img_size).reshape(-1)[0]))


# ---- temporal smoothing (acr/utils.py:1466-1527) -----------------------------------------------------------
class DeviceOneEuro(object):
    """The reference keeps one dict of OneEuroFilters per hand type on the host (acr/main.py:45-47) and filters row by
    row with a D2H/H2D per frame.  Here the filter state of the stream lives in the engine's context and
    `acrmi_smooth` updates both hands' poses/betas in one launch between decode and MANO."""

    def __init__(self, engine, smooth_coeff):
        # engine: an Engine, or a callable returning the CURRENT one (acr.model.ACR.engine: a checkpoint reload
        # replaces the context, and the filter must follow it instead of calling into the closed one)
        self._engine = engine if callable(engine) else (lambda: engine)
        self.smooth_coeff = float(smooth_coeff)
        self._bound = None
        self._bind()

    def _bind(self):
        eng = self._engine()
        if eng is not self._bound:        # first use, or the model rebuilt its engine: a fresh stream state there
            eng.set_temporal(eng.temporal, smooth_coeff=self.smooth_coeff)
            eng.smooth_reset()
            self._bound = eng
        return eng

    @property
    def engine(self):
        return self._bind()

    def process_slots(self, slots):
        return self._bind().smooth(slots)


def create_OneEuroFilter(smooth_coeff, engine=None):
    """acr/utils.py:1472-1473.  With an engine: the device filter set; the per-filter parameters (poses / global
    orient: mincutoff = smooth_coeff, betas: 0.6; beta 0.7) are fixed inside acrmi_smooth."""
    if engine is None:
        raise ValueError('create_OneEuroFilter needs the engine whose context holds the filter state')
    return DeviceOneEuro(engine, smooth_coeff)


def smooth_results(filters, slots):
    """acr/utils.py:1475-1479 over a [B,2,176] slot tensor (in place): global orient filtered in rotation-matrix
    space, fingers and betas directly, only for hands whose detection flag is set."""
    return filters.process_slots(slots)
