"""Host glue of the ACR path with the reference's function names (acr/utils.py): checkpoint loading,
pre-processing, result packaging, temporal smoothing, camera translation.  None of this is on the
conv roofline; it runs on the host (numpy/torch-CPU) exactly where the reference runs it on the host.
"""
import logging
import os
import pickle

import numpy as np
import torch
import torch.nn.functional as F


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
    pad the shorter side symmetrically (extra pixel goes to bottom/right) -> (top, right, bottom, left)."""
    h, w = shape[:2]
    top = right = bottom = left = 0
    if w / float(h) < ratio:                     # too tall -> pad width
        diff = int(np.ceil(ratio * h - w))
        right, left = int(np.ceil(diff / 2)), int(np.floor(diff / 2))
    elif w / float(h) > ratio:
        diff = int(np.ceil(w / ratio - h))
        top, bottom = int(np.floor(diff / 2)), int(np.ceil(diff / 2))
    return top, right, bottom, left


def image_pad_white_bg(image, pad_trbl=None, pad_ratio=1., pad_cval=255):
    if pad_trbl is None:
        pad_trbl = compute_paddings_to_reach_aspect_ratio(image.shape, pad_ratio)
    t, r, b, l = pad_trbl
    out = np.pad(image, ((t, b), (l, r), (0, 0)), mode='constant', constant_values=pad_cval)
    return out, np.array([*out.shape[:2], 0, 0, 0, 0, *pad_trbl])


def resize_cubic(image_u8, size):
    """Stand-in for cv2.resize(..., INTER_CUBIC) (cv2 is absent here): bicubic, a = -0.75, half-pixel
    centres, no antialias - torch's kernel is the same family as OpenCV's.  Parity with cv2 is unpinned."""
    x = torch.from_numpy(np.ascontiguousarray(image_u8)).permute(2, 0, 1)[None].float()
    y = F.interpolate(x, size=(size, size), mode='bicubic', align_corners=False)
    return y.round().clamp(0, 255).to(torch.uint8)[0].permute(1, 2, 0).contiguous()


def img_preprocess(image, imgpath=None, input_size=512, single_img_input=False, bbox=None):
    """BGR frame -> {'image': uint8 [1,512,512,3] RGB, 'offsets': [1,10]} (acr/utils.py:1315-1337)."""
    image = np.ascontiguousarray(image[:, :, ::-1])
    padded, offsets = image_pad_white_bg(image)
    img = resize_cubic(padded, input_size)
    offsets = torch.from_numpy(offsets).float()
    if single_img_input:
        img, offsets = img.unsqueeze(0).contiguous(), offsets.unsqueeze(0).contiguous()
    data = {'image': img, 'offsets': offsets, 'data_set': 'internet'}
    if imgpath is not None:
        data.update({'imgpath': imgpath, 'name': os.path.basename(imgpath)})
    return data


def img_preprocess_gpu(bgr_frames, imgpaths=None):
    """Batched device pre-processing: uint8 BGR frames [n,H,W,3] already in HBM ->
    {'image': uint8 RGB [n,512,512,3] (device), 'offsets': [n,10], 'batch_ids': [n]}.  Same arithmetic as
    img_preprocess above (white square pad, bicubic a=-0.75), one HIP kernel, no host round trip."""
    from .. import ops
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
def estimate_translation_np(joints_3d, joints_2d, joints_conf, focal_length=600, img_size=np.array([512., 512.])):
    """Weighted least squares for the translation that best projects joints_3d onto joints_2d
    (acr/utils.py:430-472) - the closed form the reference falls back to when cv2.solvePnPRansac fails."""
    n = joints_3d.shape[0]
    f = np.array([focal_length, focal_length], np.float64)
    center = np.asarray(img_size, np.float64) / 2.
    Z = np.reshape(np.tile(joints_3d[:, 2], (2, 1)).T, -1)
    XY = np.reshape(joints_3d[:, 0:2], -1)
    O = np.tile(center, n)
    Fv = np.tile(f, n)
    w2 = np.reshape(np.tile(np.sqrt(joints_conf), (2, 1)).T, -1)
    Q = np.array([Fv * np.tile(np.array([1, 0]), n), Fv * np.tile(np.array([0, 1]), n), O - np.reshape(joints_2d, -1)]).T
    c = (np.reshape(joints_2d, -1) - O) * Z - Fv * XY
    W = np.diagflat(w2)
    Q, c = np.dot(W, Q), np.dot(W, c)
    return np.linalg.solve(np.dot(Q.T, Q), np.dot(Q.T, c))


def estimate_translation(joints_3d, pj2d, focal_length=600, img_size=np.array([512., 512.])):
    """Per-hand cam_trans (acr/utils.py:399-412,474-519): 2D targets are (pj2d+1)*256.
    The reference tries cv2 EPnP+RANSAC first (non-deterministic, render-only); this build always takes
    the reference's deterministic least-squares branch - on the device (acrmi_cam_trans) for device tensors,
    with the numpy restatement below for host tensors."""
    if joints_3d.is_cuda:
        from .. import ops
        return ops.cam_trans(joints_3d, pj2d, focal_length=focal_length, img_size=float(np.asarray(img_size).reshape(-1)[0]))
    j3 = joints_3d.detach().cpu().numpy().astype(np.float64)
    j2 = (pj2d.detach().cpu().numpy().astype(np.float64) + 1) * 256
    trans = np.zeros((j3.shape[0], 3))
    for i in range(j3.shape[0]):
        conf = np.ones(j3.shape[1], np.float32)
        trans[i] = estimate_translation_np(j3[i], j2[i], conf, focal_length=focal_length, img_size=img_size)
    return torch.from_numpy(trans).float()


# ---- temporal smoothing (acr/utils.py:1466-1527) -----------------------------------------------------------
class LowPassFilter(object):
    def __init__(self):
        self.prev_raw_value = None
        self.prev_filtered_value = None

    def process(self, value, alpha):
        s = value if self.prev_raw_value is None else alpha * value + (1.0 - alpha) * self.prev_filtered_value
        self.prev_raw_value = value
        self.prev_filtered_value = s
        return s


class OneEuroFilter(object):
    def __init__(self, mincutoff=1.0, beta=0.0, dcutoff=1.0, freq=30):
        self.freq, self.mincutoff, self.beta, self.dcutoff = freq, mincutoff, beta, dcutoff
        self.x_filter, self.dx_filter = LowPassFilter(), LowPassFilter()

    def compute_alpha(self, cutoff):
        te = 1.0 / self.freq
        tau = 1.0 / (2 * np.pi * cutoff)
        return 1.0 / (1.0 + tau / te)

    def process(self, x):
        prev_x = self.x_filter.prev_raw_value
        dx = 0.0 if prev_x is None else (x - prev_x) * self.freq
        edx = self.dx_filter.process(dx, self.compute_alpha(self.dcutoff))
        cutoff = self.mincutoff + self.beta * (torch.abs(edx) if isinstance(edx, torch.Tensor) else np.abs(edx))
        return self.x_filter.process(x, self.compute_alpha(cutoff))


def create_OneEuroFilter(smooth_coeff):
    return {'poses': OneEuroFilter(smooth_coeff, 0.7), 'betas': OneEuroFilter(0.6, 0.7),
            'global_orient': OneEuroFilter(smooth_coeff, 0.7)}


def _rodrigues_host(aa):
    """mano/manolayer.py:423-434 on one axis-angle vector (host tensor) -> [3,3]."""
    angle = torch.norm(aa + 1e-8)
    axis = aa / angle
    half = angle * 0.5
    q = torch.cat([torch.cos(half)[None], torch.sin(half) * axis])
    q = q / q.norm()
    w, x, y, z = q
    return torch.stack([w * w + x * x - y * y - z * z, 2 * x * y - 2 * w * z, 2 * w * y + 2 * x * z,
                        2 * w * z + 2 * x * y, w * w - x * x + y * y - z * z, 2 * y * z - 2 * w * x,
                        2 * x * z - 2 * w * y, 2 * w * x + 2 * y * z, w * w - x * x - y * y + z * z]).view(3, 3)


def _rotmat_to_aa_host(R):
    """acr/utils.py:334-360 on one matrix (via quaternion, NaN -> 0)."""
    t = R.t()
    if t[2, 2] < 1e-6:
        if t[0, 0] > t[1, 1]:
            tr = 1 + t[0, 0] - t[1, 1] - t[2, 2]
            q = torch.stack([t[1, 2] - t[2, 1], tr, t[0, 1] + t[1, 0], t[2, 0] + t[0, 2]])
        else:
            tr = 1 - t[0, 0] + t[1, 1] - t[2, 2]
            q = torch.stack([t[2, 0] - t[0, 2], t[0, 1] + t[1, 0], tr, t[1, 2] + t[2, 1]])
    elif t[0, 0] < -t[1, 1]:
        tr = 1 - t[0, 0] - t[1, 1] + t[2, 2]
        q = torch.stack([t[0, 1] - t[1, 0], t[2, 0] + t[0, 2], t[1, 2] + t[2, 1], tr])
    else:
        tr = 1 + t[0, 0] + t[1, 1] + t[2, 2]
        q = torch.stack([tr, t[1, 2] - t[2, 1], t[2, 0] - t[0, 2], t[0, 1] - t[1, 0]])
    q = q / torch.sqrt(tr) * 0.5
    s2 = q[1] * q[1] + q[2] * q[2] + q[3] * q[3]
    s = torch.sqrt(s2)
    two_theta = 2.0 * (torch.atan2(-s, -q[0]) if q[0] < 0 else torch.atan2(s, q[0]))
    k = two_theta / s if s2 > 0 else torch.tensor(2.0)
    aa = q[1:] * k
    return torch.where(torch.isnan(aa), torch.zeros_like(aa), aa)


def smooth_results(filters, body_pose=None, body_shape=None):
    """acr/utils.py:1475-1479: global orient filtered in rotation-matrix space, fingers/betas directly."""
    rot = filters['global_orient'].process(_rodrigues_host(body_pose[:3]))
    body_pose = torch.cat([_rotmat_to_aa_host(rot), filters['poses'].process(body_pose[3:])], 0)
    return body_pose, filters['betas'].process(body_shape)
