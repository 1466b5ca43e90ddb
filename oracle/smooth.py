"""TEST INFRASTRUCTURE (oracle): CPU restatement of the reference's temporal smoothing and camera translation.

  * One-Euro smoothing of (poses, betas) per hand type: acr/utils.py:1466-1527 (smooth_global_rot_matrix,
    create_OneEuroFilter, smooth_results, LowPassFilter, OneEuroFilter), called from acr/main.py:69-83; the global
    orientation goes through acr/utils.py:602-638 (batch_rodrigues, quat2mat) and :334-360,773-906
    (rotation_matrix_to_angle_axis).
  * estimate_translation_np: acr/utils.py:430-472, the closed-form least squares the reference falls back to when
    cv2.solvePnPRansac is unavailable (acr/utils.py:512-517).

Plain numpy float32 (float64 for the least squares, like the reference), state held in small dicts.  Pinned by
tests/test_oracle_pinned.py against sequences captured from the real reference (tests/golden/smooth_seq.npz,
e2e_batch1.npz cam_trans).  Only tests/ (and bench.py's cpu_baseline) may import this.
"""
import numpy as np

F = np.float32
FREQ = 30.0


def _alpha_scalar(cutoff):
    """OneEuroFilter.compute_alpha on a python/numpy double (the first-sample and dx-filter path)."""
    te = 1.0 / FREQ
    tau = 1.0 / (2 * np.pi * cutoff)
    return 1.0 / (1.0 + tau / te)


def _alpha_tensor(cutoff):
    """compute_alpha on a float32 tensor: every python scalar is rounded to float32 when it meets the tensor."""
    te = F(1.0 / FREQ)
    tau = F(1.0) / (F(2 * np.pi) * cutoff)
    return F(1.0) / (F(1.0) + tau / te)


def new_filter(mincutoff, beta=0.7, dcutoff=1.0):
    return {'mincutoff': float(mincutoff), 'beta': float(beta), 'dcutoff': float(dcutoff),
            'x_raw': None, 'x_filt': None, 'dx_filt': None}


def new_filters(smooth_coeff):
    """create_OneEuroFilter (acr/utils.py:1472-1473)."""
    return {'poses': new_filter(smooth_coeff), 'betas': new_filter(0.6), 'global_orient': new_filter(smooth_coeff)}


def one_euro(f, x):
    """OneEuroFilter.process on a float32 array (acr/utils.py:1513-1527)."""
    x = np.asarray(x, F)
    if f['x_raw'] is None:                    # dx = 0.0, edx = 0.0, first sample passes through
        f['x_raw'], f['x_filt'], f['dx_filt'] = x.copy(), x.copy(), np.zeros_like(x)
        return x.copy()
    dx = (x - f['x_raw']) * F(FREQ)
    ad = _alpha_scalar(f['dcutoff'])
    edx = F(ad) * dx + F(1.0 - ad) * f['dx_filt']
    cutoff = F(f['mincutoff']) + F(f['beta']) * np.abs(edx)
    a = _alpha_tensor(cutoff)
    s = a * x + (F(1.0) - a) * f['x_filt']
    f['x_raw'], f['x_filt'], f['dx_filt'] = x.copy(), s.astype(F), edx.astype(F)
    return s.astype(F)


def rodrigues(aa):
    """acr/utils.py:602-638 on one axis-angle vector -> [3,3] float32 (norm of aa + 1e-8, via a quaternion)."""
    aa = np.asarray(aa, F)
    t = aa + F(1e-8)
    angle = np.sqrt((t * t).sum(dtype=F), dtype=F)
    n = aa / angle
    half = angle * F(0.5)
    q = np.concatenate([[np.cos(half, dtype=F)], np.sin(half, dtype=F) * n]).astype(F)
    q = q / np.sqrt((q * q).sum(dtype=F), dtype=F)
    w, x, y, z = q
    w2, x2, y2, z2 = w * w, x * x, y * y, z * z
    wx, wy, wz, xy, xz, yz = w * x, w * y, w * z, x * y, x * z, y * z
    two = F(2)
    return np.array([[w2 + x2 - y2 - z2, two * xy - two * wz, two * wy + two * xz],
                     [two * wz + two * xy, w2 - x2 + y2 - z2, two * yz - two * wx],
                     [two * xz - two * wy, two * wx + two * yz, w2 - x2 - y2 + z2]], F)


def rotmat_to_aa(R):
    """acr/utils.py:334-360 -> :826-906 (quaternion from the transposed matrix, 4-way mask, eps 1e-6) -> :773-823
    (angle-axis with the atan2 sign handling); NaN -> 0."""
    t = np.asarray(R, F).T
    one = F(1)
    if t[2, 2] < F(1e-6):
        if t[0, 0] > t[1, 1]:
            tr = one + t[0, 0] - t[1, 1] - t[2, 2]
            q = np.array([t[1, 2] - t[2, 1], tr, t[0, 1] + t[1, 0], t[2, 0] + t[0, 2]], F)
        else:
            tr = one - t[0, 0] + t[1, 1] - t[2, 2]
            q = np.array([t[2, 0] - t[0, 2], t[0, 1] + t[1, 0], tr, t[1, 2] + t[2, 1]], F)
    elif t[0, 0] < -t[1, 1]:
        tr = one - t[0, 0] - t[1, 1] + t[2, 2]
        q = np.array([t[0, 1] - t[1, 0], t[2, 0] + t[0, 2], t[1, 2] + t[2, 1], tr], F)
    else:
        tr = one + t[0, 0] + t[1, 1] + t[2, 2]
        q = np.array([tr, t[1, 2] - t[2, 1], t[2, 0] - t[0, 2], t[0, 1] - t[1, 0]], F)
    with np.errstate(all='ignore'):
        q = (q / np.sqrt(tr, dtype=F)) * F(0.5)
        s2 = q[1] * q[1] + q[2] * q[2] + q[3] * q[3]
        s = np.sqrt(s2, dtype=F)
        two_theta = F(2) * (np.arctan2(-s, -q[0], dtype=F) if q[0] < 0 else np.arctan2(s, q[0], dtype=F))
        k = two_theta / s if s2 > 0 else F(2)
        aa = (q[1:] * k).astype(F)
    aa[np.isnan(aa)] = 0
    return aa


def smooth_results(filters, pose48, betas10):
    """acr/utils.py:1475-1479: global orient filtered as a rotation matrix, fingers and betas directly."""
    rot = one_euro(filters['global_orient'], rodrigues(pose48[:3]))
    pose = np.concatenate([rotmat_to_aa(rot), one_euro(filters['poses'], pose48[3:])]).astype(F)
    return pose, one_euro(filters['betas'], betas10)


def estimate_translation_np(joints_3d, joints_2d, joints_conf, focal_length=600, img_size=(512., 512.)):
    """acr/utils.py:430-472: weighted least squares for the translation that projects joints_3d onto joints_2d."""
    j3 = np.asarray(joints_3d, np.float64)
    j2 = np.asarray(joints_2d, np.float64)
    n = j3.shape[0]
    focal = np.array([focal_length, focal_length], np.float64)
    center = np.asarray(img_size, np.float64) / 2.
    depth = np.repeat(j3[:, 2], 2)
    plane = j3[:, :2].reshape(-1)
    origin = np.tile(center, n)
    fl = np.tile(focal, n)
    wgt = np.repeat(np.sqrt(np.asarray(joints_conf, np.float64)), 2)
    ex, ey = np.tile([1., 0.], n), np.tile([0., 1.], n)
    lhs = np.stack([fl * ex, fl * ey, origin - j2.reshape(-1)], 1) * wgt[:, None]
    rhs = ((j2.reshape(-1) - origin) * depth - fl * plane) * wgt
    return np.linalg.solve(lhs.T @ lhs, lhs.T @ rhs)


def estimate_translation(joints_3d, pj2d, focal_length=600, img_size=(512., 512.)):
    """acr/utils.py:474-519 with the least-squares branch: pj2d in [-1,1] -> pixels (pj2d+1)*img/2, unit confidences."""
    j3 = np.asarray(joints_3d, np.float64)
    half = float(np.asarray(img_size, np.float64).reshape(-1)[0]) / 2.
    j2 = (np.asarray(pj2d, np.float64) + 1) * half
    out = np.zeros((j3.shape[0], 3))
    for i in range(j3.shape[0]):
        out[i] = estimate_translation_np(j3[i], j2[i], np.ones(j3.shape[1]), focal_length, img_size)
    return out.astype(np.float32)
