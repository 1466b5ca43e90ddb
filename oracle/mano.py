"""ORACLE (test infrastructure, not product): CPU restatement of the MANO forward and projection.

ManoLayer configuration fixed by acr/mano_wrapper.py:17-35: use_pca=False,
flat_hand_mean=False, root_rot_mode='axisang', center_idx=9.  Pinned by
tests/golden/mano_*.npz captured from the imported reference.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import numpy as np
import torch

PARENTS = [-1, 0, 1, 2, 0, 4, 5, 0, 7, 8, 0, 10, 11, 0, 13, 14]          # mano/manolayer.py:100-102
JOINT_REORDER = [0, 13, 14, 15, 16, 1, 2, 3, 17, 4, 5, 6, 18, 10, 11, 12, 19, 7, 8, 9, 20]  # :254
TIPS = {'right': [745, 317, 444, 556, 673], 'left': [745, 317, 445, 556, 673]}              # :244-247


def batch_rodrigues(aa):
    """mano/manolayer.py:423-434 + quat2mat :396-421.  aa [N,3] -> [N,9]."""
    angle = torch.norm(aa + 1e-8, p=2, dim=1, keepdim=True)
    axis = aa / angle
    half = angle * 0.5
    q = torch.cat([torch.cos(half), torch.sin(half) * axis], 1)
    q = q / q.norm(p=2, dim=1, keepdim=True)
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    w2, x2, y2, z2 = w * w, x * x, y * y, z * z
    wx, wy, wz, xy, xz, yz = w * x, w * y, w * z, x * y, x * z, y * z
    return torch.stack([w2 + x2 - y2 - z2, 2 * xy - 2 * wz, 2 * wy + 2 * xz,
                        2 * wz + 2 * xy, w2 - x2 + y2 - z2, 2 * yz - 2 * wx,
                        2 * xz - 2 * wy, 2 * wx + 2 * yz, w2 - x2 - y2 + z2], 1)


def batch_rotprojs(rotmats):
    """mano/manolayer.py:436-453: [N,J,3,3] -> U V^T of each matrix's (CPU) SVD, last column negated for reflections."""
    m = torch.as_tensor(rotmats, dtype=torch.float32)
    out = torch.empty_like(m)
    for b in range(m.shape[0]):
        for r in range(m.shape[1]):
            U, S, V = m[b, r].svd()
            rot = torch.matmul(U, V.transpose(0, 1))
            if rot.det() < 0:
                rot[:, 2] = -1 * rot[:, 2]
            out[b, r] = rot
    return out


@torch.no_grad()
def mano_forward(tables, side, poses, betas, center_idx=9):
    """mano/manolayer.py:104-276.  tables: float32 arrays of one side (left: shapedirs already
    x-flipped by the caller, acr/mano_wrapper.py:35).  poses [N,48], betas [N,10]
    -> verts [N,778,3], joints [N,21,3], center [N,1,3] (numpy float32).
    poses [N,16,3,3]: joint_rot_mode='rotmat' (:151-162) - matrices through batch_rotprojs, no th_hands_mean.
    center_idx None: no root alignment (center = None)."""
    T = {k: torch.as_tensor(np.asarray(v), dtype=torch.float32) for k, v in tables.items()
         if k in ('v_template', 'shapedirs', 'posedirs', 'J_regressor', 'weights', 'hands_mean')}
    poses = torch.as_tensor(poses, dtype=torch.float32)
    betas = torch.as_tensor(betas, dtype=torch.float32)
    N = poses.shape[0]
    if poses.dim() == 4:
        rot = batch_rotprojs(poses).reshape(N, 16, 9)
    else:
        full = torch.cat([poses[:, :3], T['hands_mean'].view(1, 45) + poses[:, 3:48]], 1)
        rot = batch_rodrigues(full.reshape(-1, 3)).view(N, 16, 9)
    eye = torch.eye(3).view(1, 1, 9)
    pose_map = (rot[:, 1:] - eye).reshape(N, 135)
    R = rot.view(N, 16, 3, 3)
    v_shaped = torch.matmul(T['shapedirs'], betas.t()).permute(2, 0, 1) + T['v_template'].unsqueeze(0)
    J = torch.matmul(T['J_regressor'], v_shaped)                        # [N,16,3]
    v_posed = v_shaped + torch.matmul(T['posedirs'], pose_map.t()).permute(2, 0, 1)
    # forward kinematics in kinematic-tree order (reference builds 3 levels then reorders; same result)
    G = [None] * 16
    for j in range(16):
        p = PARENTS[j]
        t = J[:, j] if p < 0 else J[:, j] - J[:, p]
        M = torch.zeros(N, 4, 4)
        M[:, :3, :3] = R[:, j]
        M[:, :3, 3] = t
        M[:, 3, 3] = 1.0
        G[j] = M if p < 0 else torch.matmul(G[p], M)
    G = torch.stack(G, 1)                                               # [N,16,4,4]
    Jh = torch.cat([J, torch.zeros(N, 16, 1)], 2).unsqueeze(3)          # [N,16,4,1]
    tmp = torch.matmul(G, Jh)
    A = G - torch.cat([torch.zeros(N, 16, 4, 3), tmp], 3)               # rest-pose removed
    Tv = torch.matmul(A.permute(0, 2, 3, 1), T['weights'].t())          # [N,4,4,778]
    rest = torch.cat([v_posed.transpose(2, 1), torch.ones(N, 1, 778)], 1)
    verts = (Tv * rest.unsqueeze(1)).sum(2).transpose(2, 1)[:, :, :3]
    jtr = torch.cat([G[:, :, :3, 3], verts[:, TIPS[side]]], 1)[:, JOINT_REORDER]
    if center_idx is None:
        return verts.numpy(), jtr.numpy(), None
    center = jtr[:, center_idx].unsqueeze(1)
    return (verts - center).numpy(), (jtr - center).numpy(), center.numpy()


def project(verts, joints, cam, offsets):
    """acr/utils.py:384-412: weak-perspective projection.
    verts [H,778,3], joints [H,21,3], cam [H,3] (s,tx,ty), offsets [H,10]
    -> verts_camed [H,778,3], pj2d [H,21,2], pj2d_org [H,21,2]."""
    verts, joints, cam, offsets = [np.asarray(a, np.float32) for a in (verts, joints, cam, offsets)]
    s = cam[:, None, 0:1]
    t = cam[:, None, 1:3]
    vc = np.concatenate([verts[:, :, :2] * s + t, verts[:, :, 2:3]], -1)
    pj = joints[:, :, :2] * s + t
    pad = offsets[:, None, 0:2]
    crop, padt = offsets[:, 2:6], offsets[:, 6:10]
    left_top = np.stack([crop[:, 3] - padt[:, 3], crop[:, 0] - padt[:, 0]], 1)[:, None]
    org = (pj + 1) * pad / 2 + left_top
    return vc.astype(np.float32), pj.astype(np.float32), org.astype(np.float32)
