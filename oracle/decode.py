"""ORACLE (test infrastructure, not product): CPU restatement of the ACR decode step.

center-map NMS / top-1 / threshold, parameter sampling, cross-hand prior, 109-split and
6D -> axis-angle.  decode(maps) has PER-FRAME semantics (= the reference run N times at batch 1,
which is the only way acr/main.py:126-141 ever calls it); decode(maps, batch_semantics='reference')
restates what the reference's parse_maps does with a batch > 1 (prior gated on every flag of the
batch, determine_coeff reading row 0 of each side's list; acr/result_parser.py:42-47,131) and
slots_to_rows_batch the whole-batch placeholder rows (:102-120) - pinned by tests/golden/decode_batches.npz.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import numpy as np
import torch
import torch.nn.functional as F

CONF_THRESH = 0.35     # acr/config.py centermap_conf_thresh
MAP = 64               # centermap_size


def nms_top1(center_map):
    """acr/result_parser.py:218-249.  center_map [B,1,64,64] -> (flat_ind [B] int64, score [B])."""
    maxm = F.max_pool2d(center_map, 5, 1, 2)
    det = center_map * torch.eq(maxm, center_map).float()
    score, ind = torch.topk(det.reshape(det.shape[0], -1), 1)
    return ind[:, 0], score[:, 0]


def rot6d_to_rotmat(x):
    """acr/utils.py:362-376: view(-1,3,2): b1 = elements (0,2,4), a2 = (1,3,5)."""
    x = x.reshape(-1, 3, 2)
    b1 = F.normalize(x[:, :, 0], dim=1, eps=1e-6)
    dot = torch.sum(b1 * x[:, :, 1], dim=1, keepdim=True)
    b2 = F.normalize(x[:, :, 1] - dot * b1, dim=-1, eps=1e-6)
    b3 = torch.cross(b1, b2, dim=1)
    return torch.stack([b1, b2, b3], dim=-1)


def rotmat_to_quat(R, eps=1e-6):
    """acr/utils.py:826-906 (works on the transpose of R; four masked branches)."""
    t = R.transpose(1, 2)
    m_d2 = t[:, 2, 2] < eps
    m_d0_d1 = t[:, 0, 0] > t[:, 1, 1]
    m_d0_nd1 = t[:, 0, 0] < -t[:, 1, 1]
    t0 = 1 + t[:, 0, 0] - t[:, 1, 1] - t[:, 2, 2]
    q0 = torch.stack([t[:, 1, 2] - t[:, 2, 1], t0, t[:, 0, 1] + t[:, 1, 0], t[:, 2, 0] + t[:, 0, 2]], -1)
    t1 = 1 - t[:, 0, 0] + t[:, 1, 1] - t[:, 2, 2]
    q1 = torch.stack([t[:, 2, 0] - t[:, 0, 2], t[:, 0, 1] + t[:, 1, 0], t1, t[:, 1, 2] + t[:, 2, 1]], -1)
    t2 = 1 - t[:, 0, 0] - t[:, 1, 1] + t[:, 2, 2]
    q2 = torch.stack([t[:, 0, 1] - t[:, 1, 0], t[:, 2, 0] + t[:, 0, 2], t[:, 1, 2] + t[:, 2, 1], t2], -1)
    t3 = 1 + t[:, 0, 0] + t[:, 1, 1] + t[:, 2, 2]
    q3 = torch.stack([t3, t[:, 1, 2] - t[:, 2, 1], t[:, 2, 0] - t[:, 0, 2], t[:, 0, 1] - t[:, 1, 0]], -1)
    c0 = (m_d2 & m_d0_d1).view(-1, 1).float()
    c1 = (m_d2 & ~m_d0_d1).view(-1, 1).float()
    c2 = (~m_d2 & m_d0_nd1).view(-1, 1).float()
    c3 = (~m_d2 & ~m_d0_nd1).view(-1, 1).float()
    q = q0 * c0 + q1 * c1 + q2 * c2 + q3 * c3
    q = q / torch.sqrt(t0.view(-1, 1) * c0 + t1.view(-1, 1) * c1 + t2.view(-1, 1) * c2 + t3.view(-1, 1) * c3)
    return q * 0.5


def quat_to_aa(q):
    """acr/utils.py:773-823"""
    q1, q2, q3 = q[..., 1], q[..., 2], q[..., 3]
    s2 = q1 * q1 + q2 * q2 + q3 * q3
    s = torch.sqrt(s2)
    c = q[..., 0]
    two_theta = 2.0 * torch.where(c < 0.0, torch.atan2(-s, -c), torch.atan2(s, c))
    k = torch.where(s2 > 0.0, two_theta / s, 2.0 * torch.ones_like(s))
    return torch.stack([q1 * k, q2 * k, q3 * k], -1)


def rot6d_to_aa(x6):
    """acr/utils.py:378-382 + :334-360 (NaN -> 0).  x6 [N, 6*J] -> [N, 3*J]."""
    n = x6.shape[0]
    R = rot6d_to_rotmat(x6)
    aa = quat_to_aa(rotmat_to_quat(R))
    aa = torch.where(torch.isnan(aa), torch.zeros_like(aa), aa)
    return aa.reshape(n, -1)


def sample(maps, b, flat):
    """acr/result_parser.py:49-57: maps [B,C,64,64] -> [C] at (frame b, flat index)."""
    return maps[b].reshape(maps.shape[1], -1)[:, flat]


def reference_gate(flag, flat):
    """acr/result_parser.py:124-145 at batch > 1: which frames' rows take their cross-hand prior.
    flag [B,2] bool, flat [B,2] int64 (y*64+x) -> bool [B]."""
    B = flag.shape[0]
    gate = torch.zeros(B, dtype=torch.bool)
    l_ids = [b for b in range(B) if flag[b, 0]]            # l_batch_ids (:93), ascending
    r_ids = [b for b in range(B) if flag[b, 1]]
    # detection_flag holds a False for a side without any hit in the batch (:106,116); :131 wants every entry True
    if not l_ids or not r_ids:
        return gate
    both = [b for b in l_ids if b in r_ids]                # all_hand_valid_batch_ids (:126-128)
    if not both:
        return gate
    # determine_coeff (:42-47): l_centers[0] vs r_centers[0] = the first detected left / right center of the batch
    fl, fr = int(flat[l_ids[0], 0]), int(flat[r_ids[0], 1])
    diff = torch.sqrt(torch.tensor(float((fl // MAP - fr // MAP) ** 2 + (fl % MAP - fr % MAP) ** 2)))
    if diff > 32:
        return gate                                        # both priors := 0 for the whole batch
    for b in both:
        gate[b] = True
    return gate


@torch.no_grad()
def decode(maps, batch_semantics='frame'):
    """maps: head dict (l/r_params_maps, l/r_center_map, l/r_prior_maps).
    batch_semantics: 'frame' (below) | 'reference' (the prior decided batch-wide by reference_gate).
    Returns per-frame, per-hand slots (hand 0 = left, 1 = right), all float32/np:
      flag [B,2] bool, flat_ind [B,2] int64, score [B,2], params_pred [B,2,109],
      cam [B,2,3], poses [B,2,48], betas [B,2,10]
    An undetected hand samples pixel 0 of its own frame (acr/result_parser.py:106-120 at batch 1).
    """
    B = maps['l_center_map'].shape[0]
    ind = {}
    score = {}
    for s in 'lr':
        ind[s], score[s] = nms_top1(maps[s + '_center_map'])
    flag = torch.stack([score['l'] > CONF_THRESH, score['r'] > CONF_THRESH], 1)
    flat = torch.stack([ind['l'], ind['r']], 1)
    flat = torch.where(flag, flat, torch.zeros_like(flat))
    pred = torch.zeros(B, 2, 109)
    gate = reference_gate(flag, flat) if batch_semantics == 'reference' else None
    for b in range(B):
        l = sample(maps['l_params_maps'], b, flat[b, 0]).clone()
        r = sample(maps['r_params_maps'], b, flat[b, 1]).clone()
        if gate is not None:
            use = bool(gate[b])
        else:
            use = False
            if flag[b, 0] and flag[b, 1]:
                # cross prior (acr/result_parser.py:131-145) + determine_coeff (:42-47): centers as (y,x)
                ly, lx = flat[b, 0] // MAP, flat[b, 0] % MAP
                ry, rx = flat[b, 1] // MAP, flat[b, 1] % MAP
                diff = torch.sqrt(((ly - ry).float()) ** 2 + ((lx - rx).float()) ** 2)
                use = not diff > 32
        if use:
            l[3:] += sample(maps['l_prior_maps'], b, flat[b, 1])
            r[3:] += sample(maps['r_prior_maps'], b, flat[b, 0])
        pred[b, 0], pred[b, 1] = l, r
    flatp = pred.reshape(B * 2, 109)
    cam = flatp[:, :3]
    orient = rot6d_to_aa(flatp[:, 3:9].contiguous())
    pose = rot6d_to_aa(flatp[:, 9:99].contiguous())
    betas = flatp[:, 99:109]
    poses = torch.cat([orient, pose], 1)
    return {
        'flag': flag.numpy(), 'flat_ind': flat.numpy(),
        'score': torch.stack([score['l'], score['r']], 1).numpy(),
        'params_pred': pred.numpy(), 'cam': cam.reshape(B, 2, 3).numpy().copy(),
        'poses': poses.reshape(B, 2, 48).numpy(), 'betas': betas.reshape(B, 2, 10).numpy().copy(),
    }


def slots_to_rows(slots):
    """Re-pack per-frame slots into the reference's row order for ONE frame (B == 1):
    all left rows then all right rows; a side with no detection keeps its placeholder row
    with detection_flag False (acr/result_parser.py:102-120,166-168)."""
    assert slots['flag'].shape[0] == 1
    rows = {k: np.concatenate([slots[k][:, 0], slots[k][:, 1]], 0) for k in
            ('params_pred', 'cam', 'poses', 'betas')}
    rows['detection_flag'] = np.concatenate([slots['flag'][:, 0], slots['flag'][:, 1]], 0)
    rows['flat_ind'] = np.concatenate([slots['flat_ind'][:, 0], slots['flat_ind'][:, 1]], 0)
    return rows


def slots_to_rows_batch(slots):
    """Per-frame slots of a batch -> the reference's rows (acr/result_parser.py:102-120,166-183): all left rows of the
    frames with a left hand (ascending), then the right rows; a side without any hit in the WHOLE batch keeps one
    placeholder row (frame 0, pixel 0, flag False).  Also returns the frame of every row (reorganize_idx at
    batch_ids = arange)."""
    B = slots['flag'].shape[0]
    rows = {k: [] for k in ('params_pred', 'cam', 'poses', 'betas', 'detection_flag', 'flat_ind', 'frame', 'hand_type')}
    for h in (0, 1):
        ids = [b for b in range(B) if slots['flag'][b, h]]
        det = bool(ids)
        for b in (ids or [0]):
            for k in ('params_pred', 'cam', 'poses', 'betas'):
                rows[k].append(slots[k][b, h])
            rows['detection_flag'].append(det)
            rows['flat_ind'].append(slots['flat_ind'][b, h] if det else 0)
            rows['frame'].append(b)
            rows['hand_type'].append(h)
    return {k: np.asarray(v) for k, v in rows.items()}
