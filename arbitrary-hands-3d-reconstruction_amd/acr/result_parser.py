"""ResultParser: the reference's parse() contract (acr/result_parser.py:21-40,85-190) over the HIP
decode kernel.  The kernel writes fixed per-(frame,hand) slots; this module re-packs them into the
reference's variable-length rows: all left rows (ascending frame), then all right rows.

Default semantics are per frame (= the reference run at batch 1 for every frame, the only way
acr/main.py:126-141 calls it).  ResultParser(batch_semantics='reference') / args().batch_semantics
= 'reference' reproduces what the reference's parse_maps does when it is handed a batch > 1: the
cross-hand prior only when EVERY flag of the batch is set (:131) and determine_coeff deciding for
the whole batch from row 0 of each side's list (:42-47) - see reference_prior_gate below.  (The
whole-batch placeholder rows, :102-120, are reproduced in both modes.)
"""
import numpy as np
import torch

from .. import _lib
from ..config import args

PART_IDX = [3, 6, 90, 10]     # cam, global_orient (6D), hand_pose (15x6D), betas  (acr/result_parser.py:12)


def rows_from_slots(slots, meta_data=None, map_size=64):
    """slots [B,2,176] (any device) -> dict of the row tensors parse_maps/parse produce.
    A side with no detection in the whole batch keeps ONE placeholder row (frame 0, pixel 0,
    detection_flag False), as acr/result_parser.py:102-120 does."""
    dev = slots.device
    B = slots.shape[0]
    flag = slots[:, :, _lib.SLOT_FLAG] > 0.5
    out = {}
    ids, flags = [], []
    for h in (0, 1):
        idx = torch.nonzero(flag[:, h]).flatten()
        if idx.numel() == 0:
            idx = torch.zeros(1, dtype=torch.long, device=dev)
            flags.append(torch.zeros(1, device=dev))
        else:
            flags.append(torch.ones(idx.numel(), device=dev))
        ids.append(idx)
    rows = torch.cat([slots[ids[0], 0], slots[ids[1], 1]], 0)            # [H,176]
    L, R = ids[0].numel(), ids[1].numel()
    batch_ids = torch.cat(ids)
    S = _lib
    out['l_params_pred'] = rows[:L, S.SLOT_PARAMS:S.SLOT_PARAMS + 109]
    out['r_params_pred'] = rows[L:, S.SLOT_PARAMS:S.SLOT_PARAMS + 109]
    out['params_pred'] = rows[:, S.SLOT_PARAMS:S.SLOT_PARAMS + 109].contiguous()
    out['detection_flag'] = torch.cat(flags)
    out['detection_flag_cache'] = out['detection_flag'].bool()
    flat = rows[:, S.SLOT_FLATIND].long()
    centers = torch.stack([flat % map_size, torch.div(flat, map_size, rounding_mode='floor')], 1)   # (x, y)
    out['l_centers_pred'], out['r_centers_pred'] = centers[:L], centers[L:]
    conf = rows[:, S.SLOT_SCORE:S.SLOT_SCORE + 1]
    out['l_centers_conf'], out['r_centers_conf'] = conf[:L], conf[L:]
    out['left_hand_num'] = torch.tensor([L], device=dev)
    out['right_hand_num'] = torch.tensor([R], device=dev)
    out['output_hand_type'] = torch.cat((torch.zeros(L), torch.ones(R))).to(dev).to(torch.int32)
    poses = rows[:, S.SLOT_POSES:S.SLOT_POSES + 48].contiguous()
    out['params_dict'] = {'cam': rows[:, S.SLOT_CAM:S.SLOT_CAM + 3].contiguous(),
                          'global_orient': poses[:, :3].contiguous(), 'hand_pose': poses[:, 3:].contiguous(),
                          'betas': rows[:, S.SLOT_BETAS:S.SLOT_BETAS + 10].contiguous(), 'poses': poses}
    if meta_data is not None:
        bid = meta_data['batch_ids'].to(dev) if 'batch_ids' in meta_data else torch.arange(B, device=dev)
        out['reorganize_idx'] = bid[batch_ids]
        for key in ('image', 'offsets', 'imgpath'):          # acr/result_parser.py:186-187
            if key in meta_data:
                v = meta_data[key]
                if isinstance(v, torch.Tensor):
                    meta_data[key] = v[batch_ids.to(v.device)]
                elif isinstance(v, list):
                    meta_data[key] = np.array(v)[batch_ids.cpu().numpy()]
    out['_batch_ids'] = batch_ids
    return out


def reference_prior_gate(slots, map_size=64):
    """The reference's batch-wide prior decision (acr/result_parser.py:85-145) from a first decode's flags / centers.
    slots [B,2,176] -> int32 [B] on slots' device: 1 = this frame's two rows take their cross-hand prior, 0 = not.
      * l_ids / r_ids = frames whose left / right center passed the threshold (ascending);
      * no prior at all unless both lists are non-empty - a side with no hit in the WHOLE batch carries a placeholder
        row whose flag is False, and :131 wants sum(detection_flag) == len(detection_flag);
      * determine_coeff (:42-47) compares l_cyxs[0] with r_cyxs[0]: the left center of the FIRST left-detected frame
        and the right center of the FIRST right-detected frame (not necessarily the same frame); more than 32 map
        pixels apart -> both priors become 0 for every frame of the batch;
      * otherwise the prior is added exactly in the frames that have both hands (all_hand_valid_batch_ids, :128)."""
    flag = (slots[:, :, _lib.SLOT_FLAG] > 0.5).cpu()
    flat = slots[:, :, _lib.SLOT_FLATIND].long().cpu()
    B = flag.shape[0]
    gate = torch.zeros(B, dtype=torch.int32)
    l_ids, r_ids = torch.nonzero(flag[:, 0]).flatten(), torch.nonzero(flag[:, 1]).flatten()
    if l_ids.numel() and r_ids.numel():
        fl, fr = int(flat[l_ids[0], 0]), int(flat[r_ids[0], 1])
        dy, dx = float(fl // map_size - fr // map_size), float(fl % map_size - fr % map_size)
        if not (dy * dy + dx * dx) ** 0.5 > 32:
            gate[flag[:, 0] & flag[:, 1]] = 1
    return gate.to(slots.device)


class ResultParser(object):
    def __init__(self, batch_semantics=None):
        """batch_semantics: 'frame' (default; every frame as the reference treats a batch of one) or 'reference' (the
        reference's batch-wide prior rules at batch > 1); None = args().batch_semantics."""
        a = args()
        self.batch_semantics = batch_semantics or getattr(a, 'batch_semantics', 'frame')
        if self.batch_semantics not in ('frame', 'reference'):
            raise ValueError("batch_semantics %r: 'frame' or 'reference'" % (self.batch_semantics,))
        self.map_size = a.centermap_size
        self.conf_thresh = a.centermap_conf_thresh          # CenterMap.conf_thresh (acr/result_parser.py:198-205)
        self.part_name = ['cam', 'global_orient', 'hand_pose', 'betas']
        self.part_idx = [a.cam_dim, a.rot_dim, (a.mano_theta_num - 1) * a.rot_dim, 10]
        self.kps_num = 21
        self.params_num = int(np.array(self.part_idx).sum())
        if self.part_idx != PART_IDX:
            raise ValueError('only the 3/6/90/10 parameter split is implemented')

    @torch.no_grad()
    def parse(self, outputs, meta_data, cfg):
        """outputs: dict with l/r_center_map [B,1,64,64], l/r_params_maps [B,109,64,64],
        l/r_prior_maps [B,106,64,64] (NCHW device tensors, as acr.model.ACR.head_forward returns) -
        or a ready 'slots' tensor from the fused path.  Mutates and returns (outputs, meta_data)."""
        if 'slots' in outputs:
            slots = outputs['slots']
        else:
            from .. import ops
            m = {k: ops.to_nhwc(outputs[k], device=outputs[k].device) for k in
                 ('l_center_map', 'r_center_map', 'l_params_maps', 'r_params_maps', 'l_prior_maps', 'r_prior_maps')}
            dec = lambda gate: ops.decode_maps(m['l_center_map'], m['r_center_map'], m['l_params_maps'], m['r_params_maps'],
                                               m['l_prior_maps'], m['r_prior_maps'], conf_thresh=self.conf_thresh,
                                               prior_gate=gate)
            slots = dec(None)
            if self.batch_semantics == 'reference' and slots.shape[0] > 1:
                slots = dec(ops.prior_gate(slots))       # acrmi_prior_gate: the batch-wide rules on the device
            outputs['slots'] = slots
        outputs.update(rows_from_slots(slots, meta_data, self.map_size))
        return outputs, meta_data

    __call__ = parse
