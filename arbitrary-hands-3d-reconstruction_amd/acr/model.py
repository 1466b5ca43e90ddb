"""acr.model.ACR with the reference's surface (acr/model.py:23-44): no-arg constructor,
state_dict()/load_state_dict() with the reference's key names, eval()/cuda(), and
forward(meta_data, **cfg) -> the reference's output dict.  All arithmetic runs in libacrmi.so
through Engine; this class holds the checkpoint on the host and the resident engine on one GPU.
"""
from collections import OrderedDict

import torch

from ..config import args, validate
from ..engine import Engine
from ..schema import RESNET50, backbone_channels, state_dict_schema, width_of
from .result_parser import ResultParser, reference_prior_gate, rows_from_slots


class ACR(object):
    def __init__(self, device=0, max_batch=1, **kwargs):
        self._args = validate(args())
        self._retired = None
        self._result_parser = ResultParser(batch_semantics=kwargs.get('batch_semantics'))
        self.params_num = self._result_parser.params_num
        self._init_sd(kwargs.get('width', RESNET50 if getattr(self._args, 'backbone', 'hrnet') == RESNET50 else 32))
        self._device = device
        self._max_batch = max_batch
        self._engine = None
        self._loaded = False
        self.training = False

    def _init_sd(self, width):
        """Zero-initialised tensors with the reference's key names (HRNet width 32; 48 = the BASELINE configs[4] variant;
        'resnet50' = the build-defined backbone of configs[1], schema._resnet50_backbone)."""
        self._width = width
        self._sd = OrderedDict()
        for k, shp in state_dict_schema(width).items():       # zero-initialised until a checkpoint is loaded
            self._sd[k] = torch.zeros(shp, dtype=torch.int64 if k.endswith('num_batches_tracked') else torch.float32)
        for k in self._sd:
            if k.endswith('running_var') or (k.endswith('.weight') and (k[:-7] + '.running_var') in self._sd):
                self._sd[k].fill_(1.0)

    # ---- nn.Module-like surface ----------------------------------------------------------------
    def state_dict(self):
        return OrderedDict(self._sd)

    def load_state_dict(self, sd, strict=True):
        from ..packer import strip_prefix, check_state_dict
        sd = strip_prefix(sd)
        if width_of(sd) != self._width and width_of(sd) in (32, 48, RESNET50):
            self._init_sd(width_of(sd))                  # an HRNet-W48 checkpoint re-shapes the module
        if strict:
            check_state_dict(sd)
        missing = [k for k in self._sd if k not in sd]
        unexpected = [k for k in sd if k not in self._sd]
        for k, v in sd.items():
            if k in self._sd:
                t = v.detach().cpu() if hasattr(v, 'detach') else torch.as_tensor(v)
                if tuple(t.shape) != tuple(self._sd[k].shape) and t.numel() == self._sd[k].numel():
                    t = t.reshape(self._sd[k].shape)
                if tuple(t.shape) != tuple(self._sd[k].shape):
                    raise ValueError('size mismatch for %s: %s vs %s' % (k, tuple(t.shape), tuple(self._sd[k].shape)))
                self._sd[k] = t.to(self._sd[k].dtype).clone()
        self._loaded = True
        if self._engine is not None:   # re-pack on next use; the new context adopts the MANO tables / options
            self._retired, self._engine = self._engine, None
        return missing, unexpected

    def eval(self):
        self.training = False
        return self

    def train(self, mode=True):
        if mode:
            raise ValueError('inference only: the reference ships no training code either')
        return self

    def cuda(self, device=None):
        if device is not None:
            self._device = device if isinstance(device, int) else (torch.device(device).index or 0)
        self.engine()
        return self

    def to(self, device):
        return self.cuda(device)

    def engine(self, min_batch=1):
        """The resident context.  A checkpoint reload builds a new one that adopts the old one's MANO tables and
        options, so wrappers holding the model (MANOWrapper) keep working."""
        if self._engine is None:
            eng = Engine(self._device)
            eng.load_state_dict(self._sd, max_batch=max(self._max_batch, min_batch), precision=self._args.model_precision)
            a = self._args
            eng.set_conf_thresh(a.centermap_conf_thresh)          # CenterMap.conf_thresh (acr/result_parser.py:198-205)
            eng.set_center_idx(a.align_idx if a.mano_mesh_root_align else None)
            if self._retired is not None:
                eng.adopt(self._retired)
                self._retired.close()
                self._retired = None
            else:
                # the demo loop's operating point is one frame per call (acr/main.py:126-141): lanes by measurement in
                # the host process's actual stream / hardware-queue state (Engine.tune_lanes)
                eng.tune_lanes(1)
            self._engine = eng
        else:
            self._engine.ensure_batch(min_batch)
        return self._engine

    # ---- the reference's forward ----------------------------------------------------------------
    @torch.no_grad()
    def backbone(self, image):
        """acr/model.py:831-865: uint8 [B,512,512,3] -> [B,32,128,128] (NCHW copy of the resident buffer)."""
        eng = self.engine(image.shape[0])
        B = eng.backbone_heads(image)
        return eng.buffer(eng.program['heads'].backbone_buf, B, backbone_channels(self._width)).permute(0, 3, 1, 2).float().contiguous()

    @torch.no_grad()
    def head_forward(self, x):
        """acr/model.py:47-65.  x: backbone features, float [B,32,128,128] as the reference takes them (e.g. what
        `self.backbone(image)` returned: `model.head_forward(model.backbone(img))` works as on the reference) - the heads run
        on them (acrmi_heads); or uint8 frames [B,512,512,3], the whole resident program in one go."""
        eng = self.engine(x.shape[0])
        if x.dtype == torch.uint8:
            B = eng.backbone_heads(x)
        else:
            B = eng.heads(x)
        return eng.head_maps(B)

    @torch.no_grad()
    def forward(self, meta_data, **cfg):
        """meta_data: {'image': uint8 [B,512,512,3] RGB, 'offsets': [B,10], 'batch_ids': [B], ...};
        cfg (mode/calc_loss) is accepted and ignored like the reference does in eval (acr/model.py:32-44).
        cfg['return_maps']=False skips the NCHW copies of the head maps."""
        img = meta_data['image']
        if img.dtype != torch.uint8:
            img = img.to(torch.uint8)
        eng = self.engine(img.shape[0])
        B = eng.backbone_heads(img.contiguous())
        outputs = eng.head_maps(B) if cfg.get('return_maps', True) else {}
        outputs['slots'] = eng.decode(B)
        if self._result_parser.batch_semantics == 'reference' and B > 1:
            # the reference's batch-wide prior rules (acr/result_parser.py:42-47,131): decided ON THE DEVICE from the first
            # decode's flags / centers (acrmi_prior_gate; result_parser.reference_prior_gate is the host statement of the same
            # rules, kept for the tests), applied by a second decode - no host round trip between the two
            outputs['slots'] = eng.decode(B, prior_gate=eng.prior_gate(outputs['slots']))
        # 'fp16x3' only: an activation outside the f16 range is an error, not a NaN / empty result.  Checked once, behind
        # everything this call queued (it synchronizes the stream: in front of the second decode it serialized the path)
        eng.check_range()
        if 'batch_ids' not in meta_data:
            meta_data['batch_ids'] = torch.arange(B)
        outputs.update(rows_from_slots(outputs['slots'], meta_data, self._result_parser.map_size))
        outputs['meta_data'] = meta_data
        return outputs

    __call__ = forward
