"""The flags the ACR hot path reads, with the reference's names and defaults.

Reference: acr/config.py:19-222 (argparse, ~110 mostly dead training flags), YAML `ARGS:` overlay
(:194-209, CLI wins), class-level singleton parsed at import (:232) behind args() (:269).
Here: a plain namespace with the ~25 live fields, the same YAML/CLI names, no import-time parsing
and no files written.  Unsupported combinations raise ValueError where the reference does
(acr/result_parser.py:157-164, acr/model.py:701).
"""
import argparse
import copy

DEFAULTS = dict(
    tab='process_images', demo_mode='image', inputs=None, output_dir=None, configs_yml='configs/demo.yml',
    model_path='checkpoints/wild.pkl', mano_root='mano/', backbone='hrnet', model_precision='fp32',
    input_size=512, centermap_size=64, head_block_num=2, inter_prior=True, attention_mode='pred-part',
    offset_mode='concat', merge_mano_camera_head=False, cam_dim=3, rot_dim=6, mano_theta_num=16, Rot_type='6D',
    prior_mode='cross', dataset='internet', centermap_conf_thresh=0.35, max_hand=2, kernel_sizes=[5], align_idx=9,
    mano_mesh_root_align=True, perspective_proj=False, focal_length=1265, temporal_optimization=False,
    smooth_coeff=4.0, save_dict_results=False, save_visualization_on_img=False, val_batch_size=1, GPUS=0,
    renderer='none', render_size=512,
    # not a reference flag: 'frame' = every frame as a batch of one (default); 'reference' = the reference's batch-wide
    # prior rules when a batch > 1 is parsed (acr/result_parser.py:42-47,131; result_parser.reference_prior_gate)
    batch_semantics='frame',
)

_ARGS = argparse.Namespace(**copy.deepcopy(DEFAULTS))


def args():
    """acr/config.py:269"""
    return _ARGS


def parse_args(input_args=None):
    """acr/config.py:19-222: CLI flags (same names) + YAML `ARGS:` overlay; CLI wins."""
    ap = argparse.ArgumentParser(description='ACR (MI355X) demo flags')
    for k, v in DEFAULTS.items():
        if isinstance(v, bool):
            ap.add_argument('--' + k, type=lambda s: str(s).lower() in ('1', 'true', 'yes'), default=None)
        elif isinstance(v, list):
            ap.add_argument('--' + k, type=int, nargs='+', default=None)
        else:
            ap.add_argument('--' + k, type=type(v) if v is not None else str, default=None)
    ap.add_argument('-t', dest='temporal_optimization_flag', action='store_true')
    ns = ap.parse_args(input_args)
    out = copy.deepcopy(DEFAULTS)
    yml = ns.configs_yml or DEFAULTS['configs_yml']
    try:
        import yaml
        with open(yml) as f:
            cfg = yaml.safe_load(f) or {}
        for k, v in (cfg.get('ARGS') or {}).items():
            out[k] = v
    except (IOError, OSError):
        pass
    for k in DEFAULTS:
        v = getattr(ns, k)
        if v is not None:
            out[k] = v
    if ns.temporal_optimization_flag:
        out['temporal_optimization'] = True
    return validate(argparse.Namespace(**out))


def validate(ns):
    if ns.backbone not in ('hrnet', 'hrnetv4', 'resnet50'):
        # the reference only ever builds HRNet-W32 (acr/model.py:27), whatever --backbone says (acr/config.py:95: "resnet50
        # or hrnet", default 'hrnetv4'); 'resnet50' selects the build-defined ResNet-50 of BASELINE.json configs[1]
        # (schema._resnet50_backbone) - there is no reference network or checkpoint behind it
        raise ValueError("backbone %r: 'hrnet' (HRNet-W32, the reference's only network) or 'resnet50' (build-defined)" % ns.backbone)
    if 'part' not in ns.attention_mode:
        raise ValueError('attention_mode must contain "part" (acr/model.py:698-701)')
    if ns.prior_mode != 'cross' or not ns.inter_prior or ns.dataset == 'FreiHand':
        raise ValueError('only inter_prior=True, prior_mode="cross", dataset != "FreiHand" is implemented '
                         '(acr/result_parser.py:125-164)')
    if ns.Rot_type != '6D' or ns.rot_dim != 6 or ns.cam_dim != 3 or ns.mano_theta_num != 16:
        raise ValueError('only the 6D / 109-parameter layout is implemented (acr/result_parser.py:12)')
    if ns.centermap_size != 64 or ns.input_size != 512 or ns.head_block_num != 2 or ns.offset_mode != 'concat':
        raise ValueError('only centermap_size=64, input_size=512, head_block_num=2, offset_mode=concat are implemented')
    if ns.merge_mano_camera_head or ns.perspective_proj:
        raise ValueError('merge_mano_camera_head / perspective_proj are not implemented (acr/model.py:85-88)')
    if list(ns.kernel_sizes) != [5]:
        # CenterMap's NMS window (acr/result_parser.py:199,207-216): the decode kernel implements the 5x5 pool only
        raise ValueError('kernel_sizes=%r: only the 5x5 center NMS (kernel_sizes=[5]) is implemented' % (ns.kernel_sizes,))
    # (max_hand is not validated: the reference reads it only under train_flag, acr/result_parser.py:221-224 -
    # inference always takes the top-1 center per map, whatever the reference's default of 4 says)
    if not (isinstance(ns.align_idx, int) and 0 <= ns.align_idx <= 20):
        raise ValueError('align_idx must be a joint index 0..20')
    if getattr(ns, 'batch_semantics', 'frame') not in ('frame', 'reference'):
        raise ValueError("batch_semantics %r: 'frame' or 'reference'" % (ns.batch_semantics,))
    if ns.model_precision not in ('fp32', 'fp16', 'bf16', 'fp16x3', 'bf16x3'):
        # acr/config.py:96: fp32 (configs/demo.yml) | fp16 (the argparse default: autocast, acr/model.py:33-37);
        # bf16 = the same 16-bit program on the other gfx950 MFMA type (packer.lower); fp16x3 / bf16x3 = fp32 tensors with
        # split 16-bit operands (fp32-class accuracy; fp16x3 needs |activation| <= 65504 and reports ACRMI_ERANGE otherwise)
        raise ValueError("model_precision %r: 'fp32', 'fp16', 'bf16', 'fp16x3' or 'bf16x3'" % ns.model_precision)
    return ns


class ConfigContext(object):
    """acr/config.py:225-267: swap the active args for a `with` block (nothing is written to disk)."""

    def __init__(self, parsed_args):
        self.parsed = validate(parsed_args)

    def __enter__(self):
        global _ARGS
        self.prev = _ARGS
        _ARGS = self.parsed
        return self.parsed

    def __exit__(self, *exc):
        global _ARGS
        _ARGS = self.prev
