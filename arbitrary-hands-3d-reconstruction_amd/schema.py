"""Checkpoint schema of the ACR network (HRNet-W32 + ACR heads; `width` = 48 gives the HRNet-W48 variant that
BASELINE.json configs[4] names - the reference itself hard-wires W32, acr/model.py:797-819, so W48 is build-defined:
branch widths 48/96/192/384, heads fed by 48 + 2 coordinate channels, everything else unchanged).

This enumerates, from the topology alone, every tensor of the reference's
``acr.model.ACR().state_dict()`` (names, shapes, role), so that the packer,
the synthetic-checkpoint generator and the loader all agree on the key names
a real ``wild.pkl`` carries (minus its ``module.`` prefix).

Reference topology: acr/model.py:785-829 (make_baseline), :703-736
(_make_transition_layer), :738-752 (_make_layer), :620-661 (_make_fuse_layers),
:374-463 (SegmNet), :185-313 (heads).  Nothing here is executed on the
reference; tests/test_schema.py pins the key list against a digest captured
from the imported reference (tests/golden/schema_digest.json).
"""
from collections import OrderedDict

def stage_cfg(width=32):
    """acr/model.py:797-819 (stage2/3/4_cfg) with NUM_CHANNELS scaled to `width`."""
    w = int(width)
    if w not in (32, 48):
        raise ValueError('HRNet width %r: 32 (the reference) and 48 (BASELINE.json configs[4]) are defined' % (width,))
    return {
        2: dict(modules=1, channels=[w, 2 * w]),
        3: dict(modules=4, channels=[w, 2 * w, 4 * w]),
        4: dict(modules=3, channels=[w, 2 * w, 4 * w, 8 * w]),
    }


STAGE_CFG = stage_cfg(32)
BLOCKS_PER_BRANCH = 4
HEAD_CHANNELS = 64
HEAD_BLOCKS = 2
PARAMS_NUM = 109          # acr/result_parser.py:12-14
TOWER_OUT = {1: 106, 2: 1, 3: 3, 4: 106}   # params, center, cam, prior (acr/model.py:185-202)


def _conv(d, name, cout, cin, k, bias):
    d[name + '.weight'] = (cout, cin, k, k)
    if bias:
        d[name + '.bias'] = (cout,)


def _bn(d, name, c):
    d[name + '.weight'] = (c,)
    d[name + '.bias'] = (c,)
    d[name + '.running_mean'] = (c,)
    d[name + '.running_var'] = (c,)
    d[name + '.num_batches_tracked'] = ()


def _basic_block(d, p, c):
    _conv(d, p + '.conv1', c, c, 3, False); _bn(d, p + '.bn1', c)
    _conv(d, p + '.conv2', c, c, 3, False); _bn(d, p + '.bn2', c)


def fuse_plan(nb, multi_scale):
    """[(i, j, kind, [(cin, cout, relu)])] of a HighResolutionModule's fuse layers."""
    return [(i, j) for i in range(nb if multi_scale else 1) for j in range(nb) if j != i]


RESNET50 = 'resnet50'     # the `width` of a ResNet-50 checkpoint (BASELINE.json configs[1]'s backbone; build-defined)
RESNET50_LAYERS = ((64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2))      # (planes, blocks, stride): torchvision resnet50
RESNET50_UP = (256, 128, 64)                                               # channels of the three x2 upsampling stages


def backbone_channels(width):
    """channels of the 128x128 backbone output the heads read (before the 2 coordinate channels)"""
    return RESNET50_UP[-1] if width == RESNET50 else int(width)


def width_of(sd):
    """Backbone of a checkpoint (bare keys): HRNet width (channel count of branch 0: 32 / 48) or 'resnet50'."""
    if 'backbone.layer4.0.conv1.weight' in sd:
        return RESNET50
    w = sd.get('backbone.transition1.0.0.weight')
    return 32 if w is None else int(w.shape[0])


def _resnet50_backbone(d):
    """ResNet-50 trunk in torchvision's layout and key names (conv1 7x7 stride 2, max-pool, Bottleneck x [3,4,6,3] with
    the stride on the 3x3 conv and a 1x1 strided projection shortcut in the first block of every layer) + three
    upsampling stages `deconv_layers.{k}` = bilinear x2 (align_corners) -> conv3x3 -> BN -> ReLU (256, 128, 64 channels)
    that bring the 16x16x2048 map back to the 128x128 resolution the ACR heads work at.  BUILD-DEFINED: the reference
    has no ResNet (`--backbone resnet50` is a dead flag, acr/config.py:95); BASELINE.json configs[1] names the
    backbone, this is the definition it is measured on."""
    b = 'backbone.'
    _conv(d, b + 'conv1', 64, 3, 7, False); _bn(d, b + 'bn1', 64)
    cin = 64
    for li, (planes, blocks, stride) in enumerate(RESNET50_LAYERS):
        for i in range(blocks):
            p = b + 'layer%d.%d' % (li + 1, i)
            _conv(d, p + '.conv1', planes, cin, 1, False); _bn(d, p + '.bn1', planes)
            _conv(d, p + '.conv2', planes, planes, 3, False); _bn(d, p + '.bn2', planes)
            _conv(d, p + '.conv3', 4 * planes, planes, 1, False); _bn(d, p + '.bn3', 4 * planes)
            if i == 0:
                _conv(d, p + '.downsample.0', 4 * planes, cin, 1, False); _bn(d, p + '.downsample.1', 4 * planes)
            cin = 4 * planes
    for k, c in enumerate(RESNET50_UP):
        _conv(d, b + 'deconv_layers.%d.0' % k, c, cin, 3, False); _bn(d, b + 'deconv_layers.%d.1' % k, c)
        cin = c


def state_dict_schema(width=32):
    """OrderedDict key -> shape, in the reference's registration order.  width: 32 (the reference) / 48 / 'resnet50'."""
    d = OrderedDict()
    b = 'backbone.'
    if width == RESNET50:
        _resnet50_backbone(d)
        _heads(d, backbone_channels(width))
        return d
    STAGE_CFG = stage_cfg(width)
    c0 = STAGE_CFG[2]['channels'][0]
    _conv(d, b + 'conv1', 64, 3, 3, False); _bn(d, b + 'bn1', 64)
    _conv(d, b + 'conv2', 64, 64, 3, False); _bn(d, b + 'bn2', 64)
    # layer1: 4 Bottlenecks, planes 64, expansion 4
    for i in range(4):
        p = b + 'layer1.%d' % i
        cin = 64 if i == 0 else 256
        _conv(d, p + '.conv1', 64, cin, 1, False); _bn(d, p + '.bn1', 64)
        _conv(d, p + '.conv2', 64, 64, 3, False); _bn(d, p + '.bn2', 64)
        _conv(d, p + '.conv3', 256, 64, 1, False); _bn(d, p + '.bn3', 256)
        if i == 0:
            _conv(d, p + '.downsample.0', 256, 64, 1, False); _bn(d, p + '.downsample.1', 256)
    pre = [256]
    for s in (2, 3, 4):
        ch = STAGE_CFG[s]['channels']
        # transition (registered before the stage)
        t = b + 'transition%d' % (s - 1)
        for i, c in enumerate(ch):
            if i < len(pre):
                if c != pre[i]:
                    _conv(d, t + '.%d.0' % i, c, pre[i], 3, False); _bn(d, t + '.%d.1' % i, c)
            else:
                for j in range(i + 1 - len(pre)):
                    cin = pre[-1]
                    cout = c if j == i - len(pre) else cin
                    _conv(d, t + '.%d.%d.0' % (i, j), cout, cin, 3, False)
                    _bn(d, t + '.%d.%d.1' % (i, j), cout)
        nb = len(ch)
        for m in range(STAGE_CFG[s]['modules']):
            p = b + 'stage%d.%d' % (s, m)
            multi = not (s == 4 and m == STAGE_CFG[s]['modules'] - 1)
            for br in range(nb):
                for k in range(BLOCKS_PER_BRANCH):
                    _basic_block(d, p + '.branches.%d.%d' % (br, k), ch[br])
            for i in range(nb if multi else 1):
                for j in range(nb):
                    f = p + '.fuse_layers.%d.%d' % (i, j)
                    if j > i:
                        _conv(d, f + '.0', ch[i], ch[j], 1, False); _bn(d, f + '.1', ch[i])
                    elif j < i:
                        for k in range(i - j):
                            cout = ch[i] if k == i - j - 1 else ch[j]
                            _conv(d, f + '.%d.0' % k, cout, ch[j], 3, False)
                            _bn(d, f + '.%d.1' % k, cout)
        pre = ch
    _heads(d, c0)
    return d


def _heads(d, c0):
    b = 'backbone.'
    # part-segmentation head (acr/model.py:374-463)
    u = b + 'hand_segm.segm_head.upsampler.up1.conv.double_conv'
    _conv(d, u + '.0', 16, c0, 3, True); _bn(d, u + '.1', 16)
    _conv(d, u + '.3', 64, 16, 3, True); _bn(d, u + '.4', 64)
    g = b + 'hand_segm.segm_head.segm_net.double_conv'
    _conv(d, g + '.0', 33, 64, 3, True); _bn(d, g + '.1', 33)
    _conv(d, g + '.3', 33, 33, 3, True)
    # heads (acr/model.py:168-183)
    for side in ('l', 'r'):
        for t in (1, 2, 3, 4):
            p = '%s_final_layers.%d' % (side, t)
            _conv(d, p + '.0.0', HEAD_CHANNELS, c0 + 2, 3, True); _bn(d, p + '.0.1', HEAD_CHANNELS)
            for k in range(HEAD_BLOCKS):
                _basic_block(d, p + '.1.%d.0' % k, HEAD_CHANNELS)
            _conv(d, p + '.2', TOWER_OUT[t], HEAD_CHANNELS, 1, True)
    _conv(d, 'contact_layers.1.0', 256, c0 + 2, 3, True); _bn(d, 'contact_layers.1.1', 256)
    d['contact_layers.2.weight'] = (1, 6, 256, 16, 1, 1)
    d['contact_layers.3.weight'] = (1, 6, 256, 16, 1, 1)
    _conv(d, 'contact_layers.4', PARAMS_NUM, 2 * PARAMS_NUM, 1, True)
    _conv(d, 'contact_layers.5', PARAMS_NUM, 2 * PARAMS_NUM, 1, True)
    _conv(d, 'cam_shape_layers.1.0', 64, 256, 1, True)
    for k in (2, 3):
        d['cam_shape_layers.%d.weight' % k] = (10, 1024)
        d['cam_shape_layers.%d.bias' % k] = (10,)
    # built by the reference but never called (acr/model.py:181,262-286); still in checkpoints
    _conv(d, 'segmentation_layers.1.0', 256, c0 + 2, 3, True); _bn(d, 'segmentation_layers.1.1', 256)
    _conv(d, 'segmentation_layers.2.0', 33, 256, 1, True)
    return d


def schema_digest(width=32):
    import hashlib
    h = hashlib.sha256()
    d = state_dict_schema(width)
    for k, s in d.items():
        h.update(('%s:%s;' % (k, ','.join(map(str, s)))).encode())
    n_params = 0
    for k, s in d.items():
        if k.endswith('running_mean') or k.endswith('running_var') or k.endswith('num_batches_tracked'):
            continue
        n = 1
        for v in s:
            n *= v
        n_params += n
    return {'n_keys': len(d), 'n_params': n_params, 'sha256': h.hexdigest()}
