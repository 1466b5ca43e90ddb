"""ORACLE (test infrastructure, not product): CPU fp32 restatement of the ACR network.

Functional torch-CPU code driven directly by a reference-format state dict
(BatchNorm kept un-folded, NCHW, same op order as the reference) so that it
agrees with the imported reference to float round-off.  Pinned by
tests/golden/*.npz captured from the real reference (tests/golden/make_golden.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import torch
import torch.nn.functional as F

EPS = 1e-5  # nn.BatchNorm2d default; never overridden in acr/model.py


def _bn(sd, x, p):
    return F.batch_norm(x, sd[p + '.running_mean'], sd[p + '.running_var'],
                        sd[p + '.weight'], sd[p + '.bias'], False, 0.0, EPS)


def _conv(sd, x, p, stride=1, pad=1):
    return F.conv2d(x, sd[p + '.weight'], sd.get(p + '.bias'), stride, pad)


def basic_block(sd, x, p):
    """acr/model.py:483-499"""
    out = F.relu(_bn(sd, _conv(sd, x, p + '.conv1'), p + '.bn1'))
    out = _bn(sd, _conv(sd, out, p + '.conv2'), p + '.bn2')
    return F.relu(out + x)


def bottleneck(sd, x, p, stride=1):
    """acr/model.py:519-539 (stride 1).  stride 2: the first block of a ResNet-50 layer, torchvision's layout - stride on
    the 3x3 conv and on the 1x1 projection shortcut (resnet50_backbone; not part of the reference)."""
    out = F.relu(_bn(sd, _conv(sd, x, p + '.conv1', 1, 0), p + '.bn1'))
    out = F.relu(_bn(sd, _conv(sd, out, p + '.conv2', stride, 1), p + '.bn2'))
    out = _bn(sd, _conv(sd, out, p + '.conv3', 1, 0), p + '.bn3')
    res = x
    if (p + '.downsample.0.weight') in sd:
        res = _bn(sd, _conv(sd, x, p + '.downsample.0', stride, 0), p + '.downsample.1')
    return F.relu(out + res)


def hr_module(sd, xs, p, multi_scale=True):
    """acr/model.py:668-686 (forward) with fuse layers of :620-661"""
    nb = len(xs)
    xs = list(xs)
    for i in range(nb):
        for k in range(4):
            xs[i] = basic_block(sd, xs[i], '%s.branches.%d.%d' % (p, i, k))
    outs = []
    for i in range(nb if multi_scale else 1):
        y = None
        for j in range(nb):
            f = '%s.fuse_layers.%d.%d' % (p, i, j)
            if j == i:
                t = xs[j]
            elif j > i:
                t = _bn(sd, _conv(sd, xs[j], f + '.0', 1, 0), f + '.1')
                t = F.interpolate(t, scale_factor=2 ** (j - i), mode='nearest')
            else:
                t = xs[j]
                for k in range(i - j):
                    t = _bn(sd, _conv(sd, t, '%s.%d.0' % (f, k), 2, 1), '%s.%d.1' % (f, k))
                    if k != i - j - 1:
                        t = F.relu(t)
            y = t if y is None else y + t
        outs.append(F.relu(y))
    return outs


RESNET50_LAYERS = ((3, 1), (4, 2), (6, 2), (3, 2))      # (blocks, stride of the first block): torchvision resnet50


def resnet50_backbone(sd, image_u8, taps=None):
    """NO REFERENCE COUNTERPART (the reference's `--backbone resnet50` is a dead flag, acr/config.py:95): the backbone
    BASELINE.json configs[1] names, as the build defines it (schema._resnet50_backbone) - torchvision's ResNet-50 forward
    (conv1 7x7 stride 2 pad 3 -> BN -> ReLU -> max-pool 3x3 stride 2 pad 1 -> Bottlenecks [3,4,6,3]) followed by three
    stages of bilinear x2 (align_corners=True) -> conv3x3 -> BN -> ReLU down to 64 channels at 128x128.  Own-oracle of
    the HIP path for that configuration; the normalisation and the heads are the reference's."""
    b = 'backbone.'
    x = ((image_u8.permute(0, 3, 1, 2).float() / 255.) * 2.0 - 1.0).contiguous()
    x = F.relu(_bn(sd, _conv(sd, x, b + 'conv1', 2, 3), b + 'bn1'))
    if taps is not None:
        taps['stem'] = x
    x = F.max_pool2d(x, 3, 2, 1)
    for li, (blocks, stride) in enumerate(RESNET50_LAYERS):
        for i in range(blocks):
            x = bottleneck(sd, x, b + 'layer%d.%d' % (li + 1, i), stride if i == 0 else 1)
        if taps is not None:
            taps['layer%d' % (li + 1)] = x
    for k in range(3):
        x = F.interpolate(x, scale_factor=(2, 2), mode='bilinear', align_corners=True)
        x = F.relu(_bn(sd, _conv(sd, x, b + 'deconv_layers.%d.0' % k), b + 'deconv_layers.%d.1' % k))
    return x


def backbone(sd, image_u8, taps=None):
    """acr/model.py:831-865.  image_u8: [B,512,512,3] uint8/float RGB (NHWC).  (A checkpoint with ResNet-50 keys goes
    to resnet50_backbone.)"""
    if 'backbone.layer4.0.conv1.weight' in sd:
        return resnet50_backbone(sd, image_u8, taps)
    b = 'backbone.'
    x = ((image_u8.permute(0, 3, 1, 2).float() / 255.) * 2.0 - 1.0).contiguous()
    x = F.relu(_bn(sd, _conv(sd, x, b + 'conv1', 2, 1), b + 'bn1'))
    x = F.relu(_bn(sd, _conv(sd, x, b + 'conv2', 2, 1), b + 'bn2'))
    if taps is not None:
        taps['stem'] = x
    for i in range(4):
        x = bottleneck(sd, x, b + 'layer1.%d' % i)
    if taps is not None:
        taps['layer1'] = x
    t = b + 'transition1'
    xs = [F.relu(_bn(sd, _conv(sd, x, t + '.0.0'), t + '.0.1')),
          F.relu(_bn(sd, _conv(sd, x, t + '.1.0.0', 2, 1), t + '.1.0.1'))]
    ys = hr_module(sd, xs, b + 'stage2.0')
    if taps is not None:
        taps['stage2'] = ys[0]
    t = b + 'transition2'
    xs = [ys[0], ys[1], F.relu(_bn(sd, _conv(sd, ys[-1], t + '.2.0.0', 2, 1), t + '.2.0.1'))]
    for m in range(4):
        xs = hr_module(sd, xs, b + 'stage3.%d' % m)
    ys = xs
    if taps is not None:
        taps['stage3'] = ys[0]
    t = b + 'transition3'
    xs = [ys[0], ys[1], ys[2], F.relu(_bn(sd, _conv(sd, ys[-1], t + '.3.0.0', 2, 1), t + '.3.0.1'))]
    for m in range(3):
        xs = hr_module(sd, xs, b + 'stage4.%d' % m, multi_scale=(m != 2))
    return xs[0]


def segm_head(sd, x):
    """acr/model.py:374-463: bilinear x2 (align_corners) -> DoubleConv(32,64,mid 16) -> conv-BN-ReLU-conv."""
    u = 'backbone.hand_segm.segm_head.upsampler.up1.conv.double_conv'
    g = 'backbone.hand_segm.segm_head.segm_net.double_conv'
    x = F.interpolate(x, scale_factor=(2, 2), mode='bilinear', align_corners=True)
    x = F.relu(_bn(sd, _conv(sd, x, u + '.0'), u + '.1'))
    x = F.relu(_bn(sd, _conv(sd, x, u + '.3'), u + '.4'))
    x = F.relu(_bn(sd, _conv(sd, x, g + '.0'), g + '.1'))
    return _conv(sd, x, g + '.3')


def coord_maps(size=128):
    """acr/model.py:340-369: channel 0 varies along W (x), channel 1 along H (y), 2*i/(size-1)-1."""
    r = torch.arange(size, dtype=torch.int32).float() / (size - 1) * 2 - 1
    xx = r.view(1, 1, 1, size).expand(1, 1, size, size)
    yy = r.view(1, 1, size, 1).expand(1, 1, size, size)
    return torch.cat([xx, yy], 1).contiguous()


def tower(sd, x, p):
    """acr/model.py:288-313: 3x3 s2 conv+BN+ReLU -> 2 BasicBlocks -> 1x1 conv."""
    x = F.relu(_bn(sd, _conv(sd, x, p + '.0.0', 2, 1), p + '.0.1'))
    for k in range(2):
        x = basic_block(sd, x, p + '.1.%d.0' % k)
    return _conv(sd, x, p + '.2', 1, 0)


def hadamard(features, heatmaps):
    """acr/model.py:103-113: softmax over pixels per part, weighted feature sum -> [B,C,J]."""
    B, J, H, W = heatmaps.shape
    w = F.softmax(heatmaps.reshape(B, J, -1), dim=-1)
    f = features.reshape(B, -1, H * W)
    return torch.matmul(w, f.transpose(2, 1)).transpose(2, 1)


def locally_connected(weight, x):
    """acr/model.py:559-569 with kernel 1: x [B,256,16,1], weight [1,6,256,16,1,1] -> [B,6,16,1]."""
    return (x.unsqueeze(1).unsqueeze(-1) * weight).sum([2, -1])


def pare_vector(sd, side, pooled_contact, pooled_shape):
    """acr/model.py:141-159 for one side: pooled_contact [B,256,32,1], pooled_shape [B,64,32] ->
    pare = [LocallyConnected2d offsets (96) | shape Linear (10)]  [B,106]."""
    sl, lc = (slice(16, 32), 2) if side == 'l' else (slice(0, 16), 3)
    B = pooled_contact.shape[0]
    off = locally_connected(sd['contact_layers.%d.weight' % lc], pooled_contact[:, :, sl, :])
    off = off.squeeze(-1).transpose(2, 1).reshape(B, 96)
    shp = F.linear(torch.flatten(pooled_shape[:, :, sl], start_dim=1),
                   sd['cam_shape_layers.%d.weight' % lc], sd['cam_shape_layers.%d.bias' % lc])
    return torch.cat((off, shp), 1)


def head_forward(sd, x, taps=None):
    """acr/model.py:47-166.  x: backbone output [B,32,128,128].  Returns the H11 dict."""
    B = x.shape[0]
    segm = segm_head(sd, x)
    x = torch.cat((x, coord_maps(128).repeat(B, 1, 1, 1)), 1)
    maps = {}
    for side in 'lr':
        params = tower(sd, x, '%s_final_layers.1' % side)
        center = tower(sd, x, '%s_final_layers.2' % side)
        cam = tower(sd, x, '%s_final_layers.3' % side)
        prior = tower(sd, x, '%s_final_layers.4' % side)
        cam = cam.clone()
        cam[:, 0] = torch.pow(1.1, cam[:, 0])
        maps[side] = (torch.cat([cam, params], 1), center, prior)
    # part branch (acr/model.py:116-166)
    part_att = F.interpolate(segm.clone().float(), scale_factor=(1 / 2, 1 / 2), mode='nearest')[:, 1:]
    contact = F.relu(_bn(sd, _conv(sd, x, 'contact_layers.1.0'), 'contact_layers.1.1'))
    shape_f = _conv(sd, contact, 'cam_shape_layers.1.0', 1, 0)
    wc = hadamard(contact, part_att).unsqueeze(-1)       # [B,256,32,1]
    ws = hadamard(shape_f, part_att)                      # [B,64,32]
    if taps is not None:
        taps['pooled_contact'] = wc[..., 0]
        taps['pooled_shape'] = ws
    out = {}
    for side, mix in (('l', 4), ('r', 5)):
        pare = pare_vector(sd, side, wc, ws)
        if taps is not None:
            taps[side + '_pare'] = pare
        pm, center, prior = maps[side]
        pare_f = pare.unsqueeze(-1).unsqueeze(-1).repeat(1, 1, 64, 64)
        pp = torch.cat((pm[:, :3].clone(), pare_f), 1)
        pm2 = _conv(sd, torch.cat((pm, pp), 1), 'contact_layers.%d' % mix, 1, 0)
        out[side + '_params_maps'] = pm2.float()
        out[side + '_center_map'] = center.float()
        out[side + '_prior_maps'] = prior.float()
    out['segms'] = segm.float()
    return out


@torch.no_grad()
def network(sd, image_u8, taps=None):
    return head_forward(sd, backbone(sd, image_u8, taps), taps)
