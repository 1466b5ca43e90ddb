"""Seeded synthetic assets: an ACR checkpoint and MANO tables.

The pretrained ``wild.pkl`` and the licence-gated ``MANO_{LEFT,RIGHT}.pkl`` are
not redistributable (reference README.md:36-37), so parity tests and the bench
run on assets regenerated from a seed on every box (numpy PCG64: identical
bytes everywhere).  Key names/shapes are the reference's (schema.py).

Init law (chosen so activations stay O(1) through ~60 residual layers and both
center heads fire):  conv ~ N(0, 2/fan_in); BN gamma ~ U(.8,1.2) except the last
BN of every residual block ~ U(.2,.4); BN beta, running_mean ~ N(0,.1);
running_var ~ U(.8,1.2); biases ~ N(0,.05); center-tower exit bias = center_bias.
"""
import numpy as np

from .schema import state_dict_schema

_LAST_BN_SUFFIX = ('.bn2.weight', '.bn3.weight')


def make_state_dict(seed=0, center_bias=(0.3, 0.3), as_torch=True, prefix='', width=32, law='benign'):
    """Returns an OrderedDict key -> float32 array (torch tensors if as_torch).

    center_bias: (left, right) bias of the 64->1 center-tower exit conv; use a
    large negative value to suppress detections of that hand.
    width: HRNet width (32 = the reference's network; 48 = BASELINE.json configs[4], schema.stage_cfg) or 'resnet50'
    (BASELINE.json configs[1]'s backbone as schema._resnet50_backbone defines it).
    law: 'benign' (above) or 'hostile' (make_hostile_state_dict: trained-like BatchNorm statistics).
    """
    if law == 'hostile':
        return make_hostile_state_dict(seed, center_bias, as_torch, prefix, width)
    rng = np.random.Generator(np.random.PCG64(seed))
    out = {}
    sch = state_dict_schema(width)
    for key, shape in sch.items():
        if key.endswith('num_batches_tracked'):
            v = np.array(1000, dtype=np.int64)
        elif key.endswith('running_mean'):
            v = rng.normal(0.0, 0.1, shape)
        elif key.endswith('running_var'):
            v = rng.uniform(0.8, 1.2, shape)
        elif _is_bn(key, sch):
            if key.endswith('.weight'):
                if key.endswith(_LAST_BN_SUFFIX):
                    lo, hi = 0.15, 0.3
                elif '.fuse_layers.' in key:
                    lo, hi = 0.25, 0.45
                else:
                    lo, hi = 0.8, 1.2
                v = rng.uniform(lo, hi, shape)
            else:
                v = rng.normal(0.0, 0.1, shape)
        elif key.endswith('.bias'):
            v = rng.normal(0.0, 0.05, shape)
        elif len(shape) == 6:      # LocallyConnected2d [1,6,256,16,1,1]
            v = rng.normal(0.0, 0.2 / np.sqrt(shape[2]), shape)
        elif len(shape) == 2:      # Linear
            v = rng.normal(0.0, 0.5 / np.sqrt(shape[1]), shape)
        else:                      # conv weight [cout, cin, k, k]
            fan_in = shape[1] * shape[2] * shape[3]
            v = rng.normal(0.0, np.sqrt(1.5 / fan_in), shape)
        if v.dtype != np.int64:
            v = v.astype(np.float32)
        out[prefix + key] = v
    for side, cb in zip('lr', center_bias):
        out[prefix + '%s_final_layers.2.2.bias' % side][...] = cb
        # keep cam scale/params in a sane range: small exit weights for cam and params towers
        for t in (1, 3, 4):
            out[prefix + '%s_final_layers.%d.2.weight' % (side, t)] *= 0.3
    return _as_torch(out) if as_torch else out


def _as_torch(out):
    import torch
    return {k: (torch.tensor(int(v)) if v.ndim == 0 else torch.from_numpy(np.ascontiguousarray(v)))
            for k, v in out.items()}


BN_EPS = 1e-5


def make_hostile_state_dict(seed=0, center_bias=(0.3, 0.3), as_torch=True, prefix='', width=32, stream_scale=50.0,
                            row_decades=(-1.5, 1.0), chan_decades=(-1.0, 1.0)):
    """The benign checkpoint of the same seed re-parametrised into trained-like statistics WITHOUT changing the
    function it computes (in exact arithmetic), so that the benign checkpoint's reference outputs stay the ground truth
    and every difference is fp32 / Winograd round-off of the implementation under test:

      * raw conv rows: every conv that feeds a BatchNorm has output channel c scaled by s_c = 10**U(row_decades), its
        BatchNorm's running_mean by s_c and running_var' = (var + eps) s_c**2 - eps: running_var spans 1e-3 .. 1e2,
        the raw weights span 2.5 decades per layer, the folded weights are unchanged up to rounding;
      * block-internal activations: the BatchNorm inside a BasicBlock / Bottleneck / plain conv chain gets
        (gamma, beta)_c *= f_c = 10**U(chan_decades) and the only consumer's input-channel weights /= f_c: the inputs of
        the Winograd convs (conv2 of every block) have per-channel magnitudes over two decades and gamma outliers;
      * residual stream of the backbone: every BatchNorm that writes the stream gets (gamma, beta) *= stream_scale and
        every conv that reads it has its weights /= stream_scale: activations of O(stream_scale) (up to ~1e2) wherever
        the fp32 path adds residuals, fuses branches or feeds a head.
    The re-parametrisation is done in float64 and rounded once to float32."""
    if not isinstance(width, int):
        raise ValueError('the hostile re-parametrisation is written for the HRNet topologies')
    sd = make_state_dict(seed, center_bias, as_torch=False, prefix='', width=width)
    sch = state_dict_schema(width)
    c0 = width
    rng = np.random.Generator(np.random.PCG64(7000 + seed))
    d = {k: (v.astype(np.float64) if v.dtype != np.int64 else v) for k, v in sd.items()}

    def bn_of(conv):
        """BatchNorm that follows conv `name` (by the reference's naming), or None."""
        base, leaf = conv.rsplit('.', 1)
        cands = []
        if leaf.startswith('conv'):
            cands.append(base + '.bn' + leaf[4:])
        if leaf.isdigit():
            cands.append(base + '.' + str(int(leaf) + 1))
        for c in cands:
            if (c + '.running_mean') in sch:
                return c
        return None

    convs = [k[:-7] for k, shp in sch.items() if k.endswith('.weight') and len(shp) == 4]
    pairs = [(c, bn_of(c)) for c in convs if bn_of(c) is not None and not c.startswith('segmentation_layers')]
    # ---- (1) raw row scales ------------------------------------------------------------------------
    for conv, bn in pairs:
        n = d[conv + '.weight'].shape[0]
        s = 10.0 ** rng.uniform(row_decades[0], row_decades[1], n)
        d[conv + '.weight'] *= s[:, None, None, None]
        if (conv + '.bias') in d:
            d[conv + '.bias'] *= s
        d[bn + '.running_mean'] *= s
        d[bn + '.running_var'] = (d[bn + '.running_var'] + BN_EPS) * s * s - BN_EPS
    # ---- (2) block-internal channel scales: (bn, the one conv that consumes its output) ------------
    internal = [('backbone.bn1', 'backbone.conv2')]
    for k in sch:
        if k.endswith('.conv2.weight'):
            p = k[:-len('.conv2.weight')]
            internal.append((p + '.bn1', p + '.conv2'))
            if (p + '.conv3.weight') in sch:
                internal.append((p + '.bn2', p + '.conv3'))
    u = 'backbone.hand_segm.segm_head.upsampler.up1.conv.double_conv'
    g = 'backbone.hand_segm.segm_head.segm_net.double_conv'
    internal += [(u + '.1', u + '.3'), (u + '.4', g + '.0'), (g + '.1', g + '.3')]
    for bn, consumer in internal:
        n = d[bn + '.weight'].shape[0]
        f = 10.0 ** rng.uniform(chan_decades[0], chan_decades[1], n)
        d[bn + '.weight'] *= f
        d[bn + '.bias'] *= f
        d[consumer + '.weight'] /= f[None, :, None, None]
    # ---- (3) backbone residual stream -------------------------------------------------------------
    S = float(stream_scale)
    writers, readers = ['backbone.bn2'], []
    for k in sch:
        if not k.startswith('backbone.') or not k.endswith('.weight') or len(sch[k]) != 4:
            continue
        c = k[:-7]
        if '.layer1.' in c or '.branches.' in c:
            if c.endswith('.conv1') or c.endswith('.downsample.0'):
                readers.append(c)
            last = '.conv3' if '.layer1.' in c else '.conv2'
            if c.endswith(last) or c.endswith('.downsample.0'):
                writers.append(bn_of(c))
        elif '.transition' in c or '.fuse_layers.' in c:
            readers.append(c)
            writers.append(bn_of(c))
    readers.append(u + '.0')
    for bn in writers:
        d[bn + '.weight'] *= S
        d[bn + '.bias'] *= S
    for c in readers:
        d[c + '.weight'] /= S
    for c in ['%s_final_layers.%d.0.0' % (s_, t) for s_ in 'lr' for t in (1, 2, 3, 4)] + ['contact_layers.1.0']:
        d[c + '.weight'][:, :c0] /= S        # the two coordinate channels are not part of the stream
    out = {prefix + k: (v if v.dtype == np.int64 else v.astype(np.float32)) for k, v in d.items()}
    return _as_torch(out) if as_torch else out


def plant_center_peaks(sd, left=None, right=None, radius_px=6.0, amplitude=3.0, width=32):
    """Edits a checkpoint (dict of numpy float32 arrays, bare keys) IN PLACE so that the center map of a side peaks at a
    chosen INTERIOR pixel (y, x) of the 64x64 map, whatever the frame shows - a position-dependent center bias built from
    the two coordinate channels the head towers see (acr/model.py:52,340-369), through the tower's own layers:
      entry conv (3x3 s2), channels 60-63:  a_x = relu((X - X0)/r), b_x = relu((X0 - X)/r), a_y, b_y   (channel 59: 0)
      block 0 conv1, channels 60 / 61:      tent_x = relu(1 - a_x - b_x), tent_y = relu(1 - a_y - b_y)
      block 0 conv2 + residual, channel 59: bump = relu(tent_x + tent_y)            (2 at the peak, 1 on its two ridges)
      block 1: channel 59 passes through;   exit conv: center += amplitude * bump.
    The synthetic network's own center maps are bias-dominated and peak on the map border (zero padding); with this
    the reference, the oracle and the kernels are exercised on centers whose 5x5 NMS window, 3x3 taps and 9x9 point-heads
    window lie fully inside the map.  None leaves a side untouched."""
    c0 = width
    for side, peak in (('l', left), ('r', right)):
        if peak is None:
            continue
        y0, x0 = peak
        pre = '%s_final_layers.2' % side
        sp = slice(59, 64)
        r = radius_px * 4.0 / 127.0                       # coordinate units per output pixel: 2 * 2 / 127

        def ident_bn(name, ch, beta):
            sd[name + '.weight'][ch] = 1.0
            sd[name + '.bias'][ch] = beta
            sd[name + '.running_mean'][ch] = 0.0
            sd[name + '.running_var'][ch] = 1.0
        # entry conv: coordinates at the centre tap (input pixel 2*o of the 128-map)
        X0 = np.float32(np.float32(2 * x0) / np.float32(127) * 2 - 1)
        Y0 = np.float32(np.float32(2 * y0) / np.float32(127) * 2 - 1)
        w, b = sd[pre + '.0.0.weight'], sd[pre + '.0.0.bias']
        w[sp] = 0.0
        b[sp] = 0.0
        for ch, (cin, sign, ref) in zip((60, 61, 62, 63), ((c0, 1, X0), (c0, -1, X0), (c0 + 1, 1, Y0), (c0 + 1, -1, Y0))):
            w[ch, cin, 1, 1] = sign / r
            ident_bn(pre + '.0.1', ch, -sign * ref / r)
        ident_bn(pre + '.0.1', 59, 0.0)
        for blk in (0, 1):
            for conv, bn in (('conv1', 'bn1'), ('conv2', 'bn2')):
                wk = sd['%s.1.%d.0.%s.weight' % (pre, blk, conv)]
                wk[sp] = 0.0                                # special output channels: built below
                wk[:, sp] = 0.0                             # ... and invisible to the ordinary channels
                for ch in range(59, 64):
                    ident_bn('%s.1.%d.0.%s' % (pre, blk, bn), ch, 0.0)
        w1 = sd[pre + '.1.0.0.conv1.weight']
        w1[60, 60, 1, 1] = w1[60, 61, 1, 1] = -1.0
        w1[61, 62, 1, 1] = w1[61, 63, 1, 1] = -1.0
        sd[pre + '.1.0.0.bn1.bias'][60] = 1.0
        sd[pre + '.1.0.0.bn1.bias'][61] = 1.0
        w2 = sd[pre + '.1.0.0.conv2.weight']
        w2[59, 60, 1, 1] = w2[59, 61, 1, 1] = 1.0
        we = sd[pre + '.2.weight']
        we[0, sp] = 0.0
        we[0, 59] = amplitude
    return sd


def _is_bn(key, sch):
    """True for BatchNorm weight/bias tensors (their module also owns a running_mean)."""
    base = key.rsplit('.', 1)[0]
    return (base + '.running_mean') in sch


KINTREE_PARENTS = [-1, 0, 1, 2, 0, 4, 5, 0, 7, 8, 0, 10, 11, 0, 13, 14]


def make_mano_tables(seed=1):
    """Synthetic MANO-shaped tables for both sides (mano/manolayer.py:61-102 buffer shapes).

    Returns {'left': {...}, 'right': {...}} with float32 arrays:
      v_template [778,3], shapedirs [778,3,10], posedirs [778,3,135],
      J_regressor [16,778], weights [778,16] (rows sum to 1, sparse-ish),
      hands_mean [45], hands_components [45,45], faces [1538,3] int64,
      kintree_table [2,16] int64.
    The left-hand x-flip of shapedirs (acr/mano_wrapper.py:35) is NOT applied
    here; MANOWrapper applies it, exactly as the reference does after loading.
    """
    tables = {}
    for si, side in enumerate(('left', 'right')):
        rng = np.random.Generator(np.random.PCG64(seed * 2 + si))
        # a hand-sized blob in metres: 16 joint anchors, vertices clustered around them
        anchors = np.zeros((16, 3))
        for f in range(5):
            base = np.array([0.09 * np.cos(0.5 * (f - 2)), 0.09 * np.sin(0.5 * (f - 2)), 0.0])
            for k in range(3):
                anchors[1 + 3 * f + k] = base * (1.0 + 0.35 * k)
        own = rng.integers(0, 16, 778)
        v_template = anchors[own] + rng.normal(0, 0.008, (778, 3))
        if side == 'left':
            v_template[:, 0] *= -1
        shapedirs = rng.normal(0, 0.004, (778, 3, 10))
        posedirs = rng.normal(0, 0.002, (778, 3, 135))
        jr = np.zeros((16, 778))
        for j in range(16):
            idx = rng.choice(778, 24, replace=False)
            w = rng.uniform(0.1, 1.0, 24)
            jr[j, idx] = w / w.sum()
        weights = np.zeros((778, 16))
        for v in range(778):
            js = [own[v], KINTREE_PARENTS[own[v]] if KINTREE_PARENTS[own[v]] >= 0 else (own[v] + 1) % 16]
            w = rng.uniform(0.2, 1.0, 2)
            weights[v, js[0]] += w[0]
            weights[v, js[1]] += w[1]
            weights[v] /= weights[v].sum()
        hands_mean = rng.normal(0, 0.15, 45)
        comps = rng.normal(0, 0.3, (45, 45))
        faces = rng.integers(0, 778, (1538, 3)).astype(np.int64)
        kt = np.stack([np.array([4294967295] + KINTREE_PARENTS[1:], dtype=np.int64), np.arange(16, dtype=np.int64)])
        tables[side] = dict(
            v_template=v_template.astype(np.float32), shapedirs=shapedirs.astype(np.float32),
            posedirs=posedirs.astype(np.float32), J_regressor=jr.astype(np.float32),
            weights=weights.astype(np.float32), hands_mean=hands_mean.astype(np.float32),
            hands_components=comps.astype(np.float32), faces=faces, kintree_table=kt)
    return tables


def make_frames(batch, seed=0, size=512, structured=True):
    """uint8 [B,size,size,3] RGB frames.  structured: smooth blobs + noise (parity runs);
    otherwise i.i.d. randint (throughput runs; conv time is data independent)."""
    rng = np.random.Generator(np.random.PCG64(1000 + seed))
    if not structured:
        return rng.integers(0, 256, (batch, size, size, 3), dtype=np.uint8)
    yy, xx = np.mgrid[0:size, 0:size].astype(np.float32)
    frames = np.empty((batch, size, size, 3), np.uint8)
    for b in range(batch):
        img = np.full((size, size, 3), 110.0, np.float32)
        for _ in range(4):
            cx, cy = rng.uniform(0.15, 0.85, 2) * size
            s = rng.uniform(0.05, 0.18) * size
            col = rng.uniform(-110, 140, 3).astype(np.float32)
            g = np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * s * s))
            img += g[..., None] * col
        img += rng.normal(0, 12.0, img.shape).astype(np.float32)
        frames[b] = np.clip(img, 0, 255).astype(np.uint8)
    return frames
