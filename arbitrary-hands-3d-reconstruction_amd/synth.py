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


def make_state_dict(seed=0, center_bias=(0.3, 0.3), as_torch=True, prefix=''):
    """Returns an OrderedDict key -> float32 array (torch tensors if as_torch).

    center_bias: (left, right) bias of the 64->1 center-tower exit conv; use a
    large negative value to suppress detections of that hand.
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    out = {}
    for key, shape in state_dict_schema().items():
        if key.endswith('num_batches_tracked'):
            v = np.array(1000, dtype=np.int64)
        elif key.endswith('running_mean'):
            v = rng.normal(0.0, 0.1, shape)
        elif key.endswith('running_var'):
            v = rng.uniform(0.8, 1.2, shape)
        elif _is_bn(key):
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
    if as_torch:
        import torch
        out = {k: (torch.tensor(int(v)) if v.ndim == 0 else torch.from_numpy(np.ascontiguousarray(v)))
               for k, v in out.items()}
    return out


def _is_bn(key):
    """True for BatchNorm weight/bias tensors (their module also owns a running_mean)."""
    sch = _schema_cache()
    base = key.rsplit('.', 1)[0]
    return (base + '.running_mean') in sch


_SCH = None


def _schema_cache():
    global _SCH
    if _SCH is None:
        _SCH = state_dict_schema()
    return _SCH


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
