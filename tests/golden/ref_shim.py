"""Import the reference (/root/reference) on CPU with harness-side shims only.

AUTHORING-CONTAINER ONLY.  Used by tests/golden/make_golden.py to capture golden
vectors from the real reference and by nothing else; /root/reference does not
exist on the GPU box.  The reference tree is never written to
(PYTHONDONTWRITEBYTECODE, no ConfigContext/main()).  Shims:
  * sys.modules stubs for cv2 / h5py / imgaug / chumpy (absent here, no network); cv2.resize(INTER_CUBIC) and
    imgaug's pad helpers are served by oracle/preprocess.py (their published algorithms restated)
  * sys.argv=['x'] before import (acr/config.py:232 parses argv at import)
  * Tensor.cuda / Module.cuda = identity (hard-coded .cuda() calls)
  * np.float / np.int aliases (acr/utils.py:493)
  * mano.manolayer.ready_arguments -> synthetic MANO tables (licence-gated pkl absent)
"""
import os
import sys
import types

REF = '/root/reference'


def import_reference(mano_tables=None):
    """Returns (acr.model, acr.result_parser, acr.mano_wrapper, mano.manolayer, acr.utils)."""
    import numpy as np
    import torch
    os.environ['PYTHONDONTWRITEBYTECODE'] = '1'
    sys.dont_write_bytecode = True
    if not os.path.isdir(REF):
        raise RuntimeError('reference tree not present (authoring container only)')

    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    # cv2 / imgaug are absent: the two calls the pre-processing makes (acr/utils.py:1294-1321) are served by the
    # oracle's restatement of their published algorithms (oracle/preprocess.py), everything else stays a stub
    from oracle import preprocess as opre
    if 'cv2' not in sys.modules:
        def cv_resize(src, dsize, interpolation=None, **kw):
            assert interpolation == 2, 'only INTER_CUBIC is restated'
            return opre.resize_cubic_u8(np.ascontiguousarray(src), dsize[1], dsize[0])
        stub('cv2', resize=cv_resize, INTER_CUBIC=2)
    if 'h5py' not in sys.modules:
        stub('h5py')
    if 'imgaug' not in sys.modules:
        class _Pad(object):
            def __init__(self, px=None, keep_size=False, pad_mode='constant', pad_cval=0):
                assert not keep_size and pad_mode == 'constant'
                self.px, self.cval = px, pad_cval

            def __call__(self, image):
                return opre.pad_trbl(image, self.px, self.cval)

        class _Crop(object):
            def __init__(self, px=None, keep_size=False):
                assert not keep_size
                self.px = px

            def __call__(self, image):
                t, r, b, l = self.px
                return image[t:image.shape[0] - b, l:image.shape[1] - r]

        class _Sequential(object):
            def __init__(self, children):
                self.children = children

            def __call__(self, image=None):
                for c in self.children:
                    image = c(image)
                return image
        ia = stub('imgaug')
        iaa = stub('imgaug.augmenters', Pad=_Pad, Crop=_Crop, Sequential=_Sequential,
                   compute_paddings_to_reach_aspect_ratio=opre.compute_paddings_to_reach_aspect_ratio)
        ia.augmenters = iaa
        size = stub('imgaug.augmenters.size',
                    compute_paddings_to_reach_aspect_ratio=opre.compute_paddings_to_reach_aspect_ratio)
        iaa.size = size
    if 'chumpy' not in sys.modules:
        class Ch(object):
            pass
        ch = stub('chumpy', Ch=Ch)
        chch = stub('chumpy.ch', MatVecMult=None, Ch=Ch)
        ch.ch = chch
    if not hasattr(np, 'float'):
        np.float = float
    if not hasattr(np, 'int'):
        np.int = int
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self

    argv = sys.argv
    sys.argv = ['x']
    cwd = os.getcwd()
    if REF not in sys.path:
        sys.path.insert(0, REF)
    try:
        os.chdir(REF)          # config.py resolves configs/demo.yml relative to its own dir; be safe
        import acr.config      # noqa: F401
        import acr.utils as ref_utils
        import acr.result_parser as ref_parser
        import acr.model as ref_model
        import mano.manolayer as ref_manolayer
        if mano_tables is not None:
            _patch_mano(ref_manolayer, mano_tables)
        import acr.mano_wrapper as ref_wrapper
    finally:
        os.chdir(cwd)
        sys.argv = argv
    return ref_model, ref_parser, ref_wrapper, ref_manolayer, ref_utils


class _R(object):
    """Mimics a chumpy array: `.r` returns the ndarray (mano/manolayer.py:63-80)."""

    def __init__(self, a):
        self.r = a


def _patch_mano(ref_manolayer, tables):
    """tables: {'left': dict, 'right': dict} of numpy arrays (see synth.make_mano_tables)."""
    import numpy as np
    import scipy.sparse as sp

    def ready_arguments(path, posekey4vposed='pose'):
        side = 'left' if 'LEFT' in os.path.basename(path) else 'right'
        t = tables[side]
        return {
            'betas': _R(np.zeros(10)),
            'shapedirs': _R(t['shapedirs'].astype(np.float64)),
            'posedirs': _R(t['posedirs'].astype(np.float64)),
            'v_template': _R(t['v_template'].astype(np.float64)),
            'J_regressor': sp.csc_matrix(t['J_regressor'].astype(np.float64)),
            'weights': _R(t['weights'].astype(np.float64)),
            'f': t['faces'].astype(np.uint32),
            'hands_components': t['hands_components'].astype(np.float64),
            'hands_mean': t['hands_mean'].astype(np.float64),
            'kintree_table': t['kintree_table'],
        }
    ref_manolayer.ready_arguments = ready_arguments
