"""TEST INFRASTRUCTURE (oracle): CPU restatement of the reference's image pre-processing.

Reference path: acr/utils.py:1303-1337 (image_pad_white_bg -> imgaug Pad, img_preprocess -> cv2.resize INTER_CUBIC).
Both arithmetic pieces live in third-party packages that are absent from /root/reference and from this image:

  * imgaug == 0.4.0 (requirements.txt:21): `imgaug.augmenters.size.compute_paddings_to_reach_aspect_ratio` and
    `iaa.Pad(px=(top, right, bottom, left), keep_size=False, pad_mode='constant', pad_cval=255)`
  * opencv-python (requirements.txt:2, unpinned): `cv2.resize(src, (512, 512), interpolation=cv2.INTER_CUBIC)` on
    uint8: OpenCV's fixed-point path (modules/imgproc/src/resize.cpp: `resizeGeneric_` with
    HResizeCubic<uchar,int,short> / VResizeCubic<uchar,int,short,FixedPtCast<int,uchar,22>>,
    INTER_RESIZE_COEF_BITS = 11, coefficients `interpolateCubic` with A = -0.75, borders clamped)

so their published algorithms are restated here.  Parity with a cv2 *binary* stays unpinned (module absent; OpenCV's
SIMD builds evaluate the vertical pass of all but the row tail in float - `VResizeCubicVec_32s8u` - which can differ
from the scalar fixed-point path by 1 LSB on rounding ties); what is pinned is this published scalar algorithm, and
the HIP kernel is bit-exact to it.  Only tests/, tests/golden/make_golden.py and bench.py's cpu_baseline may import this.
"""
import numpy as np

INTER_RESIZE_COEF_BITS = 11
INTER_RESIZE_COEF_SCALE = 1 << INTER_RESIZE_COEF_BITS


def compute_paddings_to_reach_aspect_ratio(shape, aspect_ratio=1.0):
    """imgaug 0.4.0 size.py `compute_paddings_to_reach_aspect_ratio` -> (top, right, bottom, left)."""
    height, width = shape[0:2]
    top = right = bottom = left = 0
    if height == 0:
        height = 1
    current = width / height
    if current < aspect_ratio:      # more vertical than desired: widen (extra pixel on the right)
        diff = (aspect_ratio * height) - width
        right += int(np.ceil(diff / 2))
        left += int(np.floor(diff / 2))
    elif current > aspect_ratio:    # more horizontal than desired: heighten (extra pixel at the bottom)
        diff = ((1 / aspect_ratio) * width) - height
        top += int(np.floor(diff / 2))
        bottom += int(np.ceil(diff / 2))
    return top, right, bottom, left


def pad_trbl(image, trbl, cval=255):
    """iaa.Pad(px=trbl, keep_size=False, pad_mode='constant', pad_cval=cval) on one HxWxC uint8 image."""
    t, r, b, l = trbl
    return np.pad(image, ((t, b), (l, r), (0, 0)), mode='constant', constant_values=cval)


def _cubic_coeffs(x):
    """OpenCV interpolateCubic (float arithmetic, A = -0.75) for a float32 array of fractions -> [n,4] float32."""
    A = np.float32(-0.75)
    x = x.astype(np.float32)
    one = np.float32(1)
    c = np.empty(x.shape + (4,), np.float32)
    x1 = x + one
    c[..., 0] = ((A * x1 - np.float32(5) * A) * x1 + np.float32(8) * A) * x1 - np.float32(4) * A
    c[..., 1] = ((A + np.float32(2)) * x - (A + np.float32(3))) * x * x + one
    xm = one - x
    c[..., 2] = ((A + np.float32(2)) * xm - (A + np.float32(3))) * xm * xm + one
    c[..., 3] = one - c[..., 0] - c[..., 1] - c[..., 2]
    return c


def _taps(src_n, dst_n):
    """Per destination index: source origin (floor) and the four short coefficients (saturate_cast<short> of
    coeff * 2048 = round half to even), as resize.cpp computes them for one axis."""
    scale = float(src_n) / float(dst_n)                    # double
    d = np.arange(dst_n, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)       # fx = (float)((dx+0.5)*scale_x - 0.5)
    s = np.floor(f).astype(np.int64)                       # cvFloor
    f = f - s.astype(np.float32)
    coef = np.rint(_cubic_coeffs(f) * np.float32(INTER_RESIZE_COEF_SCALE))   # float product, round-half-even
    return s, np.clip(coef, -32768, 32767).astype(np.int32)


def resize_cubic_u8(image, out_h, out_w):
    """cv2.resize(image, (out_w, out_h), interpolation=cv2.INTER_CUBIC) for uint8 HxWxC, scalar fixed-point path."""
    image = np.ascontiguousarray(image)
    assert image.dtype == np.uint8 and image.ndim == 3
    H, W, _ = image.shape
    sx, ax = _taps(W, out_w)
    sy, ay = _taps(H, out_h)
    # (integer tensor arithmetic on torch-CPU: exact, and threaded - a 1920x1920 frame takes ~0.1 s instead of seconds)
    import torch
    src = torch.from_numpy(image)
    # horizontal pass: int32 rows, columns clamped to the border
    cols = torch.from_numpy(np.clip(sx[:, None] - 1 + np.arange(4)[None, :], 0, W - 1))      # [out_w, 4]
    tx = torch.from_numpy(ax)                                                                 # [out_w, 4] int32
    hor = torch.zeros(H, out_w, image.shape[2], dtype=torch.int32)
    for k in range(4):
        hor += src[:, cols[:, k], :].to(torch.int32) * tx[:, k].view(1, -1, 1)
    # vertical pass: rows clamped; OpenCV accumulates in int (no overflow for uint8 input: checked in int64)
    rows = torch.from_numpy(np.clip(sy[:, None] - 1 + np.arange(4)[None, :], 0, H - 1))      # [out_h, 4]
    ty = torch.from_numpy(ay).to(torch.int64)
    acc = torch.zeros(out_h, out_w, image.shape[2], dtype=torch.int64)
    for k in range(4):
        acc += hor[rows[:, k]].to(torch.int64) * ty[:, k].view(-1, 1, 1)
    assert int(acc.abs().max()) < 2 ** 31 - 2 ** 21
    acc32 = acc.numpy().astype(np.int32)
    # FixedPtCast<int, uchar, 22>: (v + (1 << 21)) >> 22, saturate to [0, 255]
    shift = 2 * INTER_RESIZE_COEF_BITS
    out = (acc32 + (1 << (shift - 1))) >> shift
    return np.clip(out, 0, 255).astype(np.uint8)


def img_preprocess(bgr, input_size=512):
    """acr/utils.py:1315-1337 on one BGR uint8 frame -> (uint8 RGB [S,S,3], offsets float32 [10])."""
    rgb = np.ascontiguousarray(bgr[:, :, ::-1])
    trbl = compute_paddings_to_reach_aspect_ratio(rgb.shape, 1.0)
    padded = pad_trbl(rgb, trbl, 255)
    out = resize_cubic_u8(padded, input_size, input_size)
    offsets = np.array([padded.shape[0], padded.shape[1], 0, 0, 0, 0, *trbl], np.float32)
    return out, offsets
