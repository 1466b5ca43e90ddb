"""Checkpoint -> (packed weight blob, op program) for libacrmi.so.

Takes a reference-format state dict (schema.py; keys of acr.model.ACR().state_dict(),
optionally prefixed 'module.' as in wild.pkl, acr/utils.py:1106-1113), folds every
eval-mode BatchNorm into its convolution (w' = w*g/sqrt(var+eps), b' = beta + (b-mean)*g/sqrt(var+eps),
eps = 1e-5), packs the weights in the MFMA B-fragment order conv_mfma.hip reads, and lowers the
HRNet-W32 + ACR-head topology (acr/model.py:785-865, :47-166) to a flat list of ops over NHWC
activation buffers.  Pure numpy; runs on the host once per checkpoint.
"""
import numpy as np

from . import _lib
from .schema import RESNET50, RESNET50_LAYERS, RESNET50_UP, backbone_channels, stage_cfg, state_dict_schema, width_of

EPS = 1e-5
# 3x3 stride-1 convolutions run as Winograd F(2,3) along x (exact-arithmetic equivalent, 1.5x fewer MFMAs;
# fp32 round-off differs from the direct form by ~1e-6 relative).  Set False to lower everything to direct conv.
WINOGRAD = True
# ... and as 2-D Winograd F(2x2,3x3) (2.25x fewer MFMAs, ~4e-6 relative round-off) where the kernel supports it.
WINOGRAD_2D = True


def _np(t):
    if hasattr(t, 'detach'):
        t = t.detach().cpu().numpy()
    return np.asarray(t)


def strip_prefix(sd, prefix='module.'):
    """acr/utils.py:1106-1151 (copy_state_dict): accept both bare and 'module.'-prefixed keys,
    unwrap 'model_state_dict' / 'state_dict' containers (:1159-1163)."""
    for k in ('model_state_dict', 'state_dict'):
        if k in sd and isinstance(sd[k], dict):
            sd = sd[k]
    out = {}
    for k, v in sd.items():
        out[k[len(prefix):] if k.startswith(prefix) else k] = v
    return out


def check_state_dict(sd):
    want = state_dict_schema(width_of(sd))
    missing = [k for k in want if k not in sd and not k.startswith('segmentation_layers.')
               and not k.endswith('num_batches_tracked')]
    if missing:
        raise ValueError('checkpoint is missing %d tensors, e.g. %s' % (len(missing), missing[:3]))
    for k, shp in want.items():
        if k in sd and tuple(_np(sd[k]).shape) != tuple(shp):
            raise ValueError('checkpoint tensor %s has shape %s, expected %s' % (k, tuple(_np(sd[k]).shape), shp))


def n_tiles_for(cout):
    return 1 if cout <= 32 else ((cout + 63) // 64) * 2


def pack_conv(w, b):
    """w [Cout,Cin,k,k], b [Cout] (float64/32) -> (packed weights fp32 1-D, padded bias fp32 [n_tiles*32]).
    Packed index: [tap][s = ci/8][ntile][lane 64][e 4] with cout = ntile*32 + (lane & 31),
    ci = 8*s + 4*(lane >> 5) + e."""
    cout, cin, kh, kw = w.shape
    nt = n_tiles_for(cout)
    c8 = (cin + 7) // 8
    wp = np.zeros((nt * 32, c8 * 8, kh, kw), np.float32)
    wp[:cout, :cin] = w
    wp = wp.reshape(nt, 32, c8, 2, 4, kh, kw)           # [nt, j, s, h, e, ky, kx]
    if cout == 33 and (kh, kw) == (4, 4):
        # F(2x2,3x3) taps of a 33-cout conv: conv_wino2_kernel's ODD path feeds the 33rd channel to 4x4x1 MFMAs, which
        # take a block's weight from the row of its first lane - cout 32 replicated into every 4th row of tile 1.  The
        # extra rows are couts 36, 40, ... of the regular layout: never stored, so every other path stays correct.
        wp[1, 4::4] = wp[1, 0:1]
    wp = wp.transpose(5, 6, 2, 0, 3, 1, 4)               # [ky, kx, s, nt, h, j, e]
    bp = np.zeros(nt * 32, np.float32)
    bp[:cout] = b
    return np.ascontiguousarray(wp).reshape(-1), bp


DT_F32, DT_F16, DT_BF16 = 0, 1, 2
# 'fp16x3' / 'bf16x3': fp32 storage, split f16 / bf16 operands on the 16-bit matrix pipe where conv_x3_kernel takes the layer
# (algo 6 / 7)
PRECISIONS = {'fp32': DT_F32, 'fp16': DT_F16, 'bf16': DT_BF16, 'fp16x3': DT_F32, 'bf16x3': DT_F32}


def round_to(x, dt):
    """float64/32 array -> the values of storage type dt (round to nearest even), as float32."""
    x = np.asarray(x, np.float32)
    if dt == DT_F16:
        return x.astype(np.float16).astype(np.float32)
    if dt == DT_BF16:
        return (to_bits16(x, dt).astype(np.uint32) << 16).view(np.float32).reshape(x.shape)
    return x


def to_bits16(x, dt):
    """float array -> uint16 bit patterns of f16 / bf16 (round to nearest even; NaN/Inf never occur in weights)."""
    x = np.ascontiguousarray(x, np.float32)
    if dt == DT_F16:
        return x.astype(np.float16).view(np.uint16)
    u = x.view(np.uint32).astype(np.uint64)
    return (((u + 0x7FFF + ((u >> 16) & 1)) >> 16) & 0xFFFF).astype(np.uint16)


def pack_conv_h16(w, b, dt):
    """w [Cout,Cin,k,k], b [Cout] -> (packed 16-bit weights as uint16 1-D, padded bias fp32 [n_tiles*32]) for
    conv_h16_kernel (csrc/conv_h16.inc): the A fragments of v_mfma_f32_32x32x16_{f16,bf16}, 1 KiB each,
    [tap][s = ci/16][ntile][lane 64][e 8] with cout = ntile*32 + (lane & 31), ci = 16*s + 8*(lane >> 5) + e."""
    cout, cin, kh, kw = w.shape
    nt = n_tiles_for(cout)
    c16 = (cin + 15) // 16
    wp = np.zeros((nt * 32, c16 * 16, kh, kw), np.float64)
    wp[:cout, :cin] = w
    wp = wp.reshape(nt, 32, c16, 2, 8, kh, kw)            # [nt, j, s, h, e, ky, kx]
    wp = wp.transpose(5, 6, 2, 0, 3, 1, 4)                # [ky, kx, s, nt, h, j, e]
    bp = np.zeros(nt * 32, np.float32)
    bp[:cout] = b
    return to_bits16(np.ascontiguousarray(wp).reshape(-1), dt), bp


def split16(x, dt=DT_F16):
    """float64/32 array -> (hi, lo) as float32 arrays holding 16-bit values: hi = round16(x), lo = round16(x - hi)."""
    x = np.asarray(x, np.float64)
    hi = round_to(x, dt).astype(np.float64)
    lo = round_to(x - hi, dt)
    return hi.astype(np.float32), lo


def x3_weight_shift(ws):
    """Power-of-two pre-scaling of an op's filters for conv_x3_kernel: S with max |w| 2^S in [2^12, 2^13) (-30 <= S <= 30), so
    that lo = f16(w 2^S - hi) stays a NORMAL f16 number for every |w| >= 2^-16 max |w| (unscaled, a weight below 2^-3 has
    its lo in the f16 subnormals: an absolute error of up to 3e-8 instead of 2^-22 |w|) - and, S < 0, so that folded filters
    ABOVE the f16 range (|w| >= 2^13 already shifts down) never become inf halves."""
    m = max(float(np.max(np.abs(w))) for w in ws)
    if not np.isfinite(m) or m <= 0.0:
        return 0
    return int(min(30, max(-30, 12 - int(np.floor(np.log2(m))))))


def pack_conv_x3(wb_list, dt=DT_F16):
    """[(w [Cout,Cin,3,3], b [Cout])] per group (Cin % 16 == 0) -> (float32 1-D holding the split f16 weights + one trailing
    float, padded biases fp32 [groups][n_tiles*32]) for conv_x3_kernel (csrc/conv_x3.inc): the A fragments of
    v_mfma_f32_32x32x16_f16 of hi = f16(w 2^S) and lo = f16(w 2^S - hi), 1 KiB each,
    [group][tap][s = ci/16][ntile][hi | lo][lane 64][e 8], cout = ntile*32 + (lane & 31), ci = 16*s + 8*(lane >> 5) + e; the
    trailing float is 2^-S (x3_weight_shift), which the kernel applies to the accumulators (exact)."""
    shift = x3_weight_shift([w for (w, _) in wb_list])
    out, biases = [], []
    for (w, b) in wb_list:
        cout, cin, kh, kw = w.shape
        assert cin % 16 == 0
        nt = n_tiles_for(cout)
        c16 = cin // 16
        wp = np.zeros((nt * 32, cin, kh, kw), np.float64)
        wp[:cout] = np.asarray(w, np.float64) * 2.0 ** shift
        hi, lo = split16(wp, dt)
        both = np.stack([hi, lo])                              # [hl, cout, cin, ky, kx]
        both = both.reshape(2, nt, 32, c16, 2, 8, kh, kw)       # [hl, nt, j, s, h, e, ky, kx]
        both = both.transpose(6, 7, 3, 1, 0, 4, 2, 5)           # [ky, kx, s, nt, hl, h, j, e]
        out.append(to_bits16(np.ascontiguousarray(both).reshape(-1), dt).view(np.float32))
        bp = np.zeros(nt * 32, np.float32)
        bp[:cout] = b
        biases.append(bp)
    out.append(np.array([2.0 ** -shift], np.float32))
    return np.concatenate(out), np.concatenate(biases)


def pack_wino3(w, b):
    """3x3 filters [32, Cin <= 32, 3, 3] -> the LDS-resident layout of conv_wino3_kernel: F(2x2,3x3) taps U = G g G^T
    as [pos = py*4+px][step s][cout half][lane 64][e 4] with cout = 16*half + (lane & 15),
    cin = 16*s + 4*(lane >> 4) + e (the A fragments of v_mfma_f32_16x16x4_f32), zero padded to 32x32: 64 KiB."""
    cout, cin = w.shape[:2]
    assert w.shape[2:] == (3, 3) and cout == 32 and cin <= 32
    u = np.zeros((32, 32, 4, 4))
    u[:cout, :cin] = winograd2d_weights(w)
    u = u.reshape(2, 16, 2, 4, 4, 4, 4)                   # [half, j, s, kq, e, py, px]
    u = u.transpose(5, 6, 2, 0, 3, 1, 4)                   # [py, px, s, half, kq, j, e]
    bp = np.zeros(32, np.float32)
    bp[:cout] = b
    return np.ascontiguousarray(u, np.float32).reshape(-1), bp


def pack_stem(w, b):
    """conv1 filters [64, 3, 3, 3] (BN folded) -> the register-resident A fragments of stem_kernel (csrc/stem.hip):
    [step s 14][n-tile 2][lane 64] with cout = 32*n + (lane & 31), k = 2*s + (lane >> 5), k = (ky*3 + kx)*3 + c
    (k = 27 is the zero pad of the 28-wide reduction); bias [64]."""
    assert w.shape == (64, 3, 3, 3)
    wk = np.zeros((64, 28))
    wk[:, :27] = np.asarray(w, np.float64).transpose(0, 2, 3, 1).reshape(64, 27)       # [cout][ky][kx][c]
    frag = wk.reshape(2, 32, 14, 2).transpose(2, 0, 3, 1)                               # [s][n][lh][li]
    return np.ascontiguousarray(frag, np.float32).reshape(-1), np.asarray(b, np.float32).copy()


def pack_stem7(w, b):
    """ResNet conv1 filters [64, 3, 7, 7] (BN folded) -> the register-resident A fragments of stem7_kernel (csrc/stem7.hip):
    [step s 74][n-tile 2][lane 64] with cout = 32*n + (lane & 31), k = 2*s + (lane >> 5), k = (ky*7 + kx)*3 + c
    (k = 147 is the zero pad of the 148-wide reduction); bias [64]."""
    assert w.shape == (64, 3, 7, 7)
    wk = np.zeros((64, 148))
    wk[:, :147] = np.asarray(w, np.float64).transpose(0, 2, 3, 1).reshape(64, 147)     # [cout][ky][kx][c]
    frag = wk.reshape(2, 32, 74, 2).transpose(2, 0, 3, 1)                               # [s][n][lh][li]
    return np.ascontiguousarray(frag, np.float32).reshape(-1), np.asarray(b, np.float32).copy()


def pack_pair1x1(w3, b3, w1, b1):
    """Folded conv3 [256,64] / bias [256] of a layer1 Bottleneck and conv1 [64,256] / bias [64] of the NEXT one -> the LDS
    image of pair1x1_kernel (csrc/pair1x1.hip), 33088 floats:
      A1 [c 8][s4 8][lane 64][4]: W3[32 c + m][32 h + 4 s4 + e]           lane = 32 h + m   (k order of the first GEMM:
                                                                           lane half h reads t2 channels 32 h ..)
      A2 [c 8][nt 2][g 4][lane 64][4]: W1[32 nt + m][32 c + 8 g + 4 h + e]   (k order of the second GEMM = the row
                                                                           order of the first one's accumulator registers)
      b3 [c 8][g 4][h 2][4], b1 [nt 2][g 4][h 2][4]: biases in accumulator-register order."""
    w3 = np.asarray(w3, np.float64).reshape(256, 64)
    w1 = np.asarray(w1, np.float64).reshape(64, 256)
    a1 = w3.reshape(8, 32, 2, 8, 4).transpose(0, 3, 2, 1, 4).reshape(-1)            # c, m, h, s4, e -> c, s4, h, m, e
    a2 = w1.reshape(2, 32, 8, 4, 2, 4).transpose(2, 0, 3, 4, 1, 5).reshape(-1)      # nt, m, c, g, h, e -> c, nt, g, h, m, e
    out = np.concatenate([a1, a2, np.asarray(b3, np.float64).reshape(-1), np.asarray(b1, np.float64).reshape(-1)])
    assert out.size == 256 * 64 + 64 * 256 + 256 + 64
    return out.astype(np.float32)


# layer1's conv3 (64 -> 256 + residual + ReLU) and the next block's conv1 (256 -> 64 + ReLU) as ONE launch (OP_PAIR1X1): the
# 256-channel map is written once and not read back by the pair (two HBM-bound launches otherwise).  Large-batch fp32
# programs (Engine.load_state_dict: max_batch >= 16); with it layer1.0 keeps its projection shortcut as a separate conv (the
# pair takes it as the residual).
FUSE_PAIRS = True


# the stem (u8norm + conv1) as one kernel reading the uint8 image; False = the two generic ops of round 1
STEM_FUSED = True

WINO_G = np.array([[1.0, 0.0, 0.0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0.0, 0.0, 1.0]])


def winograd_weights(w):
    """[Cout,Cin,3,3] -> [Cout,Cin,3,4]: Winograd F(2,3) weight transform along kx, U[ky][v] = sum_kx G[v][kx] w[ky][kx]
    (fp64 in, rounded once by pack_conv).  conv_wino_kernel consumes them as 3x4 'taps'."""
    assert w.shape[2:] == (3, 3)
    return np.einsum('vk,oiyk->oiyv', WINO_G, np.asarray(w, np.float64))


def winograd2d_weights(w):
    """[Cout,Cin,3,3] -> [Cout,Cin,4,4]: Winograd F(2x2,3x3) weight transform U = G g G^T (fp64 in, rounded once by
    pack_conv).  conv_wino2_kernel consumes them as 4x4 'taps' (py, px)."""
    assert w.shape[2:] == (3, 3)
    return np.einsum('uy,vx,oiyx->oiuv', WINO_G, WINO_G, np.asarray(w, np.float64))


WINO_G4 = np.array([[1 / 4.0, 0.0, 0.0], [-1 / 6.0, -1 / 6.0, -1 / 6.0], [-1 / 6.0, 1 / 6.0, -1 / 6.0],
                    [1 / 24.0, 1 / 12.0, 1 / 6.0], [1 / 24.0, -1 / 12.0, 1 / 6.0], [0.0, 0.0, 1.0]])


def winograd24_weights(w):
    """[Cout,Cin,3,3] -> [Cout,Cin,4,6]: Winograd F(2x4,3x3) weight transform U = G2 g G4^T - F(2,3) along y, F(4,3) along
    x (fp64 in, rounded once by pack_conv).  conv_wino24_kernel consumes them as 4x6 'taps' (py, px)."""
    assert w.shape[2:] == (3, 3)
    return np.einsum('uy,vx,oiyx->oiuv', WINO_G, WINO_G4, np.asarray(w, np.float64))


def polyphase2_weights(w):
    """[Cout,Cin,3,3] -> [Cout,Cin,4,7]: the weights of conv_pp2_kernel (csrc/conv_pp2.inc), 3x3 stride 2 in polyphase form
    with F(2,2) on the two-tap phases.  Per axis five 1-D terms: A0 = (d0 - d1) w0, A1 = d1 (w0 + w2), A2 = (d2 - d1) w2 on
    the odd input rows, B0 / B1 = w1 on the even ones.  Wave (oy, ox) holds, in this order, its corner products
    (Ya, Xa), (Ya, B), (B, Xa), (B, B) [Ya = A0 / A2 for oy = 0 / 1], two edge products (Ya, A1), (B, A1) - the waves with
    oy != ox read the TRANSPOSED pixel set, which makes theirs (B, Xa) before (Ya, B) and (A1, Xa), (A1, B) - and the centre
    (A1, A1) (every wave computes it, wave 0's copy is used).  fp64 in, rounded once by pack_conv."""
    assert w.shape[2:] == (3, 3)
    w = np.asarray(w, np.float64)
    c = {'A0': (1.0, 0.0, 0.0), 'A1': (1.0, 0.0, 1.0), 'A2': (0.0, 0.0, 1.0), 'B': (0.0, 1.0, 0.0)}
    out = np.zeros(w.shape[:2] + (4, 7))
    for oy in range(2):
        for ox in range(2):
            ya, xa = ('A0', 'A2')[oy], ('A0', 'A2')[ox]
            if oy == ox:
                prods = [(ya, xa), (ya, 'B'), ('B', xa), ('B', 'B'), (ya, 'A1'), ('B', 'A1'), ('A1', 'A1')]
            else:
                prods = [(ya, xa), ('B', xa), (ya, 'B'), ('B', 'B'), ('A1', xa), ('A1', 'B'), ('A1', 'A1')]
            for kk, (ty, tx) in enumerate(prods):
                out[:, :, 2 * oy + ox, kk] = np.einsum('oiyx,y,x->oi', w, np.array(c[ty]), np.array(c[tx]))
    return out


# 3x3 stride-2 convs on the four-wave frame in polyphase form (algo 5, conv_pp2.inc: 25 products per 2x2 output block
# instead of 36).  ACRMI_PP2=0 keeps the direct kernel (A/B runs).
POLYPHASE2 = __import__('os').environ.get('ACRMI_PP2', '1') != '0'


POLYPHASE2_SMALL = __import__('os').environ.get('ACRMI_PP2_SMALL', '1') != '0'      # ... also in small-batch programs (A/B: 0)


def polyphase2_ok(cin, cout, ho, wo):
    """conv_pp2_kernel takes the layer (csrc/conv_pp2.inc pp2_ok; ho, wo = OUTPUT map, the input is twice that)."""
    return cin % 16 == 0 and cout % 32 == 0 and ho % 8 == 0 and wo % 16 == 0


# Split-operand 16-bit program ('fp16x3': fp32 storage, every operand of the eligible convolutions split into two f16
# numbers, three products per MAC on the 16-bit matrix pipe with fp32 accumulation - csrc/conv_x3.inc, algo 6).
# 3x3 stride-2 layers of the split-operand programs on conv_x3s2_kernel (round 6); ACRMI_X3_STRIDE2=0 keeps the fp32 polyphase
# kernel for them (A/B runs)
SPLIT16_STRIDE2 = __import__('os').environ.get('ACRMI_X3_STRIDE2', '1') != '0'


def split16_ok(k, stride, cin, cout, ho, wo):
    """conv_x3_kernel (3x3) / conv_x3p_kernel (1x1) / conv_x3s2_kernel (3x3 stride 2) takes the layer (csrc/conv_x3.inc x3_ok,
    conv_x3p.inc x3p_ok, conv_x3s2.inc x3s2_ok)."""
    if not (cin % 32 == 0 and cin >= 32 and cout % 32 == 0):
        return False
    if stride == 2:
        return SPLIT16_STRIDE2 and k == 3 and ho % 8 == 0 and wo % 32 == 0
    if stride != 1:
        return False
    return (k == 3 and ((ho % 8 == 0 and wo % 32 == 0) or (ho % 16 == 0 and wo % 16 == 0))) or (
        k == 1 and ho * wo > 0 and (ho * wo) % 256 == 0)


def use_winograd(k, stride):
    return k == 3 and stride == 1


# ... and, for Cin <= 32 / Cout = 32 layers on tile-aligned maps, on the kernel that keeps the layer's taps in LDS and a
# wave's 16 positions in registers (conv_wino3.inc).
WINOGRAD_LDS = True


# F(2x4,3x3) (algo 4, conv_wino24.inc: F(2,3) along y, F(4,3) along x - 1.33x fewer MFMAs than F(2x2,3x3)) for the layers
# with Cin > 32 on maps that tile into 8x32-pixel items (branches 1 and 2, the head towers, layer1's 3x3, the contact and
# transition convs: 133 launches).  Round 3: first built with 16-channel chunks (no gain: 44 % of an item's cycles in MFMAs),
# then with 32-channel chunks and the exchange in two rounds: 0.104 vs 0.108 ms (64->64 @64x64), 0.092 vs 0.099
# (128->128 @32x32), 0.388 vs 0.413 (64->64 @128x128) per launch at batch 64; whole path 1631 -> 1662 frames/s on the same
# box.  Round-off, hostile checkpoint: worst layer 1.2e-6 of its largest output (F(2x2): 9.5e-7), vertices 2.9e-7 m.
# ACRMI_WINO24=0 lowers those layers to F(2x2,3x3) again (A/B runs).
WINOGRAD_24 = __import__('os').environ.get('ACRMI_WINO24', '1') != '0'


def wino24b_width(cin, cout, ho, wo):
    """Item width of conv_wino24b_kernel for a 3x3 stride-1 conv (csrc/conv_wino24b.inc wino24b_ok): 32, 16 or 0 = not taken."""
    if not (cin % 32 == 0 and cin >= 32 and cout % 32 == 0):
        return 0
    if ho % 8 == 0 and wo % 32 == 0:
        return 32                                     # two n-tiles per wave when Cout % 64 == 0, else one
    if ho % 16 == 0 and wo % 16 == 0 and cin >= 64 and cout % 64 == 0:
        return 16
    return 0


def conv_algo(k, stride, cin, cout, groups=1, ho=0, wo=0, per_frame_bias=False, wino24=None, split16=False):
    """0 direct, 1 Winograd F(2,3) along x, 2 Winograd F(2x2,3x3), 3 F(2x2,3x3) with LDS-resident taps, 4 F(2x4,3x3),
    5 polyphase F(2,2) (3x3 stride 2), 6 / 7 split f16 / bf16 operands on the 16-bit matrix pipe (split16 = True | 'fp16' / 'bf16':
    the 'fp16x3' / 'bf16x3' programs only).
    wino24: None = WINOGRAD_24; False keeps the F(2x2,3x3) kernels for the layers F(2x4,3x3) would take (small batches:
    its 8x32-pixel, one-n-tile items are half as many as conv_wino2's small-batch items)."""
    if split16 and split16_ok(k, stride, cin, cout, ho, wo):
        return 7 if split16 == 'bf16' else 6
    if k == 3 and stride == 2:
        # round 4 took the polyphase kernel for the large-batch lowering only; its items (8x16 output pixels x 32 or 64 couts)
        # are no fewer than the direct kernel's at small batches and run 25 products per 2x2 block instead of 36: the 24
        # stride-2 launches on a batch-1 call's dependency chain were 22.8 us each as direct convolutions (tools/critical_path.py)
        big = (WINOGRAD_24 if wino24 is None else wino24) or POLYPHASE2_SMALL
        return 5 if (POLYPHASE2 and big and polyphase2_ok(cin, cout, ho, wo)) else 0
    if not (WINOGRAD and use_winograd(k, stride)):
        return 0
    if (WINOGRAD_2D and WINOGRAD_LDS and groups == 1 and cin <= 32 and cout == 32 and ho % 8 == 0 and wo % 16 == 0
            and not per_frame_bias):
        return 3
    if WINOGRAD_2D and (WINOGRAD_24 if wino24 is None else wino24) and cout != 33:
        if cin > 32 and wo % 32 == 0 and ho % 8 == 0:
            return 4
        # what only the four-wave frame takes: maps narrower than 32 pixels (HRNet branch 3, 16 x 16: conv_wino24_kernel would
        # waste half of its 32 slots there) and single-chunk items (Cin = 32 with Cout a multiple of 64: the contact conv)
        if wino24b_width(cin, cout, ho, wo) and (cin > 32 or cout % 64 == 0):
            return 4
    return 2 if WINOGRAD_2D else 1


# The head convs that read the 32 + 2 channel map (backbone output + coordinate maps, acr/model.py:52): the coordinate
# channels are the same for every frame, so their share of the contraction is a per-pixel bias - conv(coord maps, their
# filter columns), zero padding included - computed here once and added by the kernels like a residual shared by all
# frames (ACRMI_CONV_BIAS_MAP).  Cin 34 -> 32: one 32-channel chunk instead of two for the Winograd kernels, two
# 16-channel chunks instead of three for the stride-2 one.  fp32 programs only (a 16-bit map would be one more rounding).
COORD_BIAS_MAP = True


def coord_bias_map(wc, size, stride):
    """wc [Cout,2,3,3] (fp64, BN-folded filter columns of the x and y coordinate channels) -> [Ho,Wo,round4(Cout)] fp64:
    the 3x3 / padding-1 convolution of the size x size coordinate maps (the fp32 values coordfill_kernel writes,
    acr/model.py:340-369) with wc."""
    f32 = np.float32
    t = ((np.arange(size, dtype=f32) / f32(size - 1)) * f32(2.0) - f32(1.0)).astype(np.float64)
    cm = np.zeros((2, size + 2, size + 2))
    cm[0, 1:-1, 1:-1] = t[None, :]
    cm[1, 1:-1, 1:-1] = t[:, None]
    ho = (size - 1) // stride + 1
    cout = wc.shape[0]
    out = np.zeros((ho, ho, (cout + 3) // 4 * 4))
    for ky in range(3):
        for kx in range(3):
            patch = cm[:, ky:ky + stride * (ho - 1) + 1:stride, kx:kx + stride * (ho - 1) + 1:stride]
            out[:, :, :cout] += np.einsum('chw,oc->hwo', patch, wc[:, :, ky, kx])
    return out


# Split-K (ACRMI_CONV_SPLITK; small-batch programs only - Engine.load_state_dict asks for it below 16 frames): a 3x3
# stride-1 Winograd layer whose launch would be at most SPLITK_MAX_ITEMS work items on one frame is lowered with its input
# channels in 64-channel slices that run as separate items (256 -> 256 at 16x16: 16 items of 8 chunks -> 64 of 2; 128 ->
# 128 at 32x32: 32 of 4 -> 64 of 2).  tools/critical_path.py: at batch 1 these layers ARE the dependency chain the call's
# latency follows (24 launches of 31 us + 32 of 20 us of its 2.4 ms).
SPLITK_MAX_ITEMS = 32


def splitk_slices(k, stride, cin, cout, groups, ho, wo, per_frame_bias):
    """number of K-slices a small-batch program gives this conv (1 = not split)"""
    if not (k == 3 and stride == 1 and groups == 1 and not per_frame_bias and cout != 33 and cin >= 128 and cin % 64 == 0):
        return 1
    n_tiles = 1 if cout <= 32 else (cout + 63) // 64 * 2
    items = ((ho + 7) // 8) * ((wo + 15) // 16) * n_tiles
    if items > SPLITK_MAX_ITEMS:
        return 1
    # the largest slice count <= 8 whose slices are whole 32-channel chunks (the kernel needs Cin % 32 == 0 per slice:
    # 576 or 640 input channels must not become 9 or 10 ragged slices cut down to 8)
    for s in range(min(cin // 64, 8), 1, -1):
        if cin % (32 * s) == 0:
            return s
    return 1


# HR-module fuse sums of the lower resolutions in the epilogue of the x0 chain's last stride-2 convolution (Program.hr_module);
# ACRMI_FUSE_EPI=0 keeps every sum an OP_FUSESUM launch (A/B runs)
FUSE_EPILOGUE = __import__('os').environ.get('ACRMI_FUSE_EPI', '1') != '0'
# ... and the full-resolution sum as the second output of branch 0's last conv2 (Program.fuse0_ok); ACRMI_FUSE_EPI0=0: a launch
FUSE_FULLRES = __import__('os').environ.get('ACRMI_FUSE_EPI0', '1') != '0'


# layer1.0's projection shortcut as extra input channels of its last 1x1 conv (Program.bottleneck): batch 64, fp32:
# downsample 0.46 ms + conv3 0.51 ms -> one 128 -> 256 conv.
FUSE_PROJECTION = True


class Blob(object):
    def __init__(self):
        self.parts = [np.zeros(64, np.float32)]   # offset 0 reserved
        self.n = 64

    def add16(self, bits):
        """uint16 bit patterns (an even number of them) stored in the fp32 blob two per float slot."""
        bits = np.ascontiguousarray(bits, np.uint16).reshape(-1)
        assert bits.size % 2 == 0
        return self.add(bits.view(np.float32))

    def add(self, arr):
        arr = np.ascontiguousarray(arr, np.float32).reshape(-1)
        pad = (-arr.size) % 64                     # keep every tensor 256-byte aligned
        off = self.n
        self.parts.append(arr)
        if pad:
            self.parts.append(np.zeros(pad, np.float32))
        self.n += arr.size + pad
        return off

    def finish(self):
        return np.concatenate(self.parts)


class Program(object):
    """Op list + buffer table under construction."""

    def __init__(self, sd, dt=DT_F32, keep_weights=False, keep_all=False, wino24=None, splitk=False, pairs=False, split16=False):
        self.sd = {k: _np(v) for k, v in sd.items()}
        self.dt = dt             # storage type of the activations between layers (DT_*); head outputs stay fp32
        self.keep_weights = keep_weights
        self.splitk = splitk      # small-batch program: split-K lowering of the low-resolution 3x3 layers
        self.pairs = pairs and dt == DT_F32      # large-batch fp32 program: layer1's conv3 / next conv1 pairs as one op
        self.wino24 = wino24      # None = packer.WINOGRAD_24
        self.split16 = split16 if dt == DT_F32 else False      # 'fp16x3' / 'bf16x3' program (True | 'fp16' | 'bf16')
        self.keep_all = keep_all  # no lifetime-based buffer reuse: every intermediate map survives the run (tests)
        self.chan_pad = {}        # backbone-internal channel padding of the large-batch lowering (see pad_channels)
        self.blob = Blob()
        self.bufs = []           # (h, w, cs, persistent, dtype)
        self.free = {}           # (h, w, cs, dtype) -> [ids]
        self.ops = []
        self.op_info = []        # python-side description (name, flops) per op

    # ---- buffers -----------------------------------------------------------------------------
    def buf(self, h, w, c, persistent=False, f32=False):
        """f32: an fp32 buffer also in a 16-bit program (head outputs, pooled features, bias rows)."""
        dt = DT_F32 if f32 else self.dt
        q = 4 if dt == DT_F32 else 8                     # 16-byte vectors: 4 floats / 8 halfs
        cs = (c + q - 1) // q * q
        key = (h, w, cs, dt)
        if not persistent and self.free.get(key):
            return self.free[key].pop()
        self.bufs.append((h, w, cs, 1 if persistent else 0, dt))
        return len(self.bufs) - 1

    def release(self, *ids):
        for i in ids:
            h, w, cs, p, dt = self.bufs[i]
            if not p and not self.keep_all:
                assert i not in self.free.setdefault((h, w, cs, dt), []), 'double release of buffer %d' % i
                self.free[(h, w, cs, dt)].append(i)

    def pin(self, i):
        """Excludes buffer i from lifetime-based reuse from now on (its contents survive the whole program)."""
        h, w, cs, _, dt = self.bufs[i]
        self.bufs[i] = (h, w, cs, 1, dt)
        return i

    def dtype_of(self, i):
        return self.bufs[i][4]

    def dims(self, i):
        return self.bufs[i][:3]

    # ---- weights -----------------------------------------------------------------------------
    def folded(self, conv, bn=None):
        w = self.sd[conv + '.weight'].astype(np.float64)
        b = self.sd.get(conv + '.bias')
        b = np.zeros(w.shape[0]) if b is None else b.astype(np.float64)
        if bn is not None:
            g = self.sd[bn + '.weight'].astype(np.float64)
            scale = g / np.sqrt(self.sd[bn + '.running_var'].astype(np.float64) + EPS)
            w = w * scale[:, None, None, None]
            b = self.sd[bn + '.bias'].astype(np.float64) + (b - self.sd[bn + '.running_mean'].astype(np.float64)) * scale
        return w, b

    # ---- ops ---------------------------------------------------------------------------------
    def _op(self, name, flops=0.0, **kw):
        op = _lib.Op()
        for f in ('in_buf', 'out_buf', 'res_buf', 'aux_buf'):
            setattr(op, f, -1)
        for k, v in kw.items():
            setattr(op, k, v)
        self.ops.append(op)
        self.op_info.append({'name': name, 'flops': float(flops), 'kind': int(op.kind), 'mode': int(op.mode)})
        return op

    def set_mode(self, mode, first):
        """Tags ops[first:] with a head-program variant (_lib.MODE_*)."""
        for op, info in zip(self.ops[first:], self.op_info[first:]):
            op.mode = info['mode'] = mode

    def clone_op(self, i, name, flops_scale, **kw):
        """Copy of op i (same packed weights / buffers) with some fields changed."""
        op = _lib.Op.from_buffer_copy(bytes(self.ops[i]))
        for k, v in kw.items():
            setattr(op, k, v)
        self.ops.append(op)
        info = dict(self.op_info[i])
        info.update(name=name, flops=info['flops'] * flops_scale, mode=int(op.mode))
        self.op_info.append(info)
        return op

    def conv(self, name, src, wb_list, k, stride, relu, out=None, out_c=None, in_coff=0, out_coff=0, res=None,
             res_coff=0, cin=None, bias_buf=None, bias_map=None, terms=None, dual=None, fp32_kernel=False):
        """wb_list: [(w, b)] one entry per group (all same shape).  bias_map: [Ho,Wo,round4(groups*Cout)] added to every
        frame before the ReLU (ACRMI_CONV_BIAS_MAP; fp32 programs, no residual).  terms: [(buf, shift)] up to three extra
        residual maps at 1 / 2^shift of the output size, added behind `res` before the ReLU in this order (fp32 3x3
        stride-2 convolutions: the HR fuse sum in the epilogue of the x0 downsampling chain, fuse_epilogue_ok).
        dual: (out2, [(buf, shift)]) - ACRMI_CONV_DUAL: `out` gets the convolution as usual, out2 = relu(out + the terms) - the
        full-resolution HR fuse sum written by conv_wino3_kernel's store waves (algo 3 only, fuse0_ok).
        fp32_kernel: keep the fp32 kernel in a split-operand program (the last convolution of an x0 downsampling chain: it
        is the HR fuse host where fuse_epilogue_ok, on conv_pp2_kernel - and stays on that kernel where it is not, so that
        both lowerings compute the same bits)."""
        h, w_, _ = self.dims(src)
        cout, cin_w = wb_list[0][0].shape[:2]
        cin = cin_w if cin is None else cin
        ho, wo = (h + 2 * (k // 2) - k) // stride + 1, (w_ + 2 * (k // 2) - k) // stride + 1
        flop_cout = cout
        if out is None and out_c is None and len(wb_list) == 1 and k == 3 and stride == 1 and cout < 32 and res is None:
            # a Cout < 32 tile takes the Winograd kernel's element-wise epilogue (16 scalar stores per lane); padded to
            # a full 32-cout tile with zero filters it takes the vector one (the extra channels are written as
            # relu(0) = 0 into a 32-wide buffer and never read: the consumer's Cin stays cout)
            w0, b0 = wb_list[0]
            wp = np.zeros((32,) + w0.shape[1:], w0.dtype)
            wp[:cout] = w0
            bp = np.zeros(32, b0.dtype)
            bp[:cout] = b0
            wb_list, cout = [(wp, bp)], 32
        slices = 1
        if self.splitk and self.dt == DT_F32 and bias_map is None and len(wb_list) == 1:
            slices = splitk_slices(k, stride, cin, cout, 1, ho, wo, bias_buf is not None)
            if slices > 1 and conv_algo(k, stride, cin, cout, 1, ho, wo, bias_buf is not None, False) != 2:
                slices = 1
        if slices > 1:      # K-slices as the "groups" of the op: slice s = input channels [s*ks, (s+1)*ks), same Cout
            w0, b0 = wb_list[0]
            ks = cin // slices
            wb_list, cin = [(w0[:, s * ks:(s + 1) * ks], b0) for s in range(slices)], ks
        if out is None:
            out = self.buf(ho, wo, (out_c or (cout if slices > 1 else cout * len(wb_list))))
        if self.dt != DT_F32:
            # 16-bit program: every conv is the direct f16 / bf16 MFMA kernel (the matrix pipe is 16x faster than
            # in fp32 and the layers are HBM-bound: Winograd would only add arithmetic error).  A residual has the
            # type of the output (16-bit between layers, fp32 where a head map accumulates in place).
            assert self.dtype_of(src) == self.dt, 'conv input must be a 16-bit buffer in a 16-bit program'
            assert res is None or self.dtype_of(res) == self.dtype_of(out)
            algo = 0
            packed = [pack_conv_h16(w, b, self.dt) for (w, b) in wb_list]
            w_off = self.blob.add16(np.concatenate([p[0] for p in packed]))
        else:
            # (a stride-2 conv that hosts HR fuse terms keeps the polyphase kernel: conv_x3s2_kernel has no extra residual maps)
            algo = 2 if slices > 1 else conv_algo(k, stride, cin, cout, len(wb_list), ho, wo, bias_buf is not None, self.wino24,
                                                    False if ((terms or fp32_kernel) and stride == 2) else self.split16)
            if algo in (6, 7):
                packed = [pack_conv_x3(wb_list, DT_BF16 if algo == 7 else DT_F16)]      # (one power-of-two weight scale for the op: trailing float)
            elif algo == 3:
                packed = [pack_wino3(w, b) for (w, b) in wb_list]
            else:
                tr = (lambda t: t, winograd_weights, winograd2d_weights, None, winograd24_weights, polyphase2_weights)[algo]
                packed = [pack_conv(tr(w), b) for (w, b) in wb_list]
            w_off = self.blob.add(np.concatenate([p[0] for p in packed]))
        b_off = self.blob.add(np.concatenate([p[1] for p in packed]))
        flops = 2.0 * ho * wo * flop_cout * cin_w * k * k * (1 if slices > 1 else len(wb_list))     # algorithmic (direct-conv) FLOPs
        self._op(name, flops, kind=_lib.OP_CONV, in_buf=src, out_buf=out, res_buf=-1 if res is None else res,
                 in_coff=in_coff, out_coff=out_coff, res_coff=res_coff, cin=cin, cout=cout, ksize=k, stride=stride,
                 relu=int(relu), groups=len(wb_list), w_off=w_off, b_off=b_off, flags=algo,
                 bias_per_frame=0 if bias_buf is None else 1, aux_buf=-1 if bias_buf is None else bias_buf)
        if slices > 1:
            self.ops[-1].flags = algo | _lib.CONV_SPLITK
        if dual:
            out2, terms = dual
            assert algo == 3 and bias_buf is None and self.dims(out2)[:2] == (ho, wo) and out2 != out
            self.ops[-1].flags = algo | _lib.CONV_DUAL
            self.ops[-1].aux_buf = out2
        if terms:
            assert len(terms) <= 3 and self.dt == DT_F32 and len(wb_list) == 1
            assert dual or (k == 3 and stride == 2 and algo in (0, 5))
            op = self.ops[-1]
            op.nterms = len(terms)
            for t, (tb, sh) in enumerate(terms):
                th, tw, tcs = self.dims(tb)
                assert (th << sh, tw << sh) == (ho, wo) and tcs >= cout and tb != out
                op.term_buf[t], op.term_coff[t], op.term_shift[t] = tb, 0, sh
        if bias_map is not None:
            assert res is None and self.dt == DT_F32 and algo != 3
            assert bias_map.shape == (ho, wo, (cout * len(wb_list) + 3) // 4 * 4), bias_map.shape
            self.ops[-1].flags = algo | _lib.CONV_BIAS_MAP
            self.ops[-1].w_off2 = self.blob.add(bias_map)
        self.op_info[-1]['algo'] = ('direct', 'winograd_f23x', 'winograd_f2x2_3x3', 'winograd_f2x2_3x3_lds',
                                    'winograd_f2x4_3x3', 'polyphase_f2x2_s2', 'split_f16x3', 'split_bf16x3')[algo] + ('_splitk%d' % slices if slices > 1 else '')
        # what bench.py prints next to the PMC traffic: the kernel family launch_conv picks for this op (conv_mfma.hip /
        # conv_wino24b.inc wino24b_ok) and the op's ALGORITHMIC HBM bytes per frame - input slice + output (+ residual)
        # once, in their storage types
        ng = len(wb_list)
        esz = lambda b: 4 if self.dtype_of(b) == DT_F32 else 2
        fam = ('conv_ws2_kernel', 'conv_wino_kernel', 'conv_wino2_kernel', 'conv_wino3_kernel', 'conv_wino24_kernel',
               'conv_pp2_kernel', 'conv_x3_kernel', 'conv_x3_kernel')[algo]
        if algo == 4 and wino24b_width(cin, cout, ho, wo):
            fam = 'conv_wino24b_kernel'
        if algo in (6, 7) and k == 1:
            fam = 'conv_x3p_kernel'
        if algo in (6, 7) and stride == 2:
            fam = 'conv_x3s2_kernel'
        if (algo == 0 and k == 1 and stride == 1 and cin % 32 == 0 and cout % 32 == 0 and (ho * wo) % 256 == 0 and
                (WINOGRAD_24 if self.wino24 is None else self.wino24) and
                (ho * wo // 256) * 64 * ng * ((cout // 32) // (2 if cout % 64 == 0 else 1)) >= 512):
            # (batch 64 on 256 CUs: conv_mfma.hip launch_conv routes the layer to the streaming frame, conv_p1.inc, when its
            #  items fill the chip twice)
            fam = 'conv_p1_kernel'
        if self.dt != DT_F32:
            fam = 'conv_h16_kernel'
        self.op_info[-1]['kernel'] = fam
        self.op_info[-1]['bytes'] = float(h * w_ * cin * ng * esz(src) + ho * wo * (cout if slices > 1 else cout * ng) * esz(out) *
                                          (2 if res is not None else 1) +
                                          sum((ho >> sh) * (wo >> sh) * cout * 4 for (_, sh) in (terms or [])) +
                                          (ho * wo * cout * 4 if dual else 0))
        if self.keep_weights:    # folded fp64 filters per group, for oracle/program.py (tests only)
            self.op_info[-1]['wb'] = [(np.asarray(w, np.float64), np.asarray(b, np.float64)) for (w, b) in wb_list]
        return out

    def cp(self, c):
        """channel count of a backbone-internal map that logically holds c channels (pad_channels)"""
        return self.chan_pad.get(c, c)

    def fuse_epilogue_ok(self, cin, cout):
        """The HR-module fuse sum of an output resolution i >= 1 runs in the epilogue of the last convolution of its x0
        downsampling chain (3x3 stride 2, no ReLU of its own: acr/model.py:641-666) instead of as an OP_FUSESUM launch: fp32
        storage (the 16-bit storage programs' kernel has no such epilogue), Cin > 16, Cout a multiple of 32."""
        return FUSE_EPILOGUE and self.dt == DT_F32 and cin > 16 and cin % 8 == 0 and cout % 32 == 0

    def conv_bn(self, src, conv, bn, k, stride, relu, padded=False, **kw):
        """padded: a backbone-internal convolution - its filters are zero-padded to the padded channel counts of its input and
        output maps (self.chan_pad: HRNet-W48's 48-channel branch as 64 channels): the pad channels of every such map are
        relu(0 * x + 0) = 0 (or a sum of zeros), so the function is unchanged, and every layer of the branch fits the
        kernels that want Cin % 32 == 0 and Cout % 32 == 0 (the four-wave F(2x4,3x3) frame, the split-operand kernels, the
        streaming 1x1 frame) instead of the eight-wave fallback (measured at batch 64: 864 frames/s for the fp16x3 program
        of HRNet-W48 with its 67 branch-0 launches on conv_wino24_kernel)."""
        w, b = self.folded(conv, bn)
        if padded and (self.cp(w.shape[0]) != w.shape[0] or self.cp(w.shape[1]) != w.shape[1]):
            co, ci = w.shape[:2]
            wp = np.zeros((self.cp(co), self.cp(ci)) + w.shape[2:])
            wp[:co, :ci] = w
            bp = np.zeros(self.cp(co))
            bp[:co] = b
            out = self.conv(conv, src, [(wp, bp)], k, stride, relu, **kw)
            self.op_info[-1]['flops'] *= float(co * ci) / float(wp.shape[0] * wp.shape[1])      # the algorithmic figure
            return out
        return self.conv(conv, src, [(w, b)], k, stride, relu, **kw)

    def fuse_sum(self, name, terms, c, relu, out=None, out_coff=0):
        """terms: [(buf, shift)]"""
        h, w_, _ = self.dims(terms[0][0])
        assert terms[0][1] == 0
        if out is None:
            out = self.buf(h, w_, c)
        op = self._op(name, 0.0, kind=_lib.OP_FUSESUM, out_buf=out, out_coff=out_coff, cout=c, relu=int(relu),
                      nterms=len(terms))
        for t, (b, sh) in enumerate(terms):
            op.term_buf[t], op.term_coff[t], op.term_shift[t] = b, 0, sh
        return out

    # ---- blocks ------------------------------------------------------------------------------
    def basic_block(self, x, p, dual=None):
        """acr/model.py:483-499; consumes x (released), returns the new buffer.  dual: see conv() - conv2 also writes the
        HR fuse sum that starts with this block's output."""
        t = self.conv_bn(x, p + '.conv1', p + '.bn1', 3, 1, True, padded=True)
        kw = dict(dual=dual) if dual else {}
        y = self.conv_bn(t, p + '.conv2', p + '.bn2', 3, 1, True, res=x, padded=True, **kw)
        self.release(t, x)
        return y

    def fuse0_ok(self, c, h, w):
        """The FULL-resolution HR fuse sum as the second output of branch 0's last conv2 (ACRMI_CONV_DUAL): where that conv runs
        on conv_wino3_kernel (algo 3: fp32, 32 channels, with its store waves) - in LARGE-batch programs only.  The host conv
        has to wait for the other branches' 1x1 projections, so x0 - which the three downsampling chains read - appears later
        than it would from a plain conv2: free where launches fill the chip (batch 64: 1962 -> 1999 frames/s), but on the
        dependency chain of a single-frame call (batch 1: 2.69 -> 2.90 ms with it, measured) - small-batch programs keep the
        OP_FUSESUM launch for i = 0."""
        return (FUSE_EPILOGUE and FUSE_FULLRES and self.dt == DT_F32 and not self.splitk and
                conv_algo(3, 1, self.cp(c), self.cp(c), 1, h, w, False, self.wino24, self.split16) == 3)

    def bottleneck(self, x, p, cat=None, stride=1):
        """acr/model.py:519-539.  stride 2 (ResNet-50 only, torchvision's layout): on the 3x3 conv and on the 1x1
        projection shortcut.  cat: x is channels [0, Cin) of the 2 Cin-channel buffer `cat` (a block with a projection
        shortcut): the 3x3 conv writes its output next to x and the block's last conv + projection become ONE 1x1
        convolution over the concatenated channels, y = relu([W_ds | W_3] [x ; t2] + b_ds + b_3) - the projected
        shortcut (Cout channels written, then read back as a residual: 2 x 1.07 GB at batch 64 for layer1.0) never
        exists."""
        cin = self.sd[p + '.conv1.weight'].shape[1]
        t1 = self.conv_bn(x, p + '.conv1', p + '.bn1', 1, 1, True, cin=cin)
        if cat is not None:
            assert cat == x and stride == 1 and (p + '.downsample.0.weight') in self.sd
            mid = self.sd[p + '.conv2.weight'].shape[0]
            assert mid == cin and self.dims(cat)[2] >= 2 * cin
            self.conv_bn(t1, p + '.conv2', p + '.bn2', 3, 1, True, out=cat, out_coff=cin)
            self.release(t1)
            (wd, bd), (w3, b3) = self.folded(p + '.downsample.0', p + '.downsample.1'), self.folded(p + '.conv3', p + '.bn3')
            y = self.conv(p + '.conv3+downsample', cat, [(np.concatenate([wd, w3], 1), bd + b3)], 1, 1, True)
            self.release(cat)
            return y
        t2 = self.conv_bn(t1, p + '.conv2', p + '.bn2', 3, stride, True)
        self.release(t1)
        if (p + '.downsample.0.weight') in self.sd:
            res = self.conv_bn(x, p + '.downsample.0', p + '.downsample.1', 1, stride, False)
            self.release(x)
        else:
            res = x
        y = self.conv_bn(t2, p + '.conv3', p + '.bn3', 1, 1, True, res=res)
        self.release(t2, res)
        return y

    def bottleneck_chain(self, x, prefix, n, cat=None):
        """n stride-1 Bottlenecks `prefix.{i}` in a row (HRNet's layer1, acr/model.py:738-752; ResNet-50's layer1).  With
        self.pairs and 64 -> 256 -> 64 shapes, block i's conv3 (+ residual, ReLU) and block i + 1's conv1 (+ ReLU) are one
        OP_PAIR1X1 launch; otherwise bottleneck() per block (cat: see there)."""
        shapes_ok = all(self.sd['%s.%d.conv3.weight' % (prefix, i)].shape[:2] == (256, 64) and
                        self.sd['%s.%d.conv1.weight' % (prefix, i)].shape[0] == 64 for i in range(n))
        if not (self.pairs and shapes_ok and n >= 2):
            for i in range(n):
                x = self.bottleneck(x, '%s.%d' % (prefix, i), cat=cat if i == 0 else None)
            return x
        t1 = None
        for i in range(n):
            p = '%s.%d' % (prefix, i)
            if t1 is None:
                t1 = self.conv_bn(x, p + '.conv1', p + '.bn1', 1, 1, True, cin=self.sd[p + '.conv1.weight'].shape[1])
            t2 = self.conv_bn(t1, p + '.conv2', p + '.bn2', 3, 1, True)
            self.release(t1)
            if (p + '.downsample.0.weight') in self.sd:
                res = self.conv_bn(x, p + '.downsample.0', p + '.downsample.1', 1, 1, False,
                                   cin=self.sd[p + '.downsample.0.weight'].shape[1])
                self.release(x)
            else:
                res = x
            if i == n - 1:
                y = self.conv_bn(t2, p + '.conv3', p + '.bn3', 1, 1, True, res=res)
                self.release(t2, res)
                return y
            q = '%s.%d' % (prefix, i + 1)
            (w3, b3), (w1, b1) = self.folded(p + '.conv3', p + '.bn3'), self.folded(q + '.conv1', q + '.bn1')
            h, w_, _ = self.dims(t2)
            y, t1 = self.buf(h, w_, 256), self.buf(h, w_, 64)
            self._op(p + '.conv3+' + q + '.conv1', 2.0 * h * w_ * (256 * 64 + 64 * 256), kind=_lib.OP_PAIR1X1, in_buf=t2, res_buf=res,
                     out_buf=y, aux_buf=t1, cin=64, cout=256, ksize=1, stride=1, relu=1, groups=1,
                     w_off=self.blob.add(pack_pair1x1(w3[:, :, 0, 0], b3, w1[:, :, 0, 0], b1)))
            self.op_info[-1]['algo'] = 'pair1x1'
            if self.keep_weights:
                self.op_info[-1]['wb'] = [(np.asarray(w3, np.float64), np.asarray(b3, np.float64)),
                                          (np.asarray(w1, np.float64), np.asarray(b1, np.float64))]
            self.release(t2, res)
            x = y

    def hr_module(self, xs, p, ch, multi_scale=True, final_out=None):
        """acr/model.py:668-686.  xs consumed; returns the fused outputs.  Sum order = the reference's (j ascending):
        y_i = (((t_i0 + t_i1) + t_i2) + t_i3), ReLU.  For i >= 1 the first term is the x0 downsampling chain's output: where
        fuse_epilogue_ok, that chain's last convolution takes the other terms as extra residuals and the ReLU, and writes
        y_i itself - bit for bit what OP_FUSESUM computes from the stored conv output, one launch and one round trip of
        the map through HBM less."""
        nb = len(xs)
        xs = list(xs)
        h0, w0, _ = self.dims(xs[0])
        fold0 = nb > 1 and self.fuse0_ok(ch[0], h0, w0)
        for i in range(nb):
            for k in range(3 if (i == 0 and fold0) else 4):
                xs[i] = self.basic_block(xs[i], '%s.branches.%d.%d' % (p, i, k))
        outs = []
        temps = []

        def term(i, j, hosted=None):
            """fuse_layers[i][j] applied to xs[j] -> (buffer, shift); hosted: the extra terms the LAST conv of a j < i chain
            adds in its epilogue (it then writes the fused map, ReLU included)"""
            f = '%s.fuse_layers.%d.%d' % (p, i, j)
            if j == i:
                return xs[j], 0
            if j > i:
                t = self.conv_bn(xs[j], f + '.0', f + '.1', 1, 1, False, padded=True)
                temps.append(t)
                return t, j - i
            t = xs[j]
            for k in range(i - j):
                last = k == i - j - 1
                kw = dict(terms=hosted) if (last and hosted) else {}
                if last and j == 0:
                    kw['fp32_kernel'] = True      # the x0 chain's last conv: the fuse host's kernel in every lowering (Program.conv)
                t2 = self.conv_bn(t, '%s.%d.0' % (f, k), '%s.%d.1' % (f, k), 3, 2, (not last) or bool(hosted), padded=True, **kw)
                if t is not xs[j]:
                    self.release(t)
                t = t2
            if not hosted:
                temps.append(t)
            return t, 0

        for i in range(nb if multi_scale else 1):
            f0 = '%s.fuse_layers.%d.0' % (p, i)
            if i == 0 and fold0:
                # branch 0's last block: its conv2 writes x0 (the downsampling chains read it) AND, from the store waves,
                # y0 = relu(((x0 + up(t01)) + up(t02)) + up(t03)) - the 1x1-projected lower branches are computed first
                others = [term(0, j) for j in range(1, nb)]
                y0 = final_out if final_out is not None else self.buf(h0, w0, self.cp(ch[0]))
                xs[0] = self.basic_block(xs[0], '%s.branches.0.3' % p, dual=(y0, others))
                self.op_info[-1]['name'] += '+fuse0'
                outs.append(y0)
                continue
            if i >= 1 and final_out is None:
                wl = self.sd['%s.%d.0.weight' % (f0, i - 1)]      # the chain's last conv: [C_i, C_0, 3, 3]
                if self.fuse_epilogue_ok(self.cp(wl.shape[1]), self.cp(wl.shape[0])):
                    others = [term(i, j) for j in range(1, nb)]
                    y, _ = term(i, 0, hosted=others)
                    self.op_info[-1]['name'] += '+fuse%d' % i
                    outs.append(y)
                    continue
            terms = [term(i, j) for j in range(nb)]
            # fuse_sum wants the full-resolution term first for geometry; keep the reference's sum order
            # (j ascending) by letting the kernel take per-term shifts.
            if terms[0][1] != 0:
                raise AssertionError('term 0 must be at output resolution')
            if final_out is not None:      # the backbone's output map keeps its logical channel count (the heads read c0)
                outs.append(self.fuse_sum(p + '.fuse%d' % i, terms, ch[i], True, out=final_out))
            else:
                outs.append(self.fuse_sum(p + '.fuse%d' % i, terms, self.cp(ch[i]), True))
        self.release(*temps)
        self.release(*xs)
        return outs


def point_tower(P, side, k):
    """Head tower (side, k) (acr/model.py:288-313) in the layout of tower_point_kernel (csrc/kernels.h TP_*):
    entry W [tap][cin/4][cout][4] (cin 34 -> 36) + b, 4 x (3x3 W [tap][16][64][4] + b), exit W [64][112] + b[112]."""
    def rows(w, cpad):
        co, ci = w.shape[:2]
        wp = np.zeros((co, cpad, 3, 3))
        wp[:, :ci] = w
        return wp.reshape(co, cpad // 4, 4, 9).transpose(3, 1, 0, 2).reshape(-1)
    pre = '%s_final_layers.%d' % (side, k)
    w, b = P.folded(pre + '.0.0', pre + '.0.1')
    parts = [rows(w, 36), b]
    for blk in range(2):
        for c in (1, 2):
            w, b = P.folded(pre + '.1.%d.0.conv%d' % (blk, c), pre + '.1.%d.0.bn%d' % (blk, c))
            parts += [rows(w, 64), b]
    w, b = P.folded(pre + '.2')
    we = np.zeros((64, 112))
    we[:, :w.shape[0]] = w[:, :, 0, 0].T
    be = np.zeros(112)
    be[:w.shape[0]] = b
    out = np.concatenate([np.asarray(x, np.float64).reshape(-1) for x in parts + [we, be]])
    assert out.size == 9 * 9 * 64 * 4 + 64 + 4 * (9 * 16 * 64 * 4 + 64) + 64 * 112 + 112
    return out


def lower(sd, check=True, point_heads=True, keep_taps=False, precision='fp32', keep_weights=False, keep_all=False,
          wino24=None, splitk=False, pairs=None):
    """state dict -> dict(blob, bufs, ops, heads, op_info, taps, precision, width).  See module docstring.
    point_heads: also emit the MODE_POINT variant of the head program (ops tagged MODE_DENSE / MODE_POINT; fp32 W32 only).
    keep_taps: pin the buffers of the backbone taps the golden vectors hold (stem / layer1 / stage2 / stage3 branch 0,
    tests/golden/make_golden.py) so a test can read them after the run; costs ~0.6 GB at batch 64, off by default.
    precision: 'fp32' (the reference's configs/demo.yml), or 'fp16' / 'bf16' = the reference's autocast branch
    (acr/model.py:33-37, --model_precision fp16) re-stated for gfx950: activations between layers are stored in 16
    bits (NHWC, channel stride a multiple of 8), weights are the BN-folded filters rounded once to 16 bits, every
    conv accumulates in fp32 on v_mfma_f32_32x32x16_{f16,bf16} and applies bias / residual / ReLU in fp32 before the one
    rounding of its output; the stem conv (K = 27, reads uint8), the exits of the head towers (center, params x mix,
    prior, segm logits), attention pooling, pare bias, decode and MANO stay fp32 (the reference's .float() at
    acr/model.py:56-62).  The HRNet width (32 / 48) is read off the checkpoint.
    keep_weights: op_info[i]['wb'] keeps the folded fp64 filters of every conv (oracle/program.py, tests only).
    wino24: None = packer.WINOGRAD_24 (on); False lowers the 3x3 stride-1 layers with Cin > 32 to F(2x2,3x3) - what
    Engine.load_state_dict asks for when max_batch < 16 (single-frame / small-batch latency: batch 1 3.5 vs 3.7 ms).
    pairs: None = FUSE_PAIRS for large-batch programs (not splitk): layer1's conv3 / next conv1 pairs as one launch.
    splitk: small-batch program (Engine.load_state_dict: max_batch < 16) - the low-resolution 3x3 layers are lowered
    with their input channels in slices that run as separate work items (ACRMI_CONV_SPLITK, splitk_slices()).
    keep_all: no buffer is reused, so every intermediate map can be read after a run (per-op parity tests; ~2x the
    activation memory)."""
    sd = strip_prefix(sd)
    if check:
        check_state_dict(sd)
    if precision not in PRECISIONS:
        raise ValueError('precision %r: one of %s' % (precision, sorted(PRECISIONS)))
    width = width_of({k: _np(v) for k, v in sd.items() if k in ('backbone.transition1.0.0.weight', 'backbone.layer4.0.conv1.weight')})
    STAGE_CFG = None if width == RESNET50 else stage_cfg(width)
    c0 = backbone_channels(width)
    dt = PRECISIONS[precision]
    point_heads = point_heads and dt == DT_F32 and width == 32     # (the point-heads kernels are fp32, 34-channel)
    P = Program(sd, dt, keep_weights, keep_all, wino24, splitk, FUSE_PAIRS and not splitk if pairs is None else pairs,
                split16={'fp16x3': 'fp16', 'bf16x3': 'bf16'}.get(precision, False))
    b = 'backbone.'
    taps = {}
    x34 = None
    if width == RESNET50:
        # ---- ResNet-50 (BASELINE.json configs[1]; build-defined, schema._resnet50_backbone / oracle resnet50_backbone) ----
        w, bb = P.folded(b + 'conv1', b + 'bn1')
        x = P.buf(256, 256, 64)
        wp, bp = pack_stem7(w, bb)
        P._op(b + 'conv1', 2.0 * 256 * 256 * 64 * 3 * 49, kind=_lib.OP_STEM, out_buf=x, cin=3, cout=64, ksize=7, stride=2,
              relu=1, groups=1, w_off=P.blob.add(wp), b_off=P.blob.add(bp))
        P.op_info[-1]['algo'] = 'stem7_u8'
        if keep_weights:
            P.op_info[-1]['wb'] = [(np.asarray(w, np.float64), np.asarray(bb, np.float64))]
        if keep_taps:
            taps['stem'] = P.pin(x)
        # layer1.0 has a stride-1 projection shortcut: the pooled map goes into channels 0..63 of a 128-channel buffer and
        # the block takes conv3 + downsample as one convolution (Program.bottleneck cat=)
        cat = P.buf(128, 128, 128) if FUSE_PROJECTION else None
        pooled = cat if cat is not None else P.buf(128, 128, 64)
        P._op(b + 'maxpool', 0.0, kind=_lib.OP_MAXPOOL, in_buf=x, out_buf=pooled, cin=64)
        P.release(x)
        x = pooled
        for li, (planes, blocks, stride) in enumerate(RESNET50_LAYERS):
            for i in range(blocks):
                x = P.bottleneck(x, b + 'layer%d.%d' % (li + 1, i), cat=cat if (li == 0 and i == 0) else None,
                                 stride=stride if i == 0 else 1)
        if keep_taps:
            taps['layer4'] = P.pin(x)
        x34 = P.buf(128, 128, c0 + 2, persistent=True)      # backbone output (64) + coord maps (2), acr/model.py:52
        P._op('coordfill', 0.0, kind=_lib.OP_COORDFILL, out_buf=x34, out_coff=c0)
        for k, c in enumerate(RESNET50_UP):
            h, w_, cs = P.dims(x)
            cin = P.sd[b + 'deconv_layers.%d.0.weight' % k].shape[1]
            up = P.buf(2 * h, 2 * w_, cin)
            P._op(b + 'deconv_layers.%d.bilinear2x' % k, 0.0, kind=_lib.OP_BILINEAR2X, in_buf=x, out_buf=up, cin=cin)
            P.release(x)
            x = P.conv_bn(up, b + 'deconv_layers.%d.0' % k, b + 'deconv_layers.%d.1' % k, 3, 1, True,
                          out=x34 if k == len(RESNET50_UP) - 1 else None)
            P.release(up)
    else:
        x34 = _lower_hrnet(P, sd, b, STAGE_CFG, c0, keep_taps, keep_weights, taps)
    return _lower_heads(P, sd, b, x34, c0, dt, point_heads, precision, width, taps)


def width_of_sd(sd):
    return width_of({k: v for k, v in sd.items() if k in ('backbone.transition1.0.0.weight', 'backbone.layer4.0.conv1.weight')})


def _lower_hrnet(P, sd, b, STAGE_CFG, c0, keep_taps, keep_weights, taps):
    """acr/model.py:785-865: stem, layer1, transitions, 8 HR modules -> the persistent [128,128,c0 + 2] map"""
    # ---- stem -----------------------------------------------------------------------------------
    w, bb = P.folded(b + 'conv1', b + 'bn1')
    if STEM_FUSED:
        x = P.buf(256, 256, 64)
        wp, bp = pack_stem(w, bb)
        P._op(b + 'conv1', 2.0 * 256 * 256 * 64 * 3 * 9, kind=_lib.OP_STEM, out_buf=x, cin=3, cout=64, ksize=3, stride=2,
              relu=1, groups=1, w_off=P.blob.add(wp), b_off=P.blob.add(bp))
        P.op_info[-1]['algo'] = 'stem_u8'
        if keep_weights:
            P.op_info[-1]['wb'] = [(np.asarray(w, np.float64), np.asarray(bb, np.float64))]
    else:
        x0 = P.buf(512, 512, 4)
        P._op('u8norm', 0.0, kind=_lib.OP_U8NORM, out_buf=x0)
        x = P.conv(b + 'conv1', x0, [(w, bb)], 3, 2, True, cin=3)
        P.op_info[-1]['flops'] = 2.0 * 256 * 256 * 64 * 3 * 9
        P.release(x0)
    # the stem's second conv writes channels 0..63 of a 128-channel map: layer1.0 puts its 3x3 output next to them and
    # takes conv3 and the projection shortcut as one 128 -> 256 convolution (Program.bottleneck)
    cat = P.buf(128, 128, 128) if (FUSE_PROJECTION and not P.pairs) else None
    x1 = P.conv_bn(x, b + 'conv2', b + 'bn2', 3, 2, True, out=cat)
    P.release(x)
    x = x1
    if keep_taps:
        taps['stem'] = P.pin(x)
    x = P.bottleneck_chain(x, b + 'layer1', 4, cat=cat)
    if keep_taps:
        taps['layer1'] = P.pin(x)
    # ---- stages ---------------------------------------------------------------------------------
    t = b + 'transition1'
    # HRNet-W48, large-batch fp32-storage lowering: the 48-channel branch runs as 64 channels (Program.conv_bn padded=)
    if width_of_sd(P.sd) == 48 and P.dt == DT_F32 and (WINOGRAD_24 if P.wino24 is None else P.wino24) and not P.splitk:
        P.chan_pad = {48: 64}
    xs = [P.conv_bn(x, t + '.0.0', t + '.0.1', 3, 1, True, padded=True), P.conv_bn(x, t + '.1.0.0', t + '.1.0.1', 3, 2, True)]
    P.release(x)
    x34 = P.buf(128, 128, c0 + 2, persistent=True)      # backbone output (32) + coord maps (2), acr/model.py:52
    P._op('coordfill', 0.0, kind=_lib.OP_COORDFILL, out_buf=x34, out_coff=c0)
    for s in (2, 3, 4):
        ch = STAGE_CFG[s]['channels']
        if s > 2:
            t = b + 'transition%d' % (s - 1)
            i = len(ch) - 1
            xs = xs + [P.conv_bn(xs[-1], t + '.%d.0.0' % i, t + '.%d.0.1' % i, 3, 2, True, padded=True)]
        nmod = STAGE_CFG[s]['modules']
        for m in range(nmod):
            last = (s == 4 and m == nmod - 1)
            xs = P.hr_module(xs, b + 'stage%d.%d' % (s, m), ch, multi_scale=not last,
                             final_out=x34 if last else None)
        if keep_taps and s < 4:
            taps['stage%d' % s] = P.pin(xs[0])
    return x34


def _lower_heads(P, sd, b, x34, c0, dt, point_heads, precision, width, taps):
    """acr/model.py:47-166, 374-463: segmentation head, the eight towers, the part branch - on any backbone's
    [128,128,c0 + 2] map"""
    # ---- part-segmentation head (acr/model.py:374-463) ------------------------------------------
    u = b + 'hand_segm.segm_head.upsampler.up1.conv.double_conv'
    g = b + 'hand_segm.segm_head.segm_net.double_conv'
    up = P.buf(256, 256, c0)
    P._op('segm.bilinear2x', 0.0, kind=_lib.OP_BILINEAR2X, in_buf=x34, out_buf=up, cin=c0)
    s1 = P.conv_bn(up, u + '.0', u + '.1', 3, 1, True)
    P.release(up)
    s2 = P.conv_bn(s1, u + '.3', u + '.4', 3, 1, True)
    P.release(s1)
    s3 = P.conv_bn(s2, g + '.0', g + '.1', 3, 1, True)
    P.release(s2)
    segm = P.buf(256, 256, 33, persistent=True, f32=True)
    P.conv(g + '.3', s3, [P.folded(g + '.3')], 3, 1, False, out=segm)
    P.release(s3)
    # ---- 8 head towers (acr/model.py:71-92, 288-313), batched as one 34->512 conv + grouped blocks --
    # The two center towers come first: the point-heads variant runs the same grouped convs with
    # groups=2 on channels 0..127 and evaluates the other six towers only at the decoded centers (OP_POINTHEADS).
    towers = [('l', 2), ('r', 2)] + [(side, k) for side in 'lr' for k in (1, 3, 4)]
    w_list = [P.folded('%s_final_layers.%d.0.0' % t, '%s_final_layers.%d.0.1' % t) for t in towers]
    wcat = np.concatenate([w for w, _ in w_list], 0)
    bcat = np.concatenate([bb for _, bb in w_list], 0)
    coord_map = COORD_BIAS_MAP and dt == DT_F32

    def head_conv(name, wb, stride, **kw):
        """3x3 conv over the c0 + 2 channel map; fp32 programs: over its c0 backbone channels + the position-bias map"""
        w, bb = wb
        if not coord_map:
            return P.conv(name, x34, [(w, bb)], 3, stride, True, cin=c0 + 2, **kw)
        out = P.conv(name, x34, [(w[:, :c0], bb)], 3, stride, True, cin=c0, bias_map=coord_bias_map(w[:, c0:], 128, stride), **kw)
        P.op_info[-1]['flops'] *= (c0 + 2.0) / c0          # the algorithmic figure keeps the coordinate channels
        return out

    n0 = len(P.ops)
    t0 = head_conv('towers.entry', (wcat, bcat), 2)
    P.set_mode(_lib.MODE_DENSE, n0)
    if point_heads:
        n0 = len(P.ops)
        head_conv('towers.entry.centers', (wcat[:128], bcat[:128]), 2, out=t0)
        P.set_mode(_lib.MODE_POINT, n0)
    for k in range(2):
        c1 = [P.folded('%s_final_layers.%d.1.%d.0.conv1' % (s_, t_, k), '%s_final_layers.%d.1.%d.0.bn1' % (s_, t_, k))
              for s_, t_ in towers]
        c2 = [P.folded('%s_final_layers.%d.1.%d.0.conv2' % (s_, t_, k), '%s_final_layers.%d.1.%d.0.bn2' % (s_, t_, k))
              for s_, t_ in towers]
        n0 = len(P.ops)
        t1 = P.conv('towers.block%d.conv1' % k, t0, c1, 3, 1, True)
        t2 = P.conv('towers.block%d.conv2' % k, t1, c2, 3, 1, True, res=t0)
        P.set_mode(_lib.MODE_DENSE, n0)
        if point_heads and P.split16:
            # split-operand programs: conv_x3_kernel finds the op's ONE power-of-two weight scale behind the fragments of ALL its
            # groups (a.w[groups * 9 * tap floats], pack_conv_x3) - a 2-group clone of the 8-group pack would read the "scale"
            # from inside group 2's halves.  The two center towers get their own packs (and their own scale).
            n1 = len(P.ops)
            P.conv('towers.block%d.conv1.centers' % k, t0, c1[:2], 3, 1, True, out=t1)
            P.conv('towers.block%d.conv2.centers' % k, t1, c2[:2], 3, 1, True, res=t0, out=t2)
            P.set_mode(_lib.MODE_POINT, n1)
        elif point_heads:
            for j in (n0, n0 + 1):
                P.clone_op(j, P.op_info[j]['name'] + '.centers', 2.0 / len(towers), groups=2, mode=_lib.MODE_POINT)
        P.release(t1, t0)
        t0 = t2
    heads = _lib.HeadLayout()

    def padded(w, b, n):
        """[Cout,...] filters zero-padded to n output channels: whole 32-cout tiles take the kernels' vector epilogue
        (the element-wise one costs 2x on these small layers); the extra channels land in the buffer's pad."""
        wp = np.zeros((n,) + w.shape[1:])
        wp[:w.shape[0]] = w
        bp = np.zeros(n)
        bp[:b.shape[0]] = b
        return wp, bp

    # Exits (acr/model.py:305-311) and the mix conv (:160-164).  Between the params exit and the mix conv there is
    # no non-linearity, so the dense program applies their product to the tower features directly:
    #   final = (W_mix[:, params] W_exit_p) t_p + W_mix[:, params] b_exit_p          "params_mix", at the exit stage
    #         + W_mix[:, cam] cam' + per-frame pare bias                             "cam_mix", once the bias exists
    # with cam' = the 3-channel cam exit after 1.1**x on channel 0 (:95-96).  The 106-channel intermediate map is never
    # written.  (The point-heads variant evaluates exit and mix per pixel and keeps its own weights.)
    p109, final, camb = {}, {}, {}
    mix_w = {}
    for si, (side, mix) in enumerate((('l', 4), ('r', 5))):
        wm = _np(sd['contact_layers.%d.weight' % mix]).astype(np.float64).reshape(109, 218)
        wa = wm[:, :109].copy()
        wa[:, :3] += wm[:, 109:112]                 # cam3 appears twice in the concat (acr/model.py:160-163)
        mix_w[side] = (wa, wm[:, 112:])
        p109[side] = P.buf(64, 64, 109, f32=True) if point_heads else -1   # point heads: raw exits of the sampled pixel
        final[side] = P.buf(64, 64, 128, persistent=True, f32=True)
        camb[side] = P.buf(64, 64, 32)
        center = P.buf(64, 64, 32, persistent=True, f32=True)
        prior = P.buf(64, 64, 128, persistent=True, f32=True)
        name = '%s_final_layers.%%d.2' % side
        tin = lambda k: 64 * towers.index((side, k))
        P.conv(name % 2, t0, [padded(*P.folded(name % 2), 32)], 1, 1, False, out=center, in_coff=tin(2), cin=64)
        P.op_info[-1]['flops'] = 2.0 * 64 * 64 * 64 * 1
        n0 = len(P.ops)
        we, be = P.folded(name % 1)
        wc = wa[:, 3:] @ we[:, :, 0, 0]              # [109,64]
        P.conv(side + '.params_mix', t0, [padded(wc[:, :, None, None], wa[:, 3:] @ be, 128)], 1, 1, False,
               out=final[side], in_coff=tin(1), cin=64)
        P.op_info[-1]['flops'] = 2.0 * 64 * 64 * 64 * 106
        P.conv(name % 3, t0, [padded(*P.folded(name % 3), 32)], 1, 1, False, out=camb[side], in_coff=tin(3), cin=64)
        P.op_info[-1]['flops'] = 2.0 * 64 * 64 * 64 * 3
        P._op('%s.cam_pow' % side, 0.0, kind=_lib.OP_POW11, out_buf=camb[side], out_coff=0)
        P.conv(name % 4, t0, [padded(*P.folded(name % 4), 128)], 1, 1, False, out=prior, in_coff=tin(4), cin=64)
        P.op_info[-1]['flops'] = 2.0 * 64 * 64 * 64 * 106
        P.set_mode(_lib.MODE_DENSE, n0)
        heads.center_buf[si], heads.prior_buf[si] = center, prior
    P.release(t0)
    # ---- part branch (acr/model.py:116-166) ---------------------------------------------------------
    # The 1x1 shape conv (cam_shape_layers.1.0, acr/model.py:132: no BN, no ReLU) is linear and a part's softmax
    # weights sum to 1, so it commutes with the attention pooling: pooled_shape = W_cs pooled_contact + b_cs.  It is
    # folded into the Linear that consumes it (cam_shape_layers.2/3) below; the 256->64 conv over the 128x128 map and
    # 64 of the 320 pooled channels disappear.
    feat = P.buf(128, 128, 256)
    head_conv('contact_layers.1.0', P.folded('contact_layers.1.0', 'contact_layers.1.1'), 1, out=feat)
    pooled = P.buf(1, 32, 256, f32=True)
    P._op('attpool', 2.0 * 32 * 16384 * 320 + 2.0 * 128 * 128 * 256 * 64, kind=_lib.OP_ATTPOOL, in_buf=segm, res_buf=feat,
          out_buf=pooled, cin=256)
    P.op_info[-1]['name'] = 'attpool(+cam_shape_layers.1.0)'
    P.release(feat)
    wcs, bcs = P.folded('cam_shape_layers.1.0')
    wcs = wcs[:, :, 0, 0]                                # [64, 256]
    for si, (side, lc, mix, part0) in enumerate((('l', 2, 4, 16), ('r', 3, 5, 0))):
        wa, wp = mix_w[side]
        bias_buf = P.buf(1, 1, 128, persistent=True, f32=True)     # one per side: acrmi_point_heads re-reads both
        lw = _np(sd['cam_shape_layers.%d.weight' % lc]).astype(np.float64).reshape(10, 64, 16)    # [k, c, j]
        lin_w = np.einsum('kcj,cd->kdj', lw, wcs).reshape(10, 256 * 16)                            # [k, c' * 16 + j]
        lin_b = _np(sd['cam_shape_layers.%d.bias' % lc]).astype(np.float64) + np.einsum('kcj,c->k', lw, bcs)
        P._op('%s.parebias' % side, 0.0, kind=_lib.OP_PAREBIAS, in_buf=pooled, out_buf=bias_buf, cin=256, flags=part0,
              w_off=P.blob.add(_np(sd['contact_layers.%d.weight' % lc]).reshape(6, 256, 16)),
              w_off2=P.blob.add(lin_w), b_off2=P.blob.add(lin_b),
              w_off3=P.blob.add(wp), b_off=P.blob.add(_np(sd['contact_layers.%d.bias' % mix])))
        n0 = len(P.ops)
        P.conv('contact_layers.%d' % mix, camb[side], [padded(wa[:, :3, None, None], np.zeros(109), 128)], 1, 1, False,
               out=final[side], cin=3, bias_buf=bias_buf, res=final[side])
        P.op_info[-1]['flops'] = 2.0 * 64 * 64 * 109 * 218
        P.set_mode(_lib.MODE_DENSE, n0)
        P.release(camb[side])
        if point_heads:
            mixw = np.zeros((109, 112))
            mixw[:, :109] = wa.T
            P._op('%s.point_heads' % side, 0.0, kind=_lib.OP_POINTHEADS, in_buf=x34, res_buf=p109[side], out_buf=final[side],
                  aux_buf=bias_buf, flags=si, mode=_lib.MODE_POINT,
                  w_off=P.blob.add(np.concatenate([point_tower(P, side, k) for k in (1, 3, 4)])),
                  w_off2=P.blob.add(mixw))
        heads.params_buf[si] = final[side]
    heads.segm_buf, heads.backbone_buf = segm, x34
    return {'blob': P.blob.finish(), 'bufs': P.bufs, 'ops': P.ops, 'heads': heads, 'op_info': P.op_info, 'taps': taps,
            'precision': precision, 'width': width}
