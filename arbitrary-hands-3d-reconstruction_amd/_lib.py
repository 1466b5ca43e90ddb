"""ctypes binding of libacrmi.so (include/acrmi.h).  No CPU fallback: if the HIP library is
missing or fails to load, every entry point raises."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('ACRMI_LIB') or os.path.join(HERE, 'libacrmi.so')   # ACRMI_LIB: kernel experiments

OP_U8NORM, OP_CONV, OP_FUSESUM, OP_BILINEAR2X, OP_POW11, OP_ATTPOOL, OP_PAREBIAS, OP_COORDFILL, OP_POINTHEADS, OP_STEM = range(1, 11)
OP_MAXPOOL, OP_PAIR1X1 = 11, 12
MODE_BOTH, MODE_DENSE, MODE_POINT = 0, 1, 2
CONV_BIAS_MAP = 8      # acrmi_op.flags of a CONV / acrmi_conv2d's algo: ACRMI_CONV_BIAS_MAP
CONV_SPLITK = 16       # acrmi_op.flags of a CONV: ACRMI_CONV_SPLITK
CONV_DUAL = 32         # acrmi_op.flags of a CONV: ACRMI_CONV_DUAL (second output = the full-resolution HR fuse sum)
OPT_POINT_HEADS, OPT_LANES, OPT_CENTER_IDX, OPT_TEMPORAL, OPT_CONF_THRESH, OPT_SMOOTH_COEFF, OPT_MANO_FP16 = 1, 2, 3, 4, 5, 6, 7
OPT_LANE_PLAN = 8
OPT_BATCH_PRIOR = 9
VERSION = 303
DT_F32, DT_F16, DT_BF16 = 0, 1, 2
SLOT = 176
SLOT_FLAG, SLOT_FLATIND, SLOT_SCORE, SLOT_CAM, SLOT_POSES, SLOT_BETAS, SLOT_PARAMS = 0, 1, 2, 3, 6, 54, 64
E_INVAL, E_HIP, E_STATE, E_NOMEM, E_RANGE = -1, -2, -3, -4, -5


class BufferDesc(C.Structure):
    _fields_ = [('h', C.c_int32), ('w', C.c_int32), ('cs', C.c_int32), ('persistent', C.c_int32), ('dtype', C.c_int32)]


class Op(C.Structure):
    _fields_ = [('kind', C.c_int32),
                ('in_buf', C.c_int32), ('out_buf', C.c_int32), ('res_buf', C.c_int32),
                ('in_coff', C.c_int32), ('out_coff', C.c_int32), ('res_coff', C.c_int32),
                ('cin', C.c_int32), ('cout', C.c_int32),
                ('ksize', C.c_int32), ('stride', C.c_int32), ('relu', C.c_int32), ('groups', C.c_int32),
                ('w_off', C.c_int64), ('b_off', C.c_int64),
                ('bias_per_frame', C.c_int32), ('aux_buf', C.c_int32), ('nterms', C.c_int32),
                ('term_buf', C.c_int32 * 4), ('term_coff', C.c_int32 * 4), ('term_shift', C.c_int32 * 4),
                ('w_off2', C.c_int64), ('b_off2', C.c_int64), ('w_off3', C.c_int64),
                ('flags', C.c_int32), ('mode', C.c_int32)]


class Frame(C.Structure):      # acrmi_frame
    _fields_ = [('bgr_dev', C.c_void_p), ('H', C.c_int32), ('W', C.c_int32)]


class HeadLayout(C.Structure):
    _fields_ = [('center_buf', C.c_int32 * 2), ('params_buf', C.c_int32 * 2), ('prior_buf', C.c_int32 * 2),
                ('segm_buf', C.c_int32), ('backbone_buf', C.c_int32)]


EXPORTS = ['acrmi_version', 'acrmi_last_error', 'acrmi_create', 'acrmi_destroy', 'acrmi_load_weights',
           'acrmi_set_program', 'acrmi_load_mano', 'acrmi_backbone_heads', 'acrmi_buffer_ptr', 'acrmi_decode',
           'acrmi_decode_maps', 'acrmi_mano', 'acrmi_forward', 'acrmi_conv2d', 'acrmi_u8norm', 'acrmi_bilinear2x',
           'acrmi_fuse_sum', 'acrmi_attpool', 'acrmi_attpool_ws_floats', 'acrmi_stem_conv', 'acrmi_stream_create', 'acrmi_stream_destroy', 'acrmi_profile_ops', 'acrmi_tune', 'acrmi_preprocess', 'acrmi_cam_trans',
           'acrmi_set_option', 'acrmi_point_heads', 'acrmi_set_option_f', 'acrmi_smooth', 'acrmi_smooth_reset',
           'acrmi_comm_unique_id', 'acrmi_comm_init', 'acrmi_comm_destroy', 'acrmi_allgather', 'acrmi_parebias',
           'acrmi_buffer_dtype', 'acrmi_conv2d_h16', 'acrmi_conv2d_splitk', 'acrmi_conv2d_splitk_workspace',
           'acrmi_decode_gated', 'acrmi_decode_maps_gated', 'acrmi_share_weights', 'acrmi_mano_rotmat', 'acrmi_heads',
           'acrmi_backbone_channels', 'acrmi_check_range', 'acrmi_prior_gate', 'acrmi_preprocess_frames']

_lib = None


class AcrmiError(RuntimeError):
    pass


class AcrmiRangeError(AcrmiError, OverflowError):
    """ACRMI_ERANGE: an 'fp16x3' program met an activation outside the f16 range (acrmi_check_range); the results since the
    last check are invalid (and were written as NaN)."""


def lib():
    """Loads libacrmi.so (built in-tree by build.py).  Raises if it is absent: there is no fallback path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise AcrmiError('libacrmi.so not built: run `python __graft_entry__.py build` (needs hipcc). '
                         'The ACR path has no CPU fallback.')
    L = C.CDLL(LIB_PATH)
    vp, i32, f32p, u8p = C.c_void_p, C.c_int, C.c_void_p, C.c_void_p
    L.acrmi_version.restype = C.c_int
    if L.acrmi_version() != VERSION:
        # the argument lists below are those of VERSION: a stale build (or an ACRMI_LIB override from another round)
        # would be called with shifted arguments
        raise AcrmiError('%s is ABI version %d, this package binds version %d: rebuild it (`python __graft_entry__.py build`)'
                         % (LIB_PATH, L.acrmi_version(), VERSION))
    L.acrmi_last_error.restype = C.c_char_p
    L.acrmi_last_error.argtypes = [vp]
    L.acrmi_create.argtypes = [C.POINTER(vp), i32]
    L.acrmi_destroy.argtypes = [vp]
    L.acrmi_destroy.restype = None
    L.acrmi_load_weights.argtypes = [vp, vp, C.c_size_t]
    L.acrmi_set_program.argtypes = [vp, C.POINTER(BufferDesc), i32, C.POINTER(Op), i32, C.POINTER(HeadLayout), i32]
    L.acrmi_load_mano.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp]
    L.acrmi_backbone_heads.argtypes = [vp, u8p, i32, vp]
    L.acrmi_buffer_ptr.argtypes = [vp, i32, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]
    L.acrmi_buffer_ptr.restype = vp
    L.acrmi_decode.argtypes = [vp, i32, f32p, vp]
    L.acrmi_decode_maps.argtypes = [f32p, f32p, i32, f32p, f32p, i32, f32p, f32p, i32, i32, C.c_float, f32p, vp]
    L.acrmi_decode_gated.argtypes = [vp, i32, vp, f32p, vp]
    L.acrmi_share_weights.argtypes = [vp, vp]
    L.acrmi_decode_maps_gated.argtypes = [f32p, f32p, i32, f32p, f32p, i32, f32p, f32p, i32, i32, C.c_float, vp, f32p, vp]
    L.acrmi_mano.argtypes = [vp, f32p, i32, f32p, i32, vp, i32, i32, f32p, f32p, f32p, f32p, i32, f32p, f32p, f32p,
                             f32p, vp]
    L.acrmi_mano_rotmat.argtypes = [vp, f32p, f32p, i32, vp, i32, i32, f32p, f32p, f32p, vp]
    L.acrmi_heads.argtypes = [vp, f32p, i32, vp]
    L.acrmi_backbone_channels.argtypes = [vp]
    L.acrmi_check_range.argtypes = [vp, vp]
    L.acrmi_prior_gate.argtypes = [vp, f32p, i32, vp, vp]
    L.acrmi_preprocess_frames.argtypes = [C.POINTER(Frame), i32, vp, vp, vp]
    L.acrmi_forward.argtypes = [vp, u8p, i32, f32p, f32p, f32p, f32p, f32p, f32p, f32p, vp]
    L.acrmi_conv2d.argtypes = [f32p, i32, i32, i32, i32, i32, i32, f32p, f32p, i32, f32p, i32, i32, f32p, i32, i32,
                               i32, i32, i32, i32, i32, i32, vp]
    L.acrmi_conv2d_h16.argtypes = [vp, i32, i32, i32, i32, i32, i32, vp, f32p, i32, vp, i32, i32, vp, i32, i32,
                                   i32, i32, i32, i32, i32, i32, i32, vp]
    L.acrmi_conv2d_splitk.argtypes = [f32p, i32, i32, i32, i32, i32, i32, i32, f32p, f32p, f32p, i32, i32, f32p, i32, i32,
                                      i32, i32, vp, C.c_size_t, vp]
    L.acrmi_conv2d_splitk_workspace.argtypes = [i32, i32, i32, i32, i32]
    L.acrmi_conv2d_splitk_workspace.restype = C.c_size_t
    L.acrmi_buffer_dtype.argtypes = [vp, i32]
    L.acrmi_u8norm.argtypes = [u8p, i32, f32p, vp]
    L.acrmi_bilinear2x.argtypes = [f32p, i32, i32, i32, i32, i32, f32p, i32, vp]
    L.acrmi_fuse_sum.argtypes = [i32, C.POINTER(vp), C.POINTER(i32), C.POINTER(i32), i32, i32, i32, i32, f32p, i32,
                                 i32, vp]
    L.acrmi_attpool.argtypes = [f32p, i32, f32p, i32, i32, i32, f32p, f32p, vp]
    L.acrmi_attpool_ws_floats.argtypes = [i32, i32]
    L.acrmi_stream_create.argtypes = [i32, C.POINTER(C.c_void_p)]
    L.acrmi_stream_destroy.argtypes = [vp]
    L.acrmi_stem_conv.argtypes = [vp, i32, i32, i32, f32p, f32p, f32p, i32, i32, i32, vp]
    L.acrmi_attpool_ws_floats.restype = C.c_size_t
    L.acrmi_profile_ops.argtypes = [vp, u8p, i32, vp, i32, vp]
    L.acrmi_tune.argtypes = [i32, i32]
    L.acrmi_preprocess.argtypes = [u8p, i32, i32, i32, u8p, vp, vp]
    L.acrmi_cam_trans.argtypes = [f32p, f32p, i32, C.c_float, C.c_float, f32p, vp]
    L.acrmi_set_option.argtypes = [vp, i32, i32]
    L.acrmi_point_heads.argtypes = [vp, i32, vp]
    L.acrmi_set_option_f.argtypes = [vp, i32, C.c_float]
    L.acrmi_smooth.argtypes = [vp, f32p, i32, vp]
    L.acrmi_smooth_reset.argtypes = [vp, vp]
    L.acrmi_comm_unique_id.argtypes = [vp]
    L.acrmi_comm_init.argtypes = [vp, i32, i32, vp]
    L.acrmi_comm_destroy.argtypes = [vp]
    L.acrmi_allgather.argtypes = [vp, vp, f32p, f32p, C.c_size_t, vp]
    L.acrmi_parebias.argtypes = [f32p, i32, i32, f32p, f32p, f32p, f32p, f32p, i32, f32p, i32, vp]
    for name in EXPORTS:
        fn = getattr(L, name)
        if name not in ('acrmi_last_error', 'acrmi_destroy', 'acrmi_buffer_ptr', 'acrmi_attpool_ws_floats'):
            fn.restype = C.c_int
    _lib = L
    return L


def check(rc, ctx=None):
    """Maps the C error codes onto the exceptions the reference raises (ValueError for bad
    configuration/arguments, RuntimeError otherwise)."""
    if rc >= 0:
        return rc
    msg = lib().acrmi_last_error(ctx)
    msg = msg.decode() if msg else 'acrmi error %d' % rc
    if rc == E_INVAL:
        raise ValueError(msg)
    if rc == E_RANGE:
        raise AcrmiRangeError(msg)
    raise AcrmiError(msg)
