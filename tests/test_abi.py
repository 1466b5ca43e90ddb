"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol
include/acrmi.h declares; the ctypes structs match the C layout; the product path refuses to run
without a GPU instead of falling back."""
import ctypes
import os
import re

import pytest
import torch

from conftest import ROOT, pkg


def _declared_symbols():
    src = open(os.path.join(ROOT, 'include', 'acrmi.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(acrmi_[a-z0-9_]+)\s*\(', src)))


def test_library_exports_every_declared_symbol():
    L = pkg('_lib')
    lib = L.lib()
    declared = _declared_symbols()
    assert len(declared) >= 31
    for name in declared:
        assert hasattr(lib, name), 'libacrmi.so does not export %s' % name
    assert sorted(L.EXPORTS) == declared
    assert lib.acrmi_version() == L.VERSION == 303
    for name in ('acrmi_allgather', 'acrmi_comm_init', 'acrmi_smooth', 'acrmi_set_option_f', 'acrmi_heads', 'acrmi_mano_rotmat', 'acrmi_prior_gate',
                 'acrmi_preprocess_frames'):      # SURVEY.md 8b list
        assert name in declared


def test_struct_layout_matches_header():
    L = pkg('_lib')
    assert ctypes.sizeof(L.Op) == 168 and L.Op.w_off.offset == 56 and L.Op.w_off2.offset == 136
    assert ctypes.sizeof(L.Frame) == 16 and L.Frame.H.offset == 8 and L.Frame.W.offset == 12        # acrmi_frame
    assert ctypes.sizeof(L.BufferDesc) == 20 and L.BufferDesc.dtype.offset == 16 and ctypes.sizeof(L.HeadLayout) == 32
    assert (L.DT_F32, L.DT_F16, L.DT_BF16) == (0, 1, 2)
    src = open(os.path.join(ROOT, 'include', 'acrmi.h')).read()
    for name, val in (('ACRMI_SLOT', L.SLOT), ('ACRMI_SLOT_POSES', L.SLOT_POSES), ('ACRMI_SLOT_BETAS', L.SLOT_BETAS),
                      ('ACRMI_SLOT_PARAMS', L.SLOT_PARAMS), ('ACRMI_SLOT_CAM', L.SLOT_CAM)):
        assert re.search(r'#define %s (\d+)' % name, src).group(1) == str(val)


@pytest.mark.skipif(torch.cuda.is_available(), reason='checks the no-GPU failure mode')
def test_no_cpu_fallback():
    L = pkg('_lib')
    with pytest.raises(L.AcrmiError):
        pkg('engine').Engine(0)
    with pytest.raises(L.AcrmiError):
        pkg('ops').conv2d(torch.zeros(1, 8, 8, 8), torch.zeros(8, 8, 1, 1))
    ctx = ctypes.c_void_p()
    assert L.lib().acrmi_create(ctypes.byref(ctx), 0) == L.E_HIP       # loud failure, not a silent CPU path
    assert b'HIP' in L.lib().acrmi_last_error(None)


def test_null_arguments_are_rejected_without_a_gpu():
    L = pkg('_lib')
    lib = L.lib()
    assert lib.acrmi_create(None, 0) == L.E_INVAL
    assert lib.acrmi_load_weights(None, None, 0) == L.E_INVAL
    assert lib.acrmi_decode(None, 1, None, None) == L.E_INVAL
    assert lib.acrmi_conv2d(None, 1, 8, 8, 8, 0, 8, None, None, 0, None, 0, 0, None, 8, 0, 8, 3, 1, 0, 1, 0, None) == L.E_INVAL
    assert lib.acrmi_conv2d_h16(None, 1, 8, 8, 8, 0, 8, None, None, 0, None, 0, 0, None, 8, 0, 8, 3, 1, 0, 1, 1, 0, None) == L.E_INVAL
    assert lib.acrmi_buffer_dtype(None, 0) == -1
    assert lib.acrmi_smooth(None, None, 1, None) == L.E_INVAL
    assert lib.acrmi_set_option_f(None, L.OPT_CONF_THRESH, 0.5) == L.E_INVAL
    assert lib.acrmi_allgather(None, None, None, None, 0, None) == L.E_INVAL
    assert lib.acrmi_comm_unique_id(None) == L.E_INVAL
    assert lib.acrmi_comm_init(None, 1, 0, None) == L.E_INVAL
    assert lib.acrmi_prior_gate(None, None, 2, None, None) == L.E_INVAL
    assert lib.acrmi_preprocess_frames(None, 1, None, None, None) == L.E_INVAL
    bad = (L.Frame * 1)()
    bad[0].H, bad[0].W = 4, 4                                         # a frame without a pointer
    assert lib.acrmi_preprocess_frames(bad, 1, ctypes.c_void_p(16), None, None) == L.E_INVAL
    assert lib.acrmi_decode_maps(None, None, 4, None, None, 112, None, None, 108, 1, 0.35, None, None) == L.E_INVAL
