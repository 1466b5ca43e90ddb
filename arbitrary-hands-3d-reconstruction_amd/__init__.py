"""MI355X-native ACR inference hot path (frames -> HRNet-W32 -> ACR heads -> decode -> MANO meshes).

The directory name is not a Python identifier; import it with
    importlib.import_module('arbitrary-hands-3d-reconstruction_amd')
Sub-modules: engine (C-ABI context), ops (stand-alone operators), packer/schema/synth (checkpoint
handling), parallel (multi-GPU sharding), acr.* / mano.* (the reference's Python API surface).
"""
import os as _os

# ACRMI_OPT_LANES runs the program's independent chains on parallel HIP streams.  ROCm multiplexes all streams of a
# process onto GPU_MAX_HW_QUEUES hardware queues (default 4); once RCCL and torch's stream pools exist a lane can
# land on the caller's own queue and the lanes serialise again (measured: 1345 vs 1409 frames/s at batch 64 with an
# RCCL process group).  The variable is read when the HIP runtime initialises, so import this package (or export
# the variable) before the first GPU call.
_os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

__version__ = '0.1.0'
