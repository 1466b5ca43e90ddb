"""MI355X-native ACR inference hot path (frames -> HRNet-W32 -> ACR heads -> decode -> MANO meshes).

The directory name is not a Python identifier; import it with
    importlib.import_module('arbitrary-hands-3d-reconstruction_amd')
Sub-modules: engine (C-ABI context), ops (stand-alone operators), packer/schema/synth (checkpoint
handling), parallel (multi-GPU sharding), acr.* / mano.* (the reference's Python API surface).
"""
__version__ = '0.1.0'
