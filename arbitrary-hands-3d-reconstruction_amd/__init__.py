"""MI355X-native ACR inference hot path (frames -> HRNet-W32 -> ACR heads -> decode -> MANO meshes).

The directory name is not a Python identifier; import it with
    importlib.import_module('arbitrary-hands-3d-reconstruction_amd')
Sub-modules: engine (C-ABI context), ops (stand-alone operators), packer/schema/synth (checkpoint
handling), parallel (multi-GPU sharding), acr.* / mano.* (the reference's Python API surface).
"""
import os as _os

# ACRMI_OPT_LANES runs the program's independent chains on parallel HIP streams.  ROCm multiplexes all streams of a
# process onto GPU_MAX_HW_QUEUES hardware queues (default 4).  Rounds 2-5 raised it to 8 here (then: 1345 vs 1409 frames/s at
# batch 64 with two lanes per context next to an RCCL process group).  Round 6 measured the opposite on today's programs
# (tools/hwq_probe.sh, tools/tune_probe.py; profiles/r06_hwq_probe.txt): with 8 or more queues a process that keeps MORE than
# four streams busy collapses (a batch-1 call on 5-8 lanes: 6-7.7 ms against 2.45-2.66 ms on the default's four queues; the
# C-ABI transport's side-stream all-gather next to one context: 1789 against 1946 frames/s), and nothing measured gains from
# 8 (headline 2010 vs 2014, batch-1 call 2.57 vs 2.44 ms).  So the runtime's default stays; export the variable before the
# first GPU call to override it.
_HWQ_NOTE = 'GPU_MAX_HW_QUEUES is left at the runtime default (4) since round 6'

__version__ = '0.1.0'
