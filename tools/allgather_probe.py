#!/usr/bin/env python
"""acrmi_allgather at world size 1 (GPU box): time of the collective alone and next to a running batch."""
import ctypes as C
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

L = bench.pkg('_lib')
synth, parallel, engine = bench.pkg('synth'), bench.pkg('parallel'), bench.pkg('engine')
if os.environ.get('PG') == '1':      # a torch.distributed process group (RCCL) alive next to the library's own communicator
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29533', RANK='0', WORLD_SIZE='1')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    dist.init_process_group('nccl', device_id=torch.device('cuda', 0))
    if os.environ.get('PG_BARRIER', '1') == '1':
        dist.barrier()
    print('torch.distributed (nccl) initialised', flush=True)
eng = engine.Engine(0)
uid = (C.c_char * 128)()
L.check(L.lib().acrmi_comm_unique_id(uid))
eng.comm_init(1, 0, bytes(uid), allow_second_communicator=True)      # (PG=1 measures exactly that hazard)
B = 64
flat, views = parallel.alloc_result(B, eng.device)
gathered = torch.empty_like(flat)
raw = C.c_void_p()
L.check(L.lib().acrmi_stream_create(0, C.byref(raw)))
side = torch.cuda.ExternalStream(raw.value, device=eng.device)
for name, st in (('current stream', torch.cuda.current_stream()), ('side stream', side)):
    for _ in range(3):
        eng.allgather(flat, gathered, stream=st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record(st)
    for _ in range(20):
        eng.allgather(flat, gathered, stream=st)
    e1.record(st)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print('%s: %.3f ms per all-gather on the device, %.3f ms host time per call (%.2f MB)' % (
        name, e0.elapsed_time(e1) / 20, (t1 - t0) / 20 * 1e3, flat.numel() * 4 / 1e6), flush=True)
# next to the network
sd = synth.make_state_dict(seed=0)
tables = synth.make_mano_tables(seed=1)
eng.load_state_dict(sd, max_batch=B)
eng.load_mano(tables)
eng.set_lanes(1)
frames = torch.from_numpy(synth.make_frames(B, seed=0, structured=False)).cuda()
for with_gather in (False, True, False, True):
    for _ in range(3):
        eng.forward(frames, out=views)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        eng.forward(frames, out=views)
        if with_gather:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            side.wait_event(ev)
            eng.allgather(flat, gathered, stream=side)
    torch.cuda.synchronize()
    print('forward%s: %.3f ms per batch' % (' + side-stream all-gather' if with_gather else '', (time.perf_counter() - t0) / 10 * 1e3), flush=True)
