#!/usr/bin/env python
"""Does acrmi_allgather at world size 1 BLOCK the host until the stream reaches it?  A long batch is queued on a stream, then
the all-gather behind it on the same stream; the host time of the all-gather call is printed next to the batch's GPU time."""
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
eng = engine.Engine(0)
uid = (C.c_char * 128)()
L.check(L.lib().acrmi_comm_unique_id(uid))
eng.comm_init(1, 0, bytes(uid))
B = 64
eng.load_state_dict(synth.make_state_dict(seed=0), max_batch=B)
eng.load_mano(synth.make_mano_tables(seed=1))
x = torch.from_numpy(synth.make_frames(B, seed=0, structured=False)).cuda()
flat, views = parallel.alloc_result(B, eng.device)
gathered = torch.empty_like(flat)
for _ in range(2):
    eng.forward(x, out=views)
torch.cuda.synchronize()
for trial in range(3):
    t0 = time.perf_counter()
    eng.forward(x, out=views)                  # ~33 ms of GPU work, enqueued in ~3 ms
    t1 = time.perf_counter()
    eng.allgather(flat, gathered)              # same stream, behind the batch
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    print('enqueue batch %.2f ms | allgather call %.2f ms | until done %.2f ms' % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3), flush=True)
