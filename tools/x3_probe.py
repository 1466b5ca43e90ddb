#!/usr/bin/env python
"""The split-operand 16-bit program ('fp16x3': fp32 storage, conv_x3_kernel where it applies) next to the fp32 program
(GPU box): frames/s of one context at batch 64 and the largest vertex / joint distance between the two on the same frames."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

L = bench.pkg('_lib')
synth, parallel, engine = bench.pkg('synth'), bench.pkg('parallel'), bench.pkg('engine')
B = int(os.environ.get('B', '64'))
sd = synth.make_state_dict(seed=0)
tables = synth.make_mano_tables(seed=1)
frames = torch.from_numpy(synth.make_frames(B, seed=0, structured=False)).cuda()
res = {}
for prec in ('fp32', 'fp16x3', 'bf16x3'):
    eng = engine.Engine(0)
    eng.load_state_dict(sd, max_batch=B, precision=prec)
    eng.load_mano(tables)
    eng.set_lanes(1)
    flat, views = parallel.alloc_result(B, eng.device)
    for _ in range(3):
        eng.forward(frames, out=views)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        eng.forward(frames, out=views)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    algos = {}
    for i in eng.program['op_info']:
        algos[i.get('algo')] = algos.get(i.get('algo'), 0) + 1
    print('%s: %.3f ms per batch of %d = %.1f frames/s (one context); ops by algo: %s' % (prec, dt * 1e3, B, B / dt, algos), flush=True)
    fam = {}
    for pr in eng.profile_ops(frames):
        if pr.get('mode', 0) == L.MODE_POINT:
            continue
        info = eng.program['op_info'][pr['idx']]
        key = info.get('kernel') or info.get('algo') or str(pr['kind'])
        f = fam.setdefault(key, [0, 0.0])
        f[0] += 1
        f[1] += pr['ms']
    print('   ' + ', '.join('%s x%d %.2f ms' % (k_, v_[0], v_[1]) for k_, v_ in sorted(fam.items(), key=lambda kv: -kv[1][1])), flush=True)
    res[prec] = {k: v.cpu().numpy().copy() for k, v in views.items()}
    eng.close() if hasattr(eng, 'close') else None
for other in ('fp16x3', 'bf16x3'):
    a, b = res['fp32'], res[other]
    flag_a, flag_b = a['slots'][..., L.SLOT_FLAG] > 0.5, b['slots'][..., L.SLOT_FLAG] > 0.5
    same = (flag_a == flag_b) & (~flag_a | (a['slots'][..., L.SLOT_FLATIND] == b['slots'][..., L.SLOT_FLATIND]))
    use = same & flag_a
    dv = np.linalg.norm(a['verts'] - b['verts'], axis=-1)[use]
    dj = np.linalg.norm(a['joints'] - b['joints'], axis=-1)[use]
    print('%s vs fp32 program: decisions differing %d, hands compared %d, max vertex distance %.3e m, max joint distance %.3e m' % (
        other, int((~same).sum()), int(use.sum()), dv.max() if dv.size else -1, dj.max() if dj.size else -1))
