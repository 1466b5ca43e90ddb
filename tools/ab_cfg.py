#!/usr/bin/env python
"""A/B of conv kernel variants INSIDE the network (GPU box only): per-op time of the LDS-Winograd layers, of all ops and
the batch-64 step time for each tuning configuration (acrmi_tune key 0 = conv_bench --cfg), interleaved twice.

    python tools/ab_cfg.py [cfg ...]        (default: -1 839)
"""
import importlib, sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = lambda m: importlib.import_module('arbitrary-hands-3d-reconstruction_amd.' + m)
synth = pkg('synth'); L = pkg('_lib')
eng = pkg('engine').Engine(0)
eng.load_state_dict(synth.make_state_dict(seed=0), max_batch=64)
eng.load_mano(synth.make_mano_tables(seed=1))
x = torch.from_numpy(synth.make_frames(64, seed=0, structured=True)).cuda()
cfgs = [int(v) for v in sys.argv[1:]] or [-1, 839]
for cfg in cfgs * 2:
    L.lib().acrmi_tune(0, cfg)
    eng.profile_ops(x)
    prof = eng.profile_ops(x)
    lds = [p for p in prof if p.get('algo') == 'winograd_f2x2_3x3_lds']
    for _ in range(2): eng.forward(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): eng.forward(x)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print('cfg %d: lds-winograd %d ops %.3f ms; all ops %.3f ms; step %.3f ms' % (cfg, len(lds), sum(p['ms'] for p in lds), sum(p['ms'] for p in prof), dt * 1e3))
