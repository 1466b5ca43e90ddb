#!/usr/bin/env python
"""Engine.tune_lanes at batch 1 / 8 with every lane count 1..8: call time per (lanes, planned)."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
pkg = bench.pkg
synth = pkg('synth')
eng = pkg('engine').Engine(0)
eng.load_state_dict(synth.make_state_dict(seed=0), max_batch=8)
eng.load_mano(synth.make_mano_tables(seed=1))
for b in (1, 8):
    best, ms = eng.tune_lanes(b, candidates=(1, 2, 3, 4, 5, 6, 8), calls=10)
    print('batch', b, 'best', best, {('%d%s' % (k[0], 'p' if k[1] else 's')): round(v, 3) for k, v in sorted(ms.items())}, flush=True)
