"""Lane plan probe (GPU box): Engine.tune_lanes at batch 1 / 2 / 8 - ms per network pass for every (lanes, structural |
planned) candidate.  ACRMI_PLAN_WAIT_US overrides the planner's cross-stream charge (default 16)."""
import importlib, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench   # noqa: F401
pkg = lambda m: importlib.import_module('arbitrary-hands-3d-reconstruction_amd.' + m)
synth = pkg('synth')
eng = pkg('engine').Engine(0)
eng.load_state_dict(synth.make_state_dict(seed=0), max_batch=8)
eng.load_mano(synth.make_mano_tables(seed=1))
cands = tuple(int(c) for c in (sys.argv[1].split(',') if len(sys.argv) > 1 else '1,2,3,4'.split(',')))
for b in (1, 2, 8):
    best, ms = eng.tune_lanes(b, candidates=cands, calls=10)
    print('batch', b, 'best', best, ' '.join('%d%s=%.3f' % (n, 'p' if p else 's', v) for (n, p), v in sorted(ms.items())))
