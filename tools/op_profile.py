#!/usr/bin/env python
"""Per-op times of the dense program at a given batch (GPU box only): acrmi_profile_ops brackets every op with HIP
events on one stream.

    python tools/op_profile.py [--batch 64] [--top 30] [--json gpurun_out/ops.json]
"""
import argparse
import importlib
import json
import os
import sys
from collections import defaultdict

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = lambda m: importlib.import_module('arbitrary-hands-3d-reconstruction_amd.' + m)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--top', type=int, default=30)
    ap.add_argument('--json', default='')
    args = ap.parse_args()
    synth = pkg('synth')
    eng = pkg('engine').Engine(0)
    eng.load_state_dict(synth.make_state_dict(seed=0), max_batch=args.batch)
    eng.load_mano(synth.make_mano_tables(seed=1))
    x = torch.from_numpy(synth.make_frames(args.batch, seed=0, structured=True)).cuda()
    eng.profile_ops(x)
    prof = eng.profile_ops(x)
    tot = sum(p['ms'] for p in prof)
    print('batch %d: %d ops, %.3f ms summed' % (args.batch, len(prof), tot))
    by = defaultdict(lambda: [0, 0.0])
    for p in prof:
        key = '%s k%d s%d' % (p.get('algo') or p['name'].split('.')[0], p['ksize'], p['stride'])
        by[key][0] += 1
        by[key][1] += p['ms']
    for k, (n, ms) in sorted(by.items(), key=lambda kv: -kv[1][1]):
        print('  %-34s %4d ops %8.3f ms  (%.1f us each)' % (k, n, ms, 1e3 * ms / n))
    for p in sorted(prof, key=lambda p: -p['ms'])[:args.top]:
        print('%4d %-64s %-22s %.4f ms' % (p['idx'], p['name'], p.get('algo'), p['ms']))
    if args.json:
        with open(args.json, 'w') as f:
            json.dump(prof, f)


if __name__ == '__main__':
    main()
