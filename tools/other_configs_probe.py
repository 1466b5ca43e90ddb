#!/usr/bin/env python
"""bench.py's `other_configs` block alone (configs[1] / configs[4] workloads, split-operand entries + 16-bit storage ablation)."""
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

if __name__ == '__main__':
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    synth = importlib.import_module(bench.PKG + '.synth')
    tables = synth.make_mano_tables(seed=1)
    tables['left']['shapedirs'] = tables['left']['shapedirs'].copy()
    tables['left']['shapedirs'][:, 0, :] *= -1
    print(json.dumps(bench.other_configs(tables, steps, 2, 0), indent=1))
