#!/usr/bin/env python
"""Runs bench.config3_video_stream alone (GPU box): configs[3]'s per-GPU shard from pinned host memory."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

synth = bench.pkg('synth')
sd = synth.make_state_dict(seed=0)
tables = synth.make_mano_tables(seed=1)
tables['left']['shapedirs'] = tables['left']['shapedirs'].copy()
tables['left']['shapedirs'][:, 0, :] *= -1
print(json.dumps(bench.config3_video_stream(sd, tables, int(os.environ.get('STEPS', '10')), 3, 0), indent=1))
