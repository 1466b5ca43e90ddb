"""GPU: parallel.ShardedRunner over RCCL between PROCESSES.  One process per GPU
(tests/multi_gpu_worker.py), both transports: gathered rows == rank 0's own recomputation bit for bit, padded shards,
pipelined submits.  World size 2 needs two GPUs and is skipped on a one-GPU box; the same worker at world size 1 runs on
every box, so the script itself is always exercised.  `pytest -m gpu`."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(world, transport, timeout=900):
    port = _free_port()
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')       # dmabuf IPC between the ranks' processes
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, 'tests', 'multi_gpu_worker.py'), '--rank', str(r), '--world',
                               str(world), '--port', str(port), '--transport', transport], cwd=ROOT, env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(world)]
    outs = []
    for p in procs:
        try:
            o, e = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append((p.returncode, o, e))
    return outs


def _check(outs, world, transport):
    for r, (rc, o, e) in enumerate(outs):
        assert rc == 0, 'rank %d (%s): rc %d\n%s\n%s' % (r, transport, rc, o[-3000:], e[-3000:])
        assert 'MULTI_GPU_OK' in o, o[-2000:]
    line = next(l for l in outs[0][1].splitlines() if l.startswith('MULTI_GPU_OK'))
    rep = json.loads(line.split(' ', 1)[1])
    assert rep['world'] == world and rep['transport'] == transport
    assert rep['ragged_ok'] and rep['even_ok'] and rep['pipelined_ok'] and rep['ragged_rows'] == 7
    assert rep['gather_ms'] > 0
    return rep


@pytest.mark.parametrize('transport', ['torch', 'c'])
def test_sharded_runner_worker_at_world_size_1(transport):
    """The multi-GPU worker with ONE rank (runs on every GPU box): the script, both transports, the ragged / even / pipelined
    cases against the rank's own recomputation."""
    _check(_run(1, transport), 1, transport)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason='needs two GPUs (one process per GPU over RCCL)')
@pytest.mark.parametrize('transport', ['torch', 'c'])
def test_sharded_runner_over_rccl_world_size_2(transport):
    """Two processes, two GPUs, RCCL over xGMI: every rank ends with every rank's rows in frame order, bit-equal to a
    single-process run - for a batch that does not divide by the world size (padded shards), one that does, and the
    pipelined weak-scaling loop bench.py --gpus N runs."""
    _check(_run(2, transport), 2, transport)
