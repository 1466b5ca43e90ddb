"""Multi-process (world_size 2, gloo, CPU) test of the frame-sharding + single all-gather path
(parallel.py).  The per-rank compute is the CPU oracle standing in for Engine.forward; what is under test
is sharding, the packed result buffer, the collective and the global frame order."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import cases
from conftest import PKG, ROOT, pkg

NAMES = ['both_near', 'left_only', 'right_only', 'both_far', 'none', 'edge_corner']


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _oracle_forward(tables):
    from oracle import decode as odec, mano as omano
    L = pkg('_lib')

    def fwd(frames, views):
        ids = [int(f[0, 0, 0]) for f in frames]
        maps = {k: torch.cat([torch.from_numpy(cases.decode_maps(NAMES[i])[k]) for i in ids])
                for k in cases.decode_maps(NAMES[0])}
        s = odec.decode(maps)
        sl = views['slots']
        sl.zero_()
        sl[:, :, L.SLOT_FLAG] = torch.from_numpy(s['flag'].astype(np.float32))
        sl[:, :, L.SLOT_FLATIND] = torch.from_numpy(s['flat_ind'].astype(np.float32))
        sl[:, :, L.SLOT_POSES:L.SLOT_POSES + 48] = torch.from_numpy(s['poses'])
        sl[:, :, L.SLOT_BETAS:L.SLOT_BETAS + 10] = torch.from_numpy(s['betas'])
        sl[:, :, L.SLOT_PARAMS:L.SLOT_PARAMS + 109] = torch.from_numpy(s['params_pred'])
        for h, name in ((0, 'left'), (1, 'right')):
            v, j, _ = omano.mano_forward(tables[name], name, s['poses'][:, h], s['betas'][:, h])
            views['verts'][:, h] = torch.from_numpy(v)
            views['joints'][:, h] = torch.from_numpy(j)
    return fwd


def _frames(n):
    f = torch.zeros(n, 4, 4, 3, dtype=torch.uint8)
    for i in range(n):
        f[i, 0, 0, 0] = i
    return f


def _worker(rank, world, port, q):
    import sys
    for p in (ROOT, os.path.join(ROOT, 'tests', 'golden'), os.path.join(ROOT, 'tests')):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import importlib
        parallel = importlib.import_module(PKG + '.parallel')
        synth = importlib.import_module(PKG + '.synth')
        tables = synth.make_mano_tables(seed=1)
        runner = parallel.ShardedRunner(_oracle_forward(tables), torch.device('cpu'))
        frames = _frames(len(NAMES))
        got = runner.forward_global(frames)               # strong-scaling entry: shard of the global batch
        lo, hi = parallel.shard_range(len(NAMES), rank, world)
        got2 = runner.forward_local(frames[lo:hi])         # weak-scaling entry gives the same thing
        # pipelined form: batch k's gather is queued while batch k+1 is computed (double-buffered results)
        t0 = runner.submit(frames[lo:hi])
        t1 = runner.submit(frames[lo:hi].flip(0))
        try:
            runner.submit(frames[lo:hi])
            third = False
        except RuntimeError:
            third = True                                   # a third outstanding ticket would overwrite batch k
        r0, r1 = runner.collect(t0), runner.collect(t1)
        per = hi - lo
        flipped = torch.cat([got['slots'][r * per:(r + 1) * per].flip(0) for r in range(world)])
        same = all(torch.equal(got[k], got2[k]) and torch.equal(got[k], r0[k]) for k in got)
        # (the stand-in's torch-CPU math differs by ~1e-7 with a frame's position in the batch: allclose, not equal)
        same = same and third and torch.allclose(r1['slots'], flipped, rtol=0, atol=2e-6)
        # a global batch that does NOT divide by the world size (5 frames on 2 ranks: 3 + 2, rank 1 pads one frame): the
        # padding rows are dropped after the gather, the result is the 5 frames in order
        odd = runner.forward_global(frames[:5])
        same = same and all(v.shape[0] == 5 for v in odd.values())
        same = same and all(torch.allclose(odd[k], got[k][:5], rtol=0, atol=2e-6) for k in got)
        one = runner.forward_global(frames[:1])            # fewer frames than ranks: rank 1's shard is empty
        same = same and all(v.shape[0] == 1 and torch.allclose(v, got[k][:1], rtol=0, atol=2e-6) for k, v in one.items())
        q.put((rank, {k: v.numpy() for k, v in got.items()}, same))
    finally:
        dist.destroy_process_group()


def test_sharded_forward_equals_single_process(mano_tables):
    parallel = pkg('parallel')
    flat, views = parallel.alloc_result(len(NAMES), torch.device('cpu'))
    _oracle_forward(mano_tables)(_frames(len(NAMES)), views)
    want = {k: v.numpy().copy() for k, v in views.items()}

    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, got, same in res:
        assert same
        for k in ('slots', 'verts', 'joints'):
            assert got[k].shape == want[k].shape
            np.testing.assert_array_equal(got[k], want[k])     # every rank holds the full, ordered result


class _CpuPool(object):
    """engine.EnginePool's surface (submit / release / collect / len) over the CPU stand-in forward: what bench.py's
    N > 1 step loop drives through parallel.ShardedRunner."""

    def __init__(self, fwd, n=2):
        self.fwd, self.n, self.busy, self.turn, self.submits = fwd, n, [None] * n, 0, 0

    def __len__(self):
        return self.n

    def submit(self, frames, out=None):
        i = self.turn
        assert self.busy[i] is None, 'ticket of %d submits ago not collected / released' % self.n
        self.turn = (i + 1) % self.n
        self.fwd(frames, out)
        self.submits += 1
        t = {'slot': i, 'out': out}
        self.busy[i] = t
        return t

    def release(self, t):
        self.busy[t['slot']] = None
        return None            # (a GPU pool returns the batch's event)

    def collect(self, t):
        self.release(t)
        return t['out']


def _bench_worker(rank, world, port, q):
    import sys
    for p in (ROOT, os.path.join(ROOT, 'tests', 'golden'), os.path.join(ROOT, 'tests')):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import importlib
        bench = importlib.import_module('bench')
        parallel = importlib.import_module(PKG + '.parallel')
        synth = importlib.import_module(PKG + '.synth')
        tables = synth.make_mano_tables(seed=1)
        pool = _CpuPool(_oracle_forward(tables), n=2)
        # exactly bench.py's wiring for N > 1 with a pool (bench.py main(): `local`, `runner`, npipe = min(npipe, 2))
        local = lambda f, v: pool.release(pool.submit(f, out=v))
        runner = parallel.ShardedRunner(local, torch.device('cpu'))
        lo, hi = parallel.shard_range(len(NAMES), rank, world)
        frames = _frames(len(NAMES))[lo:hi]
        last = bench.run_pipelined(5, frames, pool=pool, runner=runner)
        ok = pool.submits == 5 and not runner._pending and all(b is None for b in pool.busy)
        # the single-context loop and the pool loop of the same function (N = 1 forms), on this rank's shard
        flat, views = parallel.alloc_result(hi - lo, torch.device('cpu'))

        class _Eng(object):
            def forward(self, f, out=None):
                _oracle_forward(tables)(f, out)
                return out
        one = bench.run_pipelined(2, frames, eng=_Eng(), vsets=[views])
        vs = [parallel.alloc_result(hi - lo, torch.device('cpu'))[1] for _ in range(2)]
        two = bench.run_pipelined(3, frames, pool=_CpuPool(_oracle_forward(tables), n=2), vsets=vs)
        ok = ok and all(torch.equal(one[k], two[k]) for k in ('slots', 'verts', 'joints'))
        q.put((rank, {k: v.numpy() for k, v in last.items()}, ok))
    finally:
        dist.destroy_process_group()


def test_bench_step_loop_over_gloo(mano_tables):
    """the N > 1 code path of bench.py itself - run_pipelined with a ShardedRunner over a pool, one
    ticket left outstanding, results double-buffered - executed at world size 2 (gloo, CPU stand-ins) before hardware
    sees it: it terminates, every ticket is collected, and every rank ends with the full ordered result."""
    parallel = pkg('parallel')
    flat, views = parallel.alloc_result(len(NAMES), torch.device('cpu'))
    _oracle_forward(mano_tables)(_frames(len(NAMES)), views)
    want = {k: v.numpy().copy() for k, v in views.items()}
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bench_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, got, ok in res:
        assert ok
        for k in ('slots', 'verts', 'joints'):
            np.testing.assert_allclose(got[k], want[k], rtol=0, atol=2e-6)


def test_padded_shard_covers_every_frame_once():
    parallel = pkg('parallel')
    for n in (1, 2, 5, 7, 64, 65, 127):
        for world in (1, 2, 3, 4, 8):
            rows = []
            for r in range(world):
                lo, hi, per = parallel.padded_shard(n, r, world)
                assert per == -(-n // world) and 0 <= hi - lo <= per
                rows += list(range(lo, hi))
            assert rows == list(range(n)), (n, world)


def test_shard_range_and_buffer_layout():
    parallel = pkg('parallel')
    assert parallel.shard_range(512, 3, 8) == (192, 256)
    with pytest.raises(ValueError):
        parallel.shard_range(10, 0, 4)
    flat, v = parallel.alloc_result(3, torch.device('cpu'))
    assert flat.numel() == 3 * 2 * parallel.PER_HAND == 3 * 2 * (176 + 778 * 3 + 21 * 3)
    assert v['slots'].shape == (3, 2, 176) and v['verts'].shape == (3, 2, 778, 3) and v['joints'].shape == (3, 2, 21, 3)
    v['joints'].fill_(7.0)
    assert flat[-1] == 7.0 and v['verts'].is_contiguous()


def _run_bench(argv, env_drop=('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')):
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in env_drop}
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + argv, env=env, cwd=ROOT, timeout=600,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    return p, (json.loads(lines[-1]) if lines and lines[-1].lstrip().startswith('{') else None)


def test_bench_gpus2_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it spawns its two ranks itself
    (torch.distributed.run on 127.0.0.1), runs bench.py's N > 1 loop (ShardedRunner + run_pipelined + barriers + MAX over
    ranks) - here with the CPU stand-in engine over gloo - and rank 0's JSON line is the LAST line of stdout, rc 0."""
    p, line = _run_bench(['--gpus', '2', '--steps', '3', '--warmup', '1', '--batch', '2', '--standin'])
    assert p.returncode == 0, p.stderr[-2000:]
    assert line is not None, p.stdout[-2000:]
    assert line['n_gpus'] == 2 and line['steps'] == 3 and line['warmup'] == 1
    assert line['config']['global_batch'] == 4 and line['gathered_rows_ok'] is True
    assert 'STAND-IN' in line['metric']      # can never be mistaken for a measurement
    # the self-diagnosis of an N > 1 line: every rank's own step time, the gather alone, the ranks whose
    # rows arrived and were verified against rank 0's recomputation
    m = line['multi_gpu']
    assert m['ranks'] == m['ranks_seen'] == m['ranks_verified'] == 2
    assert len(m['per_rank_ms_per_step']) == 2 and all(v > 0 for v in m['per_rank_ms_per_step'])
    assert m['gather_ms'] > 0 and m['gather_bytes_per_rank'] == 2 * 2 * (176 + 778 * 3 + 21 * 3) * 4
    # one rank, same entry point
    p1, line1 = _run_bench(['--gpus', '1', '--steps', '2', '--warmup', '1', '--batch', '2', '--standin'])
    assert p1.returncode == 0 and line1['n_gpus'] == 1


def test_bench_refuses_a_launcher_of_the_wrong_size():
    """--gpus N under a launcher that started M != N ranks is an error, not a silently different job."""
    import subprocess
    import sys
    env = dict(os.environ, WORLD_SIZE='1', RANK='0', LOCAL_RANK='0')
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--standin'], env=env, cwd=ROOT,
                       timeout=300, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert p.returncode != 0 and 'must agree' in p.stderr
