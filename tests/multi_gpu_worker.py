"""One rank of the multi-GPU ShardedRunner check (tests/test_gpu_multi.py starts `world` of these, one per GPU):

    python tests/multi_gpu_worker.py --rank R --world N --port P --transport torch|c

Every rank: Engine on GPU R with the synthetic checkpoint, ShardedRunner over RCCL (transport 'torch': a torch.distributed NCCL
group carries the data; transport 'c': the library's own communicator, the process group is gloo and only carries the unique
id), then
  * forward_global on 7 frames (does not divide by the world size: padded shards, padding rows dropped),
  * forward_global on 2 * N frames (divides),
  * the pipelined weak-scaling form (submit, submit, collect, collect) on per-rank frames.
Rank 0 recomputes every frame on its own GPU, one call per shard-sized chunk, and requires the gathered rows to be BIT-EQUAL
(frames are independent; the same kernels run on the same GPU model).  Prints `MULTI_GPU_OK <json>` on success."""
import argparse
import importlib
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = 'arbitrary-hands-3d-reconstruction_amd'


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rank', type=int, required=True)
    ap.add_argument('--world', type=int, required=True)
    ap.add_argument('--port', type=int, required=True)
    ap.add_argument('--transport', default='torch')
    a = ap.parse_args()
    for p in (ROOT, os.path.join(ROOT, 'tests')):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(a.port), RANK=str(a.rank), WORLD_SIZE=str(a.world))
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    import torch.distributed as dist
    torch.cuda.set_device(a.rank)
    if a.transport == 'c':
        dist.init_process_group('gloo', rank=a.rank, world_size=a.world)
    else:
        dist.init_process_group('nccl', rank=a.rank, world_size=a.world, device_id=torch.device('cuda', a.rank))
    synth = importlib.import_module(PKG + '.synth')
    parallel = importlib.import_module(PKG + '.parallel')
    engine = importlib.import_module(PKG + '.engine')
    ok, report = True, {}
    try:
        per = 4
        eng = engine.Engine(a.rank)
        eng.load_state_dict(synth.make_state_dict(seed=0), max_batch=max(per, (7 + a.world - 1) // a.world))
        tables = synth.make_mano_tables(seed=1)
        tables['left']['shapedirs'] = tables['left']['shapedirs'].copy()
        tables['left']['shapedirs'][:, 0, :] *= -1
        eng.load_mano(tables)
        dev = eng.device
        runner = parallel.ShardedRunner(lambda f, v: eng.forward(f, out=v), dev, engine=eng, transport=a.transport)

        def reference(frames, chunk):
            outs = [eng.forward(frames[i:i + chunk].contiguous()) for i in range(0, frames.shape[0], chunk)]
            torch.cuda.synchronize()
            return {k: torch.cat([o[k] for o in outs], 0) for k in ('slots', 'verts', 'joints')}

        def same(got, want):
            return all(got[k].shape == want[k].shape and torch.equal(got[k], want[k]) for k in ('slots', 'verts', 'joints'))

        ragged = torch.from_numpy(synth.make_frames(7, seed=5, structured=True)).to(dev)
        got = runner.forward_global(ragged)
        torch.cuda.synchronize()
        report['ragged_rows'] = int(got['slots'].shape[0])
        even = torch.from_numpy(synth.make_frames(2 * a.world, seed=6, structured=True)).to(dev)
        got_even = runner.forward_global(even)
        torch.cuda.synchronize()
        # weak scaling, pipelined: rank r's own frames (seed 20 + r), two batches in flight
        mine = torch.from_numpy(synth.make_frames(per, seed=20 + a.rank, structured=True)).to(dev)
        t0 = runner.submit(mine)
        t1 = runner.submit(mine.flip(0).contiguous())
        r0 = {k: v.clone() for k, v in runner.collect(t0).items()}
        r1 = {k: v.clone() for k, v in runner.collect(t1).items()}
        torch.cuda.synchronize()
        gather_ms = runner.time_gather(per)
        if a.rank == 0:
            chunk7 = (7 + a.world - 1) // a.world
            ok = ok and same(got, reference(ragged, chunk7)) and report['ragged_rows'] == 7
            report['ragged_ok'] = ok
            e_ok = same(got_even, reference(even, 2))
            report['even_ok'] = e_ok
            ok = ok and e_ok
            allf = torch.cat([torch.from_numpy(synth.make_frames(per, seed=20 + r, structured=True)) for r in range(a.world)]).to(dev)
            want = reference(allf, per)
            flipped = torch.cat([allf[r * per:(r + 1) * per].flip(0) for r in range(a.world)])
            p_ok = same(r0, want) and same(r1, reference(flipped, per))
            report['pipelined_ok'] = p_ok
            ok = ok and p_ok
            report['gather_ms'] = round(gather_ms, 4)
        runner.close()
        eng.close()
    except Exception as exc:      # noqa: BLE001 - reported, then the rank exits non-zero
        import traceback
        ok = False
        report['error'] = repr(exc)
        report['trace'] = traceback.format_exc(limit=6)
    flag = torch.tensor([1 if ok else 0])
    try:
        if a.world > 1:      # a rank that failed fails every rank
            fl = flag if a.transport == 'c' else flag.cuda()
            dist.all_reduce(fl, op=dist.ReduceOp.MIN)
            flag = fl.cpu()
        dist.barrier()
        dist.destroy_process_group()
    except Exception as exc:      # noqa: BLE001
        report['teardown'] = repr(exc)
    report.update(rank=a.rank, world=a.world, transport=a.transport)
    if ok and int(flag.item()) == 1:
        print('MULTI_GPU_OK ' + json.dumps(report), flush=True)
        return 0
    print('MULTI_GPU_FAILED ' + json.dumps(report), flush=True)
    return 1


if __name__ == '__main__':
    sys.exit(main())
