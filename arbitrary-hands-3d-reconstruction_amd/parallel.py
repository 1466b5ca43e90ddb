"""Multi-GPU: one process per GPU, frames sharded contiguously by rank, one RCCL all-gather
of fixed-size per-frame result slots over xGMI per batch.

Replaces the reference's single-process nn.DataParallel (acr/main.py:61): weights and MANO tables
are resident per GPU (no per-call replicate/scatter), frames are independent (eval-mode BN,
per-frame decode, per-hand MANO), so the only communication is the gather of results:
per frame 2 hands x (176 slot + 778*3 verts + 21*3 joints) floats = 20.6 KB.  Detections vary per
frame, all-gather needs equal counts, hence fixed slots + a flag instead of the reference's
variable-length "all left rows, then all right rows" (rebuilt on the host by rows_from_slots).

Two transports for the same collective:
  * `torch.distributed.all_gather_into_tensor` (backend nccl = RCCL; gloo in the CPU tests) - the default;
  * the library's own `acrmi_allgather` (C ABI, ncclAllGather on a communicator created by
    `acrmi_comm_init`; no torch.distributed on the data path) - `transport='c'` or ACRMI_GATHER=c.
Either way the gather of batch k is queued on a side stream behind an event, into the second of two
result buffers, so it overlaps batch k+1's backbone (`submit` / `collect`).
"""
import os

import torch
import torch.distributed as dist

SLOT = 176
PER_HAND = SLOT + 778 * 3 + 21 * 3          # floats per (frame, hand)


def shard_range(n_frames, rank, world):
    """Contiguous shard [lo, hi) of a global batch; requires n_frames % world == 0 so that the
    all-gather has equal counts (the caller pads the batch otherwise)."""
    if n_frames % world:
        raise ValueError('global batch %d is not divisible by world size %d' % (n_frames, world))
    per = n_frames // world
    return rank * per, (rank + 1) * per


def padded_shard(n_frames, rank, world):
    """Shard of a global batch that need not divide by the world size: every rank runs ceil(n / world) frames so that the
    all-gather has equal counts.  Returns (lo, hi, per): this rank's REAL frames are [lo, hi) (possibly none), it runs
    `per` frames (the caller fills per - (hi - lo) of them with padding), and the gathered result keeps rows
    [r * per, r * per + (hi_r - lo_r)) of every rank r."""
    per = (n_frames + world - 1) // world
    lo = min(rank * per, n_frames)
    return lo, min(lo + per, n_frames), per


def alloc_result(n_frames, device):
    """One flat buffer [slots | verts | joints] so that a shard is gathered with ONE collective."""
    flat = torch.empty(n_frames * 2 * PER_HAND, dtype=torch.float32, device=device)
    return flat, result_views(flat, n_frames)


def result_views(flat, n_frames):
    a = n_frames * 2 * SLOT
    b = a + n_frames * 2 * 778 * 3
    return {'slots': flat[:a].view(n_frames, 2, SLOT), 'verts': flat[a:b].view(n_frames, 2, 778, 3),
            'joints': flat[b:].view(n_frames, 2, 21, 3)}


def split_gathered(gathered, world, n_local):
    """[world * per-rank flat] -> dict of [world*n_local, 2, ...] tensors in global frame order (rank-major ==
    frame order, because shards are contiguous)."""
    per_rank = gathered.view(world, -1)
    views = [result_views(per_rank[r], n_local) for r in range(world)]
    return {k: torch.cat([v[k] for v in views], 0) for k in ('slots', 'verts', 'joints')}


def all_gather_results(flat_local, n_local, group=None):
    """flat_local: this rank's alloc_result buffer.  Returns dict of [world*n_local, 2, ...] tensors."""
    world = dist.get_world_size(group)
    gathered = torch.empty(world * flat_local.numel(), dtype=flat_local.dtype, device=flat_local.device)
    dist.all_gather_into_tensor(gathered, flat_local, group=group)
    return split_gathered(gathered, world, n_local)


def init_engine_comm(engine, group=None, allow_second_communicator=False):
    """Creates the engine's own RCCL communicator (acrmi_comm_init): rank 0 draws the 128-byte unique id and the
    existing process group (any backend) carries it to the other ranks - the host's "own means" of include/acrmi.h.
    With a torch NCCL group as the carrier the process would hold two RCCL communicators (+11 ms per batch, DESIGN.md
    section 6): Engine.comm_init refuses unless allow_second_communicator - carry the id over a gloo group instead."""
    import ctypes as C
    from . import _lib
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    box = [None]
    if rank == 0:
        uid = (C.c_char * 128)()
        _lib.check(_lib.lib().acrmi_comm_unique_id(uid))
        box[0] = bytes(uid)
    if world > 1:
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    engine.comm_init(world, rank, box[0], allow_second_communicator=allow_second_communicator)


class ShardedRunner(object):
    """Runs `local_forward(frames_local, out_views)` on this rank's shard and gathers every rank's results.
    local_forward is Engine.forward on a GPU rank; tests substitute a CPU stand-in over gloo.

    transport: 'torch' (torch.distributed) or 'c' (acrmi_allgather on `engine`'s communicator); default from
    ACRMI_GATHER, else 'torch'.  transport 'c' next to a torch NCCL group is refused (two RCCL communicators in one
    process: +11 ms per batch) unless allow_second_communicator."""

    def __init__(self, local_forward, device, group=None, engine=None, transport=None, allow_second_communicator=False):
        self.local_forward = local_forward
        self.device = torch.device(device)
        self.group = group
        self.engine = engine
        self.transport = transport or os.environ.get('ACRMI_GATHER', 'torch')
        if self.transport not in ('torch', 'c'):
            raise ValueError("transport must be 'torch' or 'c'")
        if self.transport == 'c':
            if engine is None:
                raise ValueError("transport 'c' needs the Engine whose context owns the communicator")
            if not engine.comm_ranks:
                init_engine_comm(engine, group, allow_second_communicator=allow_second_communicator)
        self._buf = {}          # n_local -> two (flat, views, gathered) sets
        self._turn = 0
        self._comm_stream = None
        self._pending = {}

    def _make_comm_stream(self):
        """The side stream the gathers run on.  With an Engine at hand it is a plain HIP stream of the library
        (acrmi_stream_create) seen by torch as an external stream: torch.cuda.Stream() instantiates torch's whole pool of 32
        streams per priority, after which the few streams in use share hardware queues (DESIGN.md "Parallel lanes")."""
        if self.engine is None:
            return torch.cuda.Stream(self.device)
        import ctypes as C
        from . import _lib
        raw = C.c_void_p()
        # (an index-less 'cuda' device means the CURRENT device, not GPU 0)
        index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        _lib.check(_lib.lib().acrmi_stream_create(index, C.byref(raw)))
        self._raw_comm_stream = raw
        return torch.cuda.ExternalStream(raw.value, device=self.device)

    def close(self):
        """Releases the library stream the gathers ran on (after every ticket has been collected).  Also runs when the
        runner is garbage-collected or leaves a `with` block: a runner never leaks its stream."""
        raw = getattr(self, '_raw_comm_stream', None)
        if raw is not None:
            torch.cuda.synchronize(self.device)
            from . import _lib
            self._comm_stream = None
            _lib.lib().acrmi_stream_destroy(raw)
            self._raw_comm_stream = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def __del__(self):
        try:
            self.close()
        except Exception:      # interpreter shutdown: the library / torch may already be gone
            pass

    def _set(self, n_local, turn):
        world = dist.get_world_size(self.group)
        if n_local not in self._buf:
            sets = []
            for _ in range(2):
                flat, views = alloc_result(n_local, self.device)
                sets.append((flat, views, torch.empty(world * flat.numel(), dtype=torch.float32, device=self.device)))
            self._buf[n_local] = sets
        return self._buf[n_local][turn]

    def submit(self, frames_local):
        """Queues forward + all-gather of this rank's shard; returns a ticket for collect().  At most two tickets
        may be outstanding (double-buffered results)."""
        n_local = frames_local.shape[0]
        turn = self._turn
        if turn in self._pending:
            raise RuntimeError('collect() the ticket submitted two calls ago first (results are double-buffered)')
        self._turn ^= 1
        flat, views, gathered = self._set(n_local, turn)
        ret = self.local_forward(frames_local, views)
        ticket = {'turn': turn, 'n_local': n_local, 'event': None, 'work': None}
        if self.device.type == 'cuda':
            if self._comm_stream is None:
                self._comm_stream = self._make_comm_stream()
            # local_forward ran on the current stream - or, when it returns an event (engine.EnginePool: the batch runs
            # on one of the pool's own streams), that event marks its end
            done = ret if isinstance(ret, torch.cuda.Event) else torch.cuda.Event()
            if done is not ret:
                done.record(torch.cuda.current_stream(self.device))
            self._comm_stream.wait_event(done)             # the gather starts when this batch's MANO has finished
            with torch.cuda.stream(self._comm_stream):
                if self.transport == 'c':
                    self.engine.allgather(flat, gathered, stream=self._comm_stream)
                else:
                    dist.all_gather_into_tensor(gathered, flat, group=self.group)
                ev = torch.cuda.Event()
                ev.record(self._comm_stream)
            ticket['event'] = ev
        else:
            dist.all_gather_into_tensor(gathered, flat, group=self.group)
        self._pending[turn] = ticket
        return ticket

    def collect(self, ticket):
        """Results of a submit(): the caller's current stream waits for the gather; returns global-order tensors."""
        self._pending.pop(ticket['turn'], None)
        if ticket['event'] is not None:
            torch.cuda.current_stream(self.device).wait_event(ticket['event'])
        _, _, gathered = self._set(ticket['n_local'], ticket['turn'])
        return split_gathered(gathered, dist.get_world_size(self.group), ticket['n_local'])

    def time_gather(self, n_local, iters=10):
        """Milliseconds per all-gather of an n_local-frame shard BY ITSELF on the gather stream (collective: every rank calls
        it).  Diagnostics for the first multi-GPU runs (bench.py --gpus N reports it next to the per-rank step times): in
        the step loop the gather is queued behind a batch and hidden by the next one; this is what it costs uncovered."""
        import time
        flat, _, gathered = self._set(n_local, 0)

        def one():
            if self.transport == 'c':
                self.engine.allgather(flat, gathered, stream=self._comm_stream)
            else:
                dist.all_gather_into_tensor(gathered, flat, group=self.group)
        if self.device.type != 'cuda':
            one()
            t0 = time.perf_counter()
            for _ in range(iters):
                one()
            return (time.perf_counter() - t0) / iters * 1e3
        if self._comm_stream is None:
            self._comm_stream = self._make_comm_stream()
        torch.cuda.synchronize(self.device)
        with torch.cuda.stream(self._comm_stream):
            one()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(self._comm_stream)
            for _ in range(iters):
                one()
            e1.record(self._comm_stream)
        torch.cuda.synchronize(self.device)
        return e0.elapsed_time(e1) / iters

    def forward_local(self, frames_local):
        """Weak-scaling entry: every rank already holds its own frames."""
        return self.collect(self.submit(frames_local))

    def forward_global(self, frames_global):
        """Strong-scaling entry: every rank sees the global batch and takes its contiguous shard.  A batch that does not
        divide by the world size is padded per rank to ceil(n / world) frames (copies of the shard's last frame; a rank
        whose shard is empty runs copies of the batch's last frame) and the padding rows are dropped after the gather:
        the result is exactly the n frames, in order.  The padding rows run through whatever `local_forward` does per batch:
        with temporal smoothing switched on (Engine.set_temporal: ONE video stream per context, frames in order) a padded
        shard would feed repeated frames into the One-Euro state - sharded batches are sets of independent frames, so
        smoothing and sharding are not combined (acr/main.py smooths a single-frame stream)."""
        rank, world = dist.get_rank(self.group), dist.get_world_size(self.group)
        n = frames_global.shape[0]
        if n % world == 0:
            lo, hi = shard_range(n, rank, world)
            return self.forward_local(frames_global[lo:hi])
        lo, hi, per = padded_shard(n, rank, world)
        real = frames_global[lo:hi] if hi > lo else frames_global[n - 1:n]
        pad = per - real.shape[0]
        local = torch.cat([real, real[-1:].expand(pad, *real.shape[1:])], 0) if pad else real
        out = self.forward_local(local.contiguous())
        keep = torch.cat([torch.arange(r * per, r * per + (padded_shard(n, r, world)[1] - padded_shard(n, r, world)[0]))
                          for r in range(world)]).to(next(iter(out.values())).device)
        return {k: v.index_select(0, keep) for k, v in out.items()}
