"""Multi-GPU: one process per GPU, frames sharded contiguously by rank, one RCCL all-gather
of fixed-size per-frame result slots over xGMI per batch.

Replaces the reference's single-process nn.DataParallel (acr/main.py:61): weights and MANO tables
are resident per GPU (no per-call replicate/scatter), frames are independent (eval-mode BN,
per-frame decode, per-hand MANO), so the only communication is the gather of results:
per frame 2 hands x (176 slot + 778*3 verts + 21*3 joints) floats = 20.6 KB.  Detections vary per
frame, all-gather needs equal counts, hence fixed slots + a flag instead of the reference's
variable-length "all left rows, then all right rows" (rebuilt on the host by rows_from_slots).
"""
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


def alloc_result(n_frames, device):
    """One flat buffer [slots | verts | joints] so that a shard is gathered with ONE collective."""
    flat = torch.empty(n_frames * 2 * PER_HAND, dtype=torch.float32, device=device)
    return flat, result_views(flat, n_frames)


def result_views(flat, n_frames):
    a = n_frames * 2 * SLOT
    b = a + n_frames * 2 * 778 * 3
    return {'slots': flat[:a].view(n_frames, 2, SLOT), 'verts': flat[a:b].view(n_frames, 2, 778, 3),
            'joints': flat[b:].view(n_frames, 2, 21, 3)}


def all_gather_results(flat_local, n_local, group=None):
    """flat_local: this rank's alloc_result buffer.  Returns dict of [world*n_local, 2, ...] tensors in
    global frame order (rank-major == frame order, because shards are contiguous)."""
    world = dist.get_world_size(group)
    gathered = torch.empty(world * flat_local.numel(), dtype=flat_local.dtype, device=flat_local.device)
    dist.all_gather_into_tensor(gathered, flat_local, group=group)
    per_rank = gathered.view(world, -1)
    views = [result_views(per_rank[r], n_local) for r in range(world)]
    return {k: torch.cat([v[k] for v in views], 0) for k in ('slots', 'verts', 'joints')}


class ShardedRunner(object):
    """Runs `local_forward(frames_local, out_views)` on this rank's shard and gathers every rank's results.
    local_forward is Engine.forward on a GPU rank; tests substitute a CPU stand-in over gloo."""

    def __init__(self, local_forward, device, group=None):
        self.local_forward = local_forward
        self.device = device
        self.group = group
        self._buf = {}

    def _result(self, n_local):
        if n_local not in self._buf:
            self._buf[n_local] = alloc_result(n_local, self.device)
        return self._buf[n_local]

    def forward_local(self, frames_local):
        """Weak-scaling entry: every rank already holds its own frames."""
        n_local = frames_local.shape[0]
        flat, views = self._result(n_local)
        self.local_forward(frames_local, views)
        return all_gather_results(flat, n_local, self.group)

    def forward_global(self, frames_global):
        """Strong-scaling entry: every rank sees the global batch and takes its contiguous shard."""
        rank, world = dist.get_rank(self.group), dist.get_world_size(self.group)
        lo, hi = shard_range(frames_global.shape[0], rank, world)
        return self.forward_local(frames_global[lo:hi])
