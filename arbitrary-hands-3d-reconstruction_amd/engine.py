"""Engine: one libacrmi context on one GPU.  torch tensors are used only as HBM containers
(allocation, stream handle, host<->device copies); all arithmetic runs in libacrmi.so."""
import ctypes as C

import numpy as np
import torch

from . import _lib, packer


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _stream(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def torch_nccl_group_active():
    """True when torch.distributed is initialised with the NCCL (= RCCL on ROCm) backend in this process."""
    try:
        import torch.distributed as dist
        return bool(dist.is_available() and dist.is_initialized() and str(dist.get_backend()).lower() == 'nccl')
    except Exception:      # noqa: BLE001 - no torch.distributed in this build
        return False


class Engine(object):
    def __init__(self, device=0):
        if not torch.cuda.is_available():
            raise _lib.AcrmiError('no GPU visible: the ACR path runs only on the HIP kernels (no CPU fallback)')
        self.L = _lib.lib()
        self.device = torch.device('cuda', device if isinstance(device, int) else torch.device(device).index or 0)
        self.ctx = C.c_void_p()
        _lib.check(self.L.acrmi_create(C.byref(self.ctx), self.device.index))
        self.max_batch = 0
        self.program = None
        self.have_mano = False
        self.point_heads = False
        self.lanes = 0
        self.lane_plan = False
        self.conf_thresh = 0.35
        self.center_idx = 9
        self.temporal = False
        self.mano_fp16 = False
        self.batch_semantics = 'frame'
        self.smooth_coeff = None          # None = the library default (4.0)
        self.comm_ranks = 0
        self._mano_tables = {}

    def close(self):
        # (the handle is dropped first and without touching module globals: during interpreter shutdown `C` may already
        # be None, and an exception between the destroy and the reset would let a second __del__ destroy it again)
        ctx, self.ctx = self.ctx, None
        if ctx:
            self.L.acrmi_destroy(ctx)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- setup ---------------------------------------------------------------------------------
    def load_state_dict(self, sd, max_batch=1, keep_taps=False, precision='fp32', keep_weights=False, keep_all=False,
                        wino24='auto', splitk='auto'):
        """acr/utils.py:1153-1168 (load_model): reference-format checkpoint -> resident packed weights.
        keep_taps: see packer.lower (backbone taps stay readable through `buffer(program['taps'][name], B)`).
        precision: 'fp32' | 'fp16' | 'bf16' (args().model_precision, acr/config.py:96; packer.lower) | 'fp16x3' / 'bf16x3' (fp32
        storage, operands split into two f16 / bf16 numbers on the 16-bit matrix pipe for the 3x3 (stride 1, stride 2 outside the HR
        fuse hosts) and 1x1 stride-1 layers: csrc/conv_x3.inc, conv_x3p.inc, conv_x3s2.inc).  'fp16x3' has the f16 RANGE: every activation must stay within |x| <= 65504 - a checkpoint that
        exceeds it makes the affected calls return NaN slots / meshes and `check_range()` raise AcrmiRangeError (the
        host-facing API calls it); 'bf16x3' has fp32's range at 16-bit operand precision (8e-6 m instead of 1e-6 m on the
        bench frames)
        wino24: 'auto' (F(2x4,3x3) for contexts of max_batch >= 16), True / False to force (packer.lower).
        splitk: 'auto' (split-K lowering of the low-resolution 3x3 layers for contexts of max_batch < 16), True / False."""
        # F(2x4,3x3) is a large-batch choice: its items (8x32 pixels x one n-tile) are half as many as conv_wino2's
        # small-batch items, which costs latency when a launch cannot fill the CUs anyway (batch 1: 3.7 vs 3.5 ms)
        small = max_batch < 16        # single-frame / small-batch context: latency lowering (packer.lower)
        if wino24 == 'auto':
            wino24 = None if not small else False
        self.load_program(packer.lower(sd, keep_taps=keep_taps, precision=precision, keep_weights=keep_weights,
                                       keep_all=keep_all, wino24=wino24, splitk=small if splitk == 'auto' else splitk),
                          max_batch)

    def load_program(self, prog, max_batch=1, share_with=None):
        """A program lowered elsewhere (packer.lower, or another Engine's `program`): the same packed weights and op
        list in this context - an EnginePool lowers the checkpoint once.
        share_with: an Engine of the same device that already holds THIS program: its device copy of the weight blob is
        used instead of a second upload (acrmi_share_weights; 330 MB per context at HRNet-W32 fp32)."""
        if share_with is not None and share_with is not self and share_with.program is prog and \
                share_with.device == self.device:
            _lib.check(self.L.acrmi_share_weights(self.ctx, share_with.ctx), self.ctx)
        else:
            blob = prog['blob']
            _lib.check(self.L.acrmi_load_weights(self.ctx, blob.ctypes.data_as(C.c_void_p), blob.size), self.ctx)
        self.program = prog
        self._set_program(max_batch)

    def _set_program(self, max_batch):
        prog = self.program
        bufs = (_lib.BufferDesc * len(prog['bufs']))(*[_lib.BufferDesc(*b) for b in prog['bufs']])
        ops = (_lib.Op * len(prog['ops']))(*prog['ops'])
        with torch.cuda.device(self.device):
            torch.cuda.synchronize()
            _lib.check(self.L.acrmi_set_program(self.ctx, bufs, len(bufs), ops, len(ops), C.byref(prog['heads']),
                                                max_batch), self.ctx)
        self.max_batch = max_batch

    @property
    def has_point_heads(self):
        """True when the loaded program carries the point-heads variant (packer.lower emits it for fp32 HRNet-W32
        programs only: 16-bit programs, HRNet-W48 and the ResNet-50 backbone run the dense heads)."""
        prog = getattr(self, 'program', None)
        return bool(prog) and any(o.kind == _lib.OP_POINTHEADS for o in prog['ops'])

    def set_point_heads(self, on):
        """ACRMI_OPT_POINT_HEADS: `forward` evaluates the params/cam/prior head towers and the mix conv only at the
        pixels ResultParser samples (acr/result_parser.py:49-57,141-145).  Same slots / vertices up to fp32
        round-off; `head_maps` params/prior maps are then only valid after `backbone_heads` (always dense).
        Returns what is in effect: a program without the point-heads ops stays on its dense heads (same results)."""
        on = bool(on) and self.has_point_heads
        _lib.check(self.L.acrmi_set_option(self.ctx, _lib.OPT_POINT_HEADS, int(on)), self.ctx)
        self.point_heads = on
        return on

    def set_lanes(self, n):
        """ACRMI_OPT_LANES: independent chains of the program on n parallel HIP streams (1 = single stream,
        0 = by batch size: the library default)."""
        _lib.check(self.L.acrmi_set_option(self.ctx, _lib.OPT_LANES, int(n)), self.ctx)
        self.lanes = int(n)

    def set_lane_plan(self, on):
        """ACRMI_OPT_LANE_PLAN: small-batch schedules assign lanes from the op times the last `profile_ops` measured
        (list scheduling) instead of from the structure of the graph alone."""
        _lib.check(self.L.acrmi_set_option(self.ctx, _lib.OPT_LANE_PLAN, int(bool(on))), self.ctx)
        self.lane_plan = bool(on)

    def tune_lanes(self, batch=1, candidates=(1, 2, 4, 6), calls=5):
        """Picks ACRMI_OPT_LANES / ACRMI_OPT_LANE_PLAN by measurement IN THIS PROCESS: the ops are profiled once (the
        library keeps the times and plans the lanes from them), then every candidate lane count is timed with the
        structural and with the planned assignment - `calls` network passes (acrmi_backbone_heads on zero frames,
        after 2 warm-ups) each - and the fastest is set.  Why: what parallel lanes gain depends on state the library
        cannot see - ROCm maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4, which this package
        leaves alone since round 6: more queues make five or more busy streams collapse), and lane streams created next to other live streams (torch's
        pools, RCCL, another context) may share a queue with them: measured 7.5-9.9 ms per batch-1 call with four lanes
        in that state against 3.4 ms with their own queues and 5.0 ms on one stream.  Results do not depend on the
        assignment (same kernels, same order per buffer).  When the winner is what the library picks by itself for this
        batch (4 lanes up to 32 frames, 2 above) the lane option goes back to 0 = by batch size, so other batch
        sizes keep their own default; a measurement taken at a batch of the OTHER class than the one the context was built
        for (max_batch <= 32 / > 32) only sets the plan flag and leaves the lane count "by batch size".
        Returns ((lanes, planned), {(lanes, planned): ms}) - `lanes` is what was measured fastest, `self.lanes` what is set."""
        import time
        if self.program is None:
            raise _lib.AcrmiError('no checkpoint loaded')
        x = torch.zeros(batch, 512, 512, 3, dtype=torch.uint8, device=self.device)
        self.set_lanes(1)
        self.profile_ops(x)
        self.profile_ops(x)           # (warm) these times are what the planned schedules use
        ms = {}
        for n in candidates:
            for planned in ((False,) if n == 1 else (False, True)):
                self.set_lane_plan(planned)
                self.set_lanes(n)
                for _ in range(2):
                    self.backbone_heads(x)
                torch.cuda.synchronize(self.device)
                t0 = time.perf_counter()
                for _ in range(calls):
                    self.backbone_heads(x)
                torch.cuda.synchronize(self.device)
                ms[(int(n), planned)] = (time.perf_counter() - t0) / calls * 1e3
        best = min(ms, key=ms.get)
        self.set_lane_plan(best[1])
        # The measurement was taken at `batch`; it only overrides the library's choice for contexts that serve such batches
        # (the library's schedules: 4 lanes up to 32 frames, 2 above).  A context built for large batches keeps "by batch
        # size" after a batch-1 measurement: a batch-1 winner of 1 or 6 lanes must not replace the 2 lanes of its 64-frame
        # calls
        # ... and the other way round: tune_lanes(batch=64) on a max_batch=64 context IS a measurement of the class that context
        # serves, and is applied.  Classes as in the library: up to 32 frames / above (AUTO_SMALL_BATCH).
        same_class = (batch > 32) == (self.max_batch > 32)
        default = 4 if batch <= 32 else 2
        self.set_lanes(best[0] if same_class and best[0] != default else 0)
        return best, ms

    def set_conf_thresh(self, thresh):
        """ACRMI_OPT_CONF_THRESH = args().centermap_conf_thresh (acr/result_parser.py:198-205,241; strict >)."""
        _lib.check(self.L.acrmi_set_option_f(self.ctx, _lib.OPT_CONF_THRESH, float(thresh)), self.ctx)
        self.conf_thresh = float(thresh)

    def set_center_idx(self, idx):
        """ACRMI_OPT_CENTER_IDX: root-alignment joint of the fused path (args().align_idx; None = no alignment,
        acr/mano_wrapper.py:19-33)."""
        _lib.check(self.L.acrmi_set_option(self.ctx, _lib.OPT_CENTER_IDX, -1 if idx is None else int(idx)), self.ctx)
        self.center_idx = idx

    def set_temporal(self, on, smooth_coeff=None):
        """ACRMI_OPT_TEMPORAL: `forward` smooths the decoded poses/betas on the device before MANO
        (acr/main.py:69-83); the frames of a call are then one video stream in order."""
        if smooth_coeff is not None:
            _lib.check(self.L.acrmi_set_option_f(self.ctx, _lib.OPT_SMOOTH_COEFF, float(smooth_coeff)), self.ctx)
            self.smooth_coeff = float(smooth_coeff)
        _lib.check(self.L.acrmi_set_option(self.ctx, _lib.OPT_TEMPORAL, int(bool(on))), self.ctx)
        self.temporal = bool(on)

    def set_mano_fp16(self, on):
        """ACRMI_OPT_MANO_FP16: the MANO stage reads f16 copies of its blend-shape tables and skinning weights
        (BASELINE.json configs[4]); fp32 arithmetic."""
        _lib.check(self.L.acrmi_set_option(self.ctx, _lib.OPT_MANO_FP16, int(bool(on))), self.ctx)
        self.mano_fp16 = bool(on)

    def set_batch_semantics(self, mode):
        """ACRMI_OPT_BATCH_PRIOR: 'frame' (default: every frame decides its cross-hand prior like a batch of one) or
        'reference' (`forward` applies the reference's batch-wide rules at B > 1, acr/result_parser.py:42-47,102-145:
        decode -> acrmi_prior_gate -> gated decode on the device, one call)."""
        if mode not in ('frame', 'reference'):
            raise ValueError("batch_semantics %r: 'frame' or 'reference'" % (mode,))
        _lib.check(self.L.acrmi_set_option(self.ctx, _lib.OPT_BATCH_PRIOR, int(mode == 'reference')), self.ctx)
        self.batch_semantics = mode

    def prior_gate(self, slots):
        """acrmi_prior_gate: slots [B,2,176] of a first decode -> int32 [B] for decode(prior_gate=...), on the device (no
        host round trip): the reference's batch-wide prior decision (acr/result_parser.py:42-47,102-145)."""
        if not slots.is_cuda or slots.dtype != torch.float32 or not slots.is_contiguous():
            raise ValueError('slots must be a contiguous float32 device tensor')
        gate = torch.empty(slots.shape[0], dtype=torch.int32, device=slots.device)
        _lib.check(self.L.acrmi_prior_gate(self.ctx, _ptr(slots), slots.shape[0], _ptr(gate), _stream(self.device)), self.ctx)
        return gate

    def smooth(self, slots):
        """One-Euro smoothing of slots [B,2,176] in place (acr/utils.py:1466-1527), state resident in the context."""
        if not slots.is_cuda or slots.dtype != torch.float32 or not slots.is_contiguous():
            raise ValueError('slots must be a contiguous float32 device tensor')
        _lib.check(self.L.acrmi_smooth(self.ctx, _ptr(slots), slots.shape[0], _stream(self.device)), self.ctx)
        return slots

    def smooth_reset(self):
        _lib.check(self.L.acrmi_smooth_reset(self.ctx, _stream(self.device)), self.ctx)

    # ---- multi-GPU ----------------------------------------------------------------
    def comm_init(self, n_ranks, rank, unique_id, allow_second_communicator=False):
        """acrmi_comm_init: RCCL communicator on this context's device (collective).
        Refused when this process already runs a torch.distributed NCCL (= RCCL) process group: a process that holds TWO RCCL
        communicators pays +11 ms per batch for a side-stream all-gather next to a running batch (46.8 vs 35.4 ms at world size
        1, tools/allgather_probe.py) - a host uses ONE of the two transports (the C transport's control
        plane runs over gloo).  allow_second_communicator=True overrides (correctness is unaffected; tests use it)."""
        if not allow_second_communicator and torch_nccl_group_active():
            raise _lib.AcrmiError(
                'acrmi_comm_init: this process already runs a torch.distributed NCCL (RCCL) process group; a second RCCL '
                'communicator next to it costs ~11 ms per batch (DESIGN.md section 6).  Use ONE transport: '
                "ShardedRunner(transport='torch') on the NCCL group, or initialise torch.distributed with backend 'gloo' for the "
                "control plane of transport='c'.  (allow_second_communicator=True overrides.)")
        uid = (C.c_char * 128).from_buffer_copy(bytes(unique_id))
        _lib.check(self.L.acrmi_comm_init(self.ctx, int(n_ranks), int(rank), uid), self.ctx)
        self.comm_ranks = int(n_ranks)

    def allgather(self, send, recv, stream=None):
        """acrmi_allgather of a flat float32 device tensor into recv [n_ranks * send.numel()]."""
        if recv.numel() != self.comm_ranks * send.numel():
            raise ValueError('recv must hold n_ranks * send.numel() floats')
        st = _stream(self.device) if stream is None else C.c_void_p(stream.cuda_stream)
        _lib.check(self.L.acrmi_allgather(self.ctx, None, _ptr(send), _ptr(recv), send.numel(), st), self.ctx)
        return recv

    def run_point_heads(self, B):
        """Re-evaluates the point heads on the resident buffers for the current center maps (acrmi_point_heads)."""
        _lib.check(self.L.acrmi_point_heads(self.ctx, B, _stream(self.device)), self.ctx)

    def ensure_batch(self, B):
        if self.program is None:
            raise _lib.AcrmiError('no checkpoint loaded')
        if B > self.max_batch:
            self._set_program(B)

    def load_mano(self, tables):
        """tables: {'left': {...}, 'right': {...}} numpy arrays as registered by mano/manolayer.py:61-102.
        The left-hand shapedirs x-flip (acr/mano_wrapper.py:35) must already be applied by the caller."""
        for name in ('left', 'right'):
            self.load_mano_side(name, tables[name])

    def adopt(self, other):
        """Takes over the MANO tables and options of an engine this one replaces (a checkpoint reload builds a new
        context; MANOWrapper / callers keep working against the model's current engine)."""
        for name, t in other._mano_tables.items():
            self.load_mano_side(name, t)
        self.set_conf_thresh(other.conf_thresh)
        self.set_center_idx(other.center_idx)
        if other.lanes:
            self.set_lanes(other.lanes)
        self.set_temporal(other.temporal, smooth_coeff=other.smooth_coeff)
        self.set_mano_fp16(other.mano_fp16)
        self.set_batch_semantics(getattr(other, 'batch_semantics', 'frame'))

    def load_mano_side(self, name, t):
        """One side's tables (mano/manolayer.py:61-102 buffers) -> HBM, blend-shape tables transposed."""
        side = {'left': 0, 'right': 1}[name]
        arrs = [np.ascontiguousarray(np.asarray(t[k], np.float32)) for k in
                ('v_template', 'shapedirs', 'posedirs', 'J_regressor', 'weights', 'hands_mean')]
        shapes = [(778, 3), (778, 3, 10), (778, 3, 135), (16, 778), (778, 16), (45,)]
        for a, s in zip(arrs, shapes):
            if a.shape != s and a.reshape(-1).shape != (int(np.prod(s)),):
                raise ValueError('MANO table has shape %s, expected %s' % (a.shape, s))
        with torch.cuda.device(self.device):
            torch.cuda.synchronize()      # tables may be replaced while earlier launches still read them
            _lib.check(self.L.acrmi_load_mano(self.ctx, side, *[a.ctypes.data_as(C.c_void_p) for a in arrs]), self.ctx)
        self._mano_tables[name] = {k: a for k, a in zip(('v_template', 'shapedirs', 'posedirs', 'J_regressor', 'weights',
                                                          'hands_mean'), arrs)}
        self.have_mano = len(self._mano_tables) == 2

    # ---- hot path ------------------------------------------------------------------------------
    def _check_img(self, img):
        if img.dtype != torch.uint8 or img.dim() != 4 or tuple(img.shape[1:]) != (512, 512, 3):
            raise ValueError('image must be uint8 [B,512,512,3] RGB, got %s %s' % (img.dtype, tuple(img.shape)))
        if img.device != self.device:
            img = img.to(self.device)
        return img.contiguous()

    def backbone_heads(self, img):
        img = self._check_img(img)
        B = img.shape[0]
        self.ensure_batch(B)
        _lib.check(self.L.acrmi_backbone_heads(self.ctx, _ptr(img), B, _stream(self.device)), self.ctx)
        return B

    def check_range(self):
        """'fp16x3' programs: waits for the current stream and raises _lib.AcrmiRangeError if an activation left the f16
        range (|x| > 65504) since the last check - the split halves are then inf / -inf and the library has written the
        affected calls' slots and meshes as NaN (acrmi_check_range).  A no-op for every other precision.  Engine.forward
        stays asynchronous and does NOT call this; the host-facing API (acr.model.ACR.forward, acr.main.ACR) does."""
        _lib.check(self.L.acrmi_check_range(self.ctx, _stream(self.device)), self.ctx)

    def heads(self, features):
        """acr/model.py:47-65 on backbone features the caller holds: float [B, C0, 128, 128] (NCHW, what the reference's
        head_forward takes; C0 = 32 for HRNet-W32) -> the heads run on them, results as after backbone_heads (head_maps /
        decode).  fp32-storage programs (acrmi_heads)."""
        if self.program is None:
            raise _lib.AcrmiError('no checkpoint loaded')
        c0 = self.L.acrmi_backbone_channels(self.ctx)
        _lib.check(c0, self.ctx)
        if features.dim() != 4 or tuple(features.shape[1:]) != (c0, 128, 128) or not features.is_floating_point():
            raise ValueError('features must be float [B,%d,128,128] (NCHW), got %s %s' % (c0, features.dtype, tuple(features.shape)))
        B = features.shape[0]
        self.ensure_batch(B)
        f = features.to(self.device, torch.float32).contiguous()
        _lib.check(self.L.acrmi_heads(self.ctx, _ptr(f), B, _stream(self.device)), self.ctx)
        return B

    def buffer(self, buf_id, B, channels=None):
        """Zero-copy torch view [B,h,w,cs] of a program buffer (NHWC) in its storage type (float32 / float16 /
        bfloat16: 16-bit programs keep the activations between layers in 16 bits)."""
        h, w, cs = C.c_int(), C.c_int(), C.c_int()
        p = self.L.acrmi_buffer_ptr(self.ctx, buf_id, C.byref(h), C.byref(w), C.byref(cs))
        if not p:
            raise ValueError('no such buffer %d' % buf_id)
        dt = self.L.acrmi_buffer_dtype(self.ctx, buf_id)
        n = B * h.value * w.value * cs.value
        arr = _DevArray(p, n, self.device, '<f4' if dt == _lib.DT_F32 else '<i2')
        t = torch.as_tensor(arr, device=self.device)
        if dt != _lib.DT_F32:
            t = t.view(torch.float16 if dt == _lib.DT_F16 else torch.bfloat16)
        t = t.view(B, h.value, w.value, cs.value)
        return t if channels is None else t[..., :channels]

    def head_maps(self, B, nchw=True):
        """The reference's H11 output dict (acr/model.py:56-63) from the resident head buffers."""
        hl = self.program['heads']
        out = {}
        for si, side in enumerate('lr'):
            out[side + '_params_maps'] = self.buffer(hl.params_buf[si], B, 109)
            out[side + '_center_map'] = self.buffer(hl.center_buf[si], B, 1)
            out[side + '_prior_maps'] = self.buffer(hl.prior_buf[si], B, 106)
        out['segms'] = self.buffer(hl.segm_buf, B, 33)
        if nchw:
            out = {k: v.permute(0, 3, 1, 2).contiguous() for k, v in out.items()}
        return out

    def decode(self, B, prior_gate=None):
        """prior_gate: int32 device tensor [B] (acrmi_decode_gated: < 0 per-frame rule, 0 no prior, 1 prior when the frame
        has both hands) - how acr.result_parser applies the reference's batch-wide prior rules; None = per frame."""
        slots = torch.empty(B, 2, _lib.SLOT, dtype=torch.float32, device=self.device)
        if prior_gate is None:
            _lib.check(self.L.acrmi_decode(self.ctx, B, _ptr(slots), _stream(self.device)), self.ctx)
        else:
            g = prior_gate.to(device=self.device, dtype=torch.int32).contiguous()
            if g.numel() != B:
                raise ValueError('prior_gate must hold one int per frame')
            _lib.check(self.L.acrmi_decode_gated(self.ctx, B, _ptr(g), _ptr(slots), _stream(self.device)), self.ctx)
        return slots

    def mano(self, poses, betas, side, center_idx=9, cam=None, offsets=None, rotmat=False):
        """poses [H,48], betas [H,10] float32 on device; side: int tensor [H] (0 left, 1 right).
        rotmat: poses are [H,16,3,3] orthonormal rotation matrices (acrmi_mano_rotmat; no projection outputs)."""
        H = poses.shape[0]
        dev = self.device
        poses = poses.to(dev, torch.float32).contiguous()
        if rotmat:
            if tuple(poses.shape[1:]) != (16, 3, 3) or cam is not None:
                raise ValueError('rotmat poses are [H,16,3,3]; the projection is not part of this mode')
            betas = betas.to(dev, torch.float32).contiguous()
            verts = torch.empty(H, 778, 3, dtype=torch.float32, device=dev)
            joints = torch.empty(H, 21, 3, dtype=torch.float32, device=dev)
            center = torch.empty(H, 1, 3, dtype=torch.float32, device=dev)
            if H:
                side = side.to(dev, torch.int32).contiguous()
                _lib.check(self.L.acrmi_mano_rotmat(self.ctx, _ptr(poses), _ptr(betas), 10, _ptr(side), H,
                                                    -1 if center_idx is None else int(center_idx), _ptr(verts), _ptr(joints),
                                                    _ptr(center), _stream(dev)), self.ctx)
            return verts, joints, center, {}
        betas = betas.to(dev, torch.float32).contiguous()
        verts = torch.empty(H, 778, 3, dtype=torch.float32, device=dev)
        joints = torch.empty(H, 21, 3, dtype=torch.float32, device=dev)
        center = torch.empty(H, 1, 3, dtype=torch.float32, device=dev)
        extra = {}
        if H == 0:
            return verts, joints, center, extra
        side = side.to(dev, torch.int32).contiguous()
        vc = pj = org = None
        if cam is not None:
            cam = cam.to(dev, torch.float32).contiguous()
            vc = torch.empty(H, 778, 3, dtype=torch.float32, device=dev)
            pj = torch.empty(H, 21, 2, dtype=torch.float32, device=dev)
            extra = {'verts_camed': vc, 'pj2d': pj}
            if offsets is not None:
                offsets = offsets.to(dev, torch.float32).contiguous()
                org = torch.empty(H, 21, 2, dtype=torch.float32, device=dev)
                extra['pj2d_org'] = org
        _lib.check(self.L.acrmi_mano(self.ctx, _ptr(poses), 48, _ptr(betas), 10, _ptr(side), H,
                                     -1 if center_idx is None else int(center_idx), _ptr(verts), _ptr(joints),
                                     _ptr(center), _ptr(cam), 3, _ptr(offsets), _ptr(vc), _ptr(pj), _ptr(org),
                                     _stream(dev)), self.ctx)
        return verts, joints, center, extra

    def forward(self, img, offsets=None, project=False, out=None, stream=None):
        """frames -> (slots [B,2,176], verts [B,2,778,3], joints [B,2,21,3][, verts_camed, pj2d, pj2d_org]).
        stream: raw hipStream_t (int) to queue the call on instead of torch's current stream; tensors this call
        allocates still come from the current stream's pool."""
        img = self._check_img(img)
        B = img.shape[0]
        self.ensure_batch(B)
        dev = self.device
        if out is None:
            out = {'slots': torch.empty(B, 2, _lib.SLOT, dtype=torch.float32, device=dev),
                   'verts': torch.empty(B, 2, 778, 3, dtype=torch.float32, device=dev),
                   'joints': torch.empty(B, 2, 21, 3, dtype=torch.float32, device=dev)}
            if project:
                out['verts_camed'] = torch.empty(B, 2, 778, 3, dtype=torch.float32, device=dev)
                out['pj2d'] = torch.empty(B, 2, 21, 2, dtype=torch.float32, device=dev)
                if offsets is not None:
                    out['pj2d_org'] = torch.empty(B, 2, 21, 2, dtype=torch.float32, device=dev)
        if offsets is not None:
            offsets = offsets.to(dev, torch.float32).contiguous()
        _lib.check(self.L.acrmi_forward(self.ctx, _ptr(img), B, _ptr(offsets), _ptr(out['slots']), _ptr(out['verts']),
                                        _ptr(out['joints']), _ptr(out.get('verts_camed')), _ptr(out.get('pj2d')),
                                        _ptr(out.get('pj2d_org')),
                                        _stream(dev) if stream is None else C.c_void_p(stream)), self.ctx)
        return out

    def profile_ops(self, img):
        img = self._check_img(img)
        B = img.shape[0]
        self.ensure_batch(B)
        n = len(self.program['ops'])
        ms = (C.c_float * n)()
        _lib.check(self.L.acrmi_profile_ops(self.ctx, _ptr(img), B, ms, n, _stream(self.device)), self.ctx)
        ops = self.program['ops']
        return [dict(info, ms=float(ms[i]), idx=i, ksize=int(ops[i].ksize), stride=int(ops[i].stride))
                for i, info in enumerate(self.program['op_info'])]


class EnginePool(object):
    """n contexts on ONE GPU that take batches in turn, each on its own HIP stream (serving throughput).

    A batch ends in a tail of kernels with few work items (attention pooling, the per-frame pare bias, decode, MANO: ~1
    ms at batch 64) and the next one starts with pipeline fill; on one stream the 256 CUs idle through both.  With two
    independent program instances in flight the hardware scheduler fills those gaps with the other batch's kernels:
    39.8 -> 39.0 ms per batch of 64 (1607 -> 1642 frames/s), and the parallel lanes inside a context are no longer
    needed (1 lane per context measured best).  Every batch is computed by one context exactly as Engine.forward does
    (bit-identical results, tests/test_gpu_api.py); the price is a second set of activations (9.9 GB at batch 64 of the
    288 GB) and one more batch of latency.  The reference has no counterpart (one nn.Module call per image,
    acr/main.py:92-96).  A non-Python host does the same with two acrmi_ctx and two streams.

        pool = EnginePool(0, n=2); pool.load_state_dict(sd, max_batch=64); pool.load_mano(tables)
        t = pool.submit(frames)          # queued on the context's stream, behind the caller's current stream
        out = pool.collect(t)            # the caller's current stream waits for that batch; dict of device tensors
    At most n tickets may be outstanding.
    The contexts run on plain HIP streams of the library (acrmi_stream_create): torch.cuda.Stream() would instantiate
    torch's whole stream pool, and with that many streams alive the few in use share hardware queues."""

    def __init__(self, device=0, n=2, first=None):
        """first: an existing Engine to use as context 0 (its program, if loaded, is shared with the others)."""
        if n < 1:
            raise ValueError('n must be >= 1')
        self.engines = ([first] if first is not None else []) + [Engine(device) for _ in range(n - (first is not None))]
        self.device = self.engines[0].device
        # plain HIP streams from the library (acrmi_stream_create), seen by torch as external streams
        self._raw = []
        for _ in range(n):
            st = C.c_void_p()
            _lib.check(self.engines[0].L.acrmi_stream_create(self.device.index, C.byref(st)))
            self._raw.append(st)
        self.streams = [torch.cuda.ExternalStream(st.value, device=self.device) for st in self._raw]
        self._turn = 0
        self._busy = [None] * n
        self._inflight = []      # released tickets whose batch may still be running: their tensors stay referenced

    def __len__(self):
        return len(self.engines)

    def close(self, keep_first=False):
        """Releases the streams and the contexts (keep_first: all but context 0, e.g. an Engine passed as `first`)."""
        if self._raw:
            with torch.cuda.device(self.device):
                torch.cuda.synchronize()
            self._inflight = []          # every batch has run: the tensors the tickets kept alive may go
            self.streams = []
            L = _lib.lib()
            for st in self._raw:
                L.acrmi_stream_destroy(st)
            self._raw = []
        for e in self.engines[1 if keep_first else 0:]:
            e.close()
        self.engines = []      # (keep_first: context 0 stays open and belongs to whoever passed it in)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def load_state_dict(self, sd, max_batch=1, lanes=1, precision='fp32', wino24='auto'):
        """sd = None: share the program context 0 already holds.  wino24 as Engine.load_state_dict (the same
        checkpoint at the same max_batch lowers to the same program on an Engine and on a pool: bit-equal results)."""
        if wino24 == 'auto':
            wino24 = None if max_batch >= 16 else False
        prog = self.engines[0].program if sd is None else packer.lower(sd, precision=precision, wino24=wino24,
                                                                       splitk=max_batch < 16)
        if prog is None:
            raise _lib.AcrmiError('no checkpoint loaded')
        holder = next((e for e in self.engines if e.program is prog), None)      # a context that already holds the blob
        for e in self.engines:
            if e.program is not prog:
                e.load_program(prog, max_batch, share_with=holder)      # one device copy of the weights for the pool
                holder = holder or e
            else:
                e.ensure_batch(max_batch)
            e.set_lanes(lanes)

    def load_mano(self, tables=None):
        """tables = None: context 0's tables."""
        for e in self.engines:
            if tables is not None:
                e.load_mano(tables)
            elif e is not self.engines[0]:
                for name, t in self.engines[0]._mano_tables.items():
                    e.load_mano_side(name, t)

    def configure(self, fn):
        """fn(engine) on every context (options: set_conf_thresh, set_center_idx, set_point_heads, ...)."""
        for e in self.engines:
            fn(e)

    def submit(self, img, offsets=None, project=False, out=None):
        i = self._turn
        if self._busy[i] is not None:
            raise RuntimeError('collect() the ticket submitted %d calls ago first' % len(self.engines))
        self._turn = (i + 1) % len(self.engines)
        # host-side conversions run on the caller's stream: they must be queued before `ready` is recorded
        img = self.engines[i]._check_img(img)
        if offsets is not None:
            offsets = offsets.to(self.device, torch.float32).contiguous()
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(self.device))   # the frames (and the out tensors) as the caller left them
        st = self.streams[i]
        st.wait_event(ready)
        # (no `with torch.cuda.stream(st)`: tensors must not come from the caching allocator's pool of a stream that
        # close() destroys; everything this call allocates belongs to the caller's stream, which collect() orders behind
        # the batch)
        res = self.engines[i].forward(img, offsets=offsets, project=project, out=out, stream=self._raw[i].value)
        done = torch.cuda.Event()
        done.record(st)
        # Everything the batch reads or writes stays referenced by the ticket until its event has completed: the call
        # runs on the pool's stream, so torch's caching allocator (which only knows the caller's stream) would hand a
        # dropped `img` / `offsets` block to the caller's next allocation while the kernels still read it.
        ticket = {'slot': i, 'event': done, 'out': res, 'img': img, 'offsets': offsets}
        self._busy[i] = ticket
        return ticket

    def _reap(self):
        self._inflight = [t for t in self._inflight if not t['event'].query()]

    def release(self, ticket):
        """Frees the ticket's context for the next submit without making any stream wait: for callers that order
        their consumers on ticket['event'] themselves (parallel.ShardedRunner queues the all-gather behind it)."""
        if self._busy[ticket['slot']] is ticket:
            self._busy[ticket['slot']] = None
            self._reap()
            self._inflight.append(ticket)      # kept until its event has completed (see submit)
        return ticket['event']

    def collect(self, ticket):
        self.release(ticket)
        torch.cuda.current_stream(self.device).wait_event(ticket['event'])
        return ticket['out']


class _DevArray(object):
    """__cuda_array_interface__ shim so torch can view library-owned HBM without copying."""

    def __init__(self, ptr, n, device, typestr='<f4'):
        self.__cuda_array_interface__ = {'shape': (n,), 'typestr': typestr, 'data': (int(ptr), False), 'version': 2}
        self.device = device
