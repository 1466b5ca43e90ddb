"""GPU parity of the 16-bit path (the reference's --model_precision fp16 branch, acr/model.py:18-19,33-37, and its bf16
twin) and of the HRNet-W48 variant (BASELINE.json configs[4]).

NO REFERENCE ORACLE exists for either: autocast is CUDA-only and the reference hard-wires HRNet-W32.
What is checked instead:
  * kernel level: every 16-bit convolution against an exact fp64 convolution of the 16-bit inputs - the result must be
    the correctly rounded value up to the fp32 accumulation error (ONE rounding per layer);
  * program level: the resident 16-bit program against oracle/program.py (the op-list interpreter, itself pinned on
    fp32 programs to the reference-pinned oracle/acr_net.py), a few 16-bit ulps of drift allowed;
  * against the fp32 reference fixtures: the 16-bit error is REPORTED (vertices / joints / maps), with loose bounds.
Run on the MI355X box with `pytest -m gpu`."""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import cases
from conftest import ROOT, golden, pkg
from oracle import acr_net, decode as odec, mano as omano, program as oprog

pytestmark = pytest.mark.gpu

TD = {'fp16': torch.float16, 'bf16': torch.bfloat16}
MANT = {'fp16': 10, 'bf16': 7}


def ulp(v, precision):
    """Spacing of the storage type at |v| (float64 tensor)."""
    e = torch.floor(torch.log2(v.abs().clamp_min(1e-30)))
    if precision == 'fp16':
        e = e.clamp_min(-14.0)             # f16 subnormals: fixed spacing 2^-24
    return torch.pow(2.0, e - MANT[precision])


@pytest.fixture(scope='module')
def ops():
    assert torch.cuda.is_available(), 'gpu tests need a GPU'
    return pkg('ops')


H16_CASES = [
    # (B, Cin, Cout, H, W, k, stride, groups, relu, residual, out_f32)
    (2, 32, 32, 128, 128, 3, 1, 1, True, True, False),      # branch-0 BasicBlock conv (one chunk, 16x16 tiles)
    (2, 64, 64, 64, 64, 3, 1, 1, True, True, False),        # branch-1 (two n-tiles per wave)
    (3, 128, 128, 32, 32, 3, 1, 1, True, False, False),     # branch-2 (chunks + n-blocks)
    (3, 256, 256, 16, 16, 3, 1, 1, False, True, False),     # branch-3 (small map)
    (2, 48, 48, 64, 64, 3, 1, 1, True, True, False),        # HRNet-W48 branch 0: ragged second n-tile (octets)
    (2, 96, 192, 32, 32, 3, 2, 1, True, False, False),      # W48 downsample
    (2, 64, 64, 64, 64, 3, 2, 1, True, False, False),       # stem conv2 / fuse downsample
    (2, 32, 128, 64, 64, 3, 2, 1, False, False, False),
    (2, 34, 256, 32, 32, 3, 1, 1, True, False, False),      # contact_layers.1.0 (Cin = 34: ragged element pair)
    (2, 34, 512, 64, 64, 3, 2, 1, True, False, False),      # towers entry
    (1, 33, 33, 48, 48, 3, 1, 1, False, False, True),       # last segm conv: fp32 logits, ragged both
    (1, 16, 64, 32, 32, 3, 1, 1, True, False, False),
    (1, 32, 32, 256, 256, 3, 1, 1, True, False, False),     # segm head first conv (padded 16 -> 32)
    (2, 64, 256, 32, 32, 1, 1, 1, True, True, False),       # bottleneck conv3 (+ residual), n-blocks from one patch
    (2, 256, 64, 32, 32, 1, 1, 1, True, False, False),
    (2, 128, 32, 16, 16, 1, 1, 1, False, False, False),     # fuse 1x1
    (2, 64, 128, 64, 64, 1, 1, 1, False, False, True),      # tower exit (params x mix): fp32 out
    (2, 64, 32, 64, 64, 1, 1, 1, False, False, True),       # center exit: fp32 out
    (2, 3, 128, 64, 64, 1, 1, 1, False, True, True),        # cam mix: Cin = 3, fp32 residual accumulated in place
    (2, 512, 512, 32, 32, 3, 1, 8, True, True, False),      # 8 grouped head towers
    (1, 32, 32, 19, 23, 3, 1, 1, True, False, False),       # ragged spatial size (tile bounds)
    (1, 40, 72, 21, 17, 3, 2, 1, False, False, False),      # ragged + stride 2
    (1, 24, 48, 9, 31, 1, 1, 1, True, False, False),
    (2, 256, 512, 32, 32, 1, 2, 1, False, False, False),    # ResNet-50 projection shortcut: 1x1 stride 2
    (1, 40, 72, 21, 17, 1, 2, 1, True, True, False),        # ... ragged, with a residual
]


@pytest.mark.parametrize('precision', ['fp16', 'bf16'])
@pytest.mark.parametrize('case', H16_CASES, ids=lambda c: 'B%d_%dto%d_%dx%d_k%ds%dg%d%s' % (c[:8] + ('_f32' if c[10] else '',)))
def test_conv2d_h16_is_the_correctly_rounded_convolution(ops, case, precision):
    B, cin, cout, H, W, k, stride, groups, relu, use_res, out_f32 = case
    g = torch.Generator().manual_seed((hash(case) + len(precision)) % (2 ** 31))
    td = TD[precision]
    x = torch.randn(B, cin, H, W, generator=g).to(td)
    w = (torch.randn(cout, cin // groups, k, k, generator=g) / np.sqrt(cin // groups * k * k)).to(td)
    b = torch.randn(cout, generator=g) * 0.1
    ref = F.conv2d(x.double(), w.double(), None, stride, k // 2, 1, groups)
    mag = F.conv2d(x.double().abs(), w.double().abs(), None, stride, k // 2, 1, groups)     # sum of |terms|
    ref = (ref + b.double()[None, :, None, None]).float().double()                          # fp32 bias add
    res = None
    if use_res:
        res = torch.randn(ref.shape, generator=g).to(torch.float32 if out_f32 else td)
        ref = ref + res.double()
    if relu:
        ref = F.relu(ref)
    xd = ops.to_nhwc16(x, precision)
    rd = None
    if res is not None:
        rd = ops.to_nhwc(res) if out_f32 else ops.to_nhwc16(res, precision)
    out = ops.conv2d_h16(xd, w.float(), b, stride=stride, relu=relu, groups=groups, cin=cin // groups, residual=rd,
                         out_f32=out_f32)
    torch.cuda.synchronize()
    got = out[..., :cout].permute(0, 3, 1, 2).cpu().double()
    err = (got - ref).abs()
    acc = 4e-7 * (mag + b.abs().double()[None, :, None, None] + 1.0)      # fp32 accumulation over K <= 2304 terms
    if out_f32:
        bound = acc + 2e-7 * ref.abs()
    else:
        bound = 0.5 * ulp(ref, precision) + acc            # ONE rounding to the storage type
    worst = (err - bound).max().item()
    assert worst <= 0, (worst, err.max().item())
    if out.shape[-1] > cout:
        assert out[..., cout:].float().abs().max().item() == 0.0     # pad channels untouched


def test_conv2d_h16_per_frame_bias(ops):
    """The mix conv's per-frame bias row (acr/model.py:160-164 pare columns) on the fp32-output kernel."""
    g = torch.Generator().manual_seed(5)
    x = torch.randn(3, 3, 64, 64, generator=g).to(torch.float16)
    w = torch.randn(109, 3, 1, 1, generator=g).to(torch.float16)
    fb = torch.randn(3, 128, generator=g)
    res = torch.randn(3, 109, 64, 64, generator=g)
    ref = F.conv2d(x.double(), w.double()) + fb[:, :109].double()[:, :, None, None] + res.double()
    wp = torch.zeros(128, 3, 1, 1)
    wp[:109] = w.float()
    out = ops.conv2d_h16(ops.to_nhwc16(x), wp, None, frame_bias=fb.cuda(), residual=ops.to_nhwc(res, cs=128), out_f32=True)
    torch.cuda.synchronize()
    assert (out[..., :109].permute(0, 3, 1, 2).cpu().double() - ref).abs().max().item() < 2e-5


# ---- program level ---------------------------------------------------------------------------------------------
def _flip_left(tables):
    t = {k: dict(v) for k, v in tables.items()}
    t['left']['shapedirs'] = t['left']['shapedirs'].copy()
    t['left']['shapedirs'][:, 0, :] *= -1
    return t


REPORT = os.path.join(ROOT, 'gpurun_out', 'h16_report.json')


def _report(key, value):
    """Measured deviations land in gpurun_out/h16_report.json (copied to profiles/ for the record)."""
    os.makedirs(os.path.dirname(REPORT), exist_ok=True)
    data = {}
    if os.path.exists(REPORT):
        with open(REPORT) as f:
            data = json.load(f)
    data[key] = value
    with open(REPORT, 'w') as f:
        json.dump(data, f, indent=1, sort_keys=True)


@pytest.mark.parametrize('precision,width', [('fp16', 32), ('bf16', 32), ('fp16', 48), ('bf16', 'resnet50'), ('fp32', 'resnet50')])
def test_every_op_of_the_16bit_program_is_correctly_rounded(precision, width, frames2):
    """Per-op parity of the resident 16-bit program (HRNet-W32 with the reference's checkpoint schema; HRNet-W48 =
    BASELINE.json configs[4]'s backbone): the program is lowered without buffer reuse, run once on the GPU, and every
    op is then re-evaluated by oracle/program.py ON THE GPU'S OWN INPUT BUFFERS.  Exact arithmetic with one rounding
    differs from the kernels' fp32 accumulation only where a value sits on a rounding boundary: every element must be
    within ONE ulp of the storage type and all but a small fraction bit-equal.  (Whole-network agreement cannot be
    tighter than the 16-bit quantisation noise - a flipped rounding anywhere decorrelates everything behind it - which is
    why the check is per op; the whole network is compared in the next test.)
    'resnet50' = BASELINE.json configs[1]'s backbone as the build defines it (schema._resnet50_backbone): its bf16 program
    (configs[1]'s dtype) and its fp32 program (there every op is compared at fp32 accumulation-order tolerance) - this is
    also the kernel-level check of the 7x7 stem, the max-pool and the strided 1x1 projections, which only it uses."""
    synth = pkg('synth')
    sd = synth.make_state_dict(seed=0, width=width)
    eng = pkg('engine').Engine(0)
    eng.load_state_dict(sd, max_batch=1, precision=precision, keep_weights=True, keep_all=True)
    x = torch.from_numpy(frames2[:1])
    B = eng.backbone_heads(x.cuda())
    torch.cuda.synchronize()
    prog = eng.program
    hipbufs = [eng.buffer(i, B).float().cpu() for i in range(len(prog['bufs']))]
    it = oprog.Interp(prog, B)
    it.bufs = [b.clone() for b in hipbufs]
    checked, skipped, worst_frac = 0, 0, 0.0
    ops = [(op, info) for op, info in zip(prog['ops'], prog['op_info']) if op.mode != oprog.MODE_POINT]

    def in_place(op):
        return op.kind == oprog.OP_POW11 or (op.kind == oprog.OP_CONV and op.res_buf == op.out_buf)

    for n, (op, info) in enumerate(ops):
        k = op.kind
        out = op.out_buf
        # A buffer that a LATER op updates in place (the cam exit before 1.1**x, the params x mix map before the cam / pare
        # term is accumulated) no longer holds this op's result on the GPU: the interpreter's value is kept instead and
        # flows into that in-place op, whose result is then compared - the chain is checked as one unit.
        chained = any(in_place(o) and o.out_buf == out for o, _ in ops[n + 1:])
        if k == oprog.OP_POW11:
            it.pow11(op)
        elif k == oprog.OP_CONV:
            it.conv(op, info)
        elif k == oprog.OP_STEM:
            it.stem(op, info, x)
        elif k == oprog.OP_FUSESUM:
            it.fuse_sum(op)
        elif k == oprog.OP_BILINEAR2X:
            it.bilinear2x(op)
        elif k == oprog.OP_MAXPOOL:
            it.maxpool(op)
        elif k == oprog.OP_COORDFILL:
            it.coordfill(op)
        elif k == oprog.OP_ATTPOOL:
            it.attpool(op)
        elif k == oprog.OP_PAREBIAS:
            it.parebias(op)
        else:
            raise AssertionError('unexpected op kind %d' % k)
        if chained:
            skipped += 1
            continue
        want, got = it.bufs[out], hipbufs[out]
        d = (want - got).abs()
        if prog['bufs'][out][4] == 0:      # fp32 output: accumulation order only
            tol = 2e-5 * max(1.0, float(want.abs().max()))
            assert float(d.max()) <= tol, (info['name'], float(d.max()), tol)
        else:
            # one ulp of the storage type + the kernels' fp32 accumulation error (absolute: it exceeds the spacing of the
            # storage type where large terms cancel to a tiny result)
            u = ulp(want.double(), precision).float() + 2e-6 * max(1.0, float(want.abs().max()))
            bad = d > u * 1.001
            assert not bool(bad.any()), (info['name'], float(d.max()), float((d / u).max()))
            frac = float((d > 0).float().mean())
            worst_frac = max(worst_frac, frac)
            assert frac < 0.02, (info['name'], frac)
        it.bufs[out] = hipbufs[out].clone()      # later ops see the GPU's values
        checked += 1
    _report('per_op_%s_w%s' % (precision, width), {'ops_checked': checked, 'ops_checked_through_their_in_place_successor': skipped,
                                                    'worst_fraction_of_elements_off_by_one_ulp': worst_frac})
    assert checked >= (320 if width != 'resnet50' else 75) and skipped <= 8
    eng.close()


@pytest.mark.parametrize('precision', ['fp16', 'bf16'])
def test_16bit_program_matches_the_op_list_interpreter(precision, synth_sd, frames2):
    """The whole 16-bit program (HRNet-W32) against oracle/program.py run end to end: backbone taps and head maps.
    Both sides implement the same one-rounding-per-layer semantics; they differ in accumulation order, and through ~60
    layers any flipped rounding decorrelates what follows, so the two agree to within the 16-bit quantisation noise
    (measured here against the fp32 oracle), not tighter."""
    eng = pkg('engine').Engine(0)
    eng.load_state_dict(synth_sd, max_batch=2, keep_taps=True, precision=precision, keep_weights=True)
    assert eng.program['precision'] == precision
    x = torch.from_numpy(frames2)
    B = eng.backbone_heads(x.cuda())
    torch.cuda.synchronize()
    hip = {k: v.cpu() for k, v in eng.head_maps(B).items()}
    it = oprog.run_program(eng.program, x)
    ref = it.head_maps()
    with torch.no_grad():
        f32 = acr_net.network(synth_sd, x)
    eps = 2.0 ** -MANT[precision]
    rep = {}
    for name, buf in eng.program['taps'].items():
        a = eng.buffer(buf, B).float().cpu()
        b = it.bufs[buf]
        assert eng.buffer(buf, B).dtype == TD[precision]
        err, scale = float((a - b).abs().max()), float(b.abs().max())
        rep['tap_' + name] = [err, scale]
        assert err <= 16 * eps * scale, (name, err, scale)
    for k in ref:
        err, scale = float((hip[k] - ref[k]).abs().max()), float(ref[k].abs().max())
        qerr = float((ref[k] - f32[k]).abs().max())           # what 16-bit storage costs against the fp32 network
        rep[k] = {'hip_vs_interpreter': err, 'scale': scale, 'interpreter_vs_fp32_oracle': qerr,
                  'hip_vs_fp32_oracle': float((hip[k] - f32[k]).abs().max())}
        assert err <= 3.0 * qerr and rep[k]['hip_vs_fp32_oracle'] <= 3.0 * qerr, (k, rep[k])
    # coordinate channels: the fp32 expression rounded once
    x34 = eng.buffer(eng.program['heads'].backbone_buf, B, 34).float().cpu()
    cm = acr_net.coord_maps(128)[0].permute(1, 2, 0).to(TD[precision]).float()
    assert torch.equal(x34[0, :, :, 32:34], cm)
    _report('interp_' + precision, rep)
    eng.close()


def _reference_frames():
    """(checkpoint seed, frame uint8 [512,512,3], golden dict, key prefix) of the 10 synthetic-checkpoint frames the real
    reference was run on (tests/golden/make_golden.py): e2e_batch1 (2) + e2e_states (4 states x 2)."""
    synth = pkg('synth')
    g1 = golden('e2e_batch1.npz')
    f2 = synth.make_frames(2, seed=0)
    out = [(0, f2[b], g1, 'f%d_' % b) for b in range(2)]
    gs = golden('e2e_states.npz')
    fs = synth.make_frames(2, seed=cases.STATE_FRAME_SEED)
    for name, seed in cases.STATE_CHECKPOINTS.items():
        out += [(seed, fs[b], gs, '%s_f%d_' % (name, b)) for b in range(2)]
    return out


@pytest.mark.parametrize('precision', ['fp16', 'bf16'])
def test_16bit_vertex_error_against_the_reference_frames(precision, mano_tables):
    """the 16-bit W32 path on the frames the REAL reference (fp32, configs/demo.yml) was run on -
    the reference's fp16 branch cannot be run here (autocast is CUDA-only), so this measures what 16-bit storage costs
    against the fp32 reference: detection flags / centers (a decision may flip when the margin is inside the 16-bit
    error: counted, not hidden), sampled parameters, vertices, joints.  The numbers go to the report; the bounds are
    sanity bounds (fp16 1e-2 m, bf16 1e-1 m on frames whose decisions agree: bf16 is 3-7 cm here depending on which
    roundings a lowering happens to take), NOT the 1e-4 m fp32 bar."""
    L = pkg('_lib')
    synth = pkg('synth')
    frames = _reference_frames()
    rep = {'frames': len(frames), 'decision_mismatch': 0, 'max_vertex_err_m': 0.0, 'max_joint_err_m': 0.0,
           'max_params_err': 0.0, 'hands_compared': 0}
    by_seed = {}
    for seed, frame, g, key in frames:
        by_seed.setdefault(seed, []).append((frame, g, key))
    for seed, items in by_seed.items():
        eng = pkg('engine').Engine(0)
        eng.load_state_dict(synth.make_state_dict(seed=seed), max_batch=len(items), precision=precision)
        eng.load_mano(_flip_left(mano_tables))
        out = eng.forward(torch.from_numpy(np.stack([i[0] for i in items])).cuda())
        torch.cuda.synchronize()
        slots = out['slots'].cpu().numpy()
        for b, (_, g, key) in enumerate(items):
            flags = g[key + 'detection_flag'].astype(bool)
            lc, rc = g[key + 'l_centers_pred'][0], g[key + 'r_centers_pred'][0]
            same = (np.array_equal(slots[b, :, L.SLOT_FLAG] > 0.5, flags) and
                    (not flags[0] or slots[b, 0, L.SLOT_FLATIND] == lc[1] * 64 + lc[0]) and
                    (not flags[1] or slots[b, 1, L.SLOT_FLATIND] == rc[1] * 64 + rc[0]))
            if not same:
                rep['decision_mismatch'] += 1
                continue
            rep['max_params_err'] = max(rep['max_params_err'], float(np.abs(
                slots[b, :, L.SLOT_PARAMS:L.SLOT_PARAMS + 109] - g[key + 'params_pred'])[flags].max(initial=0.0)))
            if flags.any():
                dv = np.linalg.norm(out['verts'][b].cpu().numpy() - g[key + 'verts'], axis=-1)[flags]
                dj = np.linalg.norm(out['joints'][b].cpu().numpy() - g[key + 'j3d'], axis=-1)[flags]
                rep['max_vertex_err_m'] = max(rep['max_vertex_err_m'], float(dv.max()))
                rep['max_joint_err_m'] = max(rep['max_joint_err_m'], float(dj.max()))
                rep['hands_compared'] += int(flags.sum())
        eng.close()
    _report('vs_reference_' + precision, rep)
    assert rep['hands_compared'] >= 8, rep
    assert rep['max_vertex_err_m'] < (1e-2 if precision == 'fp16' else 1e-1), rep


@pytest.mark.parametrize('width,precision,max_batch', [(48, 'fp32', 2), ('resnet50', 'fp32', 2), (48, 'fp32', 16), (48, 'fp16x3', 16),
                                                       ('resnet50', 'bf16x3', 16)])
def test_hrnet_w48_fp32_matches_the_oracle(width, precision, max_batch, mano_tables):
    """BASELINE.json configs[4]'s backbone (HRNet-W48: 48/96/192/384, heads on 48 + 2 channels) in fp32 against
    oracle/acr_net.py, which is state-dict driven and needs no change for the wider network - NO REFERENCE ORACLE (the
    reference hard-wires W32, acr/model.py:797-819); the W32 reading of the same code is pinned to the reference.
    'resnet50': configs[1]'s backbone (build-defined: torchvision's ResNet-50 trunk + three bilinear x2 / conv3x3 stages,
    heads on 64 + 2 channels) against oracle/acr_net.resnet50_backbone - equally without a reference oracle.
    max_batch 16: the LARGE-batch lowering (four-wave F(2x4,3x3), polyphase stride 2, streaming 1x1) of the wider networks;
    'fp16x3' / 'bf16x3': the split-operand programs bench.py reports for configs[4] / configs[1] (other_configs) - under the
    SAME tolerances as fp32 (head maps 1e-4 relative, vertices 1e-4 m against the fp32 oracle)."""
    synth = pkg('synth')
    sd = synth.make_state_dict(seed=0, width=width)
    x = torch.from_numpy(synth.make_frames(2, seed=0))
    eng = pkg('engine').Engine(0)
    eng.load_state_dict(sd, max_batch=max_batch, precision=precision)
    assert eng.program['width'] == width
    if precision != 'fp32':
        assert sum(i.get('kernel') in ('conv_x3_kernel', 'conv_x3p_kernel') for i in eng.program['op_info']) >= 60
    eng.load_mano(_flip_left(mano_tables))
    out = eng.forward(x.cuda())
    torch.cuda.synchronize()
    hip = {k: v.cpu() for k, v in eng.head_maps(2).items()}
    with torch.no_grad():
        ref = acr_net.network(sd, x)
    for k in ref:
        err, scale = float((hip[k] - ref[k]).abs().max()), float(ref[k].abs().max())
        assert err < 1e-4 * max(1.0, scale), (k, err, scale)
    slots = odec.decode(ref)
    t = _flip_left(mano_tables)
    n = 0
    for b in range(2):
        for h, name in ((0, 'left'), (1, 'right')):
            if slots['flag'][b, h]:
                v, j, _ = omano.mano_forward(t[name], name, slots['poses'][b, h:h + 1], slots['betas'][b, h:h + 1])
                assert np.abs(out['verts'][b, h].cpu().numpy() - v[0]).max() < 1e-4
                n += 1
    _report('%s_%s_b%d_hands' % ('w48' if width == 48 else width, precision, max_batch), n)
    assert n >= 2
    eng.close()


def test_config4_workload_w48_fp16_batch64_with_fp16_mano(mano_tables):
    """BASELINE.json configs[4]'s per-GPU workload: 64 frames, HRNet-W48, 16-bit program, fp16 MANO LBS
    (ACRMI_OPT_MANO_FP16) - HIP vs OWN oracle (oracle/program.py on frames 0 and 63 of the batch; no reference oracle,
    see module docstring).  Vertices: against the fp32 MANO oracle on the interpreter's decoded parameters."""
    synth = pkg('synth')
    L = pkg('_lib')
    B = 64
    sd = synth.make_state_dict(seed=0, width=48)
    frames = synth.make_frames(B, seed=3)
    eng = pkg('engine').Engine(0)
    eng.load_state_dict(sd, max_batch=B, precision='fp16', keep_weights=True)
    t = _flip_left(mano_tables)
    eng.load_mano(t)
    eng.set_mano_fp16(True)
    out = eng.forward(torch.from_numpy(frames).cuda())
    torch.cuda.synchronize()
    pick = [0, B - 1]
    it = oprog.run_program(eng.program, torch.from_numpy(frames[pick]))
    ref = it.head_maps()
    hip = {k: v[pick].cpu() for k, v in eng.head_maps(B).items()}
    eps = 2.0 ** -10
    for k in ref:
        err, scale = float((hip[k] - ref[k]).abs().max()), float(ref[k].abs().max())
        assert err <= 8 * eps * scale, (k, err, scale)
    slots = odec.decode(ref)
    hs = out['slots'].cpu().numpy()
    worst, n = 0.0, 0
    for i, b in enumerate(pick):
        for h, name in ((0, 'left'), (1, 'right')):
            if slots['flag'][i, h] and hs[b, h, L.SLOT_FLAG] > 0.5 and hs[b, h, L.SLOT_FLATIND] == slots['flat_ind'][i, h]:
                v, j, _ = omano.mano_forward(t[name], name, slots['poses'][i, h:h + 1], slots['betas'][i, h:h + 1])
                worst = max(worst, float(np.abs(out['verts'][b, h].cpu().numpy() - v[0]).max()))
                n += 1
    _report('config4_w48_fp16_b64', {'hands': n, 'max_vertex_abs_err_m_vs_own_oracle': worst})
    assert n >= 1 and worst < 5e-3, (n, worst)
    eng.close()


@pytest.mark.parametrize('width', [32, 'resnet50'])
def test_config1_batch_and_dtype(width, mano_tables):
    """BASELINE.json configs[1]: batch 32, bf16, ResNet-50 backbone.  The reference contains no ResNet (`--backbone
    resnet50` is a dead flag, acr/config.py:95); the build defines one (schema._resnet50_backbone) and this is its
    workload - and, as in round 2, the same batch and dtype on HRNet-W32: 32 frames through the bf16 program, every frame
    bit-equal to its own batch-1 run (frames are independent in 16 bits too), the first and the last frame's head maps
    against the op-list interpreter within the 16-bit quantisation noise.  NO REFERENCE ORACLE for either."""
    synth = pkg('synth')
    B = 32
    synth_sd = synth.make_state_dict(seed=0, width=width)
    frames = synth.make_frames(B, seed=9)
    eng = pkg('engine').Engine(0)
    eng.load_state_dict(synth_sd, max_batch=B, precision='bf16', keep_weights=True)
    assert eng.program['width'] == width
    eng.load_mano(_flip_left(mano_tables))
    x = torch.from_numpy(frames).cuda()
    out = {k: v.clone() for k, v in eng.forward(x).items()}
    maps = {k: v[[0, B - 1]].cpu() for k, v in eng.head_maps(B).items()}
    for i in (0, 13, B - 1):
        one = eng.forward(x[i:i + 1].contiguous())
        for k in ('slots', 'verts', 'joints'):
            assert torch.equal(one[k][0], out[k][i]), (i, k)
    ref = oprog.run_program(eng.program, torch.from_numpy(frames[[0, B - 1]])).head_maps()
    with torch.no_grad():
        f32 = acr_net.network(synth_sd, torch.from_numpy(frames[[0, B - 1]]))
    for k in ref:
        qerr = float((ref[k] - f32[k]).abs().max())
        assert float((maps[k] - ref[k]).abs().max()) <= 3.0 * qerr, k
    eng.close()


def test_mano_fp16_lbs_against_the_reference_vectors(mano_tables):
    """ACRMI_OPT_MANO_FP16 (BASELINE.json configs[4] "fp16 MANO LBS") on the MANO vectors captured from the real
    reference (tests/golden/mano_cases.npz): f16 blend-shape tables and skinning weights, fp32 arithmetic.  The
    deviation is reported; it must stay inside the 1e-4 m budget of the path."""
    g = golden('mano_cases.npz')
    eng = pkg('engine').Engine(0)
    eng.load_mano(_flip_left(mano_tables))
    worst = {}
    for fp16 in (False, True):
        eng.set_mano_fp16(fp16)
        w = 0.0
        for n, seed in ((1, 1), (2, 2), (16, 3)):          # the cases of tests/test_gpu_kernels.py
            poses, betas = cases.mano_inputs(n, seed)
            for side, sid in (('l', 0), ('r', 1)):
                key = 'n%d_%s_' % (n, side)
                v, j, c, _ = eng.mano(torch.from_numpy(poses), torch.from_numpy(betas), torch.full((n,), sid))
                torch.cuda.synchronize()
                w = max(w, float(np.abs(v.cpu().numpy() - g[key + 'verts']).max()),
                        float(np.abs(j.cpu().numpy() - g[key + 'joints']).max()))
        worst['fp16' if fp16 else 'fp32'] = w
    _report('mano_lbs_max_abs_err_m', worst)
    assert worst['fp32'] < 2e-6 and worst['fp16'] < 2e-5, worst      # (r5: 6.4e-5 with plain-f16 skinning weights)
    eng.close()
