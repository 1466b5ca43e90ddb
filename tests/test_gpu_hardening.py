"""GPU parity where the round-2 fixtures were blind: interior centers through the whole network
against the real reference, a trained-like ("hostile") checkpoint with per-layer Winograd error, the bench batch itself
against the oracle.  `pytest -m gpu`."""
import json
import os

import numpy as np
import pytest
import torch

import cases
from conftest import ROOT, golden, pkg
from oracle import acr_net, decode as odec, mano as omano, program as oprog

pytestmark = pytest.mark.gpu
REPORT = os.path.join(ROOT, 'gpurun_out', 'hardening_report.json')


def _report(key, value):
    os.makedirs(os.path.dirname(REPORT), exist_ok=True)
    data = {}
    if os.path.exists(REPORT):
        with open(REPORT) as f:
            data = json.load(f)
    data[key] = value
    with open(REPORT, 'w') as f:
        json.dump(data, f, indent=1, sort_keys=True)


def _flip_left(tables):
    t = {k: dict(v) for k, v in tables.items()}
    t['left']['shapedirs'] = t['left']['shapedirs'].copy()
    t['left']['shapedirs'][:, 0, :] *= -1
    return t


@pytest.mark.parametrize('name', list(cases.INTERIOR_CASES))
def test_interior_centers_through_the_network(name, mano_tables):
    """Centers >= 9 px from every border (planted center bias, tests/golden/cases.py INTERIOR_CASES) through the WHOLE
    network against the real reference (e2e_interior.npz), dense heads and point heads, the golden frames scattered in
    a batch of 8: the 5x5 NMS window, the 3x3 taps of the towers and the 9x9 point-heads window all lie inside the map."""
    g = golden('e2e_interior.npz')
    L = pkg('_lib')
    synth = pkg('synth')
    seed, lp, rp = cases.INTERIOR_CASES[name]
    B = 8
    eng = pkg('engine').Engine(0)
    eng.load_state_dict(cases.interior_state_dict(synth, name), max_batch=B)
    eng.load_mano(_flip_left(mano_tables))
    gold = synth.make_frames(2, seed=cases.STATE_FRAME_SEED)
    batch = synth.make_frames(B, seed=78, structured=True)
    spots = {0: (1, B - 1), 1: (0, B // 2)}
    for b, pos in spots.items():
        for p in pos:
            batch[p] = gold[b]
    x = torch.from_numpy(batch).cuda()
    for point in (False, True):
        eng.set_point_heads(point)
        out = eng.forward(x)
        torch.cuda.synchronize()
        slots = out['slots'].cpu().numpy()
        for b, pos in spots.items():
            key = '%s_f%d_' % (name, b)
            for p in pos:
                assert (slots[p, :, L.SLOT_FLAG] > 0.5).all()
                assert slots[p, 0, L.SLOT_FLATIND] == lp[0] * 64 + lp[1] and slots[p, 1, L.SLOT_FLATIND] == rp[0] * 64 + rp[1]
                lc, rc = g[key + 'l_centers_pred'][0], g[key + 'r_centers_pred'][0]
                assert (lc[1], lc[0]) == lp and (rc[1], rc[0]) == rp
                np.testing.assert_allclose(slots[p, :, L.SLOT_PARAMS:L.SLOT_PARAMS + 109], g[key + 'params_pred'], 2e-4, 2e-4)
                assert np.abs(out['verts'][p].cpu().numpy() - g[key + 'verts']).max() < 1e-4
                assert np.abs(out['joints'][p].cpu().numpy() - g[key + 'j3d']).max() < 1e-4
    eng.set_point_heads(False)
    eng.close()


def test_hostile_checkpoint_against_the_reference_and_per_layer_winograd_error(mano_tables, frames2):
    """VERDICT r2 2(c).  The hostile re-parametrisation of checkpoint 0 (synth.make_hostile_state_dict: running_var over
    1e-3..1e2, raw filter rows over 2.5 decades, block-internal channel magnitudes over two decades, a residual stream of
    O(50..150)) computes the SAME function, so the reference's fixtures for checkpoint 0 remain the ground truth.
    (1) end to end against the real reference (e2e_batch1.npz): decisions identical, vertices / joints within the 1e-4 m
    budget; (2) per layer: every 3x3 stride-1 (Winograd F(2x2,3x3)) convolution of the program against an exact fp64
    convolution of the GPU's own input buffer - the worst relative error over the 222 launches is reported and bounded."""
    synth = pkg('synth')
    L = pkg('_lib')
    hs = synth.make_state_dict(seed=0, law='hostile')
    eng = pkg('engine').Engine(0)
    eng.load_state_dict(hs, max_batch=2, keep_weights=True, keep_all=True, wino24=True, splitk=False)      # the large-batch lowering
    eng.load_mano(_flip_left(mano_tables))
    g = golden('e2e_batch1.npz')
    x = torch.from_numpy(frames2)
    out = eng.forward(x.cuda())
    torch.cuda.synchronize()
    slots = out['slots'].cpu().numpy()
    worst_v = 0.0
    for b in range(2):
        np.testing.assert_array_equal(slots[b, :, L.SLOT_FLAG] > 0.5, g['f%d_detection_flag' % b].astype(bool))
        lc, rc = g['f%d_l_centers_pred' % b][0], g['f%d_r_centers_pred' % b][0]
        assert slots[b, 0, L.SLOT_FLATIND] == lc[1] * 64 + lc[0] and slots[b, 1, L.SLOT_FLATIND] == rc[1] * 64 + rc[0]
        worst_v = max(worst_v, float(np.abs(out['verts'][b].cpu().numpy() - g['f%d_verts' % b]).max()),
                      float(np.abs(out['joints'][b].cpu().numpy() - g['f%d_j3d' % b]).max()))
    assert worst_v < 1e-4, worst_v
    # ---- per-layer error on the GPU's own inputs
    prog = eng.program
    B = 1
    eng.backbone_heads(x[:1].cuda())
    torch.cuda.synchronize()
    hip = [eng.buffer(i, B).float().cpu() for i in range(len(prog['bufs']))]
    it = oprog.Interp(prog, B)
    it.bufs = [b.clone() for b in hip]
    rows = []
    for i, (op, info) in enumerate(zip(prog['ops'], prog['op_info'])):
        if op.mode == oprog.MODE_POINT or op.kind != oprog.OP_CONV or op.res_buf == op.out_buf:
            continue
        later_in_place = any(o.kind == oprog.OP_CONV and o.res_buf == o.out_buf and o.out_buf == op.out_buf
                             for o in prog['ops'][i + 1:]) or any(o.kind == oprog.OP_POW11 and o.out_buf == op.out_buf
                                                                   for o in prog['ops'][i + 1:])
        if later_in_place:
            continue
        it.conv(op, info)
        n = op.groups * op.cout
        want = it.bufs[op.out_buf][..., op.out_coff:op.out_coff + n]
        got = hip[op.out_buf][..., op.out_coff:op.out_coff + n]
        scale = float(want.abs().max())
        rows.append({'op': info['name'], 'algo': info.get('algo'), 'rel_err': float((want - got).abs().max()) / max(scale, 1e-20),
                     'out_absmax': scale, 'in_absmax': float(hip[op.in_buf][..., op.in_coff:op.in_coff + op.groups * op.cin].abs().max())})
        it.bufs[op.out_buf] = hip[op.out_buf].clone()
    wino = [r for r in rows if r['algo'] and r['algo'].startswith('winograd')]
    direct = [r for r in rows if r['algo'] == 'direct']
    wino.sort(key=lambda r: -r['rel_err'])
    rep = {'end_to_end_max_vertex_joint_abs_err_m': worst_v, 'winograd_layers': len(wino), 'direct_layers': len(direct),
           'worst_winograd_rel_err': wino[0]['rel_err'], 'worst_direct_rel_err': max(r['rel_err'] for r in direct),
           'max_activation': max(r['in_absmax'] for r in rows), 'worst_winograd_layers': wino[:5]}
    _report('hostile_checkpoint', rep)
    assert len(wino) >= 200 and sum(r['algo'] == 'winograd_f2x4_3x3' for r in wino) >= 100
    # F(2x2,3x3) in fp32: ~4x the round-off of the direct form, F(2x4,3x3) ~10x; the budget is relative to the layer's largest output
    assert rep['worst_winograd_rel_err'] < 2e-5, rep
    eng.close()


@pytest.mark.parametrize('precision', ['fp16x3', 'bf16x3'])
def test_split_operand_program_on_the_hostile_checkpoint(mano_tables, frames2, precision):
    """The 'fp16x3' program (fp32 storage; the 3x3 stride-1 layers on conv_x3_kernel: operands split into f16 hi + lo, three
    products per MAC on the 16-bit matrix pipe, fp32 accumulation) on the hostile checkpoint: (1) end to end against the
    real reference (e2e_batch1.npz) - decisions identical, vertices / joints inside the 1e-4 m budget of the fp32 program;
    (2) per layer: every split-operand launch against an exact fp64 convolution of the GPU's own input buffer - the same
    bound relative to the layer's largest output is 5e-6 (the fp32 Winograd kernels are held to 2e-5)."""
    synth = pkg('synth')
    L = pkg('_lib')
    hs = synth.make_state_dict(seed=0, law='hostile')
    eng = pkg('engine').Engine(0)
    eng.load_state_dict(hs, max_batch=2, keep_weights=True, keep_all=True, wino24=True, splitk=False, precision=precision)
    tol_v, tol_layer = (1e-5, 5e-6) if precision == 'fp16x3' else (2e-5, 1e-4)      # (bf16 halves: 16-bit operands; measured 2.4e-6 m, 2.0e-5)
    eng.load_mano(_flip_left(mano_tables))
    g = golden('e2e_batch1.npz')
    x = torch.from_numpy(frames2)
    out = eng.forward(x.cuda())
    torch.cuda.synchronize()
    slots = out['slots'].cpu().numpy()
    worst_v = 0.0
    for b in range(2):
        np.testing.assert_array_equal(slots[b, :, L.SLOT_FLAG] > 0.5, g['f%d_detection_flag' % b].astype(bool))
        lc, rc = g['f%d_l_centers_pred' % b][0], g['f%d_r_centers_pred' % b][0]
        assert slots[b, 0, L.SLOT_FLATIND] == lc[1] * 64 + lc[0] and slots[b, 1, L.SLOT_FLATIND] == rc[1] * 64 + rc[0]
        worst_v = max(worst_v, float(np.abs(out['verts'][b].cpu().numpy() - g['f%d_verts' % b]).max()),
                      float(np.abs(out['joints'][b].cpu().numpy() - g['f%d_j3d' % b]).max()))
    assert worst_v < tol_v, worst_v      # (fp16x3 measured 3.7e-7 m; the fp32 program: 2.9e-7 m)
    prog = eng.program
    B = 1
    eng.backbone_heads(x[:1].cuda())
    torch.cuda.synchronize()
    hip = [eng.buffer(i, B).float().cpu() for i in range(len(prog['bufs']))]
    it = oprog.Interp(prog, B)
    it.bufs = [b.clone() for b in hip]
    rows = []
    for i, (op, info) in enumerate(zip(prog['ops'], prog['op_info'])):
        if op.mode == oprog.MODE_POINT or op.kind != oprog.OP_CONV or op.res_buf == op.out_buf or not str(info.get('algo')).startswith('split_'):
            continue
        later_in_place = any(o.kind == oprog.OP_CONV and o.res_buf == o.out_buf and o.out_buf == op.out_buf
                             for o in prog['ops'][i + 1:]) or any(o.kind == oprog.OP_POW11 and o.out_buf == op.out_buf
                                                                   for o in prog['ops'][i + 1:])
        if later_in_place:
            continue
        it.conv(op, info)
        n = op.groups * op.cout
        want = it.bufs[op.out_buf][..., op.out_coff:op.out_coff + n]
        got = hip[op.out_buf][..., op.out_coff:op.out_coff + n]
        scale = float(want.abs().max())
        rows.append({'op': info['name'], 'rel_err': float((want - got).abs().max()) / max(scale, 1e-20), 'out_absmax': scale,
                     'in_absmax': float(hip[op.in_buf][..., op.in_coff:op.in_coff + op.groups * op.cin].abs().max())})
        it.bufs[op.out_buf] = hip[op.out_buf].clone()
    rows.sort(key=lambda r: -r['rel_err'])
    rep = {'end_to_end_max_vertex_joint_abs_err_m': worst_v, 'split_operand_layers': len(rows),
           'worst_rel_err': rows[0]['rel_err'], 'max_activation': max(r['in_absmax'] for r in rows), 'worst_layers': rows[:5]}
    _report('hostile_checkpoint_' + precision, rep)
    assert len(rows) >= 150
    assert precision != 'fp16x3' or rep['max_activation'] < 65504.0          # the f16 range the split needs
    assert rep['worst_rel_err'] < tol_layer, rep      # (fp16x3 measured 1.5e-6; F(2x4,3x3) on the fp32 pipe: 1.2e-6)
    eng.close()


def test_hostile_checkpoint_on_the_small_batch_lowering(mano_tables, frames2):
    """The same hostile checkpoint on a small-batch context (F(2x2,3x3) everywhere, the low-resolution branches as
    split-K launches whose partial tiles are summed by the last arriver): decisions identical to the real reference's,
    vertices / joints within the 1e-4 m budget."""
    synth = pkg('synth')
    L = pkg('_lib')
    eng = pkg('engine').Engine(0)
    eng.load_state_dict(synth.make_state_dict(seed=0, law='hostile'), max_batch=2)
    assert sum(1 for o in eng.program['ops'] if o.kind == L.OP_CONV and o.flags & L.CONV_SPLITK) == 80
    eng.load_mano(_flip_left(mano_tables))
    g = golden('e2e_batch1.npz')
    out = eng.forward(torch.from_numpy(frames2).cuda())
    torch.cuda.synchronize()
    slots = out['slots'].cpu().numpy()
    worst_v = 0.0
    for b in range(2):
        np.testing.assert_array_equal(slots[b, :, L.SLOT_FLAG] > 0.5, g['f%d_detection_flag' % b].astype(bool))
        lc, rc = g['f%d_l_centers_pred' % b][0], g['f%d_r_centers_pred' % b][0]
        assert slots[b, 0, L.SLOT_FLATIND] == lc[1] * 64 + lc[0] and slots[b, 1, L.SLOT_FLATIND] == rc[1] * 64 + rc[0]
        worst_v = max(worst_v, float(np.abs(out['verts'][b].cpu().numpy() - g['f%d_verts' % b]).max()),
                      float(np.abs(out['joints'][b].cpu().numpy() - g['f%d_j3d' % b]).max()))
    _report('hostile_checkpoint_small_batch_lowering', {'end_to_end_max_vertex_joint_abs_err_m': worst_v})
    assert worst_v < 1e-4, worst_v
    eng.close()


@pytest.mark.parametrize('law', ['benign', 'hostile'])
def test_layer1_pair_kernel_matches_its_two_convolutions(law, frames2):
    """OP_PAIR1X1 (csrc/pair1x1.hip; large-batch fp32 programs): block i's conv3 + residual + ReLU chained with block
    i + 1's conv1 + ReLU, the second GEMM reading the first one's epilogue registers as its B operand.  The three pair ops
    of layer1 are re-evaluated by the interpreter (exact fp64 products on the GPU's own input buffers): both outputs - the
    256-channel map and the next block's 64-channel map - within fp32 accumulation-order tolerance, also on the hostile
    checkpoint (stream activations of O(100))."""
    synth = pkg('synth')
    L = pkg('_lib')
    eng = pkg('engine').Engine(0)
    eng.load_state_dict(synth.make_state_dict(seed=0, law=law), max_batch=3, keep_weights=True, keep_all=True, wino24=True,
                        splitk=False)
    prog = eng.program
    pairs = [(i, o) for i, o in enumerate(prog['ops']) if o.kind == L.OP_PAIR1X1]
    assert len(pairs) == 3
    x = torch.from_numpy(np.concatenate([frames2, frames2[:1]]))
    B = eng.backbone_heads(x.cuda())
    torch.cuda.synchronize()
    hip = [eng.buffer(i, B).float().cpu() for i in range(len(prog['bufs']))]
    it = oprog.Interp(prog, B)
    it.bufs = [b.clone() for b in hip]
    rep = {}
    for i, op in pairs:
        it.pair1x1(op, prog['op_info'][i])
        for name, buf, n in (('out', op.out_buf, 256), ('aux', op.aux_buf, 64)):
            want, got = it.bufs[buf][..., :n], hip[buf][..., :n]
            err, scale = float((want - got).abs().max()), float(want.abs().max())
            rep['%s.%s' % (prog['op_info'][i]['name'], name)] = [err, scale]
            assert err <= 2e-6 * max(1.0, scale), (prog['op_info'][i]['name'], name, err, scale)
        it.bufs[op.out_buf], it.bufs[op.aux_buf] = hip[op.out_buf].clone(), hip[op.aux_buf].clone()
    _report('pair1x1_' + law, rep)
    eng.close()


def test_the_bench_batch_against_the_oracle(synth_sd, mano_tables):
    """the 64 i.i.d.-noise frames bench.py times (synth.make_frames(64, seed=0, structured=False))
    through the HIP path and through the oracle: decisions identical on every frame, vertices / joints within 1e-4 m on
    every detected hand."""
    synth = pkg('synth')
    L = pkg('_lib')
    B = 64
    frames = synth.make_frames(B, seed=0, structured=False)
    eng = pkg('engine').Engine(0)
    eng.load_state_dict(synth_sd, max_batch=B)
    t = _flip_left(mano_tables)
    eng.load_mano(t)
    out = eng.forward(torch.from_numpy(frames).cuda())
    torch.cuda.synchronize()
    torch.set_num_threads(16)
    maps = {}
    with torch.no_grad():
        for i in range(0, B, 8):
            m = acr_net.network(synth_sd, torch.from_numpy(frames[i:i + 8]))
            for k, v in m.items():
                maps.setdefault(k, []).append(v)
    maps = {k: torch.cat(v) for k, v in maps.items()}
    slots = odec.decode(maps)
    got = out['slots'].cpu().numpy()
    np.testing.assert_array_equal(got[..., L.SLOT_FLAG] > 0.5, slots['flag'])
    np.testing.assert_array_equal(got[..., L.SLOT_FLATIND], slots['flat_ind'].astype(np.float32))
    worst, hands = 0.0, 0
    for h, side in ((0, 'left'), (1, 'right')):
        v, j, _ = omano.mano_forward(t[side], side, slots['poses'][:, h], slots['betas'][:, h])
        sel = slots['flag'][:, h]
        if sel.any():
            worst = max(worst, float(np.abs(out['verts'][:, h].cpu().numpy() - v)[sel].max()),
                        float(np.abs(out['joints'][:, h].cpu().numpy() - j)[sel].max()))
            hands += int(sel.sum())
    _report('bench_batch_vs_oracle', {'frames': B, 'hands': hands, 'max_vertex_joint_abs_err_m': worst})
    assert worst < 1e-4, worst
    eng.close()


def test_fp16x3_range_overflow_is_an_error_not_a_silent_result(synth_sd, mano_tables, frames2):
    """VERDICT r4 weak 9 / ADVICE r4: the 'fp16x3' program splits every activation into two f16 numbers, so |x| > 65504 cannot
    be represented (hi = inf, lo = -inf, the layer's output NaN - which the next ReLU would silently turn into 0).  A
    FUNCTION-PRESERVING re-parametrisation of the checkpoint - stem conv2's BatchNorm x K, layer1.0's two 1x1 convolutions
    that read it x 1/K (ReLU is positively homogeneous) - puts values of ~3e5 into layer1's input map: the split kernels flag
    it, the library writes the call's slots and meshes as NaN, Engine.check_range() raises AcrmiRangeError (ACRMI_ERANGE) once
    and the flag is cleared; the SAME checkpoint as a 'bf16x3' program (fp32's exponent range) and as the fp32 program
    reproduces the reference's end-to-end fixture; the plain checkpoint never trips the guard."""
    L = pkg('_lib')
    K = 3e5
    sd = {k: (v.clone() if hasattr(v, 'clone') else np.array(v)) for k, v in synth_sd.items()}
    for k in ('backbone.bn2.weight', 'backbone.bn2.bias'):
        sd[k] = sd[k] * K
    for k in ('backbone.layer1.0.conv1.weight', 'backbone.layer1.0.downsample.0.weight'):
        sd[k] = sd[k] / K
    x = torch.from_numpy(frames2).cuda()
    g = golden('e2e_batch1.npz')
    eng = pkg('engine').Engine(0)
    eng.load_state_dict(sd, max_batch=16, precision='fp16x3')
    eng.load_mano(_flip_left(mano_tables))
    out = eng.forward(x)
    with pytest.raises(L.AcrmiRangeError):
        eng.check_range()
    assert torch.isnan(out['slots']).all() and torch.isnan(out['verts']).all() and torch.isnan(out['joints']).all()
    eng.check_range()                                   # reported once, cleared
    eng.load_state_dict(synth_sd, max_batch=16, precision='fp16x3')      # the plain checkpoint on the same context: clean
    clean = eng.forward(x)
    eng.check_range()
    assert torch.isfinite(clean['verts']).all()
    for prec, tol in (('bf16x3', 1e-4), ('fp32', 1e-4)):
        eng.load_state_dict(sd, max_batch=16, precision=prec)
        o = eng.forward(x)
        eng.check_range()
        slots = o['slots'].cpu().numpy()
        for b in range(2):
            np.testing.assert_array_equal(slots[b, :, L.SLOT_FLAG] > 0.5, g['f%d_detection_flag' % b].astype(bool))
            assert np.abs(o['verts'][b].cpu().numpy() - g['f%d_verts' % b]).max() < tol, prec
    eng.close()
