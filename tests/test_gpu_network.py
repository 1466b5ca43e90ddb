"""GPU parity of the whole path: uint8 frames -> HRNet-W32 -> heads -> decode -> MANO, HIP vs oracle
and vs the golden vectors captured from the reference.  `pytest -m gpu`."""
import numpy as np
import pytest
import torch

import cases
from conftest import golden, pkg
from oracle import acr_net, decode as odec, mano as omano

pytestmark = pytest.mark.gpu


def _flip_left(tables):
    t = {k: dict(v) for k, v in tables.items()}
    t['left']['shapedirs'] = t['left']['shapedirs'].copy()
    t['left']['shapedirs'][:, 0, :] *= -1
    return t


@pytest.fixture(scope='module')
def engine(synth_sd, mano_tables):
    eng = pkg('engine').Engine(0)
    eng.load_state_dict(synth_sd, max_batch=2)
    eng.load_mano(_flip_left(mano_tables))
    return eng


@pytest.fixture(scope='module')
def hip_maps(engine, frames2):
    B = engine.backbone_heads(torch.from_numpy(frames2).cuda())
    torch.cuda.synchronize()
    return {k: v.cpu() for k, v in engine.head_maps(B).items()}


@pytest.fixture(scope='module')
def oracle_maps(synth_sd, frames2):
    torch.set_num_threads(max(1, torch.get_num_threads()))
    with torch.no_grad():
        return acr_net.network(synth_sd, torch.from_numpy(frames2))


def test_head_maps_match_oracle(hip_maps, oracle_maps):
    for k in ('l_center_map', 'r_center_map', 'l_params_maps', 'r_params_maps', 'l_prior_maps', 'r_prior_maps', 'segms'):
        a, b = hip_maps[k], oracle_maps[k]
        assert a.shape == b.shape, k
        err = (a - b).abs().max().item()
        scale = b.abs().max().item()
        assert err < 1e-4 * max(1.0, scale), (k, err, scale)      # fp32 re-association over ~60 conv layers


def test_split_operand_program_matches_oracle_and_reference_golden(synth_sd, mano_tables, frames2, oracle_maps):
    """The 'fp16x3' program (fp32 storage, conv_x3_kernel: split f16 operands, three products per MAC on the 16-bit matrix
    pipe) under the SAME tolerances as the fp32 program: head maps against the oracle and against the reference's golden
    vectors, detections and vertices against the reference's end-to-end fixture."""
    L = pkg('_lib')
    eng = pkg('engine').Engine(0)
    eng.load_state_dict(synth_sd, max_batch=2, precision='fp16x3', wino24=True, splitk=False)
    eng.load_mano(_flip_left(mano_tables))
    assert sum(i.get('kernel') == 'conv_x3_kernel' for i in eng.program['op_info']) >= 190
    B = eng.backbone_heads(torch.from_numpy(frames2).cuda())
    torch.cuda.synchronize()
    maps = {k: v.cpu() for k, v in eng.head_maps(B).items()}
    for k in ('l_center_map', 'r_center_map', 'l_params_maps', 'r_params_maps', 'l_prior_maps', 'r_prior_maps', 'segms'):
        a, b = maps[k], oracle_maps[k]
        err = (a - b).abs().max().item()
        assert err < 1e-4 * max(1.0, b.abs().max().item()), (k, err)
    g = golden('net_frame0.npz')
    for k in ('l_center_map', 'r_center_map'):
        np.testing.assert_allclose(maps[k][:1].numpy(), g[k], rtol=1e-4, atol=1e-4)
    ge = golden('e2e_batch1.npz')
    out = eng.forward(torch.from_numpy(frames2).cuda())
    torch.cuda.synchronize()
    slots = out['slots'].cpu().numpy()
    for b in range(2):
        np.testing.assert_array_equal(slots[b, :, L.SLOT_FLAG] > 0.5, ge['f%d_detection_flag' % b].astype(bool))
        assert np.abs(out['verts'][b].cpu().numpy() - ge['f%d_verts' % b]).max() < 1e-4
        assert np.abs(out['joints'][b].cpu().numpy() - ge['f%d_j3d' % b]).max() < 1e-4
    # the point-heads variant of a split-operand program (what acr.main.ACR.forward_batch runs by default) -
    # its center-tower convs carry their own packs and weight scale; same decisions, same meshes as the dense variant
    assert eng.has_point_heads
    dense = {k: v.clone() for k, v in out.items()}
    eng.set_point_heads(True)
    try:
        pt = {k: v.clone() for k, v in eng.forward(torch.from_numpy(frames2).cuda()).items()}
        torch.cuda.synchronize()
    finally:
        eng.set_point_heads(False)
    _assert_point_matches_dense(pt, dense)
    for b in range(2):
        assert np.abs(pt['verts'][b].cpu().numpy() - ge['f%d_verts' % b]).max() < 1e-4
    eng.close()


def test_head_maps_match_reference_golden(hip_maps):
    g = golden('net_frame0.npz')
    for k in ('l_center_map', 'r_center_map'):
        np.testing.assert_allclose(hip_maps[k][:1].numpy(), g[k], rtol=1e-4, atol=1e-4)
    for k in ('l_params_maps', 'r_params_maps', 'l_prior_maps', 'r_prior_maps', 'segms'):
        s, cs = cases.sub(hip_maps[k][:1].numpy(), 8192)
        np.testing.assert_allclose(s, g[k], rtol=1e-4, atol=2e-4)


def test_backbone_matches_reference_golden(engine, hip_maps):
    g = golden('net_frame0.npz')
    x = engine.buffer(engine.program['heads'].backbone_buf, 2, 34).cpu()
    bb = x[:1, :, :, :32].permute(0, 3, 1, 2).contiguous().numpy()
    s, cs = cases.sub(bb)
    np.testing.assert_allclose(s, g['tap_backbone'], rtol=1e-4, atol=2e-4)
    # coord maps appended in channels 32/33 (acr/model.py:52,340-369)
    cm = acr_net.coord_maps(128)[0].permute(1, 2, 0)
    assert torch.equal(x[0, :, :, 32:34], cm) and torch.equal(x[1, :, :, 32:34], cm)


def test_forward_matches_reference_end_to_end(engine, frames2):
    """Vertices/joints within 1e-4 m of the reference (BASELINE.json north_star tolerance)."""
    g = golden('e2e_batch1.npz')
    L = pkg('_lib')
    offsets = torch.tensor([[512., 512, 0, 0, 0, 0, 0, 0, 0, 0]]).repeat(2, 1)
    out = engine.forward(torch.from_numpy(frames2).cuda(), offsets=offsets.cuda(), project=True)
    torch.cuda.synchronize()
    slots = out['slots'].cpu().numpy()
    for b in range(2):
        np.testing.assert_array_equal(slots[b, :, L.SLOT_FLAG] > 0.5, g['f%d_detection_flag' % b].astype(bool))
        lc, rc = g['f%d_l_centers_pred' % b][0], g['f%d_r_centers_pred' % b][0]
        assert slots[b, 0, L.SLOT_FLATIND] == lc[1] * 64 + lc[0] and slots[b, 1, L.SLOT_FLATIND] == rc[1] * 64 + rc[0]
        np.testing.assert_allclose(slots[b, :, L.SLOT_PARAMS:L.SLOT_PARAMS + 109], g['f%d_params_pred' % b], 2e-4, 2e-4)
        assert np.abs(out['verts'][b].cpu().numpy() - g['f%d_verts' % b]).max() < 1e-4
        assert np.abs(out['joints'][b].cpu().numpy() - g['f%d_j3d' % b]).max() < 1e-4
        np.testing.assert_allclose(out['verts_camed'][b].cpu().numpy(), g['f%d_verts_camed' % b], 1e-3, 2e-4)
        np.testing.assert_allclose(out['pj2d'][b].cpu().numpy(), g['f%d_pj2d' % b], 1e-3, 2e-4)
        np.testing.assert_allclose(out['pj2d_org'][b].cpu().numpy(), g['f%d_pj2d_org' % b], 1e-3, 5e-2)


def test_forward_matches_oracle_and_is_batch_invariant(engine, frames2, oracle_maps, mano_tables):
    out2 = engine.forward(torch.from_numpy(frames2).cuda())
    out1 = engine.forward(torch.from_numpy(frames2[1:2]).cuda())
    torch.cuda.synchronize()
    assert torch.equal(out2['verts'][1], out1['verts'][0])        # frames are independent (eval-mode BN)
    slots = odec.decode(oracle_maps)
    t = _flip_left(mano_tables)
    for b in range(2):
        for h, name in ((0, 'left'), (1, 'right')):
            v, j, _ = omano.mano_forward(t[name], name, slots['poses'][b, h:h + 1], slots['betas'][b, h:h + 1])
            assert np.abs(out2['verts'][b, h].cpu().numpy() - v[0]).max() < 1e-4
            assert np.abs(out2['joints'][b, h].cpu().numpy() - j[0]).max() < 1e-4


def _assert_point_matches_dense(pt, dense):
    L = pkg('_lib')
    ps, ds = pt['slots'].cpu().numpy(), dense['slots'].cpu().numpy()
    np.testing.assert_array_equal(ps[..., L.SLOT_FLAG], ds[..., L.SLOT_FLAG])
    np.testing.assert_array_equal(ps[..., L.SLOT_FLATIND], ds[..., L.SLOT_FLATIND])
    np.testing.assert_array_equal(ps[..., L.SLOT_SCORE], ds[..., L.SLOT_SCORE])
    # fp32 FMA chains over the same 5 conv layers in a different summation order (direct vs Winograd/MFMA)
    np.testing.assert_allclose(ps[..., L.SLOT_PARAMS:L.SLOT_PARAMS + 109], ds[..., L.SLOT_PARAMS:L.SLOT_PARAMS + 109],
                               rtol=5e-5, atol=5e-5)
    assert (pt['verts'] - dense['verts']).abs().max().item() < 2e-5
    assert (pt['joints'] - dense['joints']).abs().max().item() < 2e-5


def test_point_heads_match_dense_path_and_reference(engine, frames2):
    """the params/cam/prior towers + mix conv evaluated only at the pixels the decode samples
    (acr/result_parser.py:49-57,141-145) give the dense path's slots and meshes, and the reference's."""
    g = golden('e2e_batch1.npz')
    x = torch.from_numpy(frames2).cuda()
    dense = {k: v.clone() for k, v in engine.forward(x).items()}
    hl = engine.program['heads']
    for si in range(2):      # whatever the point variant does not write must not be read either
        engine.buffer(hl.params_buf[si], 2).fill_(float('nan'))
        engine.buffer(hl.prior_buf[si], 2).fill_(float('nan'))
    engine.set_point_heads(True)
    try:
        pt = {k: v.clone() for k, v in engine.forward(x).items()}
        torch.cuda.synchronize()
    finally:
        engine.set_point_heads(False)
    assert dense['slots'][..., 0].sum().item() == 4          # both hands of both golden frames: prior gate exercised
    _assert_point_matches_dense(pt, dense)
    for b in range(2):
        assert np.abs(pt['verts'][b].cpu().numpy() - g['f%d_verts' % b]).max() < 1e-4
        assert np.abs(pt['joints'][b].cpu().numpy() - g['f%d_j3d' % b]).max() < 1e-4
    again = engine.forward(x)                                # the dense program repairs the poisoned maps
    assert torch.equal(again['verts'], dense['verts'])


def test_point_heads_at_map_borders_and_without_detections(engine, synth_sd):
    """Placeholder rows sample pixel 0 (acr/result_parser.py:106-120): the 9x9 window then hangs over the map corner,
    where the dense convolutions see zero padding at every layer.  Also centers forced onto all four borders."""
    synth = pkg('synth')
    x = torch.from_numpy(synth.make_frames(2, seed=3, structured=False)).cuda()
    dense = {k: v.clone() for k, v in engine.forward(x).items()}
    engine.set_point_heads(True)
    try:
        pt = {k: v.clone() for k, v in engine.forward(x).items()}
    finally:
        engine.set_point_heads(False)
    _assert_point_matches_dense(pt, dense)
    # border centers: plant the peaks in the resident center maps and re-run the point heads on them
    hl = engine.program['heads']
    B = engine.backbone_heads(x)
    _check_point_heads_at(engine, x, [((0, 63), (5, 60)), ((63, 0), (62, 7))])       # all four borders
    # interior centers: the 9x9 window of the point heads fully inside the map, 3x3 taps never see padding (r2 2b)
    _check_point_heads_at(engine, x, [((20, 30), (32, 33)), ((40, 12), (32, 33))])


def test_point_heads_under_reference_batch_semantics(engine, synth_sd):
    """Round 6: with ACRMI_OPT_BATCH_PRIOR the batch-wide rule (acr/result_parser.py:42-47,131) can open the prior gate for a
    frame whose OWN centers are more than 32 px apart (it looks at the first left- / right-detected frames only).  The point
    heads must then have evaluated that frame's prior point although its per-frame gate is closed.  Planted centers: frame 0
    near (decides the batch), frame 1 far: point heads + gated decode == gated decode of the dense maps."""
    synth, rp = pkg('synth'), pkg('acr.result_parser')
    x = torch.from_numpy(synth.make_frames(2, seed=3, structured=False)).cuda()
    hl = engine.program['heads']
    B = engine.backbone_heads(x)                              # dense maps of both frames
    spots = [((20, 30), (24, 33)), ((5, 6), (58, 57))]        # frame 1: 73 px apart
    for b, (l, r) in enumerate(spots):
        for s, (y, xx) in enumerate((l, r)):
            cm = engine.buffer(hl.center_buf[s], B, 1)
            cm[b].fill_(0.0)
            cm[b, y, xx, 0] = 0.9
    first = engine.decode(B)
    gate = engine.prior_gate(first)
    assert gate.tolist() == [1, 1] and rp.reference_prior_gate(first).tolist() == [1, 1]
    want = engine.decode(B, prior_gate=gate).clone()
    per_frame = engine.decode(B).clone()
    assert (want[1] - per_frame[1]).abs().max().item() > 1e-4          # frame 1: the batch rule adds a prior its own rule would not
    for s in range(2):
        engine.buffer(hl.params_buf[s], B).fill_(float('nan'))
        engine.buffer(hl.prior_buf[s], B).fill_(float('nan'))
    engine.set_batch_semantics('reference')
    try:
        engine.run_point_heads(B)
        got = engine.decode(B, prior_gate=gate)
        torch.cuda.synchronize()
    finally:
        engine.set_batch_semantics('frame')
    assert not torch.isnan(got).any()
    assert (got - want).abs().max().item() < 2e-4                      # the point towers' fp32 round-off against the dense maps
    engine.forward(x)                                                  # the dense program repairs the poisoned maps


def _check_point_heads_at(engine, x, spots):
    """spots: per frame (left (y,x), right (y,x)), <= 32 px apart (the prior path runs): peaks planted in the resident
    center maps, point heads re-run on them and compared with the decode of the dense maps."""
    hl = engine.program['heads']
    B = engine.backbone_heads(x)
    for b, (l, r) in enumerate(spots):
        for s, (y, xx) in enumerate((l, r)):
            cm = engine.buffer(hl.center_buf[s], B, 1)
            cm[b].fill_(0.0)
            cm[b, y, xx, 0] = 0.9
    want = engine.decode(B).clone()
    for s in range(2):
        engine.buffer(hl.params_buf[s], B).fill_(float('nan'))
        engine.buffer(hl.prior_buf[s], B).fill_(float('nan'))
    engine.run_point_heads(B)
    got = engine.decode(B)
    torch.cuda.synchronize()
    L = pkg('_lib')
    assert want[..., L.SLOT_FLAG].sum().item() == 4
    for b, (l, r) in enumerate(spots):
        assert want[b, 0, L.SLOT_FLATIND].item() == l[0] * 64 + l[1] and want[b, 1, L.SLOT_FLATIND].item() == r[0] * 64 + r[1]
    np.testing.assert_allclose(got.cpu().numpy(), want.cpu().numpy(), rtol=5e-5, atol=5e-5)


def test_parallel_lanes_are_bit_identical(engine, frames2):
    """ACRMI_OPT_LANES: the program's independent chains (HRNet branches, head towers, segm / part heads - found from
    the ops' buffer reads and writes) on 2..8 HIP streams forked from / joined to the caller's stream give bit-identical
    maps, slots and meshes to the single-stream run, in the dense and the point-heads variant, call after call."""
    x = torch.from_numpy(frames2).cuda()
    other = torch.from_numpy(pkg('synth').make_frames(2, seed=5, structured=True)).cuda()
    offsets = torch.tensor([[512., 512, 0, 0, 0, 0, 0, 0, 0, 0]]).repeat(2, 1).cuda()
    engine.set_lanes(1)
    B = engine.backbone_heads(x)
    maps1 = {k: v.clone() for k, v in engine.head_maps(B).items()}
    want = {}
    for point in (False, True):
        engine.set_point_heads(point)
        want[point] = [{k: v.clone() for k, v in engine.forward(f, offsets=offsets, project=True).items()} for f in (x, other)]
    try:
        for lanes in (2, 4, 8):
            engine.set_lanes(lanes)
            B = engine.backbone_heads(x)
            for k, v in engine.head_maps(B).items():
                assert torch.equal(v, maps1[k]), (lanes, k)
            for point in (False, True):
                engine.set_point_heads(point)
                for rep in range(3):                     # back-to-back calls: the next call's lanes may not overtake
                    for f, w in zip((x, other), want[point]):
                        out = engine.forward(f, offsets=offsets, project=True)
                        for k in w:
                            assert torch.equal(out[k], w[k]), (lanes, point, rep, k)
        with pytest.raises(ValueError):
            engine.set_lanes(9)
    finally:
        engine.set_point_heads(False)
        engine.set_lanes(0)


def test_tune_lanes_picks_a_measured_candidate_and_keeps_the_results(engine, frames2):
    """Engine.tune_lanes: lane count and assignment (structural / planned from measured op times, ACRMI_OPT_LANE_PLAN)
    are chosen by timing the candidates in the process's actual stream state; the winner is one of them and the outputs do
    not depend on the choice - checked for every candidate, planned schedules included."""
    x = torch.from_numpy(frames2).cuda()
    engine.set_lanes(1)
    want = {k: v.clone() for k, v in engine.forward(x).items()}
    try:
        best, ms = engine.tune_lanes(1, candidates=(1, 2, 4), calls=3)
        assert set(ms) == {(1, False), (2, False), (2, True), (4, False), (4, True)}
        assert best in ms and ms[best] == min(ms.values())
        assert engine.lanes == (0 if best[0] == 4 else best[0]) and engine.lane_plan == best[1]
        for n, planned in ms:
            engine.set_lane_plan(planned)
            engine.set_lanes(n)
            for rep in range(2):
                out = engine.forward(x)
                for k in want:
                    assert torch.equal(out[k], want[k]), (n, planned, rep, k)
    finally:
        engine.set_lane_plan(True)
        engine.set_lanes(0)


def test_full_size_batch64_properties(synth_sd, mano_tables, frames2):
    """BASELINE.json's bench configuration (batch 64, 512x512): size-independent properties of the whole path.
    Frames are independent, so (i) the two golden frames planted anywhere in the batch of 64 reproduce the reference's
    vertices within the 1e-4 m budget and are bit-identical to their batch-2 run and to each other's copies - whichever
    CU / work item they land on; (ii) running the batch twice is bit-identical (no races, no atomics-order effects);
    (iii) a permutation of the frames permutes the outputs."""
    g = golden('e2e_batch1.npz')
    eng = pkg('engine').Engine(0)
    eng.load_state_dict(synth_sd, max_batch=64)
    eng.load_mano(_flip_left(mano_tables))
    rnd = pkg('synth').make_frames(64, seed=9, structured=False)
    spots = {0: (0, 17, 63), 1: (1, 31, 62)}          # golden frame -> positions in the batch
    for b, pos in spots.items():
        for p in pos:
            rnd[p] = frames2[b]
    x = torch.from_numpy(rnd).cuda()
    out = {k: v.clone() for k, v in eng.forward(x).items()}
    again = eng.forward(x)
    torch.cuda.synchronize()
    for k in ('slots', 'verts', 'joints'):
        assert torch.equal(out[k], again[k]), k
        assert torch.isfinite(out[k]).all(), k
    small = eng.forward(torch.from_numpy(frames2).cuda())
    for b, pos in spots.items():
        for p in pos:
            assert torch.equal(out['verts'][p], small['verts'][b])
            assert torch.equal(out['slots'][p], small['slots'][b])
            assert np.abs(out['verts'][p].cpu().numpy() - g['f%d_verts' % b]).max() < 1e-4
            assert np.abs(out['joints'][p].cpu().numpy() - g['f%d_j3d' % b]).max() < 1e-4
    eng.set_lanes(1)                                         # (iv) single stream vs the default lanes: bit-identical
    lan = eng.forward(x)
    eng.set_lanes(0)
    for k in ('slots', 'verts', 'joints'):
        assert torch.equal(lan[k], out[k]), k
    eng.set_point_heads(True)                                # (v) the point-heads variant agrees on all 128 hands
    pt = {k: v.clone() for k, v in eng.forward(x).items()}
    eng.set_point_heads(False)
    _assert_point_matches_dense(pt, out)
    perm = torch.randperm(64, generator=torch.Generator().manual_seed(1))
    outp = eng.forward(x[perm.cuda()].contiguous())
    torch.cuda.synchronize()
    assert torch.equal(outp['verts'], out['verts'][perm.cuda()])
    eng.close()


def test_intermediate_backbone_taps_match_reference_golden(synth_sd, frames2):
    """not only the backbone output but the taps in between - stem, layer1, stage2 and stage3
    (branch 0) - equal the reference's activations (net_frame0.npz), so a compensating error pair cannot hide."""
    g = golden('net_frame0.npz')
    eng = pkg('engine').Engine(0)
    eng.load_state_dict(synth_sd, max_batch=1, keep_taps=True)
    eng.backbone_heads(torch.from_numpy(frames2[:1]).cuda())
    torch.cuda.synchronize()
    chans = {'stem': 64, 'layer1': 256, 'stage2': 32, 'stage3': 32}
    for name, c in chans.items():
        t = eng.buffer(eng.program['taps'][name], 1, c).cpu().permute(0, 3, 1, 2).contiguous().numpy()
        sub, sums = cases.sub(t)
        scale = max(1.0, float(np.abs(g['tap_' + name]).max()))
        assert np.abs(sub - g['tap_' + name]).max() < 1e-4 * scale, name
        np.testing.assert_allclose(sums, g['tap_%s_sum' % name], rtol=2e-5)
    # pinning the taps changes buffer aliasing only: the final maps are bit-identical to the default lowering
    plain = pkg('engine').Engine(0)
    plain.load_state_dict(synth_sd, max_batch=1)
    plain.backbone_heads(torch.from_numpy(frames2[:1]).cuda())
    for k, v in plain.head_maps(1).items():
        assert torch.equal(v, eng.head_maps(1)[k]), k
    eng.close()
    plain.close()


def test_stem_kernel_lowering_matches_the_two_op_lowering(synth_sd, mano_tables, frames2):
    """packer.STEM_FUSED: conv1 inside stem_kernel (reads the uint8 frame) vs round 1's u8norm + generic conv - the same
    fp32 products in another summation order: the stem tap agrees to round-off, the results to well inside the budget."""
    packer = pkg('packer')
    x = torch.from_numpy(frames2).cuda()
    res, taps = [], []
    try:
        for fused in (True, False):
            packer.STEM_FUSED = fused
            eng = pkg('engine').Engine(0)
            eng.load_state_dict(synth_sd, max_batch=2, keep_taps=True)
            assert any(o.kind == pkg('_lib').OP_STEM for o in eng.program['ops']) == fused
            eng.load_mano(_flip_left(mano_tables))
            res.append({k: v.clone() for k, v in eng.forward(x).items()})
            taps.append(eng.buffer(eng.program['taps']['stem'], 2, 64).clone())
            eng.close()
    finally:
        packer.STEM_FUSED = True
    scale = float(taps[1].abs().max())
    assert float((taps[0] - taps[1]).abs().max()) < 2e-5 * max(1.0, scale)
    assert torch.equal(res[0]['slots'][..., :2], res[1]['slots'][..., :2])          # flags, centers
    assert float((res[0]['verts'] - res[1]['verts']).abs().max()) < 2e-5


@pytest.mark.parametrize('name', list(cases.STATE_CHECKPOINTS))
def test_detection_states_through_the_network(name, mano_tables):
    """left-only / right-only / none / both-with-prior (centres <= 32 px apart) through the WHOLE
    network against the real reference (e2e_states.npz), the golden frames scattered in a larger batch - batch 64 for
    the both-hands state, 8 for the others - with the rest of the batch checked against the frame's own batch-1 run."""
    g = golden('e2e_states.npz')
    L = pkg('_lib')
    synth = pkg('synth')
    B = 64 if name == 'both_near' else 8
    eng = pkg('engine').Engine(0)
    eng.load_state_dict(synth.make_state_dict(seed=cases.STATE_CHECKPOINTS[name]), max_batch=B)
    eng.load_mano(_flip_left(mano_tables))
    gold = synth.make_frames(2, seed=cases.STATE_FRAME_SEED)
    batch = synth.make_frames(B, seed=77, structured=True)
    spots = {0: (1, B - 1), 1: (0, B // 2)}
    for b, pos in spots.items():
        for p in pos:
            batch[p] = gold[b]
    out = eng.forward(torch.from_numpy(batch).cuda())
    torch.cuda.synchronize()
    slots = out['slots'].cpu().numpy()
    for b, pos in spots.items():
        key = '%s_f%d_' % (name, b)
        for p in pos:
            np.testing.assert_array_equal(slots[p, :, L.SLOT_FLAG] > 0.5, g[key + 'detection_flag'].astype(bool))
            lc, rc = g[key + 'l_centers_pred'][0], g[key + 'r_centers_pred'][0]
            assert slots[p, 0, L.SLOT_FLATIND] == lc[1] * 64 + lc[0] and slots[p, 1, L.SLOT_FLATIND] == rc[1] * 64 + rc[0]
            np.testing.assert_allclose(slots[p, :, L.SLOT_PARAMS:L.SLOT_PARAMS + 109], g[key + 'params_pred'], 2e-4, 2e-4)
            if name != 'none':                   # the reference does not run MANO when nothing is detected
                assert np.abs(out['verts'][p].cpu().numpy() - g[key + 'verts']).max() < 1e-4
                assert np.abs(out['joints'][p].cpu().numpy() - g[key + 'j3d']).max() < 1e-4
    one = eng.forward(torch.from_numpy(batch[3:4]).cuda())        # any other frame: batch-invariant
    assert torch.equal(one['slots'][0], out['slots'][3]) and torch.equal(one['verts'][0], out['verts'][3])
    eng.close()


def test_config3_1080p_shard_of_32_frames(synth_sd, mano_tables):
    """BASELINE.json configs[3] per-GPU workload (batch 128 over 4 GPUs = 32 frames per GPU): 32 raw 1080p BGR frames
    in HBM -> acrmi_preprocess -> acrmi_forward, at fp32 tolerance: the pre-processed frames equal the oracle's
    (OpenCV restatement) bit for bit on all 32, vertices/joints are within 1e-4 m of the oracle network + MANO on all
    32, and every frame equals its own batch-1 run."""
    from oracle import preprocess as opre
    synth = pkg('synth')
    eng = pkg('engine').Engine(0)
    eng.load_state_dict(synth_sd, max_batch=32)
    eng.load_mano(_flip_left(mano_tables))
    rs = np.random.RandomState(11)
    small = synth.make_frames(32, seed=21, size=270)[:, :, :, ::-1]                   # BGR content
    raw = np.empty((32, 1080, 1920, 3), np.uint8)
    for i in range(32):
        up = np.kron(small[i, :, :240], np.ones((4, 8, 1), np.uint8))                   # 1080 x 1920
        raw[i] = np.clip(up.astype(np.int16) + rs.randint(-6, 7, up.shape), 0, 255).astype(np.uint8)
    dev_raw = torch.from_numpy(raw).cuda()
    img, offsets = pkg('ops').preprocess(dev_raw)
    assert offsets[0].tolist() == [1920, 1920, 0, 0, 0, 0, 420, 0, 420, 0]
    out = eng.forward(img, offsets=offsets.cuda(), project=True)
    torch.cuda.synchronize()
    host_img = img.cpu().numpy()
    picks = tuple(range(32))            # every frame of the shard (~5 s of oracle)
    want = {i: opre.img_preprocess(raw[i])[0] for i in range(32)}
    for i in range(32):
        np.testing.assert_array_equal(host_img[i], want[i])
    torch.set_num_threads(max(8, torch.get_num_threads()))
    with torch.no_grad():
        maps = acr_net.network(synth_sd, torch.from_numpy(np.stack([want[i] for i in picks])))
    slots = odec.decode(maps)
    t = _flip_left(mano_tables)
    L = pkg('_lib')
    got = out['slots'].cpu().numpy()
    for k, i in enumerate(picks):
        np.testing.assert_array_equal(got[i, :, L.SLOT_FLAG] > 0.5, slots['flag'][k])
        np.testing.assert_array_equal(got[i, :, L.SLOT_FLATIND], slots['flat_ind'][k].astype(np.float32))
        for h, side in ((0, 'left'), (1, 'right')):
            v, j, _ = omano.mano_forward(t[side], side, slots['poses'][k, h:h + 1], slots['betas'][k, h:h + 1])
            assert np.abs(out['verts'][i, h].cpu().numpy() - v[0]).max() < 1e-4
            assert np.abs(out['joints'][i, h].cpu().numpy() - j[0]).max() < 1e-4
            _, pj, org = omano.project(v, j, slots['cam'][k, h:h + 1], np.array([[1920, 1920, 0, 0, 0, 0, 420, 0, 420, 0]], np.float32))
            np.testing.assert_allclose(out['pj2d_org'][i, h].cpu().numpy(), org[0], rtol=1e-4, atol=0.1)
    for i in (1, 17, 30):
        one = eng.forward(img[i:i + 1].contiguous(), offsets=offsets[i:i + 1].cuda(), project=True)
        for k in ('slots', 'verts', 'joints', 'pj2d_org'):
            assert torch.equal(one[k][0], out[k][i]), (i, k)
    eng.close()


def test_parebias_kernel_matches_oracle(synth_sd, hip_maps):
    """N10 (LocallyConnected2d, acr/model.py:559-569) + the shape Linear + the mix conv's pare columns
    (acr/model.py:141-164) as a unit: acrmi_parebias on the reference's raw weights and pooled features vs
    oracle.acr_net.pare_vector, for both sides, real pooled features and random ones."""
    ops = pkg('ops')
    torch.manual_seed(5)
    B = 5
    wc = torch.randn(B, 256, 32, 1) * 0.7
    ws = torch.randn(B, 64, 32) * 0.7
    pooled = torch.cat([wc[..., 0], ws], 1).transpose(1, 2).contiguous()             # [B,32,320]
    for side, lc, mix, part0 in (('l', 2, 4, 16), ('r', 3, 5, 0)):
        pare = acr_net.pare_vector(synth_sd, side, wc, ws).double().numpy()          # [B,106]
        wm = synth_sd['contact_layers.%d.weight' % mix].double().numpy().reshape(109, 218)
        want = synth_sd['contact_layers.%d.bias' % mix].double().numpy() + pare @ wm[:, 112:].T
        got = ops.parebias(pooled.cuda(), synth_sd['contact_layers.%d.weight' % lc].numpy().reshape(6, 256, 16),
                           synth_sd['cam_shape_layers.%d.weight' % lc].numpy(), synth_sd['cam_shape_layers.%d.bias' % lc].numpy(),
                           wm[:, 112:], synth_sd['contact_layers.%d.bias' % mix].numpy(), part0).cpu().numpy()
        assert got.shape == (B, 112) and (got[:, 109:] == 0).all()
        np.testing.assert_allclose(got[:, :109], want, rtol=2e-5, atol=2e-5)
        # a spatially constant pare vector is what the mix conv sees: LC offsets depend on this side's 16 parts only
        other = pooled.clone()
        other[:, (0 if part0 else 16):(16 if part0 else 32)] += 1.0
        got2 = ops.parebias(other.cuda(), synth_sd['contact_layers.%d.weight' % lc].numpy().reshape(6, 256, 16),
                            synth_sd['cam_shape_layers.%d.weight' % lc].numpy(), synth_sd['cam_shape_layers.%d.bias' % lc].numpy(),
                            wm[:, 112:], synth_sd['contact_layers.%d.bias' % mix].numpy(), part0).cpu().numpy()
        np.testing.assert_array_equal(got, got2)
    with pytest.raises(ValueError):
        ops.parebias(pooled[:, :, :100].cuda(), np.zeros((6, 256, 16)), np.zeros((10, 1024)), np.zeros(10), np.zeros((109, 106)),
                     np.zeros(109), 0)


def test_decision_margins_at_batch_64(synth_sd, hip_maps, oracle_maps):
    """VERDICT r1 1(d) / SURVEY.md 7: how close the discrete decisions (NMS equality, arg-max, > 0.35) come to
    flipping.  Over 64 structured frames x 2 sides: the margin between the best and the second-best NMS survivor
    and between the best score and the threshold, against the measured GPU-vs-oracle error of the center maps.
    The histogram goes to gpurun_out/r02_tie_rate.json (DESIGN.md quotes it)."""
    import json
    import os
    import torch.nn.functional as F
    eng = pkg('engine').Engine(0)
    eng.load_state_dict(synth_sd, max_batch=64)
    x = torch.from_numpy(pkg('synth').make_frames(64, seed=123, structured=True)).cuda()
    B = eng.backbone_heads(x)
    maps = eng.head_maps(B)
    err = max((hip_maps[k] - oracle_maps[k]).abs().max().item() for k in ('l_center_map', 'r_center_map'))
    margins, thr = [], []
    for k in ('l_center_map', 'r_center_map'):
        c = maps[k].float()
        keep = (F.max_pool2d(c, 5, 1, 2) == c).float() * c
        top = torch.topk(keep.reshape(B, -1), 2, dim=1).values
        margins += (top[:, 0] - top[:, 1]).cpu().tolist()
        thr += (top[:, 0] - 0.35).abs().cpu().tolist()
    edges = [0, 1e-5, 1e-4, 1e-3, 1e-2, 1e-1, 1e9]
    hist = np.histogram(margins, bins=edges)[0].tolist()
    rep = {'frames': 64, 'decisions': len(margins), 'center_map_max_abs_err_vs_oracle': err,
           'top2_margin_min': min(margins), 'top2_margin_hist_edges': edges[:-1], 'top2_margin_hist': hist,
           'threshold_margin_min': min(thr), 'ties_within_10x_err': int(sum(m < 10 * err for m in margins))}
    os.makedirs('gpurun_out', exist_ok=True)
    with open('gpurun_out/r02_tie_rate.json', 'w') as f:
        json.dump(rep, f, indent=1)
    assert err < 5e-5
    assert rep['ties_within_10x_err'] == 0 and min(thr) > 10 * err, rep
    eng.close()


def test_engine_argument_errors(engine):
    with pytest.raises(ValueError):
        engine.forward(torch.zeros(1, 256, 256, 3, dtype=torch.uint8).cuda())
    with pytest.raises(ValueError):
        engine.forward(torch.zeros(1, 512, 512, 3).cuda())


@pytest.mark.parametrize('lowering', ['large', 'small', 'w48_small', 'fp16x3_large'])
def test_fuse_sum_in_the_conv_epilogue_is_bit_equal_to_the_fuse_sum_launch(lowering, synth_sd, frames2):
    """VERDICT r4 item 6 / acr/model.py:672-686: the HR-module fuse sums of output resolutions i >= 1 run in the epilogue of
    the last convolution of their x0 downsampling chain (packer.Program.hr_module; conv_pp2_kernel<1, true> at large
    batches, the 32-cout stride-2 direct kernel at small ones; up to three extra residual maps, nearest-upsampled by
    2^shift, summed in the reference's order, then the ReLU).  The same fp32 operations in the same order as the
    OP_FUSESUM launch on the stored conv output: the backbone output and EVERY head map of the program with the fold are
    BIT-EQUAL to the program without it.  The eight FULL-resolution sums (i = 0) are the second output of branch 0's last conv2,
    written by conv_wino3_kernel's store waves (ACRMI_CONV_DUAL; large-batch fp32 W32 programs): 23 launches fewer per pass there,
    15 in the small-batch / W48 / split-operand programs."""
    packer, L = pkg('packer'), pkg('_lib')
    synth = pkg('synth')
    sd = synth_sd if not lowering.startswith('w48') else synth.make_state_dict(seed=0, width=48)
    kw = dict(max_batch=16) if lowering.endswith('large') else dict(max_batch=3)
    if lowering.startswith('fp16x3'):
        kw['precision'] = 'fp16x3'
    x = torch.from_numpy(np.concatenate([frames2, frames2[:1]])).cuda()
    res = {}
    saved = packer.FUSE_EPILOGUE
    # the full-resolution sums ride on conv_wino3_kernel's store waves (ACRMI_CONV_DUAL): fp32 HRNet-W32 programs
    full = 8 if lowering == 'large' else 0      # (large-batch fp32 W32 programs: packer.Program.fuse0_ok)
    try:
        for fold in (True, False):
            packer.FUSE_EPILOGUE = fold
            eng = pkg('engine').Engine(0)
            eng.load_state_dict(sd, **kw)
            ops = eng.program['ops']
            hosted = sum(1 for o in ops if o.kind == L.OP_CONV and o.nterms)
            sums = sum(1 for o in ops if o.kind == L.OP_FUSESUM)
            assert (hosted, sums) == ((15 + full, 8 - full) if fold else (0, 23)), (hosted, sums)
            assert sum(1 for o in ops if o.kind == L.OP_CONV and o.flags & L.CONV_DUAL) == (full if fold else 0)
            B = eng.backbone_heads(x)
            torch.cuda.synchronize()
            maps = {k: v.clone() for k, v in eng.head_maps(B).items()}
            hl = eng.program['heads']
            maps['backbone'] = eng.buffer(hl.backbone_buf, B).clone()
            res[fold] = maps
            eng.close()
    finally:
        packer.FUSE_EPILOGUE = saved
    for k in res[True]:
        assert torch.equal(res[True][k], res[False][k]), (lowering, k, (res[True][k] - res[False][k]).abs().max().item())
