"""Seeded inputs shared by make_golden.py (authoring container, real reference) and the tests.

Everything here is data generation with numpy PCG64 (identical on every box); no reference code.
"""
import numpy as np


def rng(seed):
    return np.random.Generator(np.random.PCG64(seed))


# ---- decode cases (G4): head maps with planted center peaks -------------------------------------
# name -> (left peak (y,x,score) or None, right peak or None)
DECODE_CASES = {
    'both_near': ((20, 30, 0.9), (25, 40, 0.8)),
    'both_far': ((5, 6, 0.7), (50, 55, 0.95)),
    'both_dist32': ((10, 10, 0.6), (10, 42, 0.6)),        # distance == 32 exactly -> prior kept
    'left_only': ((33, 12, 0.5), None),
    'right_only': (None, (63, 63, 0.36)),
    'none': (None, None),
    'edge_corner': ((0, 0, 0.9), (63, 0, 0.9)),
    'below_thresh': ((12, 12, 0.35), (40, 40, 0.3499)),   # strict > 0.35
}


def decode_maps(name):
    """Head-map dict (float32, NCHW, B=1) for a decode case."""
    lp, rp = DECODE_CASES[name]
    g = rng(abs(hash_name(name)))
    m = {}
    for s, pk in (('l', lp), ('r', rp)):
        c = g.uniform(-0.2, 0.2, (1, 1, 64, 64)).astype(np.float32)
        if pk is not None:
            c[0, 0, pk[0], pk[1]] = pk[2]
        m[s + '_center_map'] = c
        m[s + '_params_maps'] = g.normal(0, 0.6, (1, 109, 64, 64)).astype(np.float32)
        m[s + '_prior_maps'] = g.normal(0, 0.3, (1, 106, 64, 64)).astype(np.float32)
    return m


# Batches of the cases above: what the reference's parse_maps does with MIXED detection states in one
# batch - the prior gated on every flag of the batch (acr/result_parser.py:131), determine_coeff deciding from row 0 of each
# side's list (:42-47), whole-batch placeholder rows (:102-120).  tests/golden/make_golden_batch.py -> decode_batches.npz
DECODE_BATCHES = {
    'b2_both_leftonly': ('both_near', 'left_only'),                       # as per frame: prior in frame 0
    'b2_far_near': ('both_far', 'both_near'),                             # row 0 is far apart: NO prior anywhere (frame 1 differs)
    'b2_near_far': ('both_near', 'both_far'),                             # row 0 is near: prior everywhere (frame 1 differs)
    'b2_leftonly_dist32': ('left_only', 'both_dist32'),                   # row 0 of the two sides come from different frames: 37.8 px -> none
    'b3_rightonly_near_none': ('right_only', 'both_near', 'none'),        # l row 0 = frame 1, r row 0 = frame 0: 54 px -> none
    'b4_no_right_at_all': ('left_only', 'below_thresh', 'none', 'left_only'),   # right placeholder row: flag False, no prior
    'b4_mixed': ('both_near', 'none', 'right_only', 'both_far'),          # prior in frames 0 AND 3
    'b2_none_none': ('none', 'none'),                                     # two placeholder rows
}


# the whole reference (model.forward + MANOWrapper) on a batch > 1: name -> (checkpoint seed of STATE_CHECKPOINTS, frames)
E2E_BATCHES = {'both_b4': (10, 4), 'left_only_b3': (3, 3), 'none_b2': (7, 2)}


# ManoLayer beyond the wrapper's configuration (VERDICT r3 "missing" 5): constructor options + forward options, inputs seeded
# name -> (ctor kwargs, forward options, n, seed)
MANO_OPTION_CASES = {
    'pca6_flat_right_c9': (dict(use_pca=True, ncomps=6, flat_hand_mean=True, side='right', center_idx=9), {}, 4, 21),
    'pca12_mean_left_nocenter': (dict(use_pca=True, ncomps=12, flat_hand_mean=False, side='left', center_idx=None), {}, 3, 22),
    'palm_right_c0': (dict(use_pca=False, flat_hand_mean=False, side='right', center_idx=0), dict(root_palm=True), 3, 23),
    'palm_trans_left': (dict(use_pca=False, flat_hand_mean=True, side='left', center_idx=9), dict(root_palm=True, trans=True), 2, 24),
    'pca45_share_betas': (dict(use_pca=True, ncomps=45, flat_hand_mean=False, side='right', center_idx=9), dict(share_betas=True), 4, 25),
}


def mano_option_inputs(name):
    kw, opt, n, seed = MANO_OPTION_CASES[name]
    g = rng(1300 + seed)
    nc = kw['ncomps'] if kw.get('use_pca') else 45
    poses = g.normal(0, 0.5, (n, 3 + nc)).astype(np.float32)
    betas = g.normal(0, 1.0, (n, 10)).astype(np.float32)
    trans = g.normal(0, 0.2, (n, 3)).astype(np.float32) if opt.get('trans') else None
    return poses, betas, trans


# joint_rot_mode='rotmat' (mano/manolayer.py:151-162): name -> (ctor kwargs, n, seed, noise)
MANO_ROTMAT_CASES = {
    'rotmat_right_c9': (dict(use_pca=False, joint_rot_mode='rotmat', side='right', center_idx=9), 3, 41, 0.0),
    'rotmat_left_nocenter_noisy': (dict(use_pca=False, joint_rot_mode='rotmat', side='left', center_idx=None), 3, 42, 0.05),
    'rotmat_right_c0_reflection': (dict(use_pca=False, joint_rot_mode='rotmat', side='right', center_idx=0), 2, 43, 0.02),
}


def mano_rotmat_inputs(name):
    """[n,16,3,3] float32 matrices + betas: exact rotations (fp64 Rodrigues of seeded axis-angles), plus seeded noise so that
    batch_rotprojs' SVD projection matters, plus - in the '_reflection' case - one matrix with a negated column (det < 0:
    the reference's reflection branch)."""
    kw, n, seed, noise = MANO_ROTMAT_CASES[name]
    g = rng(1700 + seed)
    aa = g.normal(0, 0.6, (n, 16, 3))
    th = np.linalg.norm(aa, axis=-1, keepdims=True)
    k = aa / th
    K = np.zeros((n, 16, 3, 3))
    K[..., 0, 1], K[..., 0, 2], K[..., 1, 0] = -k[..., 2], k[..., 1], k[..., 2]
    K[..., 1, 2], K[..., 2, 0], K[..., 2, 1] = -k[..., 0], -k[..., 1], k[..., 0]
    R = np.eye(3) + np.sin(th)[..., None] * K + (1 - np.cos(th))[..., None] * (K @ K)
    R = R + noise * g.normal(0, 1, R.shape)
    if name.endswith('_reflection'):
        R[0, 5, :, 1] *= -1
    betas = g.normal(0, 1.0, (n, 10)).astype(np.float32)
    return R.astype(np.float32), betas


def decode_batch_maps(name):
    """Head-map dict (float32, NCHW, B = len(members)) of a decode batch: the members' maps, concatenated."""
    ms = [decode_maps(m) for m in DECODE_BATCHES[name]]
    return {k: np.concatenate([m[k] for m in ms], 0) for k in ms[0]}


def hash_name(name):
    h = 0
    for ch in name:
        h = (h * 131 + ord(ch)) % (2 ** 31)
    return h


# ---- rotation KATs (G5) ---------------------------------------------------------------------------
def rot6d_inputs():
    g = rng(55)
    x = g.normal(0, 1, (40, 6)).astype(np.float32)
    x[0] = [1, 0, 0, 1, 0, 0]                 # identity: b1=(1,0,0) a2=(0,1,0) interleaved
    x[1] = [1, 1e-4, 0, 1, 1e-4, 0]           # near identity
    x[2] = [-1, 0, 0, -1, 0, 1]               # rotation by pi about z
    x[3] = [-1, 1e-3, 0, -1, 0, 1]            # near pi
    x[4] = [1, 2, 1, 2, 1, 2]                 # parallel b1, a2 (degenerate)
    x[5] = [0, 0, 0, 0, 0, 0]                 # zero
    x[6] = [1, 0, 0, 0, 0, 1]                 # R[2,2] < eps branch
    x[7] = [0, 1, 1, 0, 0, 0]
    x[8] = [-1, 0, 0, 1, 0, 0]
    x[9] = [0, -1, -1, 0, 0, 0]
    return x


# ---- MANO cases (G6) ------------------------------------------------------------------------------
def mano_inputs(n, seed):
    g = rng(700 + seed)
    poses = g.normal(0, 0.4, (n, 48)).astype(np.float32)
    betas = g.normal(0, 1.0, (n, 10)).astype(np.float32)
    if n >= 1:
        poses[0] = 0
        betas[0] = 0
    if n >= 2:
        poses[1] = g.normal(0, 1.5, 48).astype(np.float32)   # large pose
    return poses, betas


def proj_inputs(n, seed):
    g = rng(900 + seed)
    cam = np.stack([g.uniform(0.5, 2.0, n), g.uniform(-0.5, 0.5, n), g.uniform(-0.5, 0.5, n)], 1).astype(np.float32)
    offsets = np.tile(np.array([[1920, 1920, 0, 0, 0, 0, 420, 0, 420, 0]], np.float32), (n, 1))
    return cam, offsets


def sub(t, max_elems=4096):
    """Deterministic strided subsample of an array's flattened view + fp64 checksums."""
    a = np.asarray(t, dtype=np.float32).reshape(-1)
    step = max(1, a.size // max_elems)
    return a[::step][:max_elems].copy(), np.array([a.astype(np.float64).sum(), np.abs(a.astype(np.float64)).sum()])


# ---- temporal smoothing sequence (G9) -----------------------------------------------------------------
def smooth_inputs(n_frames=14, seed=41):
    """A random-walk (poses [T,2,48], betas [T,2,10]) sequence for the two hand types plus per-frame detection
    flags [T,2]: the right hand appears at frame 2, the left hand drops out for frames 6-7 (its filters then keep
    their state, acr/main.py:78-80), a near-pi global orientation sits at frame 9."""
    g = rng(seed)
    poses = np.zeros((n_frames, 2, 48), np.float32)
    betas = np.zeros((n_frames, 2, 10), np.float32)
    p = g.normal(0, 0.5, (2, 48))
    b = g.normal(0, 0.8, (2, 10))
    for t in range(n_frames):
        p = p + g.normal(0, 0.12, (2, 48))
        b = b + g.normal(0, 0.05, (2, 10))
        poses[t], betas[t] = p, b
    poses[9, 0, :3] = [3.1, 0.02, -0.03]
    flags = np.ones((n_frames, 2), bool)
    flags[:2, 1] = False
    flags[6:8, 0] = False
    return poses, betas, flags


# ---- network states through the real reference (G10): checkpoint seed -> detection state on synth frames --
# (found by scanning seeds with the pinned oracle; seed 0 = both hands, centres 63 px apart, is e2e_batch1.npz)
STATE_CHECKPOINTS = {'both_near': 10, 'left_only': 3, 'right_only': 11, 'none': 7}
STATE_FRAME_SEED = 5


# ---- interior centers through the real reference (G12): planted center peaks (synth.plant_center_peaks) ----------------
# name -> (checkpoint seed, left (y, x), right (y, x)); every peak >= 9 px from every border of the 64x64 map
INTERIOR_CASES = {
    'near': (0, (20, 30), (40, 12)),        # 26.9 px apart: the cross-hand prior is applied
    'far': (10, (12, 10), (50, 52)),        # 56.6 px apart: priors zeroed (acr/result_parser.py:42-47)
    'mid': (3, (32, 33), (33, 20)),
}


def interior_state_dict(synth, name, as_torch=True):
    seed, lp, rp = INTERIOR_CASES[name]
    sd = synth.make_state_dict(seed=seed, as_torch=False)
    synth.plant_center_peaks(sd, left=lp, right=rp)
    return synth._as_torch(sd) if as_torch else sd
