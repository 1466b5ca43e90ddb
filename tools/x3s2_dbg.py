"""conv_x3s2_kernel (3x3 stride 2, split operands) against an fp64 direct convolution on a few shapes, printing WHERE the bad
elements sit (channels / rows / columns) - the tool that separated the one-n-tile multi-chunk f16 build with 584 bytes of
scratch (wrong sums) from its bf16 twin (right) in round 6 (GPU box only):

    python tools/x3s2_dbg.py
"""
import importlib, sys, os
import numpy as np, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module('arbitrary-hands-3d-reconstruction_amd.ops')
def run(B, cin, cout, H, W, groups, algo):
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin // groups, 3, 3, generator=g) / np.sqrt(cin // groups * 9)
    ref = F.conv2d(x.double(), w.double(), None, 2, 1, 1, groups)
    xin = x.permute(0, 2, 3, 1).contiguous().cuda()
    out = ops.conv2d(xin, w, None, stride=2, groups=groups, cin=cin // groups, algo=algo)
    torch.cuda.synchronize()
    got = out[..., :cout].permute(0, 3, 1, 2).cpu().double()
    d = (got - ref).abs()
    bad = ~torch.isfinite(got) | (d > 1e-3)
    print(algo, (B, cin, cout, H, W, groups), 'max err', d[torch.isfinite(d)].max().item(), 'bad', int(bad.sum()))
    if bad.any():
        idx = bad.nonzero()
        print('  channels', sorted(set(idx[:, 1].tolist()))[:40])
        print('  rows', sorted(set(idx[:, 2].tolist())), 'cols', sorted(set(idx[:, 3].tolist()))[:40])
for case in [(1, 128, 192, 16, 64, 2), (1, 64, 96, 16, 64, 1), (1, 64, 32, 16, 64, 1), (1, 32, 96, 16, 64, 1), (1, 64, 64, 16, 64, 1), (1, 128, 128, 16, 64, 2)]:
    for algo in ('split16', 'split_bf16'):
        run(*case, algo)
