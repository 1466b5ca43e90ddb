#!/usr/bin/env python
"""Micro-benchmark of the conv kernels on the shapes HRNet-W32 issues at batch 64 (GPU box only).

    python tools/conv_bench.py [--cfg N] [--batch 64] [--filter 3x3]
Prints one line per shape: avg ms over `--iters` launches and algorithmic TFLOP/s.
"""
import argparse
import ctypes as C
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = 'arbitrary-hands-3d-reconstruction_amd'

SHAPES = [
    # name, cin, cout, H, W, k, stride, groups, residual
    ('b0 32->32 3x3 @128', 32, 32, 128, 128, 3, 1, 1, True),
    ('b0 32->32 3x3 @128 no residual', 32, 32, 128, 128, 3, 1, 1, False),
    ('b1 64->64 3x3 @64', 64, 64, 64, 64, 3, 1, 1, True),
    ('b1 64->64 3x3 @64 no residual', 64, 64, 64, 64, 3, 1, 1, False),
    ('b2 128->128 3x3 @32', 128, 128, 32, 32, 3, 1, 1, True),
    ('b3 256->256 3x3 @16', 256, 256, 16, 16, 3, 1, 1, True),
    ('l1 64->64 3x3 @128', 64, 64, 128, 128, 3, 1, 1, False),
    ('towers 8x64->64 3x3 @64', 512, 512, 64, 64, 3, 1, 8, True),
    ('segm 16->64 3x3 @256', 16, 64, 256, 256, 3, 1, 1, False),
    ('segm 64->33 3x3 @256', 64, 33, 256, 256, 3, 1, 1, False),
    ('segm 33->33 3x3 @256', 33, 33, 256, 256, 3, 1, 1, False),
    ('contact 34->256 3x3 @128', 34, 256, 128, 128, 3, 1, 1, False),
    ('contact 32->256 3x3 @128 (+map)', 32, 256, 128, 128, 3, 1, 1, False),
    ('segm 256->32 3x3 @128', 256, 32, 128, 128, 3, 1, 1, False),
    ('l1 64->256 1x1 @128', 64, 256, 128, 128, 1, 1, 1, True),
    ('l1 256->64 1x1 @128', 256, 64, 128, 128, 1, 1, 1, False),
    ('r50 1024->256 1x1 @32', 1024, 256, 32, 32, 1, 1, 1, False),
    ('r50 256->1024 1x1 @32', 256, 1024, 32, 32, 1, 1, 1, True),
    ('fuse 128->32 1x1 @32', 128, 32, 32, 32, 1, 1, 1, False),
    ('s2 64->64 3x3s2 @256', 64, 64, 256, 256, 3, 2, 1, False),
    ('s2 256->64 3x3s2 @128', 256, 64, 128, 128, 3, 2, 1, False),
    ('entry 34->512 3x3s2 @128', 34, 512, 128, 128, 3, 2, 1, False),
    ('fuse 32->64 3x3s2 @128', 32, 64, 128, 128, 3, 2, 1, False),
    ('fuse 32->32 3x3s2 @128', 32, 32, 128, 128, 3, 2, 1, False),
    ('fuse 64->128 3x3s2 @64', 64, 128, 64, 64, 3, 2, 1, False),
    ('fuse 128->256 3x3s2 @32', 128, 256, 32, 32, 3, 2, 1, False),
    ('entry 32->512 3x3s2 @128 (+map)', 32, 512, 128, 128, 3, 2, 1, False),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--cfg', type=int, default=-1)
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--filter', default='')
    ap.add_argument('--wino', action='store_true', help='3x3 stride-1 shapes through the Winograd F(2,3) kernel')
    ap.add_argument('--wino2', action='store_true', help='... through the 2-D Winograd F(2x2,3x3) kernel')
    ap.add_argument('--wino24', action='store_true', help='... Cin > 16 shapes through the F(2x4,3x3) kernel (others as --wino2)')
    ap.add_argument('--wino3', action='store_true', help='... Cin<=32/Cout=32 shapes through the LDS-resident F(2x2,3x3) kernel')
    ap.add_argument('--pp2', action='store_true', help='3x3 stride-2 shapes through the polyphase kernel (conv_pp2.inc)')
    ap.add_argument('--x3', action='store_true', help='3x3 stride-1 shapes through the split-f16 kernel (conv_x3.inc)')
    ap.add_argument('--h16', default='', help="'fp16' | 'bf16': the 16-bit direct kernels (conv_h16.hip) on 16-bit tensors")
    ap.add_argument('--phase', type=int, default=0, help='loader-wave tuning switch: 8 = idle loader (timing ablation, wrong results), 9 = loader at priority 0')
    ap.add_argument('--stamps', action='store_true', help='print clock64 deltas of workgroup 0 (ws kernel)')
    args = ap.parse_args()
    L = importlib.import_module(PKG + '._lib')
    packer = importlib.import_module(PKG + '.packer')
    lib = L.lib()
    lib.acrmi_tune(0, args.cfg)
    lib.acrmi_tune(3, args.phase)
    B = args.batch
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    for name, cin, cout, H, W, k, stride, groups, use_res in SHAPES:
        if not any(f in name for f in args.filter.split(',')):
            continue
        cing, coutg = cin // groups, cout // groups
        if args.h16:
            td = torch.float16 if args.h16 == 'fp16' else torch.bfloat16
            dt = packer.PRECISIONS[args.h16]
            cs_in, cs_out = (cin + 7) // 8 * 8, (cout + 7) // 8 * 8
            x = torch.randn(B, H, W, cs_in, device='cuda').to(td)
            Ho, Wo = (H + 2 * (k // 2) - k) // stride + 1, (W + 2 * (k // 2) - k) // stride + 1
            out = torch.empty(B, Ho, Wo, cs_out, device='cuda', dtype=td)
            res = torch.randn(B, Ho, Wo, cs_out, device='cuda').to(td) if use_res else None
            w = (np.random.randn(coutg, cing, k, k) / np.sqrt(cing * k * k))
            packed = [packer.pack_conv_h16(w, np.zeros(coutg, np.float32), dt) for _ in range(groups)]
            wp = torch.from_numpy(np.concatenate([q[0] for q in packed]).view(np.int16)).cuda()
            bp = torch.from_numpy(np.concatenate([q[1] for q in packed])).cuda()

            def launch():
                rc = lib.acrmi_conv2d_h16(p(x), B, H, W, cs_in, 0, cing, p(wp), p(bp), 0, p(res), cs_out, 0, p(out), cs_out, 0,
                                          coutg, k, stride, 1, groups, dt, 0, None)
                assert rc == 0, lib.acrmi_last_error(None)
            for _ in range(3):
                launch()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                launch()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / args.iters
            flops = 2.0 * B * Ho * Wo * coutg * cing * k * k * groups
            byts = 2.0 * B * (H * W * cin + Ho * Wo * cout * (2 if use_res else 1))
            print('%-28s %8.3f ms  %6.1f TF  %6.2f TB/s(min traffic)' % (name, ms, flops / ms / 1e9, byts / ms / 1e9), flush=True)
            if args.stamps:
                lib.acrmi_tune(1, 1)
                launch()
                lib.acrmi_tune(2, 0)
                lib.acrmi_tune(1, 0)
            continue
        cs_in, cs_out = (cin + 3) // 4 * 4, (cout + 3) // 4 * 4
        x = torch.randn(B, H, W, cs_in, device='cuda')
        Ho, Wo = (H + 2 * (k // 2) - k) // stride + 1, (W + 2 * (k // 2) - k) // stride + 1
        out = torch.empty(B, Ho, Wo, cs_out, device='cuda')
        res = torch.randn(B, Ho, Wo, cs_out, device='cuda') if use_res else None
        w = (np.random.randn(coutg, cing, k, k) / np.sqrt(cing * k * k)).astype(np.float32)
        wino = 0
        if k == 3 and stride == 1:
            wino = 2 if (args.wino2 or args.wino3 or args.wino24) else (1 if args.wino else 0)
            if args.wino24 and coutg != 33 and ((cing > 32 and W % 32 == 0) or packer.wino24b_width(cing, coutg, H, W)):
                wino = 4      # (incl. Cin = 32 shapes: the four-wave frame's single-chunk kernels, also where the program keeps conv_wino3)
        if args.pp2 and k == 3 and stride == 2 and packer.polyphase2_ok(cing, coutg, H // 2, W // 2):
            wino = 5
        x3 = args.x3 and packer.split16_ok(k, stride, cing, coutg, Ho, Wo)
        if wino and args.wino3 and groups == 1 and cin <= 32 and cout == 32:
            wino = 3
            packed = [packer.pack_wino3(w.astype(np.float64), np.zeros(coutg, np.float32))]
        elif x3:
            wino = 6
            packed = [packer.pack_conv_x3([(w.astype(np.float64), np.zeros(coutg, np.float32)) for _ in range(groups)])]
        else:
            tr = (lambda t: t, packer.winograd_weights, packer.winograd2d_weights, None, packer.winograd24_weights,
                  packer.polyphase2_weights)[wino]
            packed = [packer.pack_conv(tr(w), np.zeros(coutg, np.float32)) for _ in range(groups)]
        wp = torch.from_numpy(np.concatenate([q[0] for q in packed])).cuda()
        bp = torch.from_numpy(np.concatenate([q[1] for q in packed])).cuda()

        def launch():
            rc = lib.acrmi_conv2d(p(x), B, H, W, cs_in, 0, cing, p(wp), p(bp), 0, p(res), cs_out, 0, p(out), cs_out, 0,
                                  coutg, k, stride, 1, groups, wino, None)
            assert rc == 0, lib.acrmi_last_error(None)
        for _ in range(3):
            launch()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            launch()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.iters
        flops = 2.0 * B * Ho * Wo * coutg * cing * k * k * groups
        byts = 4.0 * B * (H * W * cin + Ho * Wo * cout * (2 if use_res else 1))
        print('%-28s %8.3f ms  %6.1f TF  %6.2f TB/s(min traffic)' % (name, ms, flops / ms / 1e9, byts / ms / 1e9), flush=True)
        if args.stamps:
            lib.acrmi_tune(1, 1)
            launch()
            lib.acrmi_tune(2, 0)
            lib.acrmi_tune(1, 0)


if __name__ == '__main__':
    main()
