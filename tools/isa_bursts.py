#!/usr/bin/env python3
"""Run-length view of a gfx950 kernel's instruction stream: how the non-MFMA instructions sit between the MFMAs.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -S --cuda-device-only -Iinclude <file>.hip -o /tmp/k.s
    python tools/isa_bursts.py /tmp/k.s conv_wino24b_kernelILi2ELi32ELb0 [--min-mfma 8]

Per basic block (label to label) with at least --min-mfma MFMAs: the stream as runs - M = MFMA, V = VALU (incl.
v_accvgpr_*), L = LDS, G = global / buffer / scratch memory, D = LDS-DMA (buffer_load ... lds), S = scalar, W = s_waitcnt,
B = s_barrier, N = s_nop - and the cost the measured table of tools/coissue (profiles/r06_coissue.txt) assigns to it for an
fp32 MFMA stream on one wave per SIMD: MFMA 64 (32x32x2) / 32 (16x16x4) cycles, every VALU 4 cycles + 10 for the first one
behind an MFMA (a "burst"), LDS read ~1, global / buffer load ~16, LDS-DMA piece ~60.
"""
import re
import sys


def classify(op, rest):
    if op.startswith('v_mfma'):
        return 'M'
    if op.startswith('v_'):
        return 'V'
    if op.startswith('ds_'):
        return 'L'
    if op.startswith(('buffer_', 'global_', 'scratch_', 'flat_')):
        return 'D' if re.search(r'\blds\b', rest) else 'G'
    if op == 's_waitcnt':
        return 'W'
    if op == 's_barrier':
        return 'B'
    if op == 's_nop':
        return 'N'
    if op.startswith('s_'):
        return 'S'
    return '?'


def main():
    path, name = sys.argv[1], sys.argv[2]
    min_mfma = int(sys.argv[sys.argv.index('--min-mfma') + 1]) if '--min-mfma' in sys.argv else 8
    lines = open(path).read().split('\n')
    start = next(i for i, l in enumerate(lines) if l.startswith('_Z') and ':' in l and name in l.split(':')[0])
    blocks = []
    cur = ('entry', [])
    for l in lines[start + 1:]:
        s = l.strip()
        if s.startswith('.Lfunc_end'):
            break
        if s.startswith('.LBB') and s.split(':')[0].endswith(tuple('0123456789')) and ':' in s:
            blocks.append(cur)
            cur = (s.split(':')[0], [])
            continue
        if not s or s.startswith((';', '.', '//')):
            continue
        m = re.match(r'([a-z_0-9]+)\s*(.*)', s)
        if not m:
            continue
        cur[1].append((m.group(1), m.group(2)))
    blocks.append(cur)
    tot = {}
    for label, ins in blocks:
        kinds = [classify(op, rest) for op, rest in ins]
        nm = kinds.count('M')
        if nm < min_mfma:
            continue
        runs = []
        for k in kinds:
            if runs and runs[-1][0] == k:
                runs[-1][1] += 1
            else:
                runs.append([k, 1])
        mf = [op for op, _ in ins if op.startswith('v_mfma')]
        per = 32 if '16x16x4' in mf[0] or '32x32x16' in mf[0] else 64
        nv = kinds.count('V')
        bursts = 0
        last_vec = None
        for k in kinds:
            if k == 'V' and last_vec == 'M':
                bursts += 1
            if k in 'MV':
                last_vec = k
        ng, nd, nl = kinds.count('G'), kinds.count('D'), kinds.count('L')
        cost_m = nm * per
        cost_o = 4 * nv + 10 * bursts + 16 * ng + 60 * nd + 1 * nl
        print('%s: %d instructions, MFMA %d (%s), VALU %d in %d bursts, LDS %d, mem %d, DMA %d, salu %d, waitcnt %d, barrier %d' % (
            label, len(ins), nm, mf[0], nv, bursts, nl, ng, nd, kinds.count('S'), kinds.count('W'), kinds.count('B')))
        print('   modelled: MFMA %d cycles + other %d (VALU %d, bursts %d, mem %d, DMA %d, LDS %d) -> %.3f' % (
            cost_m, cost_o, 4 * nv, 10 * bursts, 16 * ng, 60 * nd, nl, cost_m / float(cost_m + cost_o)))
        print('   ' + ' '.join('%s%d' % (k, n) for k, n in runs))
        for k in 'MVLGD':
            tot[k] = tot.get(k, 0) + kinds.count(k)
    print('total over listed blocks:', tot)


if __name__ == '__main__':
    main()
