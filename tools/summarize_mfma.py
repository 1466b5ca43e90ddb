"""Per-kernel matrix-pipe occupancy from a rocprofv3 PMC pass (tools/profile_round.sh: SQ_BUSY_CU_CYCLES,
SQ_VALU_MFMA_BUSY_CYCLES).  usage: summarize_mfma.py <dir with *counter_collection.csv> <out.txt>
Also writes <out>.json (per-variant busy fraction + launch-weighted means) for bench.py's roofline.mfma_busy_pmc."""
import json
import collections
import csv
import glob
import os
import re
import sys


def main():
    src, dst = sys.argv[1], sys.argv[2]
    path = glob.glob(os.path.join(src, '**', '*counter_collection.csv'), recursive=True)[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.defaultdict(set)
    with open(path) as f:
        for r in csv.DictReader(f):
            name = re.sub(r'^void ', '', r['Kernel_Name'])
            name = re.sub(r'\(.*$', '', name).replace('acrmi::', '')
            acc[name][r['Counter_Name']] += float(r['Counter_Value'])
            launches[name].add(r['Dispatch_Id'])
    rows = sorted(acc.items(), key=lambda kv: -kv[1].get('SQ_BUSY_CU_CYCLES', 0.0))
    with open(dst, 'w') as f:
        f.write('rocprofv3 --kernel-trace --pmc SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 -- '
                'python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-point-heads --no-latency --lanes 1 (B=64), summed per kernel\n')
        for name, c in rows:
            busy, mfma = c.get('SQ_BUSY_CU_CYCLES', 0.0), c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0)
            if busy < 1e8:
                continue
            f.write('%-60s launches=%4d  SQ_BUSY_CU_CYCLES=%.3e  SQ_VALU_MFMA_BUSY_CYCLES=%.3e  mfma_busy/(4*cu_busy)=%.3f\n'
                    % (name, len(launches[name]), busy, mfma, mfma / (4 * busy) if busy else 0.0))
    variants, tot = {}, collections.defaultdict(lambda: [0.0, 0.0])
    for name, c in rows:
        busy, mfma = c.get('SQ_BUSY_CU_CYCLES', 0.0), c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0)
        if busy < 1e8 or 'conv_' not in name:
            continue
        variants[name] = {'launches': len(launches[name]), 'mfma_busy_frac': round(mfma / (4 * busy), 4)}
        fam = 'conv_wino' if 'wino' in name else 'conv_direct'
        for k in (fam, 'all_convs'):
            tot[k][0] += mfma
            tot[k][1] += 4 * busy
    with open(os.path.splitext(dst)[0] + '.json', 'w') as f:
        json.dump({'definition': 'SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES), summed over launches',
                   'cycle_weighted': {k: round(v[0] / v[1], 4) for k, v in tot.items()}, 'variants': variants}, f, indent=1)


if __name__ == '__main__':
    main()
