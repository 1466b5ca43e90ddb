#!/usr/bin/env python
"""rocprofv3 PMC passes -> HBM bytes per launch and kernel (profiles/rNN_hbm_traffic.json).

    python tools/summarize_pmc.py <fetch_dir> <write_dir> <out.json>
The two directories hold `rocprofv3 --kernel-trace --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` runs (separate passes, no
other trace domains) of `python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-point-heads --no-latency --lanes 1 --pipeline 1`.  Counter unit: KiB.
gfx950 correction (MI355X_MICROARCH.md, calibrated here on u8norm: 24589 KiB reported for 50.33 MB read):
FETCH_SIZE under-reports wide coalesced reads by 2 -> doubled; WRITE_SIZE is exact.
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def load(d, counter):
    tot, cnt = defaultdict(float), defaultdict(int)
    for path in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                if row['Counter_Name'] != counter:
                    continue
                name = re.sub(r'<.*', '', row['Kernel_Name'].replace('void ', '').replace('acrmi::', ''))
                name = re.sub(r'\(.*', '', name)
                tot[name] += float(row['Counter_Value'])
                cnt[name] += 1
    return tot, cnt


def main():
    fetch_dir, write_dir, out = sys.argv[1:4]
    ft, fc = load(fetch_dir, 'FETCH_SIZE')
    wt, wc = load(write_dir, 'WRITE_SIZE')
    kernels = {}
    for k in sorted(ft, key=lambda n: -ft[n]):
        if not k.startswith('conv_') or fc[k] == 0 or wc.get(k, 0) == 0:
            continue
        fb = 2.0 * 1024.0 * ft[k] / fc[k]
        wb = 1024.0 * wt[k] / wc[k]
        kernels[k] = {'launches_profiled': fc[k], 'fetch_bytes_per_launch': fb, 'write_bytes_per_launch': wb,
                      'hbm_bytes_per_launch': fb + wb}
    with open(out, 'w') as f:
        json.dump({'note': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) on '
                   '`python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-point-heads --no-latency --lanes 1 --pipeline 1`, B=64; FETCH_SIZE doubled per '
                   'MI355X_MICROARCH.md (gfx950 reports 1/2 of wide coalesced reads), WRITE_SIZE exact',
                   'kernels': kernels}, f, indent=1)
    print(json.dumps(kernels, indent=1))


if __name__ == '__main__':
    main()
