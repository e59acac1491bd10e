#!/usr/bin/env python3
"""HBM traffic per launch from two rocprofv3 PMC passes (collected separately, kernel-trace only):

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d <dirF> -- python3 bench.py --graph 0 ...
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d <dirW> -- python3 bench.py --graph 0 ...
    python tools/pmc_traffic.py <dirF> <dirW> profiles/r01_pmc_traffic.json

Units and corrections as MI355X_MICROARCH.md prescribes: the counters are in KB; gfx950 reports half
of wide streaming reads in FETCH_SIZE, so fetch bytes are doubled; bytes = 2*FETCH + WRITE.
Kernels are keyed by name up to the argument list (template arguments kept)."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def collect(d, counter):
    acc = defaultdict(lambda: [0, 0.0])
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        for row in csv.DictReader(open(f)):
            if row['Counter_Name'] != counter:
                continue
            name = re.sub(r'^void ', '', row['Kernel_Name'])
            name = name.replace('(anonymous namespace)::', '').split('(')[0]
            a = acc[name]
            a[0] += 1
            a[1] += float(row['Counter_Value']) * 1024.0
    return acc


def main(dir_f, dir_w, out):
    f, w = collect(dir_f, 'FETCH_SIZE'), collect(dir_w, 'WRITE_SIZE')
    res = {}
    for k in sorted(f):
        if k.startswith(('at::', '__amd', 'void at::')) or f[k][0] == 0:
            continue
        fe = 2.0 * f[k][1] / f[k][0]
        wr = w[k][1] / w[k][0] if k in w and w[k][0] else 0.0
        res[k] = {'launches': f[k][0], 'fetch_bytes_x2_avg': round(fe), 'write_bytes_avg': round(wr),
                  'hbm_bytes_avg': round(fe + wr)}
    res['_note'] = ('rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, KB units); FETCH_SIZE doubled per '
                    'MI355X_MICROARCH.md (gfx950 reports 1/2 of wide streaming reads); per-launch averages over the '
                    'eager train steps of one bench.py run')
    json.dump(res, open(out, 'w'), indent=1)
    for k, v in res.items():
        if isinstance(v, dict):
            print('%-60s n=%5d  %8.2f MB/launch' % (k[:60], v['launches'], v['hbm_bytes_avg'] / 1e6))


if __name__ == '__main__':
    main(*sys.argv[1:4])
