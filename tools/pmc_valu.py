#!/usr/bin/env python3
"""Where the waves of each kernel spend their cycles, from one rocprofv3 PMC pass (kernel-trace only):

    rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU \
        --output-format csv -d <dir> -- python3 bench.py --graph 0 --steps 3 --warmup 2 ...
    python tools/pmc_valu.py <dir> profiles/r05_pmc_valu.json

SQ_WAVE_CYCLES / SQ_ACTIVE_INST_* / SQ_WAIT_* count quad-cycles summed over the waves (MI355X_MICROARCH.md): the shares below are
of the waves' lifetime.  `valu_simd` = vector-instruction issue cycles per SIMD and launch cycle: INSTS_VALU x 4 cycles per wave64
instruction / (duration x 2.4 GHz x 1024 SIMDs) -- the fraction of the chip's vector issue slots the kernel fills (an MFMA is a VALU
instruction too)."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

CLOCK_GHZ, SIMDS = 2.4, 256 * 4


def main(d, out):
    acc = defaultdict(lambda: defaultdict(float))
    seen = defaultdict(set)
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        for row in csv.DictReader(open(f)):
            name = re.sub(r'^void ', '', row['Kernel_Name']).replace('(anonymous namespace)::', '').split('(')[0]
            a = acc[name]
            a[row['Counter_Name']] += float(row['Counter_Value'])
            did = row.get('Dispatch_Id')
            if did not in seen[name]:
                seen[name].add(did)
                a['_n'] += 1
                a['_ns'] += float(row['End_Timestamp']) - float(row['Start_Timestamp'])
    res = {}
    for k, a in sorted(acc.items(), key=lambda kv: -kv[1].get('_ns', 0)):
        wc = a.get('SQ_WAVE_CYCLES', 0.0)
        if wc == 0 or a['_ns'] == 0:
            continue
        r = {'launches': int(a['_n']), 'avg_us': round(a['_ns'] / a['_n'] / 1e3, 2),
             'valu_active_share': round(a.get('SQ_ACTIVE_INST_VALU', 0.0) / wc, 4),
             'any_active_share': round(a.get('SQ_ACTIVE_INST_ANY', 0.0) / wc, 4),
             'wait_share': round(a.get('SQ_WAIT_ANY', 0.0) / wc, 4),
             'issue_stall_share': round(a.get('SQ_WAIT_INST_ANY', 0.0) / wc, 4),
             'valu_simd': round(a.get('SQ_INSTS_VALU', 0.0) * 4 / (a['_ns'] * CLOCK_GHZ * SIMDS), 4)}
        res[k] = r
        print('%-60s n=%5d %8.1f us  VALU active %5.1f %%  any active %5.1f %%  waiting %5.1f %%  issue stall %5.1f %%  VALU issue / SIMD %5.1f %%' % (
            k[:60], r['launches'], r['avg_us'], 100 * r['valu_active_share'], 100 * r['any_active_share'], 100 * r['wait_share'],
            100 * r['issue_stall_share'], 100 * r['valu_simd']))
    res['_note'] = 'one rocprofv3 --pmc pass over eager train steps of bench.py; shares of the waves\' lifetime (quad-cycle counters)'
    json.dump(res, open(out, 'w'), indent=1)


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
