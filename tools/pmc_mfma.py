#!/usr/bin/env python3
"""MFMA-pipe utilisation per kernel from one rocprofv3 PMC pass (kernel-trace only, no other trace domains):

    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CU_CYCLES \
        --output-format csv -d <dir> -- python3 bench.py --graph 0 --steps 3 --warmup 1 ...
    python tools/pmc_mfma.py <dir> profiles/r01_pmc_mfma.json

SQ_VALU_MFMA_BUSY_CYCLES counts cycles per SIMD summed over the chip (MI355X_MICROARCH.md: = 16 x N for
v_mfma_f32_16x16x32_bf16), so utilisation = busy / (launch duration x 2.4 GHz x 1024 SIMDs); the duration
is the dispatch's own start/end in the same pass (counter collection serialises dispatches, so this is the
kernel alone on the chip).  MOPS_BF16 counts 512-FLOP units: executed MFMA TFLOP/s = MOPS x 512 / duration."""
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
    for k, a in sorted(acc.items(), key=lambda kv: -kv[1].get('SQ_VALU_MFMA_BUSY_CYCLES', 0)):
        busy = a.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0)
        if busy == 0 or a['_ns'] == 0:
            continue
        util = busy / (a['_ns'] * CLOCK_GHZ * SIMDS)
        tf = a.get('SQ_INSTS_VALU_MFMA_MOPS_BF16', 0.0) * 512 / a['_ns'] / 1e3
        res[k] = {'launches': int(a['_n']), 'avg_us': round(a['_ns'] / a['_n'] / 1e3, 2),
                  'mfma_busy_frac': round(util, 4), 'executed_bf16_mfma_tflops': round(tf, 1)}
        print('%-56s n=%5d %8.1f us  MFMA busy %5.1f %%  %7.1f TF/s executed' %
              (k[:56], a['_n'], a['_ns'] / a['_n'] / 1e3, 100 * util, tf))
    res['_note'] = ('one rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES, SQ_INSTS_VALU_MFMA_MOPS_BF16), eager train steps '
                    'of bench.py; busy / (duration x 2.4 GHz x 1024 SIMDs); dispatches are serialised under counter '
                    'collection')
    json.dump(res, open(out, 'w'), indent=1)


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
