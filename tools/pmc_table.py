#!/usr/bin/env python3
"""Per-kernel averages of every counter found in rocprofv3 PMC passes (csv output).
Usage: pmc_table.py <dir> [<dir> ...] [--match substr] [--json out.json]"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

dirs, match, out = [], None, None
it = iter(sys.argv[1:])
for a in it:
    if a == '--match':
        match = next(it)
    elif a == '--json':
        out = next(it)
    else:
        dirs.append(a)
acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
for d in dirs:
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        for row in csv.DictReader(open(f)):
            name = re.sub(r'^void ', '', row['Kernel_Name']).replace('(anonymous namespace)::', '').split('(')[0]
            if match and match not in name:
                continue
            acc[name][row['Counter_Name']] += float(row['Counter_Value'])
            cnt[name][row['Counter_Name']] += 1
res = {}
for k in sorted(acc, key=lambda k: -acc[k].get('SQ_WAVE_CYCLES', 0)):
    res[k] = {c: acc[k][c] / cnt[k][c] for c in sorted(acc[k])}
    res[k]['launches'] = max(cnt[k].values())
    print(k[:70], 'n=%d' % res[k]['launches'])
    for c in sorted(acc[k]):
        print('    %-28s %14.0f' % (c, res[k][c]))
if out:
    json.dump(res, open(out, 'w'), indent=1)
