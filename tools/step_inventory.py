#!/usr/bin/env python3
"""Kernel inventory of the LAST training step in a rocprofv3 rocpd database (graph-replayed bench):
launch count, total and average time per kernel.  Usage: step_inventory.py results.db [rows]"""
import sqlite3
import sys
from collections import Counter

db = sqlite3.connect(sys.argv[1])
rows = db.execute('select name, start, end from kernels order by start').fetchall()
marks = [i for i, r in enumerate(rows) if 'adamw_kernel' in r[0]]
seg = rows[marks[-2] + 1:marks[-1] + 1]
print('launches %d  span %.2f ms  kernel-sum %.2f ms' % (len(seg), (seg[-1][2] - seg[0][1]) / 1e6,
                                                         sum(e - s for _, s, e in seg) / 1e6))


def short(n):
    for junk in ('void (anonymous namespace)::', 'void at::native::', '(anonymous namespace)::'):
        n = n.replace(junk, '')
    return n[:78]


cnt, tim = Counter(), Counter()
for nm, s, e in seg:
    cnt[short(nm)] += 1
    tim[short(nm)] += e - s
for k, v in tim.most_common(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
    print('%4d %8.1f us %6.1f avg  %s' % (cnt[k], v / 1e3, v / 1e3 / cnt[k], k))
