#!/usr/bin/env python3
"""Where the LAST training step of a rocprofv3 rocpd database spends its time, by kernel AND launch grid (the grid tells the
resolution level a conv / GroupNorm launch works at).  Usage: level_inventory.py results.db [rows]"""
import sqlite3
import sys
from collections import Counter

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info('kernels')").fetchall()]
gx = next((c for c in cols if c.lower() in ('grid_x', 'grid_size_x', 'grid_size')), None)
wx = next((c for c in cols if c.lower() in ('workgroup_x', 'workgroup_size_x', 'workgroup_size')), None)
if gx is None:
    print('columns:', cols)
    sys.exit(1)
rows = db.execute('select name, start, end, %s, %s from kernels order by start' % (gx, wx or gx)).fetchall()
marks = [i for i, r in enumerate(rows) if 'adamw_kernel' in r[0]]
seg = rows[marks[-2] + 1:marks[-1] + 1]


def short(n):
    for junk in ('void (anonymous namespace)::', 'void at::native::', '(anonymous namespace)::'):
        n = n.replace(junk, '')
    return n.split('(')[0][:60]


cnt, tim = Counter(), Counter()
for nm, s, e, g, w in seg:
    k = (short(nm), g // max(w, 1) if wx else g)
    cnt[k] += 1
    tim[k] += e - s
tot = sum(tim.values())
print('launches %d  kernel-sum %.2f ms' % (len(seg), tot / 1e6))
buckets = Counter()
for (nm, blocks), t in tim.items():
    b = '<64' if blocks < 64 else '<128' if blocks < 128 else '<256' if blocks < 256 else '<512' if blocks < 512 else '>=512'
    buckets[b] += t
print('time by launch size (workgroups):', ', '.join('%s: %.2f ms' % (k, buckets[k] / 1e6) for k in ('<64', '<128', '<256', '<512', '>=512')))
for k, v in tim.most_common(int(sys.argv[2]) if len(sys.argv) > 2 else 60):
    print('%4d %8.1f us %6.1f avg  blocks %6d  %s' % (cnt[k], v / 1e3, v / 1e3 / cnt[k], k[1], k[0]))
