#!/usr/bin/env python3
"""How much of the LAST training step of a rocprofv3 rocpd database runs concurrently: span, kernel-sum, and for every weight-gradient
launch its interval, duration and the chain launches that ran inside it.  Usage: overlap_inventory.py results.db"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute('select name, start, end from kernels order by start').fetchall()
marks = [i for i, r in enumerate(rows) if 'adamw_kernel' in r[0]]
seg = rows[marks[-2] + 1:marks[-1] + 1]
t0 = seg[0][1]
span = (seg[-1][2] - t0) / 1e3
ksum = sum(e - s for _, s, e in seg) / 1e3
print('launches %d  span %.1f us  kernel-sum %.1f us  (overlap %.1f us)' % (len(seg), span, ksum, ksum - span))
for nm, s, e in seg:
    if 'wgrad' not in nm:
        continue
    inside = [(n2, s2, e2) for n2, s2, e2 in seg if 'wgrad' not in n2 and s2 < e and e2 > s]
    ov = sum(min(e, e2) - max(s, s2) for _, s2, e2 in inside) / 1e3
    print('%-50s start %8.1f us  dur %7.1f us  chain launches inside %3d (%.1f us of chain time)'
          % (nm.split('(')[0][-50:], (s - t0) / 1e3, (e - s) / 1e3, len(inside), ov))
