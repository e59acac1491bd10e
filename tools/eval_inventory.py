#!/usr/bin/env python3
"""Kernel inventory of the sampler's network evaluations in a rocprofv3 rocpd database: per kernel name,
launches / total / average over the whole run divided by the number of `sampler_step_kernel` launches.
Usage: eval_inventory.py results.db [rows]"""
import sqlite3
import sys
from collections import Counter

db = sqlite3.connect(sys.argv[1])
rows = db.execute('select name, start, end from kernels order by start').fetchall()
steps = sum(1 for r in rows if 'sampler_step' in r[0])
cnt, tim = Counter(), Counter()
for nm, s, e in rows:
    for junk in ('void (anonymous namespace)::', 'void at::native::', '(anonymous namespace)::'):
        nm = nm.replace(junk, '')
    cnt[nm[:78]] += 1
    tim[nm[:78]] += e - s
print('sampler steps %d  kernel-sum per step %.2f ms' % (steps, sum(tim.values()) / steps / 1e6))
for k, v in tim.most_common(int(sys.argv[2]) if len(sys.argv) > 2 else 30):
    print('%6.1f/step %8.1f us/step %7.1f avg  %s' % (cnt[k] / steps, v / steps / 1e3, v / cnt[k] / 1e3, k))
